#!/usr/bin/env python3
"""Regenerates tests/golden/marks.npz: the fixed test watermarks of the reference.

The reference's integration tests build their marks with
    ChaCha8Rng::seed_from_u64(seed) ; sample(StandardNormal) x length
(/root/reference/tests/util.rs:6-13; seeds 1, 2 and 0xBAAAAAAD in
tests/single_simple.rs:19,84 and tests/attack_*.rs).  rand 0.8.5 / rand_chacha
0.3.1 / rand_distr 0.4.3 are not vendored and there is no Rust toolchain here,
so the *published algorithms* are restated from scratch:

  * seed_from_u64: PCG32 expansion of the u64 into a 32-byte key,
  * ChaCha with 8 rounds, 64-bit block counter from 0, stream id 0,
    words consumed in order, next_u64 = lo | hi << 32,
  * StandardNormal: 256-layer ziggurat (R = 3.654152885361008796,
    V = 4.92867323399e-3), f64, cast to f32.

Validation (no Rust needed): the seed-1 mark is what the reference embedded in
its own fixture tests/watermarked_with_1.png; tests/test_oracle_golden.py
extracts it from that PNG and correlates (similarity > 25 sigma).
"""
import math
import os
import struct

import numpy as np

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(x, n):
    return ((x << n) & M32) | (x >> (32 - n))


def _quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & M32; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & M32; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & M32; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & M32; s[b] = _rotl(s[b] ^ s[c], 7)


class ChaCha8:
    def __init__(self, seed_u64: int):
        state = seed_u64 & M64
        key = []
        for _ in range(8):                      # PCG32 expansion, advance-then-output
            state = (state * 6364136223846793005 + 11634580027462260723) & M64
            xorshifted = (((state >> 18) ^ state) >> 27) & M32
            rot = state >> 59
            key.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32)
        self.key = key
        self.counter = 0
        self.buf = []

    def _block(self):
        init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + self.key + [
            self.counter & M32, (self.counter >> 32) & M32, 0, 0]
        s = list(init)
        for _ in range(4):                      # 8 rounds = 4 double rounds
            _quarter(s, 0, 4, 8, 12); _quarter(s, 1, 5, 9, 13)
            _quarter(s, 2, 6, 10, 14); _quarter(s, 3, 7, 11, 15)
            _quarter(s, 0, 5, 10, 15); _quarter(s, 1, 6, 11, 12)
            _quarter(s, 2, 7, 8, 13); _quarter(s, 3, 4, 9, 14)
        self.counter += 1
        self.buf.extend((a + b) & M32 for a, b in zip(s, init))

    def next_u64(self) -> int:
        while len(self.buf) < 2:
            self._block()
        lo, hi = self.buf[0], self.buf[1]
        del self.buf[:2]
        return lo | (hi << 32)


ZIG_R = 3.654152885361008796
ZIG_V = 4.92867323399e-3


def _pdf(x):
    return math.exp(-x * x / 2.0)


def _tables():
    x = [0.0] * 257
    x[0] = ZIG_V / _pdf(ZIG_R)
    x[1] = ZIG_R
    for i in range(2, 256):
        x[i] = math.sqrt(-2.0 * math.log(ZIG_V / x[i - 1] + _pdf(x[i - 1])))
    x[256] = 0.0
    return x, [_pdf(v) for v in x]


ZX, ZF = _tables()


def _float_exp(bits52: int, exponent: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", bits52 | ((1023 + exponent) << 52)))[0]


def standard_normal(rng: ChaCha8) -> float:
    while True:
        bits = rng.next_u64()
        i = bits & 0xFF
        u = _float_exp(bits >> 12, 1) - 3.0
        x = u * ZX[i]
        if abs(x) < ZX[i + 1]:
            return x
        if i == 0:
            xx, yy = 1.0, 0.0
            while -2.0 * yy < xx * xx:
                eps_half = 1.0 - 2.0 ** -53
                x_ = _float_exp(rng.next_u64() >> 12, 0) - eps_half
                y_ = _float_exp(rng.next_u64() >> 12, 0) - eps_half
                xx = math.log(x_) / ZIG_R
                yy = math.log(y_)
            return xx - ZIG_R if u < 0.0 else ZIG_R - xx
        u01 = (rng.next_u64() >> 11) * 2.0 ** -53
        if ZF[i + 1] + (ZF[i] - ZF[i + 1]) * u01 < _pdf(x):
            return x


def generate_fixed_normal_sequence(seed: int, length: int) -> np.ndarray:
    rng = ChaCha8(seed)
    return np.array([standard_normal(rng) for _ in range(length)], dtype=np.float64).astype(np.float32)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    for name, seed, n in (("seed_1", 1, 1000), ("seed_2", 2, 1000), ("seed_baaaaaad", 0xBAAAAAAD, 1000),
                          ("seed_1_10000", 1, 10000)):
        out[name] = generate_fixed_normal_sequence(seed, n)
        print(name, out[name][:8], float(np.linalg.norm(out[name].astype(np.float64))))
    np.savez_compressed(os.path.join(here, "marks.npz"), **out)
