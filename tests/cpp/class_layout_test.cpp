// Host-side check of the class-major column orders (csrc/dct_pair_common.hpp): position maps are bijections of
// [0, n), natural() inverts the position a launch class writes, and every frequency lands in the class whose launch
// produces it.  No device code runs: the header's maps are __host__ __device__.
#include <cstdio>
#include <vector>
#include "../../spread_spectrum_watermarking_amd/csrc/dct_pair_common.hpp"
#include "../../spread_spectrum_watermarking_amd/csrc/dct_pair_colops.hpp"

using namespace ssw;

static int fail(const char* what, unsigned n, unsigned at) { std::printf("FAIL %s n=%u at %u\n", what, n, at); return 1; }

int main() {
    for (unsigned n : {256u, 320u, 1024u, 1920u, 3840u, 7680u})
    for (unsigned t : {n, class_tile(n)})
    for (bool level2 : {false, true}) {
        if (n % t != 0) return fail("tile does not divide the line", n, t);
        if (level2 && n % 16 != 0) continue;
        typedef ForwardClassLayout F;
        const F fl{n, t, level2};
        std::vector<int> seen(n, 0);
        for (unsigned p = 0; p < n; ++p) {
            const unsigned u = fl.natural(p);
            if (u >= n || seen[u]++) return fail("forward natural() not a bijection", n, p);
            if (u / t != p / t) return fail("forward natural() leaves its tile", n, p);
        }
        // the launches' output maps (dct_pair_f64.hip, pair_class_args / fpos1 / fpos2): entry = pair index [- 1 for the
        // "-" outputs of launches of class E's shape], column = base + (entry / group) * tile + entry % group
        if (!level2) {
            for (unsigned i = 0; i < n / 8; ++i) {
                if (fl.natural(fl.pos(F::R1, i)) != 8 * i) return fail("R1", n, i);
                if (fl.natural(fl.pos(F::R2, i)) != 8 * i + 4) return fail("R2", n, i);
                if (fl.natural(fl.pos(F::OP, i)) != 8 * i + 5) return fail("O+", n, i);
                if (fl.natural(fl.pos(F::OM, i)) != 8 * i + 3) return fail("O-", n, i);
                if (fl.natural(fl.pos(F::EP, i)) != 8 * i + 1) return fail("E+", n, i);
                if (fl.natural(fl.pos(F::EM, (i + 1) - 1)) != 8 * (i + 1) - 1) return fail("E-", n, i);
            }
            for (unsigned i = 0; i < n / 16; ++i) {
                if (fl.natural(fl.pos(F::E2P, i)) != 2 * (8 * i + 1)) return fail("E'+", n, i);
                if (fl.natural(fl.pos(F::E2M, (i + 1) - 1)) != 2 * (8 * (i + 1) - 1)) return fail("E'-", n, i);
                if (fl.natural(fl.pos(F::O2P, i)) != 2 * (8 * i + 5)) return fail("O'+", n, i);
                if (fl.natural(fl.pos(F::O2M, i)) != 2 * (8 * i + 3)) return fail("O'-", n, i);
            }
        } else {
            // level 2: pair i of a launch of class E's shape with residue r -> 16 i + r and (entry i - 1 of the next class)
            // 16 i - r; class O's shape (E odd, O'): 16 i + r1 and 16 i + r2; R1 folded: 16 i and 16 i + 8
            const int eshape[5][2] = {{F::EEP, 1}, {F::O5, 5}, {F::O3, 3}, {F::R2A, 4}, {F::F_E2P, 2}};
            for (unsigned i = 0; i < n / 16; ++i) {
                for (auto& e : eshape) {
                    if (fl.natural(fl.pos(e[0], i)) != 16 * i + e[1]) return fail("level 2, E shape +", n, i);
                    if (fl.natural(fl.pos(e[0] + 1, (i + 1) - 1)) != 16 * (i + 1) - e[1]) return fail("level 2, E shape -", n, i);
                }
                if (fl.natural(fl.pos(F::EOP, i)) != 16 * i + 9) return fail("Eo+", n, i);
                if (fl.natural(fl.pos(F::EOM, i)) != 16 * i + 7) return fail("Eo-", n, i);
                if (fl.natural(fl.pos(F::F_O2P, i)) != 16 * i + 10) return fail("O'+", n, i);
                if (fl.natural(fl.pos(F::F_O2M, i)) != 16 * i + 6) return fail("O'-", n, i);
                if (fl.natural(fl.pos(F::R1A, i)) != 16 * i) return fail("R1+", n, i);
                if (fl.natural(fl.pos(F::R1B, i)) != 16 * i + 8) return fail("R1-", n, i);
            }
        }
        // the shift form of pos() the GEMM epilogue uses when the tile is a power of two
        if (t != n)
            for (int c = 0; c < fl.classes(); ++c) {
                const unsigned g = fl.group(c);
                unsigned gsh = 0;
                while ((1u << gsh) < g) ++gsh;
                if ((1u << gsh) != g) return fail("group not a power of two", n, g);
                for (unsigned e = 0; e < n / fl.mod(c); ++e)
                    if (fl.base(c) + (e >> gsh) * t + (e & (g - 1)) != fl.pos(c, e)) return fail("shift form of pos()", n, e);
            }
        // inverse row pass: by residue mod 4 (level 1) or mod 8 (level 2; t % 8 == 0)
        for (bool l2 : {false, true}) {
            if (l2 != level2 || (l2 && t % 8 != 0)) continue;
            std::vector<int> seen2(n, 0);
            for (unsigned m = 0; m < n; ++m) {
                const unsigned p = inverse_class_pos(m, n, t, l2);
                if (p >= n || seen2[p]++) return fail("inverse_class_pos not a bijection", n, m);
                if (p / t != m / t) return fail("inverse_class_pos leaves its tile", n, m);
                if (inverse_class_natural(p, n, t, l2) != m) return fail("inverse_class_natural", n, m);
            }
            if (!l2) continue;
            // the four launches of the odd part own {0,7} {4,3} {2,5} {1,6} mod 8: each pair of residues (and the mirrors
            // n - 1 - m, the same pair) occupies two neighbouring runs of t/8 positions, m / 8 ascending inside a run
            const unsigned pairs[4][2] = {{0, 7}, {4, 3}, {2, 5}, {1, 6}};
            for (unsigned c = 0; c < 4; ++c)
                for (unsigned i = 0; i < t / 8; ++i)
                    for (unsigned h = 0; h < 2; ++h) {
                        const unsigned m = 8 * i + pairs[c][h];
                        if (inverse_class_pos(m, n, t, true) != (2 * c + h) * (t / 8) + i) return fail("level-2 inverse run", n, m);
                        if ((n - 1 - m) % 8 != pairs[c][1 - h]) return fail("mirror leaves its class", n, m);
                    }
        }
    }
    // r5, the fused forward transform (dct_pair_colops.hpp): the closed form of the 128-frequency level-2 tile equals the
    // layout's pos(), and (unit, line) <-> frame row is a bijection whose line v mirrors line 15 - v
    {
        const ForwardClassLayout fl{3840, 128, true};
        for (unsigned j = 0; j < 3840; ++j)
            if (fl.natural((j / 128) * 128 + fwd_cm128_pos(j % 128)) != j) return fail("fwd_cm128_pos", 3840, j);
        for (unsigned H : {144u, 720u, 2160u, 4320u}) {
            std::vector<int> seen(H, 0);
            for (unsigned e = 0; e < H / 16; ++e)
                for (unsigned v = 0; v < 16; ++v) {
                    const unsigned y = col_unit_row(e, v, H);
                    unsigned e2 = 0, v2 = 0;
                    if (y >= H || seen[y]++) return fail("col_unit_row not a bijection", H, y);
                    col_unit_of_row(y, H, e2, v2);
                    if (e2 != e || v2 != v) return fail("col_unit_of_row", H, y);
                    if (col_unit_row(e, 15 - v, H) != H - 1 - y) return fail("line 15 - v is not the mirror row", H, y);
                }
        }
    }
    std::printf("ok\n");
    return 0;
}
