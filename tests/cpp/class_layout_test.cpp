// Host-side check of the class-major column orders (csrc/dct_pair_common.hpp): position maps are bijections of
// [0, n), natural() inverts the position a launch class writes, and every frequency lands in the class whose launch
// produces it.  No device code runs: the header's maps are __host__ __device__.
#include <cstdio>
#include <vector>
#include "../../spread_spectrum_watermarking_amd/csrc/dct_pair_common.hpp"

using namespace ssw;

static int fail(const char* what, unsigned n, unsigned at) { std::printf("FAIL %s n=%u at %u\n", what, n, at); return 1; }

int main() {
    for (unsigned n : {256u, 320u, 1024u, 1920u, 3840u, 7680u})
    for (unsigned t : {n, class_tile(n)})
    for (bool efold : {false, true}) {
        if (n % t != 0) return fail("tile does not divide the line", n, t);
        const ForwardClassLayout fl{n, t, efold};
        std::vector<int> seen(n, 0);
        for (unsigned p = 0; p < n; ++p) {
            const unsigned u = fl.natural(p);
            if (u >= n || seen[u]++) return fail("forward natural() not a bijection", n, p);
            if (u / t != p / t) return fail("forward natural() leaves its tile", n, p);
        }
        // the launches' output maps (dct_pair_f64.hip, pair_class_args / fpos1 / fpos2): entry = pair index [- 1 for the
        // "-" outputs of E], column = base + (entry / group) * tile + entry % group
        for (unsigned i = 0; i < n / 8; ++i) {
            if (fl.natural(fl.pos(ForwardClassLayout::R1, i)) != 8 * i) return fail("R1", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::R2, i)) != 8 * i + 4) return fail("R2", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::OP, i)) != 8 * i + 5) return fail("O+", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::OM, i)) != 8 * i + 3) return fail("O-", n, i);
        }
        for (unsigned i = 0; i < n / 8 && !efold; ++i) {      // class E whole: pair i -> 8 i +/- 1
            if (fl.natural(fl.pos(ForwardClassLayout::EP, i)) != 8 * i + 1) return fail("E+", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::EM, (i + 1) - 1)) != 8 * (i + 1) - 1) return fail("E-", n, i);
        }
        for (unsigned i = 0; i < n / 16; ++i) {
            if (fl.natural(fl.pos(ForwardClassLayout::E2P, i)) != 2 * (8 * i + 1)) return fail("E'+", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::E2M, (i + 1) - 1)) != 2 * (8 * (i + 1) - 1)) return fail("E'-", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::O2P, i)) != 2 * (8 * i + 5)) return fail("O'+", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::O2M, i)) != 2 * (8 * i + 3)) return fail("O'-", n, i);
            // class E folded once more: pair i of the even launch -> 16 i +/- 1 (the "-" output of pair i is entry i - 1),
            // pair i of the odd launch -> 16 i + 9 and 16 i + 7
            if (!efold) continue;
            if (fl.natural(fl.pos(ForwardClassLayout::EEP, i)) != 16 * i + 1) return fail("Ee+", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::EEM, (i + 1) - 1)) != 16 * (i + 1) - 1) return fail("Ee-", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::EOP, i)) != 16 * i + 9) return fail("Eo+", n, i);
            if (fl.natural(fl.pos(ForwardClassLayout::EOM, i)) != 16 * i + 7) return fail("Eo-", n, i);
        }
        // the shift form of pos() the GEMM epilogue uses when the tile is a power of two
        if (t != n)
            for (int c = 0; c < ForwardClassLayout::NCLASS; ++c) {
                if (!fl.has(c)) continue;
                const unsigned g = fl.group(c);
                unsigned gsh = 0;
                while ((1u << gsh) < g) ++gsh;
                if ((1u << gsh) != g) return fail("group not a power of two", n, g);
                for (unsigned e = 0; e < n / ForwardClassLayout::mod(c); ++e)
                    if (fl.base(c) + (e >> gsh) * t + (e & (g - 1)) != fl.pos(c, e)) return fail("shift form of pos()", n, e);
            }
        std::vector<int> seen2(n, 0);
        for (unsigned m = 0; m < n; ++m) {
            const unsigned p = inverse_class_pos(m, n, t);
            if (p >= n || seen2[p]++) return fail("inverse_class_pos not a bijection", n, m);
            if (p / t != m / t) return fail("inverse_class_pos leaves its tile", n, m);
            if (inverse_class_natural(p, n, t) != m) return fail("inverse_class_natural", n, m);
        }
    }
    std::printf("ok\n");
    return 0;
}
