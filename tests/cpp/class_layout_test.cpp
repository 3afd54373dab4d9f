// Host-side check of the class-major column orders (csrc/dct_pair_common.hpp): position maps are bijections of
// [0, n), natural() inverts the position a launch class writes, and every frequency lands in the class whose launch
// produces it.  No device code runs: the header's maps are __host__ __device__.
#include <cstdio>
#include <vector>
#include "../../spread_spectrum_watermarking_amd/csrc/dct_pair_common.hpp"

using namespace ssw;

static int fail(const char* what, unsigned n, unsigned at) { std::printf("FAIL %s n=%u at %u\n", what, n, at); return 1; }

int main() {
    for (unsigned n : {256u, 320u, 1024u, 1920u, 3840u, 7680u}) {
        const ForwardClassLayout fl{n};
        std::vector<int> seen(n, 0);
        for (unsigned p = 0; p < n; ++p) {
            const unsigned u = fl.natural(p);
            if (u >= n || seen[u]++) return fail("forward natural() not a bijection", n, p);
        }
        // the launches' output maps (dct_pair_f64.hip, pair_class_args): class base + pair index [- 1 for the "-" outputs of E]
        for (unsigned i = 0; i < n / 8; ++i) {
            if (fl.natural(fl.base(ForwardClassLayout::R1) + i) != 8 * i) return fail("R1", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::R2) + i) != 8 * i + 4) return fail("R2", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::EP) + i) != 8 * i + 1) return fail("E+", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::EM) - 1 + (i + 1)) != 8 * (i + 1) - 1) return fail("E-", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::OP) + i) != 8 * i + 5) return fail("O+", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::OM) + i) != 8 * i + 3) return fail("O-", n, i);
        }
        for (unsigned i = 0; i < n / 16; ++i) {
            if (fl.natural(fl.base(ForwardClassLayout::E2P) + i) != 2 * (8 * i + 1)) return fail("E'+", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::E2M) - 1 + (i + 1)) != 2 * (8 * (i + 1) - 1)) return fail("E'-", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::O2P) + i) != 2 * (8 * i + 5)) return fail("O'+", n, i);
            if (fl.natural(fl.base(ForwardClassLayout::O2M) + i) != 2 * (8 * i + 3)) return fail("O'-", n, i);
        }
        std::vector<int> seen2(n, 0);
        for (unsigned m = 0; m < n; ++m) {
            const unsigned p = inverse_class_pos(m, n);
            if (p >= n || seen2[p]++) return fail("inverse_class_pos not a bijection", n, m);
            if (inverse_class_natural(p, n) != m) return fail("inverse_class_natural", n, m);
        }
    }
    std::printf("ok\n");
    return 0;
}
