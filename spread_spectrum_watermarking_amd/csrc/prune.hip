// Pruned transform of the derived frame (batch extract path).
//
// Reader::extract (/root/reference/src/algorithm.rs:556-561) reads the derived plane at the k indices of
// the base plane's ordering and nowhere else.  The batch path therefore transforms the derived frames
// only where a chunk needs them: the row pass produces the frequency COLUMNS v = index % W that occur in
// the chunk's index lists (a few percent of W for natural spectra), the column pass runs on that compact
// plane.  Both passes use the same operand planes, the same half-basis rows (gathered) and the same GEMM
// kernel as the full transform, so every value that extract reads is bit-identical to the full
// transform's: an MFMA output element depends on its own operand line, its own basis row and the k order
// only.  Handles (Reader::derived exposes coefficients()) keep the full transform.
//
// Everything is decided on the device (no host round trip inside a batch call): capacities are static,
// a chunk whose set of columns does not fit raises a flag and the caller redoes it with the full path.
//
// Frequency classes.  With L folding levels on the row axis a frequency v is produced from the operand
// plane of its class against one row of that class's half basis:
//   L = 1:  v odd -> D x odd(W)        v even -> S x even(W)
//   L = 2:  v odd -> D x odd(W)        v = 2 mod 4 -> SD x odd(W/2)      v = 0 mod 4 -> SS x even(W/2)
//   L = 3:  v odd -> x- x odd(W)       v = 2 mod 4 -> S- x odd(W/2)      v = 4 mod 8 -> SS- x odd(W/4)
//                                                                        v = 0 mod 8 -> SSS x even(W/4)
// i.e. class = (mod, rem) and the basis row is v / mod.  Compact column of v = class offset + rank of v
// inside its class (ascending v).
// With the split odd half (f64, dct_pair_prep.hip) an odd class is two: class E holds v = 8i +/- 1 (row i of the
// quarter-length cosine and sine bases; for 8i - 1 the output is the cosine part MINUS the sine part, so the gathered
// sine row is negated -- exact), class O holds v = 8i + 5 and 8i + 3; in deep transforms the frequencies 2 mod 4 split
// the same way one level down (v = 2 (8i +/- 1), 2 (8i + 5 | 3)).  PruneClass carries the second remainder and the row
// offset: row = (v + radd) / mod.
#include "dct_pair_common.hpp"

namespace ssw {

constexpr uint32_t PRUNE_NONE = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void prune_mark_kernel(const uint32_t* __restrict__ idx, size_t count, unsigned W,
                                                        uint32_t* __restrict__ flag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < count) flag[idx[i] % W] = 1u;
}

// One block: per class an ordered compaction of the flagged frequencies.
//   rows[off + j] = basis row of the j-th member (PRUNE_NONE beyond the count)
//   pos[v]        = compact column of v (PRUNE_NONE when v is not needed or did not fit)
//   info[0] |= 1 when a class overflows its capacity; info[1 + c] = members of class c
__global__ __launch_bounds__(1024) void prune_build_kernel(const uint32_t* __restrict__ flag, PrunePlan plan,
                                                          uint32_t* __restrict__ rows, uint32_t* __restrict__ pos,
                                                          uint32_t* __restrict__ info) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t base_s;
    const unsigned t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (unsigned v = t; v < plan.W; v += 1024) pos[v] = PRUNE_NONE;
    for (unsigned j = t; j < plan.cap_total; j += 1024) rows[j] = PRUNE_NONE;
    __syncthreads();
    bool overflow = false;
    for (unsigned c = 0; c < plan.n_classes; ++c) {
        const PruneClass pc = plan.c[c];
        if (t == 0) base_s = 0;
        __syncthreads();
        // members of the class in ascending order of v
        for (unsigned v0 = 0; v0 < plan.W; v0 += 1024) {
            const unsigned v = v0 + t;
            const unsigned r = v % pc.mod;
            const bool neg = pc.rem2 != PRUNE_NO_REM && r == pc.rem2;
            const bool on = v < plan.W && (r == pc.rem || neg) && flag[v] != 0;
            const unsigned long long bal = __ballot(on);
            const unsigned before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_sum[wv] = __popcll(bal);
            __syncthreads();
            unsigned prefix = base_s;
            for (unsigned q = 0; q < wv; ++q) prefix += wave_sum[q];
            if (on) {
                const unsigned j = prefix + before;
                if (j < pc.cap) {
                    rows[pc.off + j] = ((v + pc.radd) / pc.mod) | (neg ? PRUNE_NEG : 0u);
                    pos[v] = pc.off + j;
                }
            }
            __syncthreads();
            if (t == 0) {
                unsigned tot = 0;
                for (unsigned q = 0; q < 16; ++q) tot += wave_sum[q];
                base_s += tot;
            }
            __syncthreads();
        }
        if (t == 0) {
            info[1 + c] = base_s;
            if (base_s > pc.cap) overflow = true;
        }
        __syncthreads();
    }
    if (t == 0 && overflow) info[0] = 1u;
}

// Gathered half basis of one class: dst [Kp / KB][cap][KB] <- rows rows[j] of src [Kp / KB][src_rows][KB], zero
// rows where rows[j] == PRUNE_NONE.  One thread = one 16-byte piece; k-block pieces are 64 bytes in both precisions.
__global__ __launch_bounds__(256) void prune_gather_basis_kernel(const uint32_t* __restrict__ rows, unsigned cap,
                                                                const char* __restrict__ src, unsigned src_rows,
                                                                unsigned kblocks, char* __restrict__ dst, bool negate) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t total = (size_t)kblocks * cap * 4;
    if (i >= total) return;
    const unsigned piece = (unsigned)(i & 3);
    const unsigned j = (unsigned)((i >> 2) % cap);
    const unsigned kb = (unsigned)((i >> 2) / cap);
    const uint32_t rf = rows[j];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (rf != PRUNE_NONE) {
        const uint32_t r = rf & ~PRUNE_NEG;
        v = *reinterpret_cast<const u32x4*>(src + ((size_t)kb * src_rows + r) * 64 + piece * 16);
        if (negate && (rf & PRUNE_NEG)) { v[1] ^= 0x80000000u; v[3] ^= 0x80000000u; }      // two doubles: flip the sign bits
    }
    *reinterpret_cast<u32x4*>(dst + ((size_t)kb * cap + j) * 64 + piece * 16) = v;
}

int launch_prune_build(hipStream_t st, const uint32_t* idx, size_t n_frames, size_t k, const PrunePlan& plan,
                       uint32_t* flag, uint32_t* rows, uint32_t* pos, uint32_t* info) {
    SSW_HIP_CHECK(hipMemsetAsync(flag, 0, (size_t)plan.W * sizeof(uint32_t), st));
    SSW_HIP_CHECK(hipMemsetAsync(info, 0, SSW_PRUNE_INFO * sizeof(uint32_t), st));
    const size_t count = n_frames * k;
    if (count) prune_mark_kernel<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(idx, count, plan.W, flag);
    prune_build_kernel<<<1, 1024, 0, st>>>(flag, plan, rows, pos, info);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prune_gather_basis(hipStream_t st, const uint32_t* rows, unsigned cap, const void* src, size_t src_rows,
                              size_t kblocks, void* dst, bool negate_flagged_f64) {
    const size_t total = kblocks * cap * 4;
    if (total == 0) return SSW_OK;
    prune_gather_basis_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(rows, cap, (const char*)src, (unsigned)src_rows,
                                                                                 (unsigned)kblocks, (char*)dst, negate_flagged_f64);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
