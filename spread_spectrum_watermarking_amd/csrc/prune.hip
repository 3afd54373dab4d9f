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
// The flags become one bit each (LDS, W / 64 words: coalesced loads + ballots); then wave c walks the words for class c with
// its running count in a register -- no barrier inside the walk.  (r5: the walk was block-wide, three barriers per 1024
// frequencies and class: 42 us for W = 3840 in front of a single frame's 0.2 ms of pruned row GEMMs.)
__global__ __launch_bounds__(1024) void prune_build_kernel(const uint32_t* __restrict__ flag, PrunePlan plan,
                                                          uint32_t* __restrict__ rows, uint32_t* __restrict__ pos,
                                                          uint32_t* __restrict__ info) {
    extern __shared__ unsigned long long bits[];                       // (W + 63) / 64 words
    const unsigned t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned nwords = (plan.W + 63) / 64;
    for (unsigned v = t; v < plan.W; v += 1024) pos[v] = PRUNE_NONE;
    for (unsigned j = t; j < plan.cap_total; j += 1024) rows[j] = PRUNE_NONE;
    for (unsigned w = wv; w < nwords; w += 16) {
        const unsigned v = 64 * w + lane;
        const unsigned long long bal = __ballot(v < plan.W && flag[v] != 0);
        if (lane == 0) bits[w] = bal;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    for (unsigned c = wv; c < plan.n_classes; c += 16) {               // wave-uniform
        const PruneClass pc = plan.c[c];
        unsigned count = 0;
        for (unsigned w = 0; w < nwords; ++w) {                        // members of the class in ascending order of v
            const unsigned v = 64 * w + lane;
            const unsigned r = v % pc.mod;
            const bool neg = pc.rem2 != PRUNE_NO_REM && r == pc.rem2;
            const bool on = ((bits[w] >> lane) & 1ull) && (r == pc.rem || neg);
            const unsigned long long bal = __ballot(on);
            if (on) {
                const unsigned j = count + (unsigned)__popcll(bal & below);
                if (j < pc.cap) {
                    rows[pc.off + j] = ((v + pc.radd) / pc.mod) | (neg ? PRUNE_NEG : 0u);
                    pos[v] = pc.off + j;
                }
            }
            count += (unsigned)__popcll(bal);
        }
        if (lane == 0) {
            info[1 + c] = count;
            if (count > pc.cap) atomicOr(&info[0], 1u);
        }
    }
}

// Gathered half bases of the classes of one derived-frame row pass, one launch: job j's dst [Kp / KB][cap][KB] <- rows
// rows[i] of src [Kp / KB][src_rows][KB], zero rows where rows[i] == PRUNE_NONE.  One thread = one 16-byte piece; k-block
// pieces are 64 bytes in both precisions.  (r5: one launch per class and basis was sixteen launches of 4.7 us each in front
// of a single 4K frame's 0.2 ms of row GEMMs.)
__global__ __launch_bounds__(256) void prune_gather_basis_kernel(const PruneGatherJobs jobs) {
    unsigned jn = 0;
    while (jn + 1 < jobs.n && blockIdx.x >= jobs.j[jn + 1].first_block) ++jn;                // block-uniform
    const PruneGatherJob& job = jobs.j[jn];
    const size_t i = (blockIdx.x - job.first_block) * (size_t)blockDim.x + threadIdx.x;
    const unsigned cap_out = job.frag ? (job.cap + 15u) & ~15u : job.cap;      // fragment order: whole tiles of 16 rows, zero rows behind the class
    const size_t total = (size_t)job.kblocks * cap_out * 4;
    if (i >= total) return;
    const unsigned piece = (unsigned)(i & 3);
    const unsigned r_out = (unsigned)((i >> 2) % cap_out);
    const unsigned kb = (unsigned)((i >> 2) / cap_out);
    const uint32_t rf = r_out < job.cap ? job.rows[r_out] : PRUNE_NONE;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (rf != PRUNE_NONE) {
        const uint32_t r = rf & ~PRUNE_NEG;
        v = *reinterpret_cast<const u32x4*>(job.src + ((size_t)kb * job.src_rows + r) * 64 + piece * 16);
        if (job.negate && (rf & PRUNE_NEG)) { v[1] ^= 0x80000000u; v[3] ^= 0x80000000u; }      // two doubles: flip the sign bits
    }
    if (!job.frag) {
        *reinterpret_cast<u32x4*>(job.dst + ((size_t)kb * job.cap + r_out) * 64 + piece * 16) = v;
        return;
    }
    // r5, the fused derived pass (dct_pair_derived.hip): MFMA B-fragment order -- per tile of 16 rows [k / 4][k % 4][row % 16] doubles,
    // so that a wave reads a fragment (16 rows x 4 k) as 512 contiguous bytes.  f64 only (8 k per k-block).
    const unsigned tile = r_out >> 4, li = r_out & 15u, k = 8 * kb + 2 * piece;      // this piece: k and k + 1
    u32x2* d = reinterpret_cast<u32x2*>(job.dst) + ((size_t)tile * (2 * job.kblocks) + (k >> 2)) * 64 + li;
    d[(k & 3u) * 16] = (u32x2){v[0], v[1]};
    d[((k + 1) & 3u) * 16] = (u32x2){v[2], v[3]};
}

int launch_prune_build(hipStream_t st, const uint32_t* idx, size_t n_frames, size_t k, const PrunePlan& plan,
                       uint32_t* flag, uint32_t* rows, uint32_t* pos, uint32_t* info) {
    SSW_HIP_CHECK(hipMemsetAsync(flag, 0, (size_t)plan.W * sizeof(uint32_t), st));
    SSW_HIP_CHECK(hipMemsetAsync(info, 0, SSW_PRUNE_INFO * sizeof(uint32_t), st));
    const size_t count = n_frames * k;
    if (count) prune_mark_kernel<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(idx, count, plan.W, flag);
    prune_build_kernel<<<1, 1024, ((size_t)plan.W + 63) / 64 * sizeof(unsigned long long), st>>>(flag, plan, rows, pos, info);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prune_gather_bases(hipStream_t st, PruneGatherJobs jobs) {
    unsigned blocks = 0;
    for (unsigned j = 0; j < jobs.n; ++j) {
        jobs.j[j].first_block = blocks;
        const unsigned cap_out = jobs.j[j].frag ? (jobs.j[j].cap + 15u) & ~15u : jobs.j[j].cap;
        blocks += (unsigned)(((size_t)jobs.j[j].kblocks * cap_out * 4 + 255) / 256);
    }
    if (blocks == 0) return SSW_OK;
    prune_gather_basis_kernel<<<blocks, 256, 0, st>>>(jobs);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
