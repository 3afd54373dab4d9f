/*
 * ssw.h -- C ABI of the MI355X-native spread-spectrum watermarking hot path.
 *
 * Drop-in boundary for the `yiq -> dct2d -> ordering/top-k -> embed | extract ->
 * similarity` path of iwanders/spread_spectrum_watermarking.  The reference has
 * no FFI of its own (it is a pure-Rust crate); each entry point below replaces
 * the *body* of the reference function it cites (file:line relative to the
 * reference tree), and INTEGRATION.md shows the Rust `-sys` binding a maintainer
 * would add.  Plain pointers and sizes only; no exceptions or aborts cross this
 * boundary -- every call returns an ssw_status.
 *
 * Threading: a context is bound to one GPU and must be used by one host thread
 * at a time (like the reference's `!Send` Writer/Reader).  Multi-GPU = one
 * context (and one process or thread) per GPU; frames are independent so there
 * is no collective anywhere.
 *
 * Memory: "host" pointers are ordinary CPU memory; "dev" pointers are HIP device
 * memory on the context's GPU (e.g. from ssw_dev_alloc or a torch tensor's
 * data_ptr()).  Handles own their device planes; the caller owns all I/O buffers.
 */
#ifndef SSW_H
#define SSW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ssw_status {
    SSW_OK = 0,
    SSW_ERR_BAD_ARG = 1,          /* null pointer, unknown enum value                      */
    SSW_ERR_BAD_DIMS = 2,         /* w*h == 0 or data length != w*h   (src/dct2d.rs:90)     */
    SSW_ERR_LENGTH_MISMATCH = 3,  /* derived vs base length (src/algorithm.rs:550-552),
                                     similarity lengths (:697-700)                          */
    SSW_ERR_K_TOO_LARGE = 4,      /* extraction length >= coefficients (:553-555)           */
    SSW_ERR_NOT_BASE = 5,         /* derived reader used as base (:507, :530 unwrap)        */
    SSW_ERR_UNSUPPORTED = 6,      /* Custom(closure) variants cannot cross to the device    */
    SSW_ERR_CONSUMED = 7,         /* writer already consumed by mark()/result() (:355,:361) */
    SSW_ERR_HIP = 8,              /* HIP runtime error; see ssw_last_error()                */
    SSW_ERR_NO_DEVICE = 9,        /* no usable gfx950 device: there is NO CPU fallback      */
    SSW_ERR_OUT_OF_MEMORY = 10
} ssw_status;

/* OrderingMethod (src/algorithm.rs:143-152) */
typedef enum ssw_ordering {
    SSW_ORDER_ENERGY = 0,
    SSW_ORDER_ENERGY_ORTHOGONAL = 1,
    SSW_ORDER_LEGACY = 2,
    SSW_ORDER_CUSTOM = 3          /* -> SSW_ERR_UNSUPPORTED */
} ssw_ordering;

/* Insertion / Extraction (src/algorithm.rs:68-77, :115-124) */
typedef enum ssw_method {
    SSW_OPTION1 = 1,
    SSW_OPTION2 = 2,
    SSW_OPTION3 = 3,
    SSW_METHOD_CUSTOM = 4         /* -> SSW_ERR_UNSUPPORTED */
} ssw_method;

/* dct2d::Type (src/dct2d.rs:71-79) */
typedef enum ssw_dct_type {
    SSW_DCT2 = 0,
    SSW_DCT2_ORTHOGONAL = 1,
    SSW_DCT3 = 2
} ssw_dct_type;

/* Arithmetic of the DCT basis GEMMs (no counterpart in the reference, which
   delegates to rustdct's f32 FFT kernels). */
typedef enum ssw_precision {
    SSW_PRECISION_F32 = 0,        /* v_mfma_f32_32x32x2_f32: f32 fma chains.  NOT a parity path: extracted
                                     marks within 1e-5 in the median, ~1e-3 worst case; since r3 (split
                                     odd halves in f64) it is also the slower precision at 4K */
    SSW_PRECISION_F64 = 1         /* DEFAULT.  v_mfma_f64_16x16x4_f64, f64 basis, result rounded
                                     once to f32: the correctly rounded ("canonical") transform,
                                     bit-identical extraction against the CPU restatement   */
} ssw_precision;

/* WriteConfig / ReadConfig (src/algorithm.rs:99-140).  ssw_config_default() is
   Option2(0.1) + Energy like the reference's Default impls (:104-111, :132-139), with the
   canonical (f64) transform precision. */
typedef struct ssw_config {
    int32_t ordering;             /* ssw_ordering  */
    int32_t method;               /* ssw_method    */
    float alpha;
    int32_t precision;            /* ssw_precision */
} ssw_config;

typedef struct ssw_ctx ssw_ctx;
typedef struct ssw_writer ssw_writer;
typedef struct ssw_reader ssw_reader;

/* ---- library / context ---------------------------------------------------- */
const char* ssw_version(void);
/* 1 when the loaded library is the diagnostic build (`make ALL_STRATEGIES=1`: lib/libssw_hip_all.so) that also carries the
   superseded strategies of the transform (src/dct2d.rs:83-219) -- levels 1 / 2 of ssw_ctx_set_dct_folding (in-kernel folding)
   and the f32 twin of the operand-ready GEMMs; 0 for the default library, where those requests run the dense kernels
   (same results to their precision's bars, slower). */
int ssw_build_all_strategies(void);
/* Which strategy of the 2-D transform (src/dct2d.rs:83-219) a batch of n_frames frames of w x h takes in the canonical (f64)
   precision under the context's current settings, as flags -- introspection for bench.py, DESIGN.md and the tests; every
   strategy computes the same values:
     PAIR_F64     operand-ready f64 GEMMs (otherwise the dense kernels)
     ROWS_DEEP / COLS_DEEP       one pre-pass per pass writes the operands of all launches (split odd halves)
     ROWS_LEVEL2 / COLS_LEVEL2   every launch of the pass sums len/16 terms (eight launches per pass)
     CLASS_MAJOR  the plane (or operand lines) between the passes in class-major order inside tiles of 128 columns
     FUSED_COLS   r5: no f32 plane between the passes -- the row launches' epilogue writes the column operands */
enum { SSW_PLAN_PAIR_F64 = 1, SSW_PLAN_ROWS_DEEP = 2, SSW_PLAN_COLS_DEEP = 4, SSW_PLAN_ROWS_LEVEL2 = 8, SSW_PLAN_COLS_LEVEL2 = 16,
       SSW_PLAN_CLASS_MAJOR = 32, SSW_PLAN_FUSED_COLS = 64 };
int ssw_ctx_transform_plan(ssw_ctx* ctx, size_t n_frames, size_t w, size_t h, int dct_type, uint32_t* flags);
const char* ssw_status_string(int status);
/* Text of the last failing HIP call on this thread (empty string if none). */
const char* ssw_last_error(void);
void ssw_config_default(ssw_config* cfg);

/* One context per GPU.  Fails with SSW_ERR_NO_DEVICE when no HIP device is
   present -- the library never computes on the CPU. */
int ssw_ctx_create(int device_id, ssw_ctx** out);
int ssw_ctx_destroy(ssw_ctx* ctx);
int ssw_ctx_synchronize(ssw_ctx* ctx);
/* Stream contract.  Every entry point that takes "dev" pointers (ssw_rgb_to_yiq ... ssw_batch_*,
   ssw_resize_rgb8, ssw_synth_frames) only ENQUEUES work on the context's stream and returns; by default
   that is a private hipStreamNonBlocking stream, which is ordered neither against the null stream nor
   against e.g. torch's current stream.  The caller must therefore
     (1) make sure its inputs are complete before the call -- synchronise its producer, or hand the
         library an event to wait for (ssw_ctx_wait_event), or put the library on its own stream
         (ssw_ctx_set_stream);
     (2) treat outputs as valid only after ssw_ctx_synchronize(), or after an event recorded with
         ssw_ctx_record_event() has completed, or in later work on the stream given to ssw_ctx_set_stream.
   Entry points that take HOST buffers (the Writer / Reader / Tester handles, ssw_copy_*) return when the
   caller's buffers are theirs again: inputs have been read (staged or DMA'd), outputs are complete.  The
   device work a handle constructor started may still be running -- it is ordered before every later call
   on the same context, and a failure of it is reported by the next call that waits for the device.
   Calls change the calling thread's current HIP device only for their duration.
   One exception to "only enqueues": ssw_batch_extract* with pruning on (the default) waits once for the
   device at its end (see ssw_ctx_set_prune); while the context's stream is being captured into a graph it
   takes the full transform instead and does not wait. */
/* hipStream_t the context currently enqueues on, as an opaque pointer. */
void* ssw_ctx_stream(ssw_ctx* ctx);
/* Enqueue on the caller's hipStream_t from now on (NULL: back to the private stream).  Synchronises the
   stream used so far.  The stream must belong to the context's GPU and outlive its use here. */
int ssw_ctx_set_stream(ssw_ctx* ctx, void* hip_stream);
/* hipStreamWaitEvent / hipEventRecord on the context's stream with the caller's hipEvent_t: chain the
   library behind a producer, or a consumer behind the library, without a host synchronisation. */
int ssw_ctx_wait_event(ssw_ctx* ctx, void* hip_event);
int ssw_ctx_record_event(ssw_ctx* ctx, void* hip_event);
/* Frames processed per internal pass of the batch entry points (bounds the workspace: up to 44 bytes per
   pixel of a pass in the default GEMM strategy, per lane -- four f32 planes, row and column operand planes side
   by side since the fused forward transform, the inverse's A1 / T2 / E; see ssw_ctx_set_overlap).  0 = automatic, the
   default: about 2^30 pixels per pass (128 4K frames, 514 full-HD frames, 32 8K frames; up to 47 GB of
   workspace per lane): sized for the 288 GB of an MI355X, where longer GEMM launches amortise their tails
   (2^28 pixels cost 2.8 % at 4K, 1.8 % at full HD).  The automatic size never asks for more than half of what
   the device can give at the time of the call (free memory + what the context already holds): a smaller device
   or a co-tenant gets smaller passes, not SSW_ERR_OUT_OF_MEMORY. */
int ssw_ctx_set_chunk_frames(ssw_ctx* ctx, size_t frames);
/* Frames per pass a batch call over n_frames frames of w x h would use with the current setting. */
size_t ssw_ctx_pass_frames(ssw_ctx* ctx, size_t n_frames, size_t w, size_t h);

/* Batch pipelines (ssw_batch_*): two chunks in flight, each with its own workspace -- the HBM-bound stages
   of one (operand pre-passes, selection, colour conversion) run on a second internal stream while the
   basis GEMMs of the other run on the context's stream; the context's stream is ordered after all of it
   when the call returns.  Default on, used from two passes per call upwards (2 x 36 B/px of workspace
   then); 0 = one chunk at a time on one stream
   (what the per-kernel timings of bench.py's roofline leg use).  Results are bit-identical either way. */
int ssw_ctx_set_overlap(ssw_ctx* ctx, int enable);
/* ssw_batch_extract*: transform the derived frames only where Reader::extract reads them
   (src/algorithm.rs:556-561: k coefficients) -- the row pass for the frequency columns that occur in a
   chunk's index lists, the column pass on that compact plane; same operands, basis rows, kernel and
   summation order as the full transform, so the extracted values are bit-identical.  Falls back to the
   full transform per chunk when the columns do not fit 8 sqrt(k) slots, and altogether when that exceeds
   W/4 or the shape does not take the default GEMM strategy.  Default on; the call then waits once, at
   its end, for the device.  Handles (Reader::derived exposes coefficients()) always transform fully. */
int ssw_ctx_set_prune(ssw_ctx* ctx, int enable);
/* stats[0] chunks that took the pruned path, [1] of those redone with the full transform, [2] frequency
   columns actually needed (sum over chunks); since the last ssw_ctx_reset_timing. */
int ssw_ctx_get_prune_stats(ssw_ctx* ctx, uint64_t* stats);
/* Top-k selection (the first k entries of the ordering of src/algorithm.rs:200-280): stats[0] frames selected, [1] of
   those whose sampled threshold left fewer than k or more than the candidate buffer's survivors, so that the finish ran
   the exact select over the whole plane (same result, one CU sorting the plane: a latency cliff; massive ties and
   constant planes take it by design, images should not); since the last ssw_ctx_reset_timing.  Synchronises.
   stats[0] is counted when a call enqueues its selection (replays of a captured graph are not counted), stats[1] on
   the device. */
int ssw_ctx_get_select_stats(ssw_ctx* ctx, uint64_t* stats);

/* Even/odd folding of the basis GEMMs (fewer multiply-adds for the same transform; exact in f64,
   one extra rounding per input pair in f32) where the frame shape allows (W % 8 == 0 / H % 8 == 0).
   Strategy levels:
     0  dense GEMMs
     1  one folding level inside the GEMM kernel (1/2 of the dense MACs)
     2  same as 1 (the in-kernel second level of round 1 was superseded by level 4 and removed)
     3  "operand-ready" GEMMs -- HBM-bound pre-passes write the folded operands once per pass as
        k-blocked planes in the GEMM's precision and the MFMA loop issues no VALU instruction
        (csrc/dct_pair_f64.hip, dct_pair_f32.hip, dct_pair_prep.hip); one level
     4  level 3 with the even half folded once more on row passes with W % 16 == 0 and column
        passes with H % 8 == 0 (3/8 of the dense MACs on that axis)
     5  default.  Level 4 plus a third folding level on forward row passes of at least 3072 columns
        (a multiple of 32): 11/32 of the dense MACs there
     6  level 5 without the size threshold (shorter rows lose more to the extra small launches than
        they save; for tests)
   In f64 all levels produce the same f64-accurate result rounded once to f32; in f32 each folding
   level adds one rounding per operand sum (tests/test_gpu_parity.py holds both to their bars).
   The default library carries levels 0 and 3 .. 6 in f64; levels 1 / 2 and every folded level in SSW_PRECISION_F32 are
   part of the diagnostic build only (ssw_build_all_strategies) and run dense otherwise. */
#define SSW_DCT_FOLDING_DEFAULT 5
int ssw_ctx_set_dct_folding(ssw_ctx* ctx, int level);
/* f64 precision, folding level 4 and up: the odd half of a folded transform (a DCT-IV of half the length, the one part
   that does not fold) is computed as a cosine and a sine transform of a quarter of the length each, after one plane
   rotation of its input pairs -- X[2j] = A[j] + B[j], X[2j-1] = A[j] - B[j] with a[n] = d[n] cos psi_n + d[M-1-n] sin psi_n,
   b[n] = d[M-1-n] cos psi_n - d[n] sin psi_n, psi_n = pi (2n+1) / (4M) -- and both of those fold once more: a quarter of
   the odd half's multiply-adds.  On axes whose length allows it (rows % 64 == 0 / columns % 16 == 0 forward, rows % 128 ==
   0 / columns % 16 == 0 inverse) one pre-pass applies this to the odd halves of two levels ("deep"), otherwise
   (length % 8 == 0) to the first.  The rotation is the one inexact step ahead of the f64 MFMA sums (relative error
   2^-52 of the operand); everything else stays an exact fold.  Results: the same f64-accurate transform rounded
   once to f32 -- equal to the unsplit GEMMs' output except where that rounding was within ~1e-9 ulp of a tie
   (tests/test_gpu_parity.py::test_odd_split_matches_exact_operands).  Default on; 0 = every operand an exact sum
   (the round-2 arithmetic, 1.8x the multiply-adds). */
int ssw_ctx_set_odd_split(ssw_ctx* ctx, int enable);

/* Strategy thresholds and A/B switches of the transform, process-wide (csrc/tuning.hip).  They choose between kernels that
   compute the same values (the strategies of src/dct2d.rs:83-219's one transform), never between results; tests lower a
   threshold so that small shapes the oracle finishes in seconds take the kernels of the 4K path.  Names (default):
     efold_min (1280), efold_inv_min (1280), efold_cols_min (720)   shortest forward row / inverse row / column pass at level 2
     deep_min_rows (256), deep_min_cols (256)                       shortest pass that takes the deep pre-passes
     class_tile (1)        class-major planes inside tiles of 128 columns (0: one tile per line)
     prep_staged (1)       LDS-staged pre-passes (0: the r3 kernels)
     merge_max_lines (8192) passes of at most this many lines run a stage's launches as one
     bn32 (-1)             32-pair tiles for small single-class launches: -1 automatic, 0 / 1 forced
     band_split (1)        single-image handles: row pass of the top half beside the upload of the bottom half
     fuse_cols (1)         forward transform: the row GEMMs' epilogue writes the column operands (no f32 plane between the passes)
     fuse_inv_cols (0)     inverse transform: the same (bit-identical, 8 B/px less traffic, measured no faster: off)
     upload_bands (3)      single-image handles: bands of rows a host frame is uploaded and row-transformed in (2 .. 4)
     speculate_k (1)       Reader::base queues its selection for the mark length of the context's last extraction
     prep_light (1)        level-2 RGB row pre-pass in the < 64-VGPR form that fits beside GEMM blocks (0: the register-resident kernel)
     lane_stagger (1)      two lanes: an RGB pre-pass waits for the other lane's row launches and runs beside its column launches
     derived_fused (1)     the derived frame's pruned row pass in one kernel (0: pre-pass + gathered launches)
     inv_prep_light (0)    inverse row pre-pass at level 2 with whole rows through LDS (bit-identical, measured no faster: off)
     gemm_group_m (4), gemm_group_m_rows (4)   tile rows per group of the GEMM launches' block -> tile map (column / row passes)
     tile48 (1)            48-pair GEMM tiles for classes whose 64-pair tiling ends in a tile of <= 16 pairs (135 = 48 + 48 + 39)
     merge_batch (0)       a batch pass's independent GEMM launches as one launch, class after class (r6 A/B: +1.7 % one lane, +0 two)
     gemm_stagger (0)      r6 A/B: the second resident block of every CU starts this many 3.4-us sleeps late (measured: no effect)
   An entry never set reads its SSW_<NAME> environment variable at first use (the r4 behaviour), else the default.
   ssw_tuning_set takes effect for the calls that follow; workspaces and cached plans of existing contexts were sized
   under the old values, so change a value before creating the context that should see it (tests use a fresh context).
   ssw_tuning_reset(NULL) returns every entry to environment / default.  Unknown name: SSW_ERR_BAD_ARG. */
int ssw_tuning_set(const char* name, long long value);
int ssw_tuning_get(const char* name, long long* value);
int ssw_tuning_reset(const char* name);

/* Per-stage device timers (hipEvent pairs on the context's stream).  DCT_ROW / DCT_COL cover the
   GEMM launches of a pass; at folding levels 3 / 4 the pre-passes are timed separately (DCT_PREP). */
typedef enum ssw_stage {
    SSW_STAGE_RGB_TO_YIQ = 0,     /* rgb -> y,i,q (24 B/px) or rgb -> y (16 B/px)   */
    SSW_STAGE_DCT_ROW = 1,        /* basis GEMM along the width                     */
    SSW_STAGE_DCT_COL = 2,        /* basis GEMM along the height                    */
    SSW_STAGE_SELECT = 3,         /* top-k ordering (radix select + sort)           */
    SSW_STAGE_EMBED = 4,
    SSW_STAGE_EXTRACT = 5,
    SSW_STAGE_SIMILARITY = 6,
    SSW_STAGE_YIQ_TO_RGB = 7,
    SSW_STAGE_RESIZE = 8,         /* CatmullRom resize of the attack harness         */
    SSW_STAGE_CONVERT = 9,        /* u8 <-> f32 frame conversion                     */
    SSW_STAGE_DCT_PREP = 10,      /* f64 operand pre-passes of folding levels 3 / 4 (HBM-bound) */
    SSW_STAGE_DCT_ROW_MAIN = 11,  /* the largest GEMM launch of a row pass alone (nested in DCT_ROW) */
    SSW_STAGE_DCT_COL_MAIN = 12,  /* the largest GEMM launch of a column pass alone (nested in DCT_COL) */
    SSW_STAGE_COUNT = 13
} ssw_stage;
int ssw_ctx_enable_timing(ssw_ctx* ctx, int enable);
int ssw_ctx_reset_timing(ssw_ctx* ctx);
/* Synchronises, then returns accumulated milliseconds and launch counts per stage
   (arrays of SSW_STAGE_COUNT). */
int ssw_ctx_get_timing(ssw_ctx* ctx, double* ms, uint64_t* launches);
/* Work done inside the timed regions, per stage (array of SSW_STAGE_COUNT): executed floating-point
   operations for the GEMM stages (DCT_ROW, DCT_COL and their *_MAIN launches), algorithmic bytes
   (what the stage must read and write once) for the HBM-bound ones. */
int ssw_ctx_get_work(ssw_ctx* ctx, double* work);
/* Algorithmic HBM bytes moved inside the timed regions, per stage (array of SSW_STAGE_COUNT): for the HBM-bound stages
   the same figure as ssw_ctx_get_work; for the GEMM stages (DCT_ROW, DCT_COL) the operand planes read, the results
   written and what the dependent launches of an inverse pass exchange (A1 / T2 / E out and in, I and Q in and RGB out in
   the last pass of Writer::result, src/algorithm.rs:361-379) -- each byte once, bases not counted (cache-resident).
   bench.py's `roofline_step` is built from it. */
int ssw_ctx_get_traffic(ssw_ctx* ctx, double* bytes);

/* Device memory helpers for hosts without their own allocator. */
int ssw_dev_mem_info(ssw_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int ssw_dev_alloc(ssw_ctx* ctx, size_t bytes, void** dev_ptr);
int ssw_dev_free(ssw_ctx* ctx, void* dev_ptr);
int ssw_copy_to_dev(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int ssw_copy_to_host(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);

/* Host <-> device transfers of the host-buffer entry points.  A pinned host buffer (ssw_host_alloc,
   hipHostMalloc, hipHostRegister) is the DMA source / target itself; any other buffer goes through a ring
   of pinned staging buffers in the context, filled / drained by a few host threads while the DMA of the
   previous slice runs (csrc/transfer.hip).  The reference has no counterpart: its images never leave the
   CPU (`DynamicImage` in, `DynamicImage` out, src/algorithm.rs:295, :355). */
int ssw_host_alloc(ssw_ctx* ctx, size_t bytes, void** host_ptr);      /* pinned (page-locked) host memory */
int ssw_host_free(ssw_ctx* ctx, void* host_ptr);
/* Host threads (the caller's included) that copy between pageable buffers and the staging ring:
   0 = automatic (4, or half the cores if fewer; environment: SSW_COPY_THREADS), 1 = the caller's only. */
int ssw_ctx_set_copy_threads(ssw_ctx* ctx, int threads);
typedef enum ssw_transfer_stat {
    SSW_TRANSFER_H2D_BYTES = 0,       /* bytes uploaded by host-buffer entry points                  */
    SSW_TRANSFER_D2H_BYTES = 1,
    SSW_TRANSFER_H2D_SECONDS = 2,     /* host wall time inside uploads (staging copy + DMA issue)    */
    SSW_TRANSFER_D2H_SECONDS = 3,     /* host wall time inside downloads (incl. waiting for results) */
    SSW_TRANSFER_STAGED_BYTES = 4,    /* of the above, bytes that went through the staging ring      */
    SSW_TRANSFER_DIRECT_BYTES = 5,    /* ... and bytes DMA'd straight from / to pinned caller memory */
    SSW_TRANSFER_STAT_COUNT = 6
} ssw_transfer_stat;
/* Copies the counters (array of SSW_TRANSFER_STAT_COUNT doubles; may be NULL) and optionally zeroes them. */
int ssw_ctx_get_transfer_stats(ssw_ctx* ctx, double* stats, int reset);

/* ---- transforms (device-resident, batched) --------------------------------- */

/* From<&Rgb32FImage> for YIQ32FImage, src/yiq.rs:177-186.  rgb: [n][h][w][3] f32.
   dev_i / dev_q may both be NULL (readers never use them: algorithm.rs:476). */
int ssw_rgb_to_yiq(ssw_ctx* ctx, const float* dev_rgb, size_t n_frames, size_t w, size_t h,
                   float* dev_y, float* dev_i, float* dev_q);
/* From<&YIQ32FImage> for Rgb32FImage, src/yiq.rs:187-197 (clamps to [0,1]). */
int ssw_yiq_to_rgb(ssw_ctx* ctx, const float* dev_y, const float* dev_i, const float* dev_q,
                   size_t n_frames, size_t w, size_t h, float* dev_rgb);
/* dct2d::dct2_2d, src/dct2d.rs:83-219, in place on n_frames contiguous row-major
   planes.  Same pass order (larger dimension first), same f32 store between the
   passes, same scaling points. */
int ssw_dct2d(ssw_ctx* ctx, int dct_type, int precision, size_t n_frames, size_t w, size_t h,
              float* dev_planes);

/* ---- ordering (src/algorithm.rs:200-280) ----------------------------------- */
/* First k entries of obtain_indices_by_function (src/algorithm.rs:200-210) for each of
   n_frames coefficient planes: stable descending order, DC skipped, ties -> lower index first.
   dev_indices: [n_frames][k] u32.  k <= w*h-1. */
int ssw_topk_indices(ssw_ctx* ctx, const float* dev_coef, size_t n_frames, size_t w, size_t h,
                     int ordering, size_t k, uint32_t* dev_indices);

/* ---- embed / extract / similarity on device-resident coefficient planes ---- */
/* Writer::embed_watermark, src/algorithm.rs:382-410.  marks: [n_frames][n_marks][k]
   f32 (every mark of length k); indices [n_frames][k]. */
int ssw_embed_coefficients(ssw_ctx* ctx, float* dev_coef, size_t n_frames, size_t plane_len,
                           const uint32_t* dev_indices, size_t k, int method, float alpha,
                           const float* dev_marks, size_t n_marks);
/* Reader::extract_watermark, src/algorithm.rs:543-562.  out: [n_frames][k]. */
int ssw_extract_coefficients(ssw_ctx* ctx, const float* dev_base, const float* dev_derived,
                             size_t n_frames, size_t plane_len, const uint32_t* dev_indices,
                             size_t k, int method, float alpha, float* dev_out);
/* Tester::similarity, src/algorithm.rs:696-714, n_pairs independent (extracted,
   mark) pairs of length k; sequential f32 accumulation order preserved. */
int ssw_similarity_batch(ssw_ctx* ctx, const float* dev_extracted, const float* dev_marks,
                         size_t n_pairs, size_t k, float* dev_sims);

/* One-extraction-many-marks testing (README.md:62; the CLI's loop over stored marks,
   examples/main.rs:369-415): sims[b][j] = Tester::new(extracted[b]).similarity(marks[j]) for
   n_extracted extracted marks against a database of n_marks stored marks of length k, as an
   (n_extracted x k) . (k x n_marks) GEMM on the f32 matrix cores.  Numerators are MFMA fma chains
   instead of the reference's sequential f32 sums (src/algorithm.rs:702-711): equal to 1e-4
   relative, not bit for bit; denominators keep the reference's order.  dev_sims: [n_extracted][n_marks]. */
int ssw_similarity_matrix(ssw_ctx* ctx, const float* dev_extracted, size_t n_extracted, const float* dev_marks,
                          size_t n_marks, size_t k, float* dev_sims);

/* ---- whole path, batched & device-resident (the bench path) ---------------- */
/* Writer::new + Writer::mark for n_frames frames (algorithm.rs:295-316, :355-379):
   rgb -> yiq -> DCT2 -> top-k -> embed -> DCT3 -> rgb.  One mark of length k per
   frame: dev_marks [n_frames][k].  A mark longer than w*h-1 is cut at w*h-1 entries like the
   reference's zip() does (:396); the stride of dev_marks stays k.  Optional outputs (may be NULL):
   dev_coef_out [n_frames][h][w] = Writer::coefficient_image() before embedding,
   dev_indices_out [n_frames][min(k, w*h-1)]. */
int ssw_batch_embed(ssw_ctx* ctx, const ssw_config* cfg, const float* dev_rgb, size_t n_frames,
                    size_t w, size_t h, const float* dev_marks, size_t k, float* dev_rgb_out,
                    float* dev_coef_out, uint32_t* dev_indices_out);
/* Reader::base + Reader::derived + extract (+ Tester::similarity when dev_marks
   is given) for n_frames frame pairs (algorithm.rs:462-562, :696-714).
   dev_extracted [n_frames][k]; dev_sims [n_frames] (NULL allowed with dev_marks NULL). */
int ssw_batch_extract(ssw_ctx* ctx, const ssw_config* cfg, const float* dev_base_rgb,
                      const float* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                      float* dev_extracted, const float* dev_marks, float* dev_sims);

/* ---- 8-bit frames and the resize attack (device-resident, batched) --------- */
/* Arithmetic of the third-party `image 0.24.3` crate, restated from its published behaviour
   (parity unpinned beyond the reference's similarity asserts; identical to the CPU oracle). */
/* `DynamicImage::into_rgb32f()` for 8-bit input (call sites src/algorithm.rs:308, :476): v / 255. */
int ssw_convert_rgb8_to_f32(ssw_ctx* ctx, const uint8_t* dev_in, size_t n_values, float* dev_out);
/* `DynamicImage::into_rgb8()` from Rgb32F (tests/single_simple.rs:28): round(clamp(v,0,1) * 255). */
int ssw_convert_f32_to_rgb8(ssw_ctx* ctx, const float* dev_in, size_t n_values, uint8_t* dev_out);
/* `image::imageops::resize(img, nw, nh, FilterType::CatmullRom)` (tests/attack_resize.rs:17-36) on
   n_frames RGB8 frames [h][w][3] -> [nh][nw][3]. */
int ssw_resize_rgb8(ssw_ctx* ctx, const uint8_t* dev_in, size_t n_frames, size_t w, size_t h, size_t nw,
                    size_t nh, uint8_t* dev_out);
/* ssw_batch_embed / ssw_batch_extract on 8-bit frames: the u8 -> f32 conversion is fused into the
   colour kernels (3 instead of 12 B/px at the boundary); the embedded frames come back quantised
   like `into_rgb8()`.  Same reference lines as the f32 forms. */
int ssw_batch_embed_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* dev_rgb, size_t n_frames,
                         size_t w, size_t h, const float* dev_marks, size_t k, uint8_t* dev_rgb_out);
int ssw_batch_extract_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* dev_base_rgb,
                           const uint8_t* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                           float* dev_extracted, const float* dev_marks, float* dev_sims);

/* ---- 16-bit frames (device-resident, batched) ----------------------------------- */
/* `DynamicImage::into_rgb32f()` for 16-bit input (ImageRgb16; call sites src/algorithm.rs:308, :476): v / 65535,
   and `into_rgb16()` from Rgb32F: round(clamp(v,0,1) * 65535) (`image 0.24.3`, like the 8-bit forms). */
int ssw_convert_rgb16_to_f32(ssw_ctx* ctx, const uint16_t* dev_in, size_t n_values, float* dev_out);
int ssw_convert_f32_to_rgb16(ssw_ctx* ctx, const float* dev_in, size_t n_values, uint16_t* dev_out);
/* ssw_batch_embed / ssw_batch_extract on 16-bit frames: v / 65535 fused into the first operand pre-pass (6 instead of
   12 B/px read at the boundary).  The marked frames come back as f32 -- what Writer::mark returns
   (src/algorithm.rs:355-379: Rgb32F); quantise with ssw_convert_f32_to_rgb16 / _rgb8 as the caller would.  Bit-identical
   to the f32 entry points on host-converted frames.  Same reference lines as the f32 forms. */
int ssw_batch_embed_rgb16(ssw_ctx* ctx, const ssw_config* cfg, const uint16_t* dev_rgb, size_t n_frames,
                          size_t w, size_t h, const float* dev_marks, size_t k, float* dev_rgb_out);
int ssw_batch_extract_rgb16(ssw_ctx* ctx, const ssw_config* cfg, const uint16_t* dev_base_rgb,
                            const uint16_t* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                            float* dev_extracted, const float* dev_marks, float* dev_sims);

/* ---- host-image streaming (n frames that live on the host, one call) ------------- */
/* The loops of the reference's callers -- examples/main.rs:271-278 (`watermark`: per image Writer::new -> mark ->
   into_rgb8) and :383-415 (`test`: per image Reader::base / derived -> extract -> Tester::similarity) -- as ONE call over
   n 8-bit host images [h][w][3] (frames[i]: one pointer per image; pinned buffers -- ssw_host_alloc, hipHostRegister --
   are the DMA source / target themselves, any other buffer goes through the context's staging ring): the images go
   through ssw_batch_embed_rgb8 / ssw_batch_extract_rgb8 in groups of frames (G = 8 4K frames by default; a call ramps
   up through groups of G/4 and G/2 frames, embed ramps down again), the upload of group g + 1, the kernels of group g
   and the download of group g - 1 in flight together (csrc/ssw_stream.hip).  The output side overlaps fully only
   with pinned output buffers (a pageable buffer is filled by a blocking staged copy; it is issued after the next
   group's kernels are queued, so the device keeps working, but the host thread sleeps in it).  The device ring of a
   call (3 slots x 2 buffers of G frames) stays allocated until ssw_ctx_destroy or until an allocation of the context
   runs short of device memory.  Bit-identical to the single-image handles.  Host-buffer entry points: they return when every buffer is the caller's again.
   host_marks: [n][k] f32 (mark i for frame i); extract: host_marks and host_sims both or neither (NULL). */
int ssw_batch_embed_host_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* const* host_frames, size_t n_frames,
                              size_t w, size_t h, const float* host_marks, size_t k, uint8_t* const* host_out);
int ssw_batch_extract_host_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* const* host_base,
                                const uint8_t* const* host_derived, size_t n_frames, size_t w, size_t h, size_t k,
                                float* host_extracted, const float* host_marks, float* host_sims);

/* ---- single-image handles mirroring the crate's types (host buffers) ------- */
/* Writer::new(image, config), src/algorithm.rs:295-316.  rgb_hwc: host [h][w][3]
   f32 (what `into_rgb32f()` yields, :308).  The ordering is computed lazily at
   embed time, when the mark length is known (only the first k entries are ever
   consumed, :396). */
int ssw_writer_create(ssw_ctx* ctx, const float* rgb_hwc, size_t w, size_t h,
                      const ssw_config* cfg, ssw_writer** out);
/* Writer::new on an 8-bit image: `image.into_rgb32f()` (src/algorithm.rs:308; v / 255) happens on the
   device, fused into the colour conversion, and 3 instead of 12 bytes per pixel cross PCIe.  rgb_hwc:
   host [h][w][3] u8.  Bit-identical to ssw_writer_create on the host-converted frame. */
int ssw_writer_create_rgb8(ssw_ctx* ctx, const uint8_t* rgb_hwc, size_t w, size_t h,
                           const ssw_config* cfg, ssw_writer** out);
/* Writer::new on a 16-bit image (ImageRgb16): `into_rgb32f()` (src/algorithm.rs:308; v / 65535) on the device, 6 instead
   of 12 bytes per pixel cross PCIe.  rgb_hwc: host [h][w][3] u16.  Bit-identical to ssw_writer_create on the
   host-converted frame. */
int ssw_writer_create_rgb16(ssw_ctx* ctx, const uint16_t* rgb_hwc, size_t w, size_t h,
                            const ssw_config* cfg, ssw_writer** out);
/* Writer::coefficient_image(), src/algorithm.rs:319-321 -> host [h][w]. */
int ssw_writer_coefficients(ssw_writer* wr, float* out_plane);
/* Writer::embed(&mut self, marks), src/algorithm.rs:348-352.  marks[m] has lens[m] floats (host).
   The ordering is the one Writer::new fixed from the ORIGINAL coefficients (:314): a second embed()
   (the reference says "call once", but allows it) still ranks the original plane, not the modified one. */
int ssw_writer_embed(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks);
/* Writer::result(self), src/algorithm.rs:361-379 -> host [h][w][3]; consumes the writer. */
int ssw_writer_result(ssw_writer* wr, float* out_rgb_hwc);
/* Writer::mark(self, marks), src/algorithm.rs:355-358 = embed + result. */
int ssw_writer_mark(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks,
                    float* out_rgb_hwc);
/* Writer::result(self).into_rgb8() / Writer::mark(self, marks).into_rgb8() (src/algorithm.rs:355-379 followed
   by the caller's `into_rgb8()`, examples/main.rs:271-278, tests/single_simple.rs:28): round(clamp(v,0,1)*255)
   in the epilogue of the last inverse pass -> host [h][w][3] u8.  Works on writers created from f32 or u8. */
int ssw_writer_result_rgb8(ssw_writer* wr, uint8_t* out_rgb_hwc);
int ssw_writer_mark_rgb8(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks,
                         uint8_t* out_rgb_hwc);
int ssw_writer_destroy(ssw_writer* wr);

/* Reader::base(image, config) when is_base != 0 (src/algorithm.rs:462-464), else
   Reader::derived / ReaderDerived::new (:453-455, :469-471); cfg may be NULL for a derived reader. */
int ssw_reader_create(ssw_ctx* ctx, const float* rgb_hwc, size_t w, size_t h, int is_base,
                      const ssw_config* cfg, ssw_reader** out);
/* The same on an 8-bit image (`into_rgb32f()`, src/algorithm.rs:476, on the device); host [h][w][3] u8. */
int ssw_reader_create_rgb8(ssw_ctx* ctx, const uint8_t* rgb_hwc, size_t w, size_t h, int is_base,
                           const ssw_config* cfg, ssw_reader** out);
/* ... and on a 16-bit image (v / 65535; src/algorithm.rs:476); host [h][w][3] u16. */
int ssw_reader_create_rgb16(ssw_ctx* ctx, const uint16_t* rgb_hwc, size_t w, size_t h, int is_base,
                            const ssw_config* cfg, ssw_reader** out);
/* Reader::coefficients(), src/algorithm.rs:502-504 -> host [h*w]. */
int ssw_reader_coefficients(ssw_reader* rd, float* out_plane);
/* Reader::indices(), src/algorithm.rs:506-508: first k entries (k <= w*h-1) as u64 (`usize`). */
int ssw_reader_indices(ssw_reader* rd, size_t k, uint64_t* out);
/* Reader::extract(&self, &derived, &mut [f32]), src/algorithm.rs:529-539. */
int ssw_reader_extract(ssw_reader* base, ssw_reader* derived, float* out, size_t k);
int ssw_reader_destroy(ssw_reader* rd);

/* Tester::new(extracted).similarity(mark), src/algorithm.rs:689-714 (host buffers, computed on the GPU). */
int ssw_similarity(ssw_ctx* ctx, const float* extracted, size_t n_extracted,
                   const float* mark, size_t n_mark, float* out_similarity);

/* ---- synthetic input (bench / parity plumbing, not in the reference) ------- */
/* Deterministic multi-octave value-noise frames, bit-identical to the oracle's
   generator: frame index = first_frame + i.  dev_rgb: [n_frames][h][w][3]. */
int ssw_synth_frames(ssw_ctx* ctx, uint32_t seed, uint32_t first_frame, size_t n_frames,
                     size_t w, size_t h, float* dev_rgb);

#ifdef __cplusplus
}
#endif
#endif /* SSW_H */
