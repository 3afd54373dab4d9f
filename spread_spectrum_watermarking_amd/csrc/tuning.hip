// Strategy thresholds and A/B switches of the transform: one process-wide table, settable at run time through the C ABI
// (ssw_tuning_set / ssw_tuning_get, include/ssw.h) so that tests lower a threshold in-process instead of spawning a child
// with an environment variable (sixteen getenv switches read once per process).  An entry that was never set
// takes its SSW_* environment variable (read at first use, as before), else its default.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ssw_internal.hpp"

namespace ssw {
namespace {

struct Entry {
    const char* name;
    const char* env;
    long long dflt;
    std::atomic<long long> value{0};
    std::atomic<int> state{0};             // 0: not read yet, 1: from the environment / default, 2: set through the ABI
    Entry(const char* n, const char* e, long long d) : name(n), env(e), dflt(d) {}
};

Entry g_tune[TUNE_COUNT] = {
    {"efold_min", "SSW_EFOLD_MIN", 1280},              // shortest forward row pass at level 2 (dct_pair_efold)
    {"efold_inv_min", "SSW_EFOLD_INV_MIN", 1280},      // shortest inverse row pass at level 2 (dct_pair_efold_inv)
    {"efold_cols_min", "SSW_EFOLD_COLS_MIN", 720},     // shortest column pass at level 2 (dct_pair_efold_cols)
    {"class_tile", "SSW_CLASS_TILE", 1},               // class-major order inside tiles of 128 columns (0: one tile per line)
    {"deep_min_rows", "SSW_DEEP_MIN_ROWS", 256},       // shortest row pass that takes the deep pre-passes
    {"deep_min_cols", "SSW_DEEP_MIN_COLS", 256},       // ... column pass
    {"prep_staged", "SSW_PREP_STAGED", 1},             // the LDS-staged pre-passes (0: the r3 kernels)
    {"merge_max_lines", "SSW_MERGE_MAX_LINES", 8192},  // passes of at most this many lines run a stage's classes as one launch
    {"bn32", "SSW_BN32", -1},                          // 32-pair tiles for small single-class launches: -1 automatic, 0 / 1 forced
    {"band_split", "SSW_BAND_SPLIT", 1},               // single-image handles: row pass of the top half beside the upload of the bottom half
    {"fuse_cols", "SSW_FUSE_COLS", 1},                 // forward: column operands straight from the row GEMMs' epilogue (r5)
    {"fuse_inv_cols", "SSW_FUSE_INV_COLS", 0},         // inverse: the same (r5; bit-identical, measured no faster: off by default)
    {"upload_bands", "SSW_UPLOAD_BANDS", 3},           // single-image handles: bands of rows a host frame is uploaded and row-transformed in (2 .. 4)
    {"speculate_k", "SSW_SPECULATE_K", 1},             // Reader::base queues the selection for the context's last extraction length
    {"prep_light", "SSW_PREP_LIGHT", 1},               // level-2 row pre-pass in the < 64-VGPR form that runs beside the GEMMs (0: pair_prep16_rows_kernel)
    {"lane_stagger", "SSW_LANE_STAGGER", 1},           // two lanes: RGB pre-passes beside the other lane's column launches, not its row launches (r5: +0.7 %)
    {"derived_fused", "SSW_DERIVED_FUSED", 1},         // the derived frame's pruned row pass in one kernel (marks of up to 1024 entries; 0: pre-pass + launches)
    {"inv_prep_light", "SSW_INV_PREP_LIGHT", 0},       // inverse row pre-pass at level 2: whole rows through LDS, one lane per unit (r5 A/B)
    {"gemm_stagger", "SSW_GEMM_STAGGER", 0},           // r6 A/B: the GEMM blocks 256 .. 511 of a launch (the CUs' second residents) start this many ~3.4-us sleeps late
    {"gemm_group_m", "SSW_GEMM_GROUP_M", 4},           // column passes: tile rows per group of the GEMMs' block -> tile map (1: a line tile's tile columns are consecutive blocks)
    {"gemm_group_m_rows", "SSW_GEMM_GROUP_M_ROWS", 4}, // row passes: the same (r6: 1 cuts a fused row launch's PMC FETCH_SIZE from 1.95 to 1.29 GB and costs 2.8 % of the row stage)
    {"merge_batch", "SSW_MERGE_BATCH", 0},             // r6 A/B: a batch pass's independent launches as one, class after class (PairMulti::cls_major)
    {"tile48", "SSW_TILE48", 1},                       // r6: 48-pair tiles for classes whose 64-pair tiling ends in a tile of <= 16 pairs (135 = 48 + 48 + 39)
};

// value and state change together under this lock (first read from the environment, set, reset); the fast path of a reader
// is one acquire load of a state that is no longer 0.  (a reader that had seen state 0 stored the environment's
// value unconditionally and could overwrite a concurrent ssw_tuning_set.)
std::mutex g_tune_lock;

}  // namespace

long long tuning(int which) {
    Entry& e = g_tune[which];
    if (e.state.load(std::memory_order_acquire) == 0) {
        std::lock_guard<std::mutex> lk(g_tune_lock);
        if (e.state.load(std::memory_order_relaxed) == 0) {
            const char* s = std::getenv(e.env);
            long long v = s ? std::atoll(s) : e.dflt;
            if (which == TUNE_BAND_SPLIT && std::getenv("SSW_NO_SPLIT")) v = 0;      // the r3 name of the switch
            e.value.store(v, std::memory_order_relaxed);
            e.state.store(1, std::memory_order_release);
        }
    }
    return e.value.load(std::memory_order_relaxed);
}

}  // namespace ssw

extern "C" {

int ssw_tuning_set(const char* name, long long value) {
    if (!name) return SSW_ERR_BAD_ARG;
    for (int i = 0; i < ssw::TUNE_COUNT; ++i)
        if (std::strcmp(ssw::g_tune[i].name, name) == 0) {
            std::lock_guard<std::mutex> lk(ssw::g_tune_lock);
            ssw::g_tune[i].value.store(value, std::memory_order_relaxed);
            ssw::g_tune[i].state.store(2, std::memory_order_release);
            return SSW_OK;
        }
    return SSW_ERR_BAD_ARG;
}

int ssw_tuning_get(const char* name, long long* value) {
    if (!name || !value) return SSW_ERR_BAD_ARG;
    for (int i = 0; i < ssw::TUNE_COUNT; ++i)
        if (std::strcmp(ssw::g_tune[i].name, name) == 0) { *value = ssw::tuning(i); return SSW_OK; }
    return SSW_ERR_BAD_ARG;
}

int ssw_tuning_reset(const char* name) {
    for (int i = 0; i < ssw::TUNE_COUNT; ++i)
        if (!name || std::strcmp(ssw::g_tune[i].name, name) == 0) {
            std::lock_guard<std::mutex> lk(ssw::g_tune_lock);
            ssw::g_tune[i].state.store(0, std::memory_order_release);
            if (name) return SSW_OK;
        }
    return name ? SSW_ERR_BAD_ARG : SSW_OK;
}

}  // extern "C"
