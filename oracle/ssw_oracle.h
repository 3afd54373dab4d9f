/*
 * ssw_oracle.h -- CPU restatement of the reference hot path.
 *
 * ======================  TEST INFRASTRUCTURE ONLY  ======================
 * This is the parity checker for the HIP library, NOT a product path and
 * NOT a fallback.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it.  Nothing under
 * spread_spectrum_watermarking_amd/ links, imports or executes it.
 * ========================================================================
 *
 * Reference: iwanders/spread_spectrum_watermarking (Rust).  Citations are
 * file:line relative to the reference tree.  The reference cannot be built in
 * this environment (no cargo/rustc, dependencies not vendored), so there is
 * no oracle/_ref build.  Pinning status:
 *   - yiq / ordering / embed / extract / similarity: restated op-for-op and
 *     checked against every known-answer unit test of the reference
 *     (tests/test_oracle_golden.py, vectors transcribed in tests/golden/).
 *   - 1-D DCT kernels live in the un-vendored crate rustdct 0.7.0
 *     (Cargo.lock:520); its definition is restated from the reference's own
 *     call sites and scipy-derived goldens (src/dct2d.rs:229-524, 1e-4 abs) and
 *     cross-checked against scipy.fft in this image.  Bit-level behaviour of
 *     rustfft's butterflies is unpinned by the reference (its tests use 1e-4),
 *     so the oracle's f64 backend returns the correctly-rounded transform.
 *   - 8-bit / codec / resize boundary (`image 0.24.3`): parity unpinned.
 */
#ifndef SSW_ORACLE_H
#define SSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* transform types, src/dct2d.rs:71-79 */
enum { SSWO_DCT2 = 0, SSWO_DCT2_ORTHOGONAL = 1, SSWO_DCT3 = 2 };
/* 1-D kernel back-ends standing in for rustdct */
enum { SSWO_BACKEND_F64 = 0, SSWO_BACKEND_F32 = 1, SSWO_BACKEND_NAIVE_F64 = 2 };
/* OrderingMethod, src/algorithm.rs:143-152 (Custom is a closure: not restated) */
enum { SSWO_ORDER_ENERGY = 0, SSWO_ORDER_ENERGY_ORTHOGONAL = 1, SSWO_ORDER_LEGACY = 2 };
/* Insertion / Extraction, src/algorithm.rs:68-77,115-124 */
enum { SSWO_OPTION1 = 1, SSWO_OPTION2 = 2, SSWO_OPTION3 = 3 };

/* src/yiq.rs:177-186 (+ :131-136, :157-159): interleaved RGB -> 3 planes */
void sswo_rgb_to_yiq(const float* rgb, size_t npix, float* y, float* i, float* q);
/* src/yiq.rs:187-197 (+ :139-147, :163-165): 3 planes -> interleaved RGB, clamp [0,1] */
void sswo_yiq_to_rgb(const float* y, const float* i, const float* q, size_t npix, float* rgb);

/* src/dct2d.rs:83-219: in-place separable 2-D transform of a row-major w*h plane.
   Returns 0, or -1 on bad arguments. */
int sswo_dct2d(int type, int backend, size_t w, size_t h, float* data);
/* rustdct-convention 1-D kernels (tests pin them, src/dct2d.rs:229-265) */
int sswo_dct1d(int type /*SSWO_DCT2 or SSWO_DCT3*/, int backend, size_t n, float* data);

/* src/algorithm.rs:200-210 with comparators :214-280.  Writes the first k
   entries of the stably-sorted descending index list (k <= n-1; the list has
   n-1 entries, DC skipped). Returns number written. */
size_t sswo_indices(const float* coef, size_t n, int ordering, size_t w, size_t h,
                    size_t k, uint64_t* out);
/* sortable key used by sswo_indices (exposed for tie-aware checks):
   larger int32 == sorts earlier. */
int32_t sswo_order_key(int ordering, size_t w, size_t h, size_t index, float value);

/* src/algorithm.rs:382-410 + :414-432 */
void sswo_embed(float* coef, size_t n, const uint64_t* indices, size_t n_indices,
                int method, float alpha,
                const float* const* marks, const size_t* mark_lens, size_t n_marks);
/* src/algorithm.rs:543-562 + :566-593.  0 ok, 1 length mismatch (:550-552),
   2 k too large (:553-555). */
int sswo_extract(const float* base, size_t n_base, const float* derived, size_t n_derived,
                 const uint64_t* indices, int method, float alpha, float* out, size_t k);
/* src/algorithm.rs:696-714 */
float sswo_similarity(const float* extracted, const float* mark, size_t k);

/* Synthetic bench/parity frames; bit-identical twin of
   the device generator ssw_synth_frames().  Not part of the reference. */
void sswo_synth_frame(uint32_t seed, uint32_t frame, size_t w, size_t h, float* rgb);

/* 8-bit boundary + CatmullRom resize of the attack harness (third-party `image 0.24.3` semantics,
   parity unpinned; see the .c file). */
void sswo_u8_to_f32(const uint8_t* in, size_t n, float* out);
void sswo_f32_to_u8(const float* in, size_t n, uint8_t* out);
void sswo_u16_to_f32(const uint16_t* in, size_t n, float* out);      /* into_rgb32f of ImageRgb16: v / 65535 */
void sswo_f32_to_u16(const float* in, size_t n, uint16_t* out);      /* into_rgb16: round(clamp(v,0,1) * 65535) */
size_t sswo_resize_taps(size_t in_len, size_t out_len, size_t out_idx, uint32_t* left_out, float* ws, size_t max_taps);
void sswo_resize_rgb8(const uint8_t* in, size_t w, size_t h, size_t nw, size_t nh, uint8_t* out);

/* Whole-path helpers used by the cpu_baseline leg: Writer::new + mark
   (algorithm.rs:295-379) and Reader::base + derived + extract + similarity
   (:462-562, :696-714) for one frame.  `full_sort` != 0 sorts all n-1
   coefficients like the reference; 0 uses a partial selection. */
void sswo_embed_frame(const float* rgb, size_t w, size_t h, int backend, int ordering,
                      int method, float alpha, const float* mark, size_t k,
                      int full_sort, float* out_rgb);
float sswo_extract_frame(const float* base_rgb, const float* derived_rgb, size_t w, size_t h,
                         int backend, int ordering, int method, float alpha,
                         const float* mark, size_t k, int full_sort, float* extracted);

#ifdef __cplusplus
}
#endif
#endif
