// ssw.hpp -- C++17 host-side mirror of the crate's public surface over the C ABI (ssw.h).
//
// The reference is compiled code (Rust); no Rust toolchain exists in this image, so the host side
// above the C ABI is written in C++ with the reference's names, argument meaning and error
// behaviour (file:line relative to the reference tree):
//
//   wm::Writer / WriteConfig / Insertion      src/algorithm.rs:68-112, :285-433
//   wm::Reader / ReaderDerived / ReadConfig    src/algorithm.rs:114-140, :435-594
//   wm::OrderingMethod                         src/algorithm.rs:142-191
//   wm::MarkBuf                                src/algorithm.rs:596-666
//   wm::Tester / Similarity                    src/algorithm.rs:668-715
//
// Where the reference panics, wm::Error (carrying the ssw_status) is thrown.  Images are
// interleaved RGB f32 buffers [h][w][3] -- what `DynamicImage::into_rgb32f()` yields (:308, :476).
// Header-only; link against libssw_hip.so.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ssw.h"

namespace wm {

class Error : public std::runtime_error {
public:
    Error(int status, const std::string& where)
        : std::runtime_error(where + ": " + ssw_status_string(status) +
                             (status == SSW_ERR_HIP || status == SSW_ERR_OUT_OF_MEMORY ? std::string(" [") + ssw_last_error() + "]" : std::string())),
          status_(status) {}
    int status() const { return status_; }

private:
    int status_;
};

inline void check(int status, const char* where) {
    if (status != SSW_OK) throw Error(status, where);
}

// One per GPU; not thread-safe (the reference's Writer/Reader are !Send).
class Context {
public:
    explicit Context(int device_id = 0) { check(ssw_ctx_create(device_id, &ctx_), "ssw_ctx_create"); }
    ~Context() { ssw_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ssw_ctx* get() const { return ctx_; }
    void synchronize() { check(ssw_ctx_synchronize(ctx_), "ssw_ctx_synchronize"); }
    // Stream hand-off for hosts that drive the device-pointer entry points (see the stream contract in ssw.h):
    // hipStream_t / hipEvent_t as opaque pointers, so that this header needs no HIP headers.
    void set_stream(void* hip_stream) { check(ssw_ctx_set_stream(ctx_, hip_stream), "ssw_ctx_set_stream"); }
    void wait_event(void* hip_event) { check(ssw_ctx_wait_event(ctx_, hip_event), "ssw_ctx_wait_event"); }
    void record_event(void* hip_event) { check(ssw_ctx_record_event(ctx_, hip_event), "ssw_ctx_record_event"); }
    // Batch entry points: frames per internal pass (0 = automatic), two passes in flight, pruned derived transform.
    void set_chunk_frames(size_t n) { check(ssw_ctx_set_chunk_frames(ctx_, n), "ssw_ctx_set_chunk_frames"); }
    size_t pass_frames(size_t n_frames, size_t w, size_t h) const { return ssw_ctx_pass_frames(ctx_, n_frames, w, h); }
    void set_overlap(bool on) { check(ssw_ctx_set_overlap(ctx_, on ? 1 : 0), "ssw_ctx_set_overlap"); }
    void set_prune(bool on) { check(ssw_ctx_set_prune(ctx_, on ? 1 : 0), "ssw_ctx_set_prune"); }
    void set_odd_split(bool on) { check(ssw_ctx_set_odd_split(ctx_, on ? 1 : 0), "ssw_ctx_set_odd_split"); }
    // Host-buffer entry points: copy threads of the pinned staging ring (0 = automatic) and the transfer counters.
    void set_copy_threads(int n) { check(ssw_ctx_set_copy_threads(ctx_, n), "ssw_ctx_set_copy_threads"); }
    std::vector<double> transfer_stats(bool reset = false) {
        std::vector<double> st(SSW_TRANSFER_STAT_COUNT);
        check(ssw_ctx_get_transfer_stats(ctx_, st.data(), reset ? 1 : 0), "ssw_ctx_get_transfer_stats");
        return st;
    }

private:
    ssw_ctx* ctx_ = nullptr;
};

// Pinned (page-locked) host memory from the context: an image decoded into it is the DMA source / target of the
// handles, no staging copy (include/ssw.h: ssw_host_alloc).
class PinnedBuffer {
public:
    PinnedBuffer(Context& ctx, size_t bytes) : ctx_(ctx.get()), bytes_(bytes) { check(ssw_host_alloc(ctx_, bytes, &p_), "ssw_host_alloc"); }
    ~PinnedBuffer() { ssw_host_free(ctx_, p_); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    void* data() { return p_; }
    size_t size() const { return bytes_; }

private:
    ssw_ctx* ctx_;
    void* p_ = nullptr;
    size_t bytes_;
};

// Insertion / Extraction (algorithm.rs:68-77, :115-124).  Custom(closure) exists in the reference;
// it cannot cross to the device and yields SSW_ERR_UNSUPPORTED.
struct Insertion {
    int method;
    float alpha;
    static Insertion Option1(float a) { return {SSW_OPTION1, a}; }
    static Insertion Option2(float a) { return {SSW_OPTION2, a}; }
    static Insertion Option3(float a) { return {SSW_OPTION3, a}; }
    static Insertion Custom() { return {SSW_METHOD_CUSTOM, 0.f}; }
};
using Extraction = Insertion;

enum class OrderingMethod : int {
    Energy = SSW_ORDER_ENERGY,
    EnergyOrthogonal = SSW_ORDER_ENERGY_ORTHOGONAL,
    Legacy = SSW_ORDER_LEGACY,
    Custom = SSW_ORDER_CUSTOM,
};

struct WriteConfig {                                   // algorithm.rs:99-112
    Insertion insertion = Insertion::Option2(0.1f);
    OrderingMethod ordering = OrderingMethod::Energy;
    ssw_precision precision = SSW_PRECISION_F64;
    ssw_config c() const { return {static_cast<int32_t>(ordering), insertion.method, insertion.alpha, precision}; }
};
struct ReadConfig {                                    // algorithm.rs:127-140
    Extraction extraction = Extraction::Option2(0.1f);
    OrderingMethod ordering = OrderingMethod::Energy;
    ssw_precision precision = SSW_PRECISION_F64;
    ssw_config c() const { return {static_cast<int32_t>(ordering), extraction.method, extraction.alpha, precision}; }
};

// An RGB f32 image [h][w][3].
struct ImageRgb32F {
    size_t width = 0, height = 0;
    std::vector<float> data;
    ImageRgb32F() = default;
    ImageRgb32F(size_t w, size_t h) : width(w), height(h), data(w * h * 3) {}
};

// An 8-bit RGB image [h][w][3] (what image files decode to).  Handed to the library as it is: `into_rgb32f()`
// (algorithm.rs:308, :476) runs on the device and a quarter of the bytes cross PCIe.
struct ImageRgb8 {
    size_t width = 0, height = 0;
    std::vector<uint8_t> data;
    ImageRgb8() = default;
    ImageRgb8(size_t w, size_t h) : width(w), height(h), data(w * h * 3) {}
};

// A 16-bit RGB image [h][w][3] (16-bit PNG / TIFF): `into_rgb32f()` = v / 65535 on the device, half the bytes cross PCIe.
struct ImageRgb16 {
    size_t width = 0, height = 0;
    std::vector<uint16_t> data;
    ImageRgb16() = default;
    ImageRgb16(size_t w, size_t h) : width(w), height(h), data(w * h * 3) {}
};

class MarkBuf {                                        // algorithm.rs:607-645
public:
    MarkBuf() = default;
    static MarkBuf generate_normal(size_t length) {    // :619-626 (non-deterministic by design)
        MarkBuf m;
        std::random_device rd;
        std::mt19937_64 gen(rd());
        std::normal_distribution<float> n(0.f, 1.f);
        m.data_.resize(length);
        for (auto& v : m.data_) v = n(gen);
        return m;
    }
    static MarkBuf from(const float* data, size_t n) { MarkBuf m; m.data_.assign(data, data + n); return m; }
    static MarkBuf from(const std::vector<float>& v) { return from(v.data(), v.size()); }
    const std::vector<float>& data() const { return data_; }
    void set_data(const float* data, size_t n) { data_.assign(data, data + n); }

private:
    std::vector<float> data_;
};

class Writer {                                         // algorithm.rs:285-433
public:
    Writer(Context& ctx, const ImageRgb32F& image, const WriteConfig& config = WriteConfig())
        : w_(image.width), h_(image.height) {
        if (image.data.size() != w_ * h_ * 3) throw Error(SSW_ERR_BAD_DIMS, "Writer::new");
        ssw_config c = config.c();
        check(ssw_writer_create(ctx.get(), image.data.data(), w_, h_, &c, &wr_), "Writer::new");
    }
    Writer(Context& ctx, const ImageRgb8& image, const WriteConfig& config = WriteConfig())
        : w_(image.width), h_(image.height) {
        if (image.data.size() != w_ * h_ * 3) throw Error(SSW_ERR_BAD_DIMS, "Writer::new");
        ssw_config c = config.c();
        check(ssw_writer_create_rgb8(ctx.get(), image.data.data(), w_, h_, &c, &wr_), "Writer::new");
    }
    Writer(Context& ctx, const ImageRgb16& image, const WriteConfig& config = WriteConfig())
        : w_(image.width), h_(image.height) {
        if (image.data.size() != w_ * h_ * 3) throw Error(SSW_ERR_BAD_DIMS, "Writer::new");
        ssw_config c = config.c();
        check(ssw_writer_create_rgb16(ctx.get(), image.data.data(), w_, h_, &c, &wr_), "Writer::new");
    }
    ~Writer() { ssw_writer_destroy(wr_); }
    Writer(const Writer&) = delete;
    Writer& operator=(const Writer&) = delete;

    std::vector<float> coefficient_image() const {     // :319-321
        std::vector<float> out(w_ * h_);
        check(ssw_writer_coefficients(wr_, out.data()), "Writer::coefficient_image");
        return out;
    }
    void embed(const std::vector<const MarkBuf*>& marks) {   // :348-352
        std::vector<const float*> p; std::vector<size_t> l;
        for (auto* m : marks) { p.push_back(m->data().data()); l.push_back(m->data().size()); }
        check(ssw_writer_embed(wr_, p.data(), l.data(), p.size()), "Writer::embed");
    }
    ImageRgb32F result() {                             // :361-379 (consumes the writer)
        ImageRgb32F out(w_, h_);
        check(ssw_writer_result(wr_, out.data.data()), "Writer::result");
        return out;
    }
    ImageRgb32F mark(const std::vector<const MarkBuf*>& marks) {   // :355-358
        embed(marks);
        return result();
    }
    // `writer.mark(marks).into_rgb8()` (examples/main.rs:271-278): quantised on the device, 3 bytes per pixel back
    ImageRgb8 mark_rgb8(const std::vector<const MarkBuf*>& marks) {
        embed(marks);
        ImageRgb8 out(w_, h_);
        check(ssw_writer_result_rgb8(wr_, out.data.data()), "Writer::mark");
        return out;
    }

private:
    size_t w_, h_;
    ssw_writer* wr_ = nullptr;
};

class ReaderDerived;

class Reader {                                         // algorithm.rs:441-594
public:
    static Reader base(Context& ctx, const ImageRgb32F& image, const ReadConfig& config = ReadConfig()) {   // :462-464
        return Reader(ctx, image, true, config);
    }
    static Reader base(Context& ctx, const ImageRgb8& image, const ReadConfig& config = ReadConfig()) {
        return Reader(ctx, image, true, config);
    }
    ~Reader() { ssw_reader_destroy(rd_); }
    Reader(Reader&& o) noexcept : w_(o.w_), h_(o.h_), rd_(o.rd_) { o.rd_ = nullptr; }
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;

    std::vector<float> coefficients() const {          // :502-504
        std::vector<float> out(w_ * h_);
        check(ssw_reader_coefficients(rd_, out.data()), "Reader::coefficients");
        return out;
    }
    std::vector<uint64_t> indices(size_t k) const {    // :506-508 (first k entries)
        std::vector<uint64_t> out(k);
        check(ssw_reader_indices(rd_, k, out.data()), "Reader::indices");
        return out;
    }
    void extract(const ReaderDerived& derived, std::vector<float>& extracted) const;   // :529-539

private:
    friend class ReaderDerived;
    Reader(Context& ctx, const ImageRgb32F& image, bool is_base, const ReadConfig& config)
        : w_(image.width), h_(image.height) {
        if (image.data.size() != w_ * h_ * 3) throw Error(SSW_ERR_BAD_DIMS, "Reader::new_impl");
        ssw_config c = config.c();
        check(ssw_reader_create(ctx.get(), image.data.data(), w_, h_, is_base ? 1 : 0, &c, &rd_), "Reader::new_impl");
    }
    Reader(Context& ctx, const ImageRgb8& image, bool is_base, const ReadConfig& config)
        : w_(image.width), h_(image.height) {
        if (image.data.size() != w_ * h_ * 3) throw Error(SSW_ERR_BAD_DIMS, "Reader::new_impl");
        ssw_config c = config.c();
        check(ssw_reader_create_rgb8(ctx.get(), image.data.data(), w_, h_, is_base ? 1 : 0, &c, &rd_), "Reader::new_impl");
    }
    Reader(Context& ctx, const ImageRgb16& image, bool is_base, const ReadConfig& config)
        : w_(image.width), h_(image.height) {
        if (image.data.size() != w_ * h_ * 3) throw Error(SSW_ERR_BAD_DIMS, "Reader::new_impl");
        ssw_config c = config.c();
        check(ssw_reader_create_rgb16(ctx.get(), image.data.data(), w_, h_, is_base ? 1 : 0, &c, &rd_), "Reader::new_impl");
    }
    size_t w_, h_;
    ssw_reader* rd_ = nullptr;
};

class ReaderDerived {                                  // algorithm.rs:448-456
public:
    ReaderDerived(Context& ctx, const ImageRgb32F& image, ssw_precision precision = SSW_PRECISION_F64)
        : r_(ctx, image, false, [&] { ReadConfig c; c.precision = precision; return c; }()) {}
    ReaderDerived(Context& ctx, const ImageRgb8& image, ssw_precision precision = SSW_PRECISION_F64)
        : r_(ctx, image, false, [&] { ReadConfig c; c.precision = precision; return c; }()) {}
    std::vector<float> coefficients() const { return r_.coefficients(); }

private:
    friend class Reader;
    Reader r_;
};

inline void Reader::extract(const ReaderDerived& derived, std::vector<float>& extracted) const {
    check(ssw_reader_extract(rd_, derived.r_.rd_, extracted.data(), extracted.size()), "Reader::extract");
}

struct Similarity {                                    // algorithm.rs:669-680
    float similarity;
    bool exceeds_sigma(float n_sigma) const { return similarity > n_sigma; }
};

class Tester {                                         // algorithm.rs:683-715
public:
    Tester(Context& ctx, const std::vector<float>& extracted_watermark) : ctx_(ctx), e_(extracted_watermark) {}
    Similarity similarity(const MarkBuf& comparison_watermark) const {
        float out = 0.f;
        check(ssw_similarity(ctx_.get(), e_.data(), e_.size(), comparison_watermark.data().data(),
                             comparison_watermark.data().size(), &out), "Tester::similarity");
        return {out};
    }

private:
    Context& ctx_;
    const std::vector<float>& e_;
};

// The callers' loops over host images (examples/main.rs:271-278, :383-415) as one streaming call each: uploads,
// kernels and downloads of consecutive groups of images overlap (ssw_batch_embed_host_rgb8 / ssw_batch_extract_host_rgb8).
// marks[i] is embedded into images[i]; all marks have one length.  Bit-identical to a loop over Writer / Reader.
inline std::vector<ImageRgb8> mark_many(Context& ctx, const std::vector<const ImageRgb8*>& images, const std::vector<const MarkBuf*>& marks,
                                        const WriteConfig& config = WriteConfig()) {
    if (images.empty() || images.size() != marks.size()) throw Error(SSW_ERR_BAD_ARG, "mark_many");
    const size_t w = images[0]->width, h = images[0]->height, k = marks[0]->data().size();
    std::vector<const uint8_t*> in(images.size());
    std::vector<float> m(images.size() * k);
    std::vector<ImageRgb8> out(images.size(), ImageRgb8(w, h));
    std::vector<uint8_t*> op(images.size());
    for (size_t i = 0; i < images.size(); ++i) {
        if (images[i]->width != w || images[i]->height != h || images[i]->data.size() != w * h * 3 || marks[i]->data().size() != k)
            throw Error(SSW_ERR_BAD_DIMS, "mark_many");
        in[i] = images[i]->data.data();
        std::copy(marks[i]->data().begin(), marks[i]->data().end(), m.begin() + i * k);
        op[i] = out[i].data.data();
    }
    ssw_config c = config.c();
    check(ssw_batch_embed_host_rgb8(ctx.get(), &c, in.data(), in.size(), w, h, m.data(), k, op.data()), "mark_many");
    return out;
}
struct ExtractedMany { std::vector<std::vector<float>> extracted; std::vector<float> similarity; };
inline ExtractedMany extract_many(Context& ctx, const std::vector<const ImageRgb8*>& base, const std::vector<const ImageRgb8*>& derived,
                                  size_t k, const std::vector<const MarkBuf*>& marks, const ReadConfig& config = ReadConfig()) {
    if (base.empty() || base.size() != derived.size() || (!marks.empty() && marks.size() != base.size())) throw Error(SSW_ERR_BAD_ARG, "extract_many");
    const size_t n = base.size(), w = base[0]->width, h = base[0]->height;
    std::vector<const uint8_t*> bp(n), dp(n);
    std::vector<float> m(marks.empty() ? 0 : n * k), ext(n * k), sims(marks.empty() ? 0 : n);
    for (size_t i = 0; i < n; ++i) {
        if (base[i]->width != w || base[i]->height != h || derived[i]->width != w || derived[i]->height != h ||
            base[i]->data.size() != w * h * 3 || derived[i]->data.size() != w * h * 3)
            throw Error(SSW_ERR_BAD_DIMS, "extract_many");
        bp[i] = base[i]->data.data(); dp[i] = derived[i]->data.data();
        if (!marks.empty()) {
            if (marks[i]->data().size() != k) throw Error(SSW_ERR_LENGTH_MISMATCH, "extract_many");
            std::copy(marks[i]->data().begin(), marks[i]->data().end(), m.begin() + i * k);
        }
    }
    ssw_config c = config.c();
    check(ssw_batch_extract_host_rgb8(ctx.get(), &c, bp.data(), dp.data(), n, w, h, k, ext.data(), marks.empty() ? nullptr : m.data(),
                                      marks.empty() ? nullptr : sims.data()), "extract_many");
    ExtractedMany r;
    for (size_t i = 0; i < n; ++i) r.extracted.emplace_back(ext.begin() + i * k, ext.begin() + (i + 1) * k);
    r.similarity = sims;
    return r;
}

}  // namespace wm
