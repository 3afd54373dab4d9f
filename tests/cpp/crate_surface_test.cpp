// C++ counterpart of the reference's doc-tests / tests/single_simple.rs, written against
// include/ssw.hpp (the host-side mirror of the crate surface).  Reads a raw f32 RGB frame and a
// mark from files written by the pytest driver, runs embed -> extract -> similarity on the GPU and
// prints the numbers the driver checks against the CPU oracle.
//   usage: crate_surface_test <rgb.f32> <w> <h> <mark.f32> <k> <out_marked.f32> <out_extracted.f32> <out_marked8.u8>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "ssw.hpp"

static std::vector<float> read_f32(const char* path, size_t n) {
    std::vector<float> v(n);
    std::ifstream f(path, std::ios::binary);
    f.read(reinterpret_cast<char*>(v.data()), n * sizeof(float));
    if (!f) { std::fprintf(stderr, "short read %s\n", path); std::exit(2); }
    return v;
}
static void write_f32(const char* path, const std::vector<float>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), v.size() * sizeof(float));
}

int main(int argc, char** argv) {
    if (argc != 9) return 2;
    const size_t w = std::atoi(argv[2]), h = std::atoi(argv[3]), k = std::atoi(argv[5]);
    try {
        wm::Context ctx(0);
        wm::ImageRgb32F orig(w, h);
        orig.data = read_f32(argv[1], w * h * 3);
        wm::MarkBuf mark = wm::MarkBuf::from(read_f32(argv[4], k));

        // lib.rs:24-41: embed with the default configuration
        wm::Writer watermarker(ctx, orig, wm::WriteConfig());
        wm::ImageRgb32F res = watermarker.mark({&mark});
        write_f32(argv[6], res.data);

        // a consumed writer must refuse (Writer::result takes self)
        bool consumed_ok = false;
        try { watermarker.result(); } catch (const wm::Error& e) { consumed_ok = e.status() == SSW_ERR_CONSUMED; }

        // lib.rs:45-66: extract and test
        wm::Reader reader = wm::Reader::base(ctx, orig, wm::ReadConfig());
        wm::ReaderDerived derived(ctx, res);
        std::vector<float> extracted(k);
        reader.extract(derived, extracted);
        write_f32(argv[7], extracted);
        wm::Tester tester(ctx, extracted);
        wm::Similarity sim = tester.similarity(mark);
        wm::MarkBuf other = wm::MarkBuf::generate_normal(k);
        wm::Similarity rnd = tester.similarity(other);

        // error behaviour of Reader::extract (algorithm.rs:553-555)
        bool too_large_ok = false;
        try { std::vector<float> big(w * h); reader.extract(derived, big); }
        catch (const wm::Error& e) { too_large_ok = e.status() == SSW_ERR_K_TOO_LARGE; }

        // the same flow on the 8-bit image a file would decode to (examples/main.rs:271-278: ... .mark(..).into_rgb8())
        wm::ImageRgb8 orig8(w, h);
        for (size_t i = 0; i < orig8.data.size(); ++i) {
            float v = orig.data[i] < 0.f ? 0.f : (orig.data[i] > 1.f ? 1.f : orig.data[i]);
            orig8.data[i] = (uint8_t)std::floor(v * 255.0f + 0.5f);
        }
        wm::Writer writer8(ctx, orig8, wm::WriteConfig());
        wm::ImageRgb8 marked8 = writer8.mark_rgb8({&mark});
        { std::ofstream f(argv[8], std::ios::binary); f.write(reinterpret_cast<const char*>(marked8.data.data()), marked8.data.size()); }
        wm::Reader reader8 = wm::Reader::base(ctx, orig8, wm::ReadConfig());
        wm::ReaderDerived derived8(ctx, marked8);
        std::vector<float> extracted8(k);
        reader8.extract(derived8, extracted8);
        wm::Tester tester8(ctx, extracted8);
        wm::Similarity sim8 = tester8.similarity(mark);

        std::printf("similarity %.6f exceeds6 %d random %.6f consumed_ok %d too_large_ok %d first_index %llu similarity8 %.6f\n",
                    sim.similarity, sim.exceeds_sigma(6.0f) ? 1 : 0, rnd.similarity, consumed_ok ? 1 : 0,
                    too_large_ok ? 1 : 0, (unsigned long long)reader.indices(1)[0], sim8.similarity);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
