// C++ counterpart of the reference's doc-tests / tests/single_simple.rs, written against
// include/ssw.hpp (the host-side mirror of the crate surface).  Reads a raw f32 RGB frame and a
// mark from files written by the pytest driver, runs embed -> extract -> similarity on the GPU and
// prints the numbers the driver checks against the CPU oracle.
//   usage: crate_surface_test <rgb.f32> <w> <h> <mark.f32> <k> <out_marked.f32> <out_extracted.f32>
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
    if (argc != 8) return 2;
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

        std::printf("similarity %.6f exceeds6 %d random %.6f consumed_ok %d too_large_ok %d first_index %llu\n",
                    sim.similarity, sim.exceeds_sigma(6.0f) ? 1 : 0, rnd.similarity, consumed_ok ? 1 : 0,
                    too_large_ok ? 1 : 0, (unsigned long long)reader.indices(1)[0]);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
