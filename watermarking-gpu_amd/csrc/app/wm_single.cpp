// wm_single: the reference's call pattern timed from C++ -- ONE image per call, synchronous
// Watermark::makeWatermark followed by Watermark::detectWatermark (Watermark.cpp:156-172,234-250; the loop of
// testForImage, main.cpp:165-220, without its file I/O).  Prints one JSON line; bench.py's "single_call" leg runs it.
//   wm_single <rows> <cols> <loops> [f32|u8] [ME|NVF] [work_dir]
// The environment variable WM_FUSED=0 selects the batched sweeps instead of the fused single-launch kernels.
#include "../../../include/Watermark.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

static uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: wm_single rows cols loops [f32|u8] [ME|NVF] [work_dir]\n"); return 2; }
    const int R = std::atoi(argv[1]), C = std::atoi(argv[2]), loops = std::atoi(argv[3]);
    const bool u8 = argc > 4 && std::string(argv[4]) == "u8";
    const MASK_TYPE mask = argc > 5 && std::string(argv[5]) == "NVF" ? NVF : ME;
    const std::string dir = argc > 6 ? argv[6] : "/tmp";
    const size_t n = (size_t)R * C;
    std::vector<float> w(n), x(n);
    for (size_t i = 0; i < n; ++i) {
        const float u1 = unit(hash32(77u + 2u * (uint32_t)i)) + 1e-7f, u2 = unit(hash32(77u + 2u * (uint32_t)i + 1u));
        w[i] = std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
    }
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            float v = 128.0f + 56.0f * std::sin(r * 0.0648f) * std::cos(c * 0.103f) + 36.0f * std::sin((r + 2 * c) * 0.01615f) +
                      40.0f * (unit(hash32(5u + (uint32_t)(r * C + c))) - 0.5f);
            v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            x[(size_t)r * C + c] = u8 ? std::floor(v) : v;
        }
    const std::string wpath = dir + "/wm_single_w_" + std::to_string(R) + "x" + std::to_string(C) + ".dat";
    {
        std::ofstream f(wpath, std::ios::binary);
        f.write(reinterpret_cast<const char*>(w.data()), (std::streamsize)(n * sizeof(float)));
    }
    try {
        const Watermark wm(R, C, wpath, 3, 40.0f);
        wm::Image img;
        if (u8) {
            std::vector<uint8_t> xb(n);
            for (size_t i = 0; i < n; ++i) xb[i] = (uint8_t)x[i];
            img = wm::Image::fromHost(xb.data(), R, C);
        } else img = wm::Image::fromHost(x.data(), R, C);
        float a = 0.0f, corr = 0.0f;
        wm::Image y;
        for (int i = 0; i < 5; ++i) { y = wm.makeWatermark(img, img, a, mask); corr = wm.detectWatermark(y, mask); }
        using clk = std::chrono::steady_clock;
        auto us = [](clk::time_point t0, clk::time_point t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
        auto t0 = clk::now();
        for (int i = 0; i < loops; ++i) y = wm.makeWatermark(img, img, a, mask);
        auto t1 = clk::now();
        for (int i = 0; i < loops; ++i) corr = wm.detectWatermark(y, mask);
        auto t2 = clk::now();
        for (int i = 0; i < loops; ++i) { y = wm.makeWatermark(img, img, a, mask); corr = wm.detectWatermark(y, mask); }
        auto t3 = clk::now();
        // the same pair as ONE call (wm_embed_detect: both launches back to back, one wait)
        float a2 = 0.0f, corr2 = 0.0f;
        for (int i = 0; i < 5; ++i) y = wm.makeAndDetectWatermark(img, img, a2, corr2, mask);
        auto t4 = clk::now();
        for (int i = 0; i < loops; ++i) y = wm.makeAndDetectWatermark(img, img, a2, corr2, mask);
        auto t5 = clk::now();
        if (a2 != a || corr2 != corr) { std::fprintf(stderr, "wm_single: the one-call pair differs from the two calls (a %.9g / %.9g, corr %.9g / %.9g)\n", (double)a2, (double)a, (double)corr2, (double)corr); return 1; }
        int wg = 0, th = 0;
        unsigned long long fb = 0;
        const int fused = wm_fused_info(wm.handle(), &wg, &th, &fb);
        std::printf("{\"rows\": %d, \"cols\": %d, \"dtype\": \"%s\", \"mask\": \"%s\", \"loops\": %d, \"embed_us\": %.2f, \"detect_us\": %.2f, "
                    "\"pair_us\": %.2f, \"pair_one_call_us\": %.2f, \"a\": %.6f, \"corr\": %.7f, \"fused\": %d, \"workgroups\": %d, \"tile_rows\": %d, \"fallbacks\": %llu}\n",
                    R, C, u8 ? "u8" : "f32", mask == ME ? "ME" : "NVF", loops, us(t0, t1) / loops, us(t1, t2) / loops, us(t2, t3) / loops, us(t4, t5) / loops,
                    (double)a, (double)corr, fused, wg, th, fb);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "wm_single: %s", e.what());
        return 1;
    }
    std::remove(wpath.c_str());
    return 0;
}
