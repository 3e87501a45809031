// wm_selftest: the reference's class surface exercised from C++ (include/Watermark.hpp over libwm_hip.so):
// constructor errors with the reference's messages, deep copies that share W, reinitialize, the embed()/detect()
// aliases, RGB bases, the unsolvable-system rule.  Prints one line per check; exit code = number of failed checks.
//   wm_selftest <work_dir>      (work_dir receives two generated W files)
#include "../../../include/Watermark.hpp"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

static int failures = 0;
#define CHECK(cond, what)                                                     \
    do {                                                                      \
        const bool ok_ = (cond);                                              \
        std::printf("%s  %s\n", ok_ ? "ok  " : "FAIL", what);                 \
        if (!ok_) ++failures;                                                 \
    } while (0)

static uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

static std::vector<float> make_w(int rows, int cols, uint32_t seed)
{
    std::vector<float> w((size_t)rows * cols);
    for (size_t i = 0; i < w.size(); ++i) {
        const float u1 = unit(hash32(seed + 2u * (uint32_t)i)) + 1e-7f, u2 = unit(hash32(seed + 2u * (uint32_t)i + 1u));
        w[i] = std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
    }
    return w;
}
static std::vector<float> make_frame(int rows, int cols, uint32_t seed)
{
    std::vector<float> x((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            x[(size_t)r * cols + c] = 128.0f + 60.0f * std::sin(r * 0.07f) * std::cos(c * 0.05f) + 30.0f * (unit(hash32(seed + (uint32_t)(r * cols + c))) - 0.5f);
    return x;
}
static void write_w(const std::string& path, const std::vector<float>& w)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(w.data()), (std::streamsize)(w.size() * sizeof(float)));
}
template <typename F>
static std::string message_of(F&& f)
{
    try { f(); } catch (const std::runtime_error& e) { return e.what(); }
    return "";
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int R = 96, C = 300, R2 = 130, C2 = 260;
    const std::string w1 = dir + "/w_96x300.dat", w2 = dir + "/w_130x260.dat";
    write_w(w1, make_w(R, C, 11));
    write_w(w2, make_w(R2, C2, 12));

    // ---- constructor errors (Watermark.cpp:24-25, 65-66, 70-71)
    CHECK(message_of([&] { Watermark w(R, C, w1, 4, 40.0f); }).find("Wrong p parameter: 4") != std::string::npos, "p = 4 is rejected with the reference's message");
    CHECK(message_of([&] { Watermark w(R, C, dir + "/missing.dat", 3, 40.0f); }).find("Error opening") != std::string::npos, "missing W file");
    CHECK(message_of([&] { Watermark w(R, C, w2, 3, 40.0f); }).find("W file total elements != image dimensions") != std::string::npos, "W file of another size");

    // ---- embed / detect, aliases
    Watermark wm1(R, C, w1, 3, 40.0f);
    const std::vector<float> x = make_frame(R, C, 5);
    const wm::Image gray = wm::Image::fromHost(x.data(), R, C, 1);
    float a_me = -1.0f, a_nvf = -1.0f, a_alias = -1.0f;
    const wm::Image y_me = wm1.makeWatermark(gray, gray, a_me, ME);
    const wm::Image y_nvf = wm1.makeWatermark(gray, gray, a_nvf, NVF);
    const wm::Image y_alias = wm1.embed(gray, gray, a_alias, ME);
    const float c_me = wm1.detectWatermark(y_me, ME), c_nvf = wm1.detectWatermark(y_nvf, NVF);
    CHECK(a_me > 0.0f && a_nvf > 0.0f && a_alias == a_me, "strengths are set; embed() is makeWatermark()");
    CHECK(c_me > 0.2f && c_nvf > 0.1f && wm1.detect(y_me, ME) == c_me, "marked images correlate; detect() is detectWatermark()");
    CHECK(std::fabs(wm1.detectWatermark(gray, ME)) < 0.05f, "an unmarked image does not");
    std::vector<float> yh((size_t)R * C), yh2((size_t)R * C);
    y_me.host(yh.data());
    {
        // the pair as one call (an addition to the reference's class): the same image, strength and score as the two calls
        float a_pair = -1.0f, c_pair = -1.0f;
        const wm::Image y_pair = wm1.makeAndDetectWatermark(gray, gray, a_pair, c_pair, ME);
        y_pair.host(yh2.data());
        CHECK(a_pair == a_me && std::fabs(c_pair - c_me) <= 2e-7f && yh2 == yh, "makeAndDetectWatermark: the two calls' results in one");
        wm1.setHandover(true);   // (no effect on the synchronous methods; the switch itself must work)
        float a_ho = -1.0f;
        const wm::Image y_ho = wm1.makeWatermark(gray, gray, a_ho, ME);
        CHECK(a_ho == a_me && wm1.detectWatermark(y_ho, ME) == c_me, "setHandover leaves the synchronous calls as they are");
        wm1.setHandover(false);
    }
    double mse = 0.0;
    for (size_t i = 0; i < yh.size(); ++i) mse += ((double)yh[i] - x[i]) * ((double)yh[i] - x[i]);
    const double psnr = 10.0 * std::log10(255.0 * 255.0 / (mse / (double)yh.size()));
    CHECK(std::fabs(psnr - 40.0) < 0.2, "the embed hits the requested PSNR");

    // ---- RGB base: the same a*u is added to the three channels (main.cpp:169-190)
    std::vector<float> rgb((size_t)3 * R * C);
    for (int ch = 0; ch < 3; ++ch)
        for (size_t i = 0; i < x.size(); ++i) rgb[(size_t)ch * R * C + i] = std::fmin(255.0f, std::fmax(0.0f, x[i] + 10.0f * (ch - 1)));
    const wm::Image base = wm::Image::fromHost(rgb.data(), R, C, 3);
    float a_rgb = 0.0f;
    const wm::Image y_rgb = wm1.makeWatermark(gray, base, a_rgb, ME);
    std::vector<float> yr((size_t)3 * R * C);
    y_rgb.host(yr.data());
    bool rgb_ok = y_rgb.channels() == 3 && a_rgb == a_me;
    for (size_t i = 0; i < x.size() && rgb_ok; i += 97) {
        const float d0 = yr[i] - rgb[i], d1 = yr[(size_t)R * C + i] - rgb[(size_t)R * C + i];
        if (rgb[i] > 40.f && rgb[i] < 215.f && std::fabs(d0 - d1) > 1e-3f) rgb_ok = false;
    }
    CHECK(rgb_ok, "RGB base: 3 channels out, same strength, same increment per channel");

    // ---- deep copy that shares W (Watermark.cpp:30-51): same results, independent lifetime
    float a_copy = 0.0f, a_assign = 0.0f;
    {
        Watermark copy(wm1);
        const wm::Image yc = copy.makeWatermark(gray, gray, a_copy, ME);
        yc.host(yh2.data());
        CHECK(a_copy == a_me && yh2 == yh && copy.detectWatermark(yc, ME) == c_me, "copy constructor: identical results");
    }
    Watermark other(R2, C2, w2, 3, 40.0f);
    other = wm1;
    const wm::Image ya = other.makeWatermark(gray, gray, a_assign, ME);
    CHECK(a_assign == a_me && other.size().rows == R && other.size().cols == C, "copy assignment replaces the engine");
    float a_after = 0.0f;
    wm1.makeWatermark(gray, gray, a_after, ME);
    CHECK(a_after == a_me, "the original still works after its copy was destroyed");

    // ---- reinitialize (Watermark.cpp:78-85)
    wm1.reinitialize(w2, R2, C2);
    const std::vector<float> x2 = make_frame(R2, C2, 6);
    const wm::Image gray2 = wm::Image::fromHost(x2.data(), R2, C2, 1);
    float a2 = 0.0f;
    const wm::Image y2 = wm1.makeWatermark(gray2, gray2, a2, ME);
    CHECK(a2 > 0.0f && wm1.detectWatermark(y2, ME) > 0.2f && y2.rows() == R2 && y2.cols() == C2, "reinitialize: new size, new W");
    CHECK(message_of([&] { float t; wm1.makeWatermark(gray, gray, t, ME); }).find("ERROR in makeWatermark") != std::string::npos, "an image of the old size is refused");
    CHECK(message_of([&] { wm1.reinitialize(w1, R2, C2); }).find("W file total elements != image dimensions") != std::string::npos, "reinitialize checks the W file size");

    // ---- unsolvable system (Watermark.cpp:164-165, 246-247): constant image
    const std::vector<float> flat((size_t)R2 * C2, 77.0f);
    const wm::Image fl = wm::Image::fromHost(flat.data(), R2, C2, 1);
    float a_flat = -123.0f;
    const wm::Image y_flat = wm1.makeWatermark(fl, fl, a_flat, ME);
    CHECK(a_flat == -123.0f && y_flat.same_buffer(fl), "unsolvable: the output image is returned as is, the strength is untouched");
    CHECK(wm1.detectWatermark(fl, ME) == 0.0f, "unsolvable: detect returns 0.0f");

    std::printf("%d check(s) failed\n", failures);
    return failures;
}
