// wm_genw.cpp -- watermark generator: same CLI and file format as the reference's CommonRandomMatrix tool
// (CommonRandomMatrix/main.cpp:16-68): `wm_genw <rows> <cols> <seed> <output_file>` writes rows*cols N(0,1) floats,
// raw little-endian f32, row-major -- the only contract Watermark::loadRandomMatrix relies on (Watermark.cpp:62-75).
//
// The reference seeds one mt19937 + std::normal_distribution per OpenMP thread with the SAME seed
// (CommonRandomMatrix/main.cpp:41), so its matrix is T identical chunks and depends on the thread count and on the
// standard library's distribution.  This tool is counter-based instead: element (r,c) is a pure function of
// (seed, r, c) -- a 32-bit integer hash -> two uniforms -> Box-Muller in f64 -- identical for any thread count and
// identical to watermarking-gpu_amd/synth.py:synth_watermark().
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static inline uint32_t mix(uint32_t h)  // lowbias32 finalizer
{
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
static inline uint32_t hash_u32(uint32_t seed, uint32_t stream, uint32_t r, uint32_t c)
{
    const uint32_t h = mix(seed ^ mix(stream * 0x9E3779B1u + r));
    return mix(h ^ mix(c + 0x85EBCA6Bu));
}

// one command-line number: the whole token must parse and lie in [lo, hi]
static bool parse_uint(const char* tok, unsigned long lo, unsigned long hi, unsigned long* out)
{
    if (!tok || !*tok || *tok == '-' || *tok == '+') return false;
    char* end = nullptr;
    errno = 0;
    const unsigned long v = std::strtoul(tok, &end, 10);
    if (errno != 0 || *end != '\0' || v < lo || v > hi) return false;
    *out = v;
    return true;
}

int main(int argc, char* argv[])
{
    // the same four positional arguments as CommonRandomMatrix (the contract); everything around them is this tool's own
    unsigned long rows = 0, cols = 0, seed = 0;
    const bool ok = argc == 5 && parse_uint(argv[1], 1, 32767, &rows) && parse_uint(argv[2], 1, 32767, &cols) && parse_uint(argv[3], 0, 0xFFFFFFFFul, &seed);
    if (!ok) {
        std::fprintf(stderr, "Usage: %s <rows> <cols> <seed> <output_file>\n  rows, cols: 1 .. 32767;  seed: 0 .. 4294967295;  output: raw little-endian f32, row-major\n",
                     argc > 0 ? argv[0] : "wm_genw");
        return 2;
    }
    std::FILE* f = std::fopen(argv[4], "wb");
    if (!f) {
        std::fprintf(stderr, "wm_genw: cannot create '%s': %s\n", argv[4], std::strerror(errno));
        return 1;
    }
    // generated and written in bands of rows (element (r, c) depends on (seed, r, c) only: any banding, any thread count, the same file)
    const unsigned long band = 256;
    std::vector<float> buf((size_t)band * cols);
    const double two_pi = 6.283185307179586476925286766559;
    for (unsigned long r0 = 0; r0 < rows; r0 += band) {
        const long nr = (long)(rows - r0 < band ? rows - r0 : band);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < nr; ++i)
            for (unsigned long c = 0; c < cols; ++c) {
                const uint32_t r = (uint32_t)(r0 + (unsigned long)i);
                const double u1 = ((double)hash_u32((uint32_t)seed, 0x5741u, r, (uint32_t)c) + 1.0) / 4294967297.0;
                const double u2 = (double)hash_u32((uint32_t)seed, 0x5742u, r, (uint32_t)c) / 4294967296.0;
                buf[(size_t)i * cols + c] = (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(two_pi * u2));
            }
        const size_t want = (size_t)nr * cols;
        if (std::fwrite(buf.data(), sizeof(float), want, f) != want) {
            std::fprintf(stderr, "wm_genw: short write to '%s': %s\n", argv[4], std::strerror(errno));
            std::fclose(f);
            return 1;
        }
    }
    if (std::fclose(f) != 0) {
        std::fprintf(stderr, "wm_genw: closing '%s' failed: %s\n", argv[4], std::strerror(errno));
        return 1;
    }
    std::printf("wm_genw: %lu x %lu = %lu N(0,1) values (seed %lu) -> %s\n", rows, cols, rows * cols, seed, argv[4]);
    return 0;
}
