// wm_genw.cpp -- watermark generator: same CLI and file format as the reference's CommonRandomMatrix tool
// (CommonRandomMatrix/main.cpp:16-68): `wm_genw <rows> <cols> <seed> <output_file>` writes rows*cols N(0,1) floats,
// raw little-endian f32, row-major -- the only contract Watermark::loadRandomMatrix relies on (Watermark.cpp:62-75).
//
// The reference seeds one mt19937 + std::normal_distribution per OpenMP thread with the SAME seed
// (CommonRandomMatrix/main.cpp:41), so its matrix is T identical chunks and depends on the thread count and on the
// standard library's distribution.  This tool is counter-based instead: element (r,c) is a pure function of
// (seed, r, c) -- a 32-bit integer hash -> two uniforms -> Box-Muller in f64 -- identical for any thread count and
// identical to watermarking-gpu_amd/synth.py:synth_watermark().
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
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

int main(int argc, char* argv[])
{
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <rows> <cols> <seed> <output_file>\n";
        return EXIT_FAILURE;
    }
    const int rows = std::stoi(argv[1]);
    const int cols = std::stoi(argv[2]);
    const uint32_t seed = (uint32_t)std::stoul(argv[3]);
    const std::string filename = argv[4];
    if (rows <= 0 || cols <= 0 || rows >= 32768 || cols >= 32768) {
        std::cerr << "Rows and columns must be positive integers less than or equal to 32768.\n";
        return EXIT_FAILURE;
    }
    std::vector<float> w((size_t)rows * cols);
    const double two_pi = 6.283185307179586476925286766559;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const double u1 = ((double)hash_u32(seed, 0x5741u, (uint32_t)r, (uint32_t)c) + 1.0) / 4294967297.0;
            const double u2 = (double)hash_u32(seed, 0x5742u, (uint32_t)r, (uint32_t)c) / 4294967296.0;
            w[(size_t)r * cols + c] = (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(two_pi * u2));
        }
    std::ofstream output(filename, std::ios::binary);
    if (!output) {
        std::cerr << "Error: Unable to open file " << filename << " for writing.\n";
        return EXIT_FAILURE;
    }
    output.write(reinterpret_cast<const char*>(w.data()), (std::streamsize)(w.size() * sizeof(float)));
    if (!output) {
        std::cerr << "Error: Failed to write data to " << filename << ".\n";
        return EXIT_FAILURE;
    }
    std::cout << "Successfully wrote " << (size_t)rows * cols << " random floats to " << filename << ".\n";
    return EXIT_SUCCESS;
}
