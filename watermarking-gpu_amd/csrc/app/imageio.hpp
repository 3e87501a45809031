// imageio.hpp -- minimal image file I/O for the harness: binary PPM/PGM (P6/P5) read+write and PNG read
// (8-bit grey / RGB / RGBA, non-interlaced) through zlib.  The reference loads images with ArrayFire/FreeImage
// (main.cpp:153); neither exists on the target.  Pixels come back as interleaved RGB u8 [rows][cols][3].
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

struct RgbImage {
    int rows = 0, cols = 0;
    std::vector<uint8_t> rgb;  // interleaved
};

inline RgbImage read_pnm(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open image " + path);
    std::string magic;
    f >> magic;
    if (magic != "P6" && magic != "P5") throw std::runtime_error("unsupported PNM type in " + path);
    auto next_int = [&f]() {
        int c;
        for (;;) {
            c = f.peek();
            if (c == '#') { std::string l; std::getline(f, l); }
            else if (std::isspace(c)) f.get();
            else break;
        }
        int v; f >> v; return v;
    };
    RgbImage im;
    im.cols = next_int(); im.rows = next_int();
    const int maxv = next_int();
    if (maxv != 255) throw std::runtime_error("only 8-bit PNM supported");
    f.get();
    const size_t n = (size_t)im.rows * im.cols;
    im.rgb.resize(n * 3);
    if (magic == "P6") f.read((char*)im.rgb.data(), n * 3);
    else {
        std::vector<uint8_t> g(n);
        f.read((char*)g.data(), n);
        for (size_t i = 0; i < n; ++i) im.rgb[3 * i] = im.rgb[3 * i + 1] = im.rgb[3 * i + 2] = g[i];
    }
    if (!f) throw std::runtime_error("short read in " + path);
    return im;
}

inline void write_ppm(const std::string& path, int rows, int cols, const uint8_t* rgb_interleaved)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    f << "P6\n" << cols << " " << rows << "\n255\n";
    f.write((const char*)rgb_interleaved, (size_t)rows * cols * 3);
}

inline RgbImage read_png(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open image " + path);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8)) throw std::runtime_error("not a PNG: " + path);
    auto be32 = [&](size_t o) { return (uint32_t)d[o] << 24 | (uint32_t)d[o + 1] << 16 | (uint32_t)d[o + 2] << 8 | d[o + 3]; };
    size_t o = 8;
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (o + 8 <= d.size()) {
        const uint32_t len = be32(o);
        const std::string type((const char*)&d[o + 4], 4);
        const size_t body = o + 8;
        if (body + len + 4 > d.size()) throw std::runtime_error("truncated PNG");
        if (type == "IHDR") { W = be32(body); H = be32(body + 4); depth = d[body + 8]; ctype = d[body + 9]; interlace = d[body + 12]; }
        else if (type == "IDAT") idat.insert(idat.end(), d.begin() + body, d.begin() + body + len);
        else if (type == "IEND") break;
        o = body + len + 4;
    }
    if (depth != 8 || interlace != 0 || (ctype != 0 && ctype != 2 && ctype != 6 && ctype != 4))
        throw std::runtime_error("unsupported PNG format (need 8-bit, non-interlaced, grey/RGB/RGBA)");
    const int bpp = ctype == 0 ? 1 : (ctype == 4 ? 2 : (ctype == 2 ? 3 : 4));
    const size_t stride = (size_t)W * bpp;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf outlen = raw.size();
    if (uncompress(raw.data(), &outlen, idat.data(), idat.size()) != Z_OK || outlen != raw.size()) throw std::runtime_error("PNG inflate failed");
    std::vector<uint8_t> img(stride * H);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* in = &raw[y * (stride + 1) + 1];
        uint8_t* out = &img[y * stride];
        const uint8_t* up = y ? &img[(y - 1) * stride] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error("bad PNG filter");
            }
            out[x] = (uint8_t)(in[x] + pred);
        }
    }
    RgbImage im;
    im.rows = (int)H; im.cols = (int)W;
    im.rgb.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const uint8_t* p = &img[i * bpp];
        if (ctype == 0 || ctype == 4) im.rgb[3 * i] = im.rgb[3 * i + 1] = im.rgb[3 * i + 2] = p[0];
        else { im.rgb[3 * i] = p[0]; im.rgb[3 * i + 1] = p[1]; im.rgb[3 * i + 2] = p[2]; }
    }
    return im;
}

inline RgbImage read_image(const std::string& path)
{
    const auto dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == "png") return read_png(path);
    return read_pnm(path);
}
