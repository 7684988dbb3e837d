// PNG output for the host driver: stands where Film::writeFile hands its RGBA
// vector to lodepng::encode (raytracer.cpp:212-231).  8-bit RGBA, one IDAT,
// filter type 0 on every row, zlib for the deflate stream.
#include "rt_host.h"

#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>

namespace {

void be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}

void chunk(std::vector<unsigned char>& png, const char type[4], const unsigned char* data, size_t n) {
    be32(png, (uint32_t)n);
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, png.data() + start, (uInt)(n + 4));
    be32(png, crc);
}

}  // namespace

extern "C" int rth_write_png(const char* path, const uint8_t* rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return RT_ERR_ARG;
    size_t stride = (size_t)width * 4;
    std::vector<unsigned char> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; y++) {
        raw[(stride + 1) * y] = 0;
        std::memcpy(&raw[(stride + 1) * y + 1], rgba + stride * y, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return RT_ERR_ARG;
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    const unsigned char tail[5] = {8, 6, 0, 0, 0};   // bit depth 8, colour type 6 (RGBA)
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(png, "IHDR", ihdr.data(), ihdr.size());
    chunk(png, "IDAT", z.data(), zlen);
    chunk(png, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) return RT_ERR_ARG;
    size_t w = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    return w == png.size() ? 0 : RT_ERR_ARG;
}
