// PNG output for the host driver: stands where Film::writeFile hands its RGBA
// vector to lodepng::encode (raytracer.cpp:212-231).  8-bit RGBA, filter type 0 on
// every row, zlib for the deflate stream.
//
// The reference pushes every byte into a vector and deflates on one thread; at 8K
// that is seconds per frame against milliseconds of rendering.  Here the rows are
// cut into bands that are deflated concurrently: every band is a raw deflate
// stream ended on a byte boundary (Z_SYNC_FLUSH; the last band Z_FINISH), so the
// bands concatenate into one valid stream; the zlib header is written once and
// the Adler-32 of the whole is combined from the bands' own.  One IDAT chunk.
#include "rt_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <zlib.h>

namespace {

void be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}

struct Band {
    int y0 = 0, y1 = 0;
    std::vector<unsigned char> z;     // raw deflate bytes of the band
    uLong adler = 1;                  // Adler-32 of the band's unfiltered scanlines
    size_t raw_bytes = 0;
    bool ok = false;
};

// Deflates rows [y0, y1) (each prefixed with filter byte 0) as a raw stream.
void deflate_band(Band& b, const uint8_t* rgba, int width, bool last, int level) {
    const size_t stride = (size_t)width * 4;
    const size_t rows = (size_t)(b.y1 - b.y0);
    std::vector<unsigned char> raw((stride + 1) * rows);
    for (size_t r = 0; r < rows; r++) {
        raw[(stride + 1) * r] = 0;
        std::memcpy(&raw[(stride + 1) * r + 1], rgba + stride * ((size_t)b.y0 + r), stride);
    }
    b.raw_bytes = raw.size();
    b.adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
    b.z.resize(deflateBound(&zs, (uLong)raw.size()) + 16);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = b.z.data(); zs.avail_out = (uInt)b.z.size();
    const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    b.ok = last ? (rc == Z_STREAM_END) : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0);
    b.z.resize(b.z.size() - zs.avail_out);
    deflateEnd(&zs);
}

}  // namespace

static int write_png(const char* path, const uint8_t* rgba, int width, int height) {
    const size_t stride = (size_t)width * 4;
    if ((stride + 1) * 64 > 0x7fffffffu) return RT_ERR_RANGE;          // a band must fit zlib's 32-bit counts
    // bands of at least 64 rows (and at most 1 GiB), one thread each, at most 16 threads at a time
    const int max_rows = (int)std::max<size_t>(1, std::min<size_t>(0x40000000u / (stride + 1), 1u << 20));
    int n_bands = std::max(1, std::min(height / 64, 16));
    int rows_per_band = (height + n_bands - 1) / n_bands;
    if (rows_per_band > max_rows) { rows_per_band = max_rows; }
    n_bands = (height + rows_per_band - 1) / rows_per_band;
    std::vector<Band> bands(n_bands);
    for (int k = 0; k < n_bands; k++) {
        bands[k].y0 = k * rows_per_band;
        bands[k].y1 = std::min(height, (k + 1) * rows_per_band);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = std::max(1, std::min<int>(16, std::min<int>(n_bands, hw ? (int)hw : 4)));
    // (a band whose worker ran out of memory stays !ok; a thread that cannot be started leaves
    // its bands to the calling thread)
    auto work = [&](int t) {
        for (int k = t; k < n_bands; k += n_threads) {
            try { deflate_band(bands[k], rgba, width, k == n_bands - 1, 6); } catch (...) { bands[k].ok = false; }
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(n_threads);
    int started = 0;
    try {
        for (; started < n_threads; started++) pool.emplace_back(work, started);
    } catch (...) {}
    for (int t = started; t < n_threads; t++) work(t);
    for (auto& th : pool) th.join();

    size_t zbytes = 2 + 4;
    uLong adler = adler32(0L, Z_NULL, 0);
    for (int k = 0; k < n_bands; k++) {
        if (!bands[k].ok) return RT_ERR_ARG;
        zbytes += bands[k].z.size();
        adler = k == 0 ? bands[k].adler : adler32_combine(adler, bands[k].adler, (z_off_t)bands[k].raw_bytes);
    }
    if (zbytes > 0x7fffffffu) return RT_ERR_RANGE;                    // one IDAT chunk holds at most 2^31 - 1 bytes

    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    png.reserve(zbytes + 64);
    auto begin_chunk = [&](const char type[4], uint32_t n) {
        be32(png, n);
        png.insert(png.end(), type, type + 4);
        return png.size() - 4;                                        // where the CRC's coverage starts
    };
    auto end_chunk = [&](size_t start) {
        uLong crc = crc32(0L, Z_NULL, 0);
        for (size_t at = start; at < png.size();) {                   // crc32 takes 32-bit lengths
            const size_t n = std::min<size_t>(png.size() - at, 1u << 30);
            crc = crc32(crc, png.data() + at, (uInt)n);
            at += n;
        }
        be32(png, (uint32_t)crc);
    };
    size_t at = begin_chunk("IHDR", 13);
    be32(png, (uint32_t)width); be32(png, (uint32_t)height);
    const unsigned char tail[5] = {8, 6, 0, 0, 0};                    // bit depth 8, colour type 6 (RGBA)
    png.insert(png.end(), tail, tail + 5);
    end_chunk(at);
    at = begin_chunk("IDAT", (uint32_t)zbytes);
    png.push_back(0x78); png.push_back(0x9c);                         // zlib header: deflate, 32 KiB window, default level
    for (int k = 0; k < n_bands; k++) png.insert(png.end(), bands[k].z.begin(), bands[k].z.end());
    be32(png, (uint32_t)adler);
    end_chunk(at);
    at = begin_chunk("IEND", 0);
    end_chunk(at);

    FILE* f = fopen(path, "wb");
    if (!f) return RT_ERR_ARG;
    const size_t w = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    return w == png.size() ? 0 : RT_ERR_ARG;
}

// No exception crosses the C boundary.
extern "C" int rth_write_png(const char* path, const uint8_t* rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return RT_ERR_ARG;
    try {
        return write_png(path, rgba, width, height);
    } catch (...) {
        return RT_ERR_ARG;
    }
}
