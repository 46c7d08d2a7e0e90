// png_writer.cpp — Canvas::write_image (canvas.rs:39-53): 8-bit RGB, no alpha, no gamma.
// The reference encodes through image 0.24.1 / png 0.17.5 (Cargo.lock:235,490); encoder bytes
// are not pinned by anything, decoded pixels are — any lossless PNG is pixel-equal.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_world.h"

namespace rth {
namespace {
void be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void chunk(std::vector<unsigned char>& out, const char* type, const unsigned char* data, size_t n) {
    be32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    uint32_t c = crc32(0L, out.data() + start, (uInt)(n + 4));
    be32(out, c);
}
}  // namespace

bool write_png(const std::string& path, const uint8_t* rgb, int w, int h, std::string& err) {
    if (w <= 0 || h <= 0 || !rgb) { err = "empty image"; return false; }
    std::vector<unsigned char> raw((size_t)h * (1 + (size_t)w * 3));
    for (int y = 0; y < h; y++) {
        unsigned char* row = raw.data() + (size_t)y * (1 + (size_t)w * 3);
        row[0] = 0;                                  // filter: None
        std::memcpy(row + 1, rgb + (size_t)y * w * 3, (size_t)w * 3);
    }
    uLongf zn = compressBound(raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), raw.size(), 6) != Z_OK) { err = "zlib compress failed"; return false; }
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zn);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path; return false; }
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) err = "short write to " + path;
    return ok;
}
}  // namespace rth
