// png_writer.cpp — Canvas::write_image (canvas.rs:39-53): 8-bit RGB, no alpha, no gamma.
// The reference encodes through image 0.24.1 / png 0.17.5 (Cargo.lock:235,490); encoder bytes
// are not pinned by anything, decoded pixels are — any lossless PNG is pixel-equal.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
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
    // The zlib stream is built from independently deflated row blocks, one thread each (an 8K frame is 100 MB of scanlines:
    // one thread at level 6 takes seconds, the render 11 ms).  Every block but the last ends with a sync flush (byte-aligned,
    // no final bit), the last with the final block, so the concatenation is ONE valid raw deflate stream; the zlib header
    // in front and the Adler-32 of all scanlines behind frame it.  Each block becomes an IDAT chunk of its own (consecutive
    // IDAT chunks are one stream by definition), with its CRC computed on the block's thread; rows are fed to deflate
    // straight from the image (filter byte 0 = None, then the row), so no copy of the frame is made.
    const size_t row = (size_t)w * 3;
    unsigned hw = std::thread::hardware_concurrency();
    int n_blocks = (int)std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 16), std::max<size_t>(1, (size_t)h * (row + 1) / (1u << 20)));
    if (n_blocks > h) n_blocks = h;
    struct Block { std::vector<unsigned char> data; uLong adler = 1; size_t raw_len = 0; uint32_t crc = 0; bool ok = false; };
    std::vector<Block> blocks(n_blocks);
    auto work = [&](int b) {
        Block& B = blocks[b];
        const size_t r0 = (size_t)h * b / n_blocks, r1 = (size_t)h * (b + 1) / n_blocks;
        B.raw_len = (r1 - r0) * (row + 1);
        z_stream zs; std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 3, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
        const size_t head = 4 + (b == 0 ? 2 : 0);                       // chunk type (+ the zlib header in the first block)
        B.data.resize(head + deflateBound(&zs, (uLong)B.raw_len) + 64);
        std::memcpy(B.data.data(), "IDAT", 4);
        if (b == 0) { B.data[4] = 0x78; B.data[5] = 0x5e; }             // deflate, 32K window, no dictionary (0x785e % 31 == 0)
        zs.next_out = B.data.data() + head; zs.avail_out = (uInt)(B.data.size() - head);
        bool good = true;
        unsigned char filter = 0;
        uLong ad = adler32(0L, Z_NULL, 0);
        for (size_t y = r0; y < r1 && good; y++) {
            const unsigned char* src = rgb + y * row;
            ad = adler32(ad, &filter, 1);
            for (size_t o = 0; o < row;) { const uInt k = (uInt)std::min<size_t>(row - o, 1u << 30); ad = adler32(ad, src + o, k); o += k; }
            zs.next_in = &filter; zs.avail_in = 1;
            good = deflate(&zs, Z_NO_FLUSH) == Z_OK;
            for (size_t o = 0; o < row && good;) {
                const uInt k = (uInt)std::min<size_t>(row - o, 1u << 30);
                zs.next_in = const_cast<unsigned char*>(src + o); zs.avail_in = k; o += k;
                const bool last = y + 1 == r1 && o == row;
                const int flush = !last ? Z_NO_FLUSH : (b + 1 < n_blocks ? Z_SYNC_FLUSH : Z_FINISH);
                const int rc = deflate(&zs, flush);
                good = (flush == Z_FINISH) ? rc == Z_STREAM_END : rc == Z_OK;
            }
        }
        B.data.resize(head + zs.total_out);
        deflateEnd(&zs);
        B.adler = ad;
        uLong c = crc32(0L, Z_NULL, 0);
        for (size_t o = 0; o < B.data.size();) { const uInt k = (uInt)std::min<size_t>(B.data.size() - o, 1u << 30); c = crc32(c, B.data.data() + o, k); o += k; }
        B.crc = (uint32_t)c;
        B.ok = good;
    };
    {
        std::vector<std::thread> th;
        for (int b = 1; b < n_blocks; b++) th.emplace_back(work, b);
        work(0);
        for (auto& t : th) t.join();
    }
    uLong ad = blocks[0].adler;
    for (int b = 0; b < n_blocks; b++) {
        if (!blocks[b].ok) { err = "zlib deflate failed"; return false; }
        if (b > 0) ad = adler32_combine(ad, blocks[b].adler, (z_off_t)blocks[b].raw_len);
    }
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    std::vector<unsigned char> tail;                                      // the stream's Adler-32 as a last, 4-byte IDAT; IEND
    { std::vector<unsigned char> a4; be32(a4, (uint32_t)ad); chunk(tail, "IDAT", a4.data(), a4.size()); }
    chunk(tail, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path; return false; }
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    for (int b = 0; b < n_blocks && ok; b++) {
        std::vector<unsigned char> len4, crc4;
        be32(len4, (uint32_t)(blocks[b].data.size() - 4)); be32(crc4, blocks[b].crc);
        ok = std::fwrite(len4.data(), 1, 4, f) == 4 && std::fwrite(blocks[b].data.data(), 1, blocks[b].data.size(), f) == blocks[b].data.size() &&
             std::fwrite(crc4.data(), 1, 4, f) == 4;
    }
    ok = ok && std::fwrite(tail.data(), 1, tail.size(), f) == tail.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) err = "short write to " + path;
    return ok;
}
}  // namespace rth
