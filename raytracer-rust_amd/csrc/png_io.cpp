// png_io.cpp -- PNG files for the host mirror of the reference's image helpers, behind the C ABI
// of include/rt_host.h.
//
//   rth_save_png / rth_save_png_rgb   save_image      src/imgcomparator/mod.rs:150-166
//   rth_load_png                      file_to_image   src/imgcomparator/mod.rs:128-148
//
// The reference leaves both to the `image` crate.  Here the writer compresses bands of rows in
// parallel (independent raw deflate streams ended on a byte boundary and concatenated, the Adler-32
// sums combined), because at 7680x4320 a serial encoder costs far more than the render.  The
// reader handles what the reference's fixtures and this writer produce: 8-bit (and 16-bit, high
// byte) grey, RGB, palette, grey+alpha and RGBA, not interlaced.
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_host.h"
#include "errors.h"

namespace {

int fail(const std::string& m) { rte::set_error(m); return 1; }

void put_u32(std::vector<uint8_t>* v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v->push_back((uint8_t)(x >> s)); }
uint32_t get_u32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool write_chunk(FILE* f, const char type[4], const uint8_t* data, size_t n) {
    uint8_t head[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    uLong crc = crc32(0L, head + 4, 4);
    // crc32 takes a 32-bit length: feed large chunks in pieces.
    for (size_t off = 0; off < n;) { size_t k = std::min<size_t>(n - off, 1u << 30); crc = crc32(crc, data + off, (uInt)k); off += k; }
    uint8_t tail[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    return fwrite(head, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
}

struct Band { std::vector<uint8_t> deflated; uLong adler = 1; size_t raw_bytes = 0; bool ok = true; };

// row(y, dst) writes the 3 * width RGB bytes of row y.
template <class RowFn>
int save_png(const char* path, uint32_t width, uint32_t height, RowFn row) {
    if (!path) return fail("path is NULL");
    if (width == 0 || height == 0) return fail("Width and height must be greater than zero");
    const size_t stride = 1 + 3 * (size_t)width;                  // filter byte + pixels
    // Bands of at least 256 KiB of scanlines, at most 256 of them.
    uint32_t rows_per_band = (uint32_t)std::max<size_t>(1, (256u << 10) / stride);
    rows_per_band = std::max(rows_per_band, (height + 255u) / 256u);
    const long nbands = (long)((height + rows_per_band - 1) / rows_per_band);
    std::vector<Band> bands((size_t)nbands);
#pragma omp parallel for schedule(dynamic, 1) if (nbands > 1)
    for (long b = 0; b < nbands; b++) {
        Band& band = bands[(size_t)b];
        const uint32_t y0 = (uint32_t)b * rows_per_band, y1 = std::min(height, y0 + rows_per_band);
        std::vector<uint8_t> raw((size_t)(y1 - y0) * stride);
        for (uint32_t y = y0; y < y1; y++) {
            uint8_t* line = raw.data() + (size_t)(y - y0) * stride;
            line[0] = 0;                                          // filter type None
            row(y, line + 1);
        }
        band.raw_bytes = raw.size();
        band.adler = adler32(1L, raw.data(), (uInt)raw.size());   // a band is far below 4 GiB
        z_stream zs{};
        if (deflateInit2(&zs, 3, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { band.ok = false; continue; }
        band.deflated.resize(deflateBound(&zs, (uLong)raw.size()) + 16);
        zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
        zs.next_out = band.deflated.data(); zs.avail_out = (uInt)band.deflated.size();
        const bool last = b == nbands - 1;
        int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);   // sync flush: ends on a byte boundary, no final block
        band.ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0);
        band.deflated.resize(zs.total_out);
        deflateEnd(&zs);
    }
    uLong adler = 1; bool first = true;
    for (const Band& b : bands) {
        if (!b.ok) return fail("deflate failed");
        adler = first ? b.adler : adler32_combine(adler, b.adler, (z_off_t)b.raw_bytes);
        first = false;
    }
    FILE* f = fopen(path, "wb");
    if (!f) return fail(std::string(std::strerror(errno)) + ": " + path);
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_u32(&ihdr, width); put_u32(&ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB, deflate, adaptive, no interlace
    bool ok = fwrite(SIG, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr.data(), ihdr.size());
    // One IDAT chunk per band; the zlib header rides in the first, the Adler-32 in the last.
    for (long b = 0; ok && b < nbands; b++) {
        std::vector<uint8_t> data;
        if (b == 0) { data.push_back(0x78); data.push_back(0x5E); }
        data.insert(data.end(), bands[(size_t)b].deflated.begin(), bands[(size_t)b].deflated.end());
        if (b == nbands - 1) put_u32(&data, (uint32_t)adler);
        ok = write_chunk(f, "IDAT", data.data(), data.size());
    }
    ok = ok && write_chunk(f, "IEND", nullptr, 0);
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(std::string("cannot write ") + path);
    return 0;
}

uint8_t paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

}  // namespace

extern "C" {

int rth_save_png(const char* path, const uint32_t* pixels, uint32_t width, uint32_t height) {
    if (!pixels) return fail("rth_save_png: NULL image");
    return save_png(path, width, height, [&](uint32_t y, uint8_t* dst) {
        const uint32_t* src = pixels + (size_t)y * width;
        for (uint32_t x = 0; x < width; x++) {                                             // mod.rs:156-162
            dst[3 * x] = (uint8_t)(src[x] >> 16); dst[3 * x + 1] = (uint8_t)(src[x] >> 8); dst[3 * x + 2] = (uint8_t)src[x];
        }
    });
}

int rth_save_png_rgb(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    if (!rgb) return fail("rth_save_png_rgb: NULL image");
    return save_png(path, width, height, [&](uint32_t y, uint8_t* dst) { std::memcpy(dst, rgb + 3 * (size_t)y * width, 3 * (size_t)width); });
}

int rth_load_png(const char* path, uint32_t** pixels, uint32_t* width, uint32_t* height) {
    if (!path || !pixels || !width || !height) return fail("rth_load_png: NULL argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(std::string(std::strerror(errno)) + " (os error " + std::to_string(errno) + "): " + path);
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof(buf), f)) > 0;) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), SIG, 8) != 0) return fail(std::string("not a PNG file: ") + path);
    uint32_t w = 0, h = 0; int depth = 0, color = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool end = false;
    for (size_t pos = 8; !end;) {
        if (pos + 12 > file.size()) return fail("PNG: truncated");
        uint32_t n = get_u32(&file[pos]);
        const uint8_t* type = &file[pos + 4]; const uint8_t* data = &file[pos + 8];
        if ((uint64_t)pos + 12 + n > file.size()) return fail("PNG: truncated chunk");
        if (crc32(crc32(0L, type, 4), data, n) != get_u32(data + n)) return fail("PNG: chunk checksum mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (n != 13) return fail("PNG: bad header");
            w = get_u32(data); h = get_u32(data + 4); depth = data[8]; color = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + n);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + n);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)n;
    }
    if (w == 0 || h == 0 || color < 0) return fail("PNG: no header");
    if (interlace != 0) return fail("PNG: interlaced files are not supported");
    int channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
    if (channels == 0 || !(depth == 8 || (depth == 16 && color != 3))) return fail("PNG: unsupported colour type or bit depth");
    const size_t bpp = (size_t)channels * (depth / 8), stride = bpp * w;
    // A deflate stream expands at most 1032 times: a header that promises more than the data can hold is damaged
    // (and must not size an allocation).
    const uint64_t raw_bytes = ((uint64_t)stride + 1) * (uint64_t)h;
    if (raw_bytes > (uint64_t)idat.size() * 1032u + 64u || raw_bytes > 0xFFFFFFFFull) return fail("PNG: image data does not match the header");
    std::vector<uint8_t> raw;
    try { raw.resize((size_t)raw_bytes); } catch (const std::exception&) { return fail("PNG: out of memory"); }
    {
        z_stream zs{};
        if (inflateInit(&zs) != Z_OK) return fail("PNG: inflate failed");
        zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size();
        zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
        int rc = inflate(&zs, Z_FINISH);
        size_t got = zs.total_out;
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || got != raw.size()) return fail("PNG: image data does not match the header");
    }
    // Undo the scanline filters in place.
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* cur = &raw[(stride + 1) * (size_t)y + 1];
        const uint8_t* up = y ? cur - (stride + 1) : nullptr;
        switch (cur[-1]) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: if (up) for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + up[i]); break;
        case 3: for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + (up ? up[i] : 0)) >> 1)); break;
        case 4: for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, up ? up[i] : 0, (up && i >= bpp) ? up[i - bpp] : 0)); break;
        default: return fail("PNG: unknown filter type");
        }
    }
    uint32_t* out = (uint32_t*)std::malloc(sizeof(uint32_t) * (size_t)w * h);
    if (!out) return fail("out of memory");
    const size_t step = (size_t)(depth / 8);                       // 16-bit samples: most significant byte first
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* line = &raw[(stride + 1) * (size_t)y + 1];
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t* p = line + bpp * x;
            uint32_t r, g, b;
            if (color == 2 || color == 6) { r = p[0]; g = p[step]; b = p[2 * step]; }
            else if (color == 3) {
                size_t i = p[0];
                if (3 * i + 2 >= palette.size()) { std::free(out); return fail("PNG: palette index out of range"); }
                r = palette[3 * i]; g = palette[3 * i + 1]; b = palette[3 * i + 2];
            } else r = g = b = p[0];
            out[(size_t)y * w + x] = (r << 16) | (g << 8) | b;     // mod.rs:136-145: alpha dropped
        }
    }
    *pixels = out; *width = w; *height = h;
    return 0;
}

}  // extern "C"
