// image_io.cpp -- texture decode (PNM, PNG here; JPEG in jpeg_decode.cpp; BMP and TGA in bmp_tga_decode.cpp) to 8-bit RGB, and the P6 / PNG writers.
//
// The reference decodes textures with its vendored stb_image forced to 3 channels
// (src/gpu_scene_builder.cpp:215) and writes frames as binary PPM (src/gpu_render.cu:1099-1107).
// This file is an independent implementation of the container formats we can support without
// third-party code: binary/ASCII PNM and PNG, interlaced or not (inflate through the system zlib).  What is left out of the reference
// decoder's list -- GIF, PSD, PIC, Radiance HDR, CMYK JPEG -- is a load failure, for which the reference's own behaviour is a 1x1 white
// texture plus a warning (src/gpu_scene_builder.cpp:216-221); the builder does the same and reports it (dsrt_host_scene_texture_failures).
// Channel handling matches stb's req_comp = 3: gray is replicated, alpha is dropped, 16-bit keeps the
// high byte, palette entries are expanded.
#include "host_internal.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace dsrt {

namespace {

// Largest texture accepted (pixels).  Headers come from untrusted files: sizes are checked against this and against the
// payload actually present before any buffer is sized from them.
constexpr unsigned long long kMaxTexturePixels = 1ull << 28;


bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    in.seekg(0, std::ios::end);
    std::streamoff n = in.tellg();
    if (n < 0) return false;
    in.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0) in.read((char*)out.data(), n);
    return (bool)in || in.eof();
}

// ---------------- PNM ----------------
struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    void skip_space_and_comments() {
        for (;;) {
            while (p < end && std::isspace(*p)) ++p;
            if (p < end && *p == '#') { while (p < end && *p != '\n') ++p; continue; }
            break;
        }
    }
    bool number(int& v) {
        skip_space_and_comments();
        if (p >= end || !std::isdigit(*p)) return false;
        long acc = 0;
        while (p < end && std::isdigit(*p)) { acc = acc * 10 + (*p - '0'); if (acc > (1 << 30)) return false; ++p; }
        v = (int)acc;
        return true;
    }
};

bool decode_pnm(const std::vector<uint8_t>& f, RgbImage& img) {
    if (f.size() < 3 || f[0] != 'P') return false;
    const int kind = f[1] - '0';
    if (kind != 2 && kind != 3 && kind != 5 && kind != 6) return false;
    Cursor c{f.data() + 2, f.data() + f.size()};
    int w, h, maxv;
    if (!c.number(w) || !c.number(h) || !c.number(maxv)) return false;
    if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 65535) return false;
    if ((unsigned long long)w * (unsigned long long)h > kMaxTexturePixels) return false;
    const int ch = (kind == 3 || kind == 6) ? 3 : 1;
    const size_t count = (size_t)w * h * ch;
    {   // the payload must be able to fill the claimed size BEFORE anything is allocated for it
        const size_t left = (size_t)(c.end - c.p);
        const size_t need = (kind == 5 || kind == 6) ? count * (maxv > 255 ? 2 : 1) : (count ? 2 * count - 1 : 0);   // ASCII: a digit and a separator per sample
        if (left < need) return false;
    }
    std::vector<uint8_t> raw(count);
    if (kind == 5 || kind == 6) {
        if (c.p < c.end) ++c.p;                       // the single whitespace after maxval
        const size_t bps = maxv > 255 ? 2 : 1;
        if ((size_t)(c.end - c.p) < count * bps) return false;
        for (size_t i = 0; i < count; ++i) raw[i] = c.p[i * bps];   // 16-bit samples are big-endian: high byte first
    } else {
        for (size_t i = 0; i < count; ++i) {
            int v;
            if (!c.number(v)) return false;
            raw[i] = (uint8_t)(maxv > 255 ? (v >> 8) : v);
        }
    }
    img.width = w;
    img.height = h;
    img.rgb.resize((size_t)w * h * 3);
    for (size_t px = 0; px < (size_t)w * h; ++px)
        for (int k = 0; k < 3; ++k) img.rgb[px * 3 + k] = raw[px * ch + (ch == 3 ? k : 0)];
    return true;
}

// ---------------- PNG ----------------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

bool decode_png(const std::vector<uint8_t>& f, RgbImage& img) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool seen_end = false;
    while (pos + 12 <= f.size() && !seen_end) {
        const uint32_t len = be32(&f[pos]);
        const uint8_t* tag = &f[pos + 4];
        if (pos + 12 + (size_t)len > f.size()) return false;
        const uint8_t* body = &f[pos + 8];
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len < 13) return false;
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(tag, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(tag, "IEND", 4)) seen_end = true;
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24) || interlace > 1) return false;
    if ((unsigned long long)w * (unsigned long long)h > kMaxTexturePixels) return false;
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break;
                     case 4: channels = 2; break; case 6: channels = 4; break; default: return false; }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    const size_t bits_pp = (size_t)channels * depth;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // The image is one pass, or (interlace method 1, Adam7) seven: pass p holds the pixels (x0 + i * dx, y0 + j * dy), as an image of its
    // own with its own scanlines and filter state; empty passes take no bytes.
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole[1] = {{0, 0, 1, 1}};
    const Pass* passes = interlace ? adam7 : whole;
    const int n_passes = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int p = 0; p < n_passes; ++p) {
        const uint32_t pw = (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx, ph = (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
        if (w <= passes[p].x0 || h <= passes[p].y0 || pw == 0 || ph == 0) continue;
        raw_size += (((size_t)pw * bits_pp + 7) / 8 + 1) * ph;
    }
    // deflate expands by at most ~1032:1: an IDAT stream that cannot produce the claimed image is rejected before allocating
    if ((unsigned long long)raw_size > (unsigned long long)idat.size() * 1032ull + 1024ull) return false;
    std::vector<uint8_t> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;

    img.width = (int)w;
    img.height = (int)h;
    img.rgb.resize((size_t)w * h * 3);
    size_t at = 0;
    for (int p = 0; p < n_passes; ++p) {
        const Pass& ps = passes[p];
        if (w <= ps.x0 || h <= ps.y0) continue;
        const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw == 0 || ph == 0) continue;
        const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (uint32_t y = 0; y < ph; ++y) {
            const uint8_t* line = &raw[at];
            at += stride + 1;
            const int filter = line[0];
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int v = line[1 + i];
                switch (filter) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break;
                                  case 4: v += paeth(a, b, c); break; default: return false; }
                cur[i] = (uint8_t)v;
            }
            for (uint32_t x = 0; x < pw; ++x) {
                uint8_t s[4] = {0, 0, 0, 0};
                for (int k = 0; k < channels; ++k) {
                    if (depth >= 8) s[k] = cur[(x * channels + k) * (depth / 8)];
                    else {
                        const size_t bit = (size_t)x * depth;
                        const int v = (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
                        s[k] = ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
                    }
                }
                uint8_t* o = &img.rgb[((size_t)(ps.y0 + y * ps.dy) * w + (ps.x0 + x * ps.dx)) * 3];
                if (ctype == 3) {
                    const size_t e = (size_t)s[0] * 3;
                    if (e + 2 < palette.size()) { o[0] = palette[e]; o[1] = palette[e + 1]; o[2] = palette[e + 2]; }
                    else o[0] = o[1] = o[2] = 0;
                } else if (channels <= 2) o[0] = o[1] = o[2] = s[0];
                else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
            }
            prev.swap(cur);
        }
    }
    return true;
}

}  // namespace

bool load_rgb8(const std::string& path, bool flip_vertically, RgbImage& img) {
    std::vector<uint8_t> f;
    if (!read_file(path, f)) return false;
    // the reference decoder's own order of attempts: PNG, BMP, (GIF, PSD, PIC: not supported here), JPEG, PNM, (HDR), TGA last -- it has no signature
    if (!decode_png(f, img) && !decode_bmp(f, img) && !decode_jpeg(f, img) && !decode_pnm(f, img) && !decode_tga(f, img)) return false;
    if (flip_vertically) {
        const size_t row = (size_t)img.width * 3;
        std::vector<uint8_t> tmp(row);
        for (int y = 0; y < img.height / 2; ++y) {
            uint8_t* a = &img.rgb[(size_t)y * row];
            uint8_t* b = &img.rgb[(size_t)(img.height - 1 - y) * row];
            std::memcpy(tmp.data(), a, row); std::memcpy(a, b, row); std::memcpy(b, tmp.data(), row);
        }
    }
    return true;
}

}  // namespace dsrt

extern "C" int dsrt_decode_image_file(const char* path, int flip_vertically, int* width, int* height, uint8_t* rgb, size_t cap) {
    if (!path || !width || !height) { dsrt::set_error("dsrt_decode_image_file: null argument"); return DSRT_ERR_INVALID; }
    return dsrt::guarded("dsrt_decode_image_file", [&]() -> int {
        dsrt::RgbImage img;
        if (!dsrt::load_rgb8(path, flip_vertically != 0, img)) { dsrt::set_error(std::string("cannot decode ") + path); return DSRT_ERR_IO; }
        *width = img.width; *height = img.height;
        if (rgb) {
            if (cap < img.rgb.size()) { dsrt::set_error("dsrt_decode_image_file: buffer too small"); return DSRT_ERR_INVALID; }
            std::memcpy(rgb, img.rgb.data(), img.rgb.size());
        }
        return DSRT_OK;
    });
}

extern "C" int dsrt_write_ppm(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) { dsrt::set_error("dsrt_write_ppm: bad argument"); return DSRT_ERR_INVALID; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { dsrt::set_error(std::string("cannot open ") + path + " for writing"); return DSRT_ERR_IO; }
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    const size_t n = (size_t)width * height * 3;
    const size_t done = std::fwrite(rgb, 1, n, f);
    std::fclose(f);
    if (done != n) { dsrt::set_error(std::string("short write to ") + path); return DSRT_ERR_IO; }
    return DSRT_OK;
}

// PNG writer (8-bit RGB, filter 0, one zlib stream): the reference shells out to ImageMagick to turn its PPM into a PNG
// (src/main.cpp:28-36); this is that step without the shell.
extern "C" int dsrt_write_png(const char* path, const uint8_t* rgb, int width, int height) {
    return dsrt::guarded("dsrt_write_png", [&]() -> int {
    if (!path || !rgb || width <= 0 || height <= 0) { dsrt::set_error("dsrt_write_png: bad argument"); return DSRT_ERR_INVALID; }
    const size_t row = (size_t)width * 3;
    std::vector<uint8_t> raw((row + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0;                                              // filter type: none
        std::memcpy(&raw[(row + 1) * y + 1], rgb + row * y, row);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) { dsrt::set_error("dsrt_write_png: zlib failed"); return DSRT_ERR_IO; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { dsrt::set_error(std::string("cannot open ") + path + " for writing"); return DSRT_ERR_IO; }
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    bool ok = true;
    auto chunk = [&](const char* type, const uint8_t* data, size_t n) {
        uint8_t head[8];
        be32(head, (uint32_t)n);
        std::memcpy(head + 4, type, 4);
        uLong crc = crc32(0L, head + 4, 4);
        if (n) crc = crc32(crc, data, (uInt)n);
        uint8_t tail[4];
        be32(tail, (uint32_t)crc);
        ok = ok && std::fwrite(head, 1, 8, f) == 8 && (n == 0 || std::fwrite(data, 1, n, f) == n) && std::fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    ok = std::fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    be32(ihdr, (uint32_t)width); be32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;     // 8 bits, colour type 2 (RGB), deflate, adaptive filters, no interlace
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (size_t)zn);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    if (!ok) { dsrt::set_error(std::string("short write to ") + path); return DSRT_ERR_IO; }
    return DSRT_OK;
    });
}
