// bmp_tga_decode.cpp -- Windows BMP and Truevision TGA textures to 8-bit RGB.
//
// The reference hands every texture path to its vendored stb_image with three channels forced (src/gpu_scene_builder.cpp:215), so what
// a file "means" is whatever that decoder makes of it.  These two decoders are independent code written against the formats, shaped so
// that every case stb_image v2.30 accepts gives the same bytes (tests/golden/ref_stb_decode_images.json holds the reference build's own
// decodes of tests/golden/assets/images/*) and every case it rejects is rejected:
//   BMP  info-header sizes 12 / 40 / 56 / 108 / 124; 1, 4, 8 bits with a palette; 16 and 32 bits with the default or BITFIELDS masks,
//        a channel of n <= 8 bits widened by bit replication; 24 bits; bottom-up unless the height is negative; RLE and embedded
//        JPEG / PNG are load failures (as there).
//   TGA  types 1 / 2 / 3 and their run-length forms 9 / 10 / 11; 8-bit gray, 16-bit gray + alpha, 15 / 16-bit 5-5-5 (c * 255 / 31),
//        24 and 32 bits, 8- or 16-bit indices into a 15 / 16 / 24 / 32-bit colour map; bottom-up unless descriptor bit 5 is set.
// Bytes missing at the end of a file read as zero, which is what the reference's reader returns past the end of its stream.
#include "host_internal.hpp"

#include <cstdlib>
#include <cstring>

namespace dsrt {

namespace {

constexpr unsigned long long kMaxPixels = 1ull << 28;          // as image_io.cpp: sizes come from untrusted headers
constexpr int kMaxDim = 1 << 24;                               // the reference decoder's own limit on either dimension

struct Bytes {
    const uint8_t* p;
    size_t n, at = 0;
    bool past_end = false;
    int u8() { if (at < n) return p[at++]; past_end = true; return 0; }
    int u16() { const int lo = u8(); return lo | (u8() << 8); }
    uint32_t u32() { const uint32_t lo = (uint32_t)u16(); return lo | ((uint32_t)u16() << 16); }
    void skip(long long k) { if (k < 0) at = n; else at = (size_t)k > n - at ? n : at + (size_t)k; }     // a negative skip lands on the end
};

int top_bit(uint32_t m) { int b = -1; while (m) { ++b; m >>= 1; } return b; }
int count_bits(uint32_t m) { int c = 0; while (m) { c += (int)(m & 1u); m >>= 1; } return c; }

// A channel of `bits` bits, cut out of v by `mask` whose highest bit is `top`, widened to 8 bits by repeating its bit pattern.
uint8_t widen_channel(uint32_t v, uint32_t mask, int top, int bits) {
    v &= mask;
    const int shift = top - 7;
    v = shift < 0 ? v << -shift : v >> shift;                  // the field's highest bit is now bit 7
    v &= 0xFFu;
    if (bits <= 0) return 0;
    const uint32_t field = v >> (8 - bits);
    uint32_t out = 0;
    for (int filled = 0; filled < 8; filled += bits) {          // append copies of the field until 8 bits are full
        const int room = 8 - filled;
        out |= room >= bits ? field << (room - bits) : field >> (bits - room);
    }
    return (uint8_t)out;
}

}  // namespace

bool decode_bmp(const std::vector<uint8_t>& f, RgbImage& img) {
    Bytes s{f.data(), f.size()};
    if (s.u8() != 'B' || s.u8() != 'M') return false;
    s.u32(); s.u16(); s.u16();
    const long long offset = (int32_t)s.u32();
    const int hsz = (int)s.u32();
    if (offset < 0) return false;
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return false;
    long long w, h;
    if (hsz == 12) { w = s.u16(); h = s.u16(); }
    else { w = (int32_t)s.u32(); h = (int32_t)s.u32(); }
    if (s.u16() != 1) return false;                              // planes
    const int bpp = s.u16();
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
    long long read = 14;                                         // bytes consumed before the info header (+ 12 for trailing masks)
    auto default_masks = [&]() {
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; ma = 0; }          // (ma keeps what a V4 / V5 header set: see below)
        else if (bpp == 32) { mr = 0xFFu << 16; mg = 0xFFu << 8; mb = 0xFFu; ma = 0xFFu << 24; }
        else { mr = mg = mb = ma = 0; }
    };
    if (hsz != 12) {
        const int compress = (int)s.u32();
        if (compress == 1 || compress == 2 || compress >= 4 || compress < 0) return false;           // RLE, embedded JPEG / PNG
        if (compress == 3 && bpp != 16 && bpp != 32) return false;
        for (int i = 0; i < 5; ++i) s.u32();                     // image size, resolutions, colours used / important
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) for (int i = 0; i < 4; ++i) s.u32();
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) default_masks();
                else {                                           // BITFIELDS: three masks follow the header
                    mr = s.u32(); mg = s.u32(); mb = s.u32();
                    read += 12;
                    if (mr == mg && mg == mb) return false;
                }
            }
        } else {
            mr = s.u32(); mg = s.u32(); mb = s.u32(); ma = s.u32();
            if (compress != 3) {
                const uint32_t keep = ma;
                default_masks();
                if (bpp == 16) ma = keep;                        // the 16-bit defaults leave the alpha mask as read
            }
            s.u32();
            for (int i = 0; i < 12; ++i) s.u32();
            if (hsz == 124) for (int i = 0; i < 4; ++i) s.u32();
        }
    }
    const bool bottom_up = h > 0;
    if (h < 0) h = -h;
    if (w > kMaxDim || h > kMaxDim || w < 0) return false;
    long long psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (offset - read - 24) / 3; }
    else if (bpp < 16) psize = (offset - read - hsz) >> 2;
    if (psize == 0) {
        const long long so_far = (long long)s.at;
        if (so_far <= 0 || so_far > 1024) return false;
        if (offset < so_far || offset - so_far > 1024) return false;
        s.skip(offset - so_far);
    }
    if (w == 0 || h == 0) return false;                          // nothing a texture pool could hold
    if ((unsigned long long)w * (unsigned long long)h > kMaxPixels) return false;
    // a file cannot hold fewer than one bit per pixel: refuse to size a buffer a header merely claims
    if ((unsigned long long)w * (unsigned long long)h / 8ull > (unsigned long long)f.size() + 64ull) return false;
    img.width = (int)w; img.height = (int)h;
    img.rgb.assign((size_t)w * (size_t)h * 3, 0);
    size_t z = 0;
    if (bpp < 16) {
        if (psize <= 0 || psize > 256) return false;
        uint8_t pal[256][3];
        std::memset(pal, 0, sizeof pal);
        for (long long i = 0; i < psize; ++i) {
            pal[i][2] = (uint8_t)s.u8(); pal[i][1] = (uint8_t)s.u8(); pal[i][0] = (uint8_t)s.u8();
            if (hsz != 12) s.u8();
        }
        s.skip(offset - read - hsz - psize * (hsz == 12 ? 3 : 4));
        long long row_bytes;
        if (bpp == 1) row_bytes = (w + 7) >> 3; else if (bpp == 4) row_bytes = (w + 1) >> 1; else if (bpp == 8) row_bytes = w; else return false;
        const long long pad = (-row_bytes) & 3;
        for (long long j = 0; j < h; ++j) {
            if (bpp == 1) {
                int bit = 7, v = s.u8();
                for (long long i = 0; i < w; ++i) {
                    const int c = (v >> bit) & 1;
                    img.rgb[z++] = pal[c][0]; img.rgb[z++] = pal[c][1]; img.rgb[z++] = pal[c][2];
                    if (i + 1 == w) break;
                    if (--bit < 0) { bit = 7; v = s.u8(); }
                }
            } else {
                for (long long i = 0; i < w; i += 2) {
                    int v = s.u8(), v2 = 0;
                    if (bpp == 4) { v2 = v & 15; v >>= 4; }
                    img.rgb[z++] = pal[v][0]; img.rgb[z++] = pal[v][1]; img.rgb[z++] = pal[v][2];
                    if (i + 1 == w) break;
                    v = bpp == 8 ? s.u8() : v2;
                    img.rgb[z++] = pal[v][0]; img.rgb[z++] = pal[v][1]; img.rgb[z++] = pal[v][2];
                }
            }
            s.skip(pad);
        }
    } else {
        s.skip(offset - read - hsz);
        const long long row_bytes = bpp == 24 ? 3 * w : (bpp == 16 ? 2 * w : 0);
        const long long pad = (-row_bytes) & 3;
        int easy = 0;
        if (bpp == 24) easy = 1;
        else if (bpp == 32 && mb == 0xFFu && mg == 0xFF00u && mr == 0x00FF0000u && ma == 0xFF000000u) easy = 2;
        int tr = 0, tg = 0, tb = 0, cr = 0, cg = 0, cb = 0;
        if (!easy) {
            if (!mr || !mg || !mb) return false;
            tr = top_bit(mr); cr = count_bits(mr); tg = top_bit(mg); cg = count_bits(mg); tb = top_bit(mb); cb = count_bits(mb);
            if (cr > 8 || cg > 8 || cb > 8 || count_bits(ma) > 8) return false;
        }
        for (long long j = 0; j < h; ++j) {
            if (easy) {
                for (long long i = 0; i < w; ++i) {
                    img.rgb[z + 2] = (uint8_t)s.u8(); img.rgb[z + 1] = (uint8_t)s.u8(); img.rgb[z + 0] = (uint8_t)s.u8();
                    z += 3;
                    if (easy == 2) s.u8();
                }
            } else {
                for (long long i = 0; i < w; ++i) {
                    const uint32_t v = bpp == 16 ? (uint32_t)s.u16() : s.u32();
                    img.rgb[z++] = widen_channel(v, mr, tr, cr);
                    img.rgb[z++] = widen_channel(v, mg, tg, cg);
                    img.rgb[z++] = widen_channel(v, mb, tb, cb);
                }
            }
            s.skip(pad);
        }
    }
    if (bottom_up) {
        const size_t row = (size_t)w * 3;
        std::vector<uint8_t> tmp(row);
        for (long long j = 0; j < h / 2; ++j) {
            uint8_t* a = &img.rgb[(size_t)j * row];
            uint8_t* b = &img.rgb[(size_t)(h - 1 - j) * row];
            std::memcpy(tmp.data(), a, row); std::memcpy(a, b, row); std::memcpy(b, tmp.data(), row);
        }
    }
    return true;
}

namespace {

// channels of a TGA pixel or colour-map entry: 1 gray, 2 gray + alpha, 3 / 4 colour; 0 = not a TGA this decoder (or the reference's) takes
int tga_channels(int bits, bool gray, bool& five_bit) {
    five_bit = false;
    switch (bits) {
        case 8: return 1;
        case 16: if (gray) return 2; five_bit = true; return 3;
        case 15: five_bit = true; return 3;
        case 24: return 3;
        case 32: return 4;
        default: return 0;
    }
}

// TGA has no signature: the header has to be plausible in every field the reference's own sniff test looks at
bool tga_plausible(const std::vector<uint8_t>& f) {
    Bytes s{f.data(), f.size()};
    s.u8();
    const int map_type = s.u8();
    if (map_type > 1) return false;
    int t = s.u8();
    if (map_type == 1) {
        if (t != 1 && t != 9) return false;
        s.skip(4);
        const int eb = s.u8();
        if (eb != 8 && eb != 15 && eb != 16 && eb != 24 && eb != 32) return false;
        s.skip(4);
    } else {
        if (t != 2 && t != 3 && t != 10 && t != 11) return false;
        s.skip(9);
    }
    if (s.u16() < 1 || s.u16() < 1) return false;
    const int bits = s.u8();
    if (map_type == 1 && bits != 8 && bits != 16) return false;
    return bits == 8 || bits == 15 || bits == 16 || bits == 24 || bits == 32;
}

void five_bit_rgb(int px, uint8_t* out) {
    out[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31);
    out[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31);
    out[2] = (uint8_t)(((px & 31) * 255) / 31);
}

}  // namespace

bool decode_tga(const std::vector<uint8_t>& f, RgbImage& img) {
    if (!tga_plausible(f)) return false;
    Bytes s{f.data(), f.size()};
    const int id_len = s.u8();
    const int indexed = s.u8();
    int type = s.u8();
    const int map_start = s.u16(), map_len = s.u16(), map_bits = s.u8();
    s.u16(); s.u16();
    const int w = s.u16(), h = s.u16();
    const int bits = s.u8();
    const int descriptor = s.u8();
    bool rle = false;
    if (type >= 8) { type -= 8; rle = true; }
    const bool bottom_up = ((descriptor >> 5) & 1) == 0;
    bool five = false;
    const int comp = indexed ? tga_channels(map_bits, false, five) : tga_channels(bits, type == 3, five);
    if (!comp) return false;
    if ((unsigned long long)w * (unsigned long long)h > kMaxPixels) return false;
    // an uncompressed file holds its pixels; a run-length one at least a byte per 128 of them
    if (!rle && (unsigned long long)w * h * (indexed ? (unsigned)(bits / 8) : (unsigned)((bits + 7) / 8)) > (unsigned long long)f.size() + 64ull) return false;
    if (rle && (unsigned long long)w * h / 128ull > (unsigned long long)f.size() + 64ull) return false;
    std::vector<uint8_t> px((size_t)w * h * comp, 0);
    s.skip(id_len);
    std::vector<uint8_t> palette;
    if (indexed) {
        if (map_len == 0) return false;
        s.skip(map_start);
        palette.assign((size_t)map_len * comp, 0);
        if (five) { for (int i = 0; i < map_len; ++i) five_bit_rgb(s.u16(), &palette[(size_t)i * comp]); }
        else {
            if (s.n - s.at < palette.size()) return false;       // a colour map cut short is a load failure
            for (size_t i = 0; i < palette.size(); ++i) palette[i] = (uint8_t)s.u8();
        }
    }
    uint8_t raw[4] = {0, 0, 0, 0};
    int run = 0;
    bool repeating = false;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        bool read_pixel = true;
        if (rle) {
            if (run == 0) { const int cmd = s.u8(); run = 1 + (cmd & 127); repeating = (cmd >> 7) != 0; }
            else if (repeating) read_pixel = false;
        }
        if (read_pixel) {
            if (indexed) {
                int idx = bits == 8 ? s.u8() : s.u16();
                if (idx >= map_len) idx = 0;
                for (int j = 0; j < comp; ++j) raw[j] = palette[(size_t)idx * comp + j];
            } else if (five) five_bit_rgb(s.u16(), raw);
            else for (int j = 0; j < comp; ++j) raw[j] = (uint8_t)s.u8();
        }
        for (int j = 0; j < comp; ++j) px[i * comp + j] = raw[j];
        --run;
    }
    img.width = w; img.height = h;
    img.rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
        const int src_row = bottom_up ? h - 1 - y : y;
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = &px[((size_t)src_row * w + x) * comp];
            uint8_t* o = &img.rgb[((size_t)y * w + x) * 3];
            if (comp <= 2) o[0] = o[1] = o[2] = p[0];                           // gray (+ alpha): replicated, alpha dropped
            else if (five) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }            // 5-5-5 was unpacked as R, G, B
            else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }                      // stored blue first
        }
    }
    return true;
}

}  // namespace dsrt
