// jpeg_decode.cpp -- baseline and progressive JPEG (ITU T.81, JFIF) to 8-bit RGB, for texture maps.
//
// Why it exists: the reference decodes textures with its vendored stb_image forced to 3 channels (src/gpu_scene_builder.cpp:215), and
// real OBJ/MTL packages ship JPEG maps.  Decoding them "approximately" would not do: a texel that differs by one level changes the albedo,
// hence the throughput, hence (with one LCG stream per pixel) every later sample of the pixel.  So this decoder is written to the JPEG
// standard for everything the standard defines (marker syntax, Huffman coding, progressive refinement -- any conforming decoder gets the same
// coefficients), and follows the arithmetic CHOICES the reference's decoder makes where the standard leaves them open, so that the texels are
// the reference's texels (checked byte for byte against images decoded by the reference's own stb_image build: tests/golden/ref_stb_decode.json):
//   * inverse DCT: the Loeffler-Ligtenberg-Moschytz factorisation as in the IJG "islow" method, constants at 12 fractional bits, the column
//     pass keeping 2 extra bits (round at 2^9, shift 10), the row pass rounding at 2^16 and adding the +128 level shift before its shift by 17;
//   * chroma upsampling: triangle filter ("fancy upsampling") -- horizontally (3 near + far + 2) >> 2 with the end samples copied, vertically
//     the same between the nearer and the farther source row, both at once (3 (3n+f) + (3n'+f') + 8) >> 4; other ratios replicate;
//   * YCbCr -> RGB in 20-bit fixed point with constants rounded to 12 bits (1.402, 0.71414, 0.34414, 1.772), the Cb contribution to green
//     truncated to its upper 16 bits, floor by arithmetic shift, clamp.
// Supported: SOF0/SOF1 (8-bit sequential Huffman) and SOF2 (progressive), 1 or 3 components, sampling factors 1..4, 8- and 16-bit quantisation
// tables, restart intervals, Adobe APP14 "transform 0" RGB files.  Not supported (reported as a load failure): arithmetic coding, 12-bit
// samples, lossless, 4-component CMYK/YCCK.
#include "host_internal.hpp"

#include <cstring>

namespace dsrt {
namespace {

constexpr unsigned long long kMaxJpegPixels = 1ull << 28;

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t symbols[256];
    int first_code[18], first_index[18], count[18];     // per code length 1..16 (canonical code, T.81 Annex C)
    void build(const uint8_t counts[16], const uint8_t* syms) {
        int code = 0, index = 0;
        for (int len = 1; len <= 16; ++len) {
            first_code[len] = code; first_index[len] = index; count[len] = counts[len - 1];
            code = (code + counts[len - 1]) << 1;
            index += counts[len - 1];
        }
        std::memcpy(symbols, syms, (size_t)index);
        present = true;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int have = 0;
    bool hit_marker = false;
    void fill() {
        while (have <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;             // stuffed zero
                    else { hit_marker = true; byte = 0; }                 // a marker: the entropy-coded segment ends, feed zeros
                } else ++p;
            }
            acc |= (uint32_t)byte << (24 - have);
            have += 8;
        }
    }
    int bit() { if (have < 1) fill(); const int b = (int)(acc >> 31); acc <<= 1; --have; return b; }
    int bits(int n) { if (n == 0) return 0; if (have < n) fill(); const int v = (int)(acc >> (32 - n)); acc <<= n; have -= n; return v; }
    void align_and_reset() { acc = 0; have = 0; hit_marker = false; }
};

inline int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }      // T.81 F.2.2.1

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                 // Huffman tables of the current scan
    int x = 0, y = 0;                   // size in samples
    int bw = 0, bh = 0;                 // blocks per row / column of the MCU-padded grid
    int dc_pred = 0;
    std::vector<int16_t> coef;          // [bh][bw][64], natural order, NOT yet dequantised
    std::vector<uint8_t> pix;           // [bh*8][bw*8]
};

// One 8-point inverse DCT (LLM / IJG islow), 12 fractional bits in the constants.  in[k*stride]; out as the even part x[0..3] and odd part t[0..3]
// such that sample i = x[i] + t[3-i] (i < 4) and sample 7-i = x[i] - t[3-i].
// (All of it in wrapping 32-bit arithmetic: on a valid stream nothing comes near 2^31; on a corrupted one the values are garbage either way,
// but signed overflow would be undefined behaviour.)
typedef uint32_t u32;
inline u32 c12(double v) { return (u32)(int)(v * 4096.0 + 0.5); }
inline void idct8(const int* s, int stride, u32 x[4], u32 t[4]) {
    const u32 s0 = (u32)s[0], s1 = (u32)s[stride], s2 = (u32)s[2 * stride], s3 = (u32)s[3 * stride], s4 = (u32)s[4 * stride], s5 = (u32)s[5 * stride],
              s6 = (u32)s[6 * stride], s7 = (u32)s[7 * stride];
    // even part
    const u32 z1 = (s2 + s6) * c12(0.5411961);
    const u32 e2 = z1 + s6 * c12(-1.847759065);
    const u32 e3 = z1 + s2 * c12(0.765366865);
    const u32 e0 = (s0 + s4) * 4096u, e1 = (s0 - s4) * 4096u;
    x[0] = e0 + e3; x[3] = e0 - e3; x[1] = e1 + e2; x[2] = e1 - e2;
    // odd part
    u32 o0 = s7, o1 = s5, o2 = s3, o3 = s1;
    u32 p3 = o0 + o2, p4 = o1 + o3, p1 = o0 + o3, p2 = o1 + o2;
    const u32 p5 = (p3 + p4) * c12(1.175875602);
    o0 *= c12(0.298631336); o1 *= c12(2.053119869); o2 *= c12(3.072711026); o3 *= c12(1.501321110);
    p1 = p5 + p1 * c12(-0.899976223); p2 = p5 + p2 * c12(-2.562915447);
    p3 *= c12(-1.961570560); p4 *= c12(-0.390180644);
    t[3] = o3 + p1 + p4; t[2] = o2 + p2 + p3; t[1] = o1 + p2 + p4; t[0] = o0 + p1 + p3;
}
inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int sar(u32 v, int n) { return (int)v >> n; }                                   // arithmetic shift of the two's-complement value

void idct_block(const int16_t* coef, const uint16_t* quant, uint8_t* out, int out_stride) {
    int d[64], v[64];
    for (int i = 0; i < 64; ++i) d[i] = (int16_t)(u32)((u32)(int)coef[i] * (u32)quant[i]);      // dequantise into 16 bits, as a short-based decoder does
    for (int c = 0; c < 8; ++c) {                                                     // columns: keep 2 extra bits
        u32 x[4], t[4];
        idct8(d + c, 8, x, t);
        for (int i = 0; i < 4; ++i) { v[i * 8 + c] = sar(x[i] + 512u + t[3 - i], 10); v[(7 - i) * 8 + c] = sar(x[i] + 512u - t[3 - i], 10); }
    }
    for (int r = 0; r < 8; ++r) {                                                     // rows: 12 + 2 + 3 bits to remove, level shift folded in
        u32 x[4], t[4];
        idct8(v + r * 8, 1, x, t);
        uint8_t* o = out + (size_t)r * out_stride;
        for (int i = 0; i < 4; ++i) {
            const u32 base = x[i] + 65536u + (128u << 17);
            o[i] = clamp8(sar(base + t[3 - i], 17));
            o[7 - i] = clamp8(sar(base - t[3 - i], 17));
        }
    }
}

class Decoder {
public:
    bool run(const std::vector<uint8_t>& f, RgbImage& img) {
        data_ = f.data(); size_ = f.size();
        if (size_ < 4 || data_[0] != 0xFF || data_[1] != 0xD8) return false;
        size_t pos = 2;
        while (pos + 4 <= size_) {
            if (data_[pos] != 0xFF) return false;
            while (pos < size_ && data_[pos] == 0xFF) ++pos;                         // fill bytes
            if (pos >= size_) return false;
            const int m = data_[pos++];
            if (m == 0xD9) break;                                                     // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;                      // stand-alone markers
            if (pos + 2 > size_) return false;
            const size_t len = ((size_t)data_[pos] << 8) | data_[pos + 1];
            if (len < 2 || pos + len > size_) return false;
            const uint8_t* seg = data_ + pos + 2;
            const size_t n = len - 2;
            pos += len;
            if (m == 0xDB) { if (!read_dqt(seg, n)) return false; }
            else if (m == 0xC4) { if (!read_dht(seg, n)) return false; }
            else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { if (have_frame_ || !read_sof(seg, n, m == 0xC2)) return false; }
            else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) return false;       // lossless / differential / arithmetic
            else if (m == 0xDD) { if (n < 2) return false; restart_interval_ = (seg[0] << 8) | seg[1]; }
            else if (m == 0xE0) { if (n >= 5 && !std::memcmp(seg, "JFIF\0", 5)) jfif_ = true; }
            else if (m == 0xEE) { if (n >= 12 && !std::memcmp(seg, "Adobe", 5)) adobe_transform_ = seg[11]; }
            else if (m == 0xDA) {
                if (!have_frame_ || !read_sos(seg, n)) return false;
                size_t after = 0;
                if (!decode_scan(pos, after)) return false;
                pos = after;
            }
            // everything else (APPn, COM, DNL ...) is skipped
        }
        if (!have_frame_ || !scans_) return false;
        finish();
        return output(img);
    }

private:
    const uint8_t* data_ = nullptr; size_t size_ = 0;
    bool have_frame_ = false, progressive_ = false, jfif_ = false;
    int adobe_transform_ = -1, scans_ = 0;
    int width_ = 0, height_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, mcux_ = 0, mcuy_ = 0;
    int restart_interval_ = 0;
    uint16_t quant_[4][64] = {};
    bool quant_present_[4] = {false, false, false, false};
    HuffTable dc_[4], ac_[4];
    Component comp_[3];
    // current scan
    int scan_n_ = 0, scan_comp_[3] = {0, 0, 0}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0, eobrun_ = 0;

    bool read_dqt(const uint8_t* s, size_t n) {
        while (n > 0) {
            const int pq = s[0] >> 4, tq = s[0] & 15;
            if (pq > 1 || tq > 3) return false;
            const size_t need = 1 + (pq ? 128 : 64);
            if (n < need) return false;
            for (int i = 0; i < 64; ++i) quant_[tq][kZigzag[i]] = pq ? (uint16_t)((s[1 + 2 * i] << 8) | s[2 + 2 * i]) : s[1 + i];
            quant_present_[tq] = true;
            s += need; n -= need;
        }
        return true;
    }
    bool read_dht(const uint8_t* s, size_t n) {
        while (n > 0) {
            if (n < 17) return false;
            const int tc = s[0] >> 4, th = s[0] & 15;
            if (tc > 1 || th > 3) return false;
            int total = 0;
            for (int i = 0; i < 16; ++i) total += s[1 + i];
            if (total > 256 || n < (size_t)(17 + total)) return false;
            (tc ? ac_[th] : dc_[th]).build(s + 1, s + 17);
            s += 17 + total; n -= 17 + (size_t)total;
        }
        return true;
    }
    bool read_sof(const uint8_t* s, size_t n, bool progressive) {
        if (n < 6 || s[0] != 8) return false;                                          // 8-bit samples only
        height_ = (s[1] << 8) | s[2]; width_ = (s[3] << 8) | s[4]; ncomp_ = s[5];
        if (width_ <= 0 || height_ <= 0 || (ncomp_ != 1 && ncomp_ != 3) || n < (size_t)(6 + 3 * ncomp_)) return false;
        if ((unsigned long long)width_ * (unsigned long long)height_ > kMaxJpegPixels) return false;
        for (int i = 0; i < ncomp_; ++i) {
            Component& c = comp_[i];
            c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
            if (c.h > hmax_) hmax_ = c.h;
            if (c.v > vmax_) vmax_ = c.v;
        }
        for (int i = 0; i < ncomp_; ++i) if (hmax_ % comp_[i].h || vmax_ % comp_[i].v) return false;      // non-integer ratios: not handled
        mcux_ = (width_ + 8 * hmax_ - 1) / (8 * hmax_); mcuy_ = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
        for (int i = 0; i < ncomp_; ++i) {
            Component& c = comp_[i];
            c.x = (width_ * c.h + hmax_ - 1) / hmax_; c.y = (height_ * c.v + vmax_ - 1) / vmax_;
            c.bw = mcux_ * c.h; c.bh = mcuy_ * c.v;
            // the coefficient store must be coverable by the file: an entropy-coded block needs at least a couple of bits
            if ((unsigned long long)c.bw * c.bh > (unsigned long long)size_ * 8ull + 4096ull) return false;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        progressive_ = progressive;
        have_frame_ = true;
        return true;
    }
    bool read_sos(const uint8_t* s, size_t n) {
        if (n < 1) return false;
        scan_n_ = s[0];
        if (scan_n_ < 1 || scan_n_ > ncomp_ || n < (size_t)(4 + 2 * scan_n_)) return false;
        for (int k = 0; k < scan_n_; ++k) {
            int which = -1;
            for (int i = 0; i < ncomp_; ++i) if (comp_[i].id == s[1 + 2 * k]) which = i;
            if (which < 0) return false;
            scan_comp_[k] = which;
            comp_[which].td = s[2 + 2 * k] >> 4; comp_[which].ta = s[2 + 2 * k] & 15;
            if (comp_[which].td > 3 || comp_[which].ta > 3) return false;
        }
        ss_ = s[1 + 2 * scan_n_]; se_ = s[2 + 2 * scan_n_]; ah_ = s[3 + 2 * scan_n_] >> 4; al_ = s[3 + 2 * scan_n_] & 15;
        if (progressive_) {
            if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) return false;
            if (ss_ == 0 && se_ != 0) return false;                                    // DC scans carry no AC
            if (ss_ > 0 && scan_n_ != 1) return false;                                 // AC scans are never interleaved
        } else { ss_ = 0; se_ = 63; ah_ = al_ = 0; }
        return true;
    }

    int decode_symbol(BitReader& br, const HuffTable& t) {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | br.bit();
            if (t.count[len] && code - t.first_code[len] < t.count[len] && code >= t.first_code[len]) return t.symbols[t.first_index[len] + code - t.first_code[len]];
        }
        return -1;
    }

    bool block_sequential(BitReader& br, Component& c, int16_t* blk) {
        const HuffTable& dc = dc_[c.td]; const HuffTable& ac = ac_[c.ta];
        if (!dc.present || !ac.present) return false;
        const int t = decode_symbol(br, dc);
        if (t < 0 || t > 15) return false;
        c.dc_pred += extend(br.bits(t), t);
        blk[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(br, ac);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) return false;
            blk[kZigzag[k++]] = (int16_t)extend(br.bits(s), s);
        }
        return true;
    }
    bool block_dc_progressive(BitReader& br, Component& c, int16_t* blk) {
        if (ah_ == 0) {
            const HuffTable& dc = dc_[c.td];
            if (!dc.present) return false;
            const int t = decode_symbol(br, dc);
            if (t < 0 || t > 15) return false;
            c.dc_pred += extend(br.bits(t), t);
            blk[0] = (int16_t)(c.dc_pred * (1 << al_));
        } else if (br.bit()) blk[0] = (int16_t)(blk[0] + (1 << al_));
        return true;
    }
    bool block_ac_progressive(BitReader& br, Component& c, int16_t* blk) {
        const HuffTable& ac = ac_[c.ta];
        if (!ac.present) return false;
        if (ah_ == 0) {                                                                 // first pass of this band (T.81 G.1.2.2)
            if (eobrun_ > 0) { --eobrun_; return true; }
            for (int k = ss_; k <= se_;) {
                const int rs = decode_symbol(br, ac);
                if (rs < 0) return false;
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobrun_ = (1 << r) - 1; if (r) eobrun_ += br.bits(r); break; }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) return false;
                    blk[kZigzag[k++]] = (int16_t)(extend(br.bits(s), s) * (1 << al_));
                }
            }
            return true;
        }
        const int bit = 1 << al_;                                                       // refinement pass (T.81 G.1.2.3)
        auto refine = [&](int16_t& v) {
            if (br.bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
        };
        int k = ss_;
        if (eobrun_ == 0) {
            for (; k <= se_;) {
                const int rs = decode_symbol(br, ac);
                if (rs < 0) return false;
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) { eobrun_ = (1 << r); if (r) eobrun_ += br.bits(r); break; }   // this block counts as the first of the run (decremented below)
                } else {
                    if (s != 1) return false;
                    value = br.bit() ? bit : -bit;
                }
                while (k <= se_) {                                                     // skip r zero-history coefficients, refining the non-zero ones passed
                    int16_t& v = blk[kZigzag[k++]];
                    if (v != 0) refine(v);
                    else { if (r == 0) { v = (int16_t)value; break; } --r; }
                }
            }
        }
        if (eobrun_ > 0) {
            for (; k <= se_; ++k) { int16_t& v = blk[kZigzag[k]]; if (v != 0) refine(v); }
            --eobrun_;
        }
        return true;
    }

    bool decode_scan(size_t start, size_t& after) {
        BitReader br{data_ + start, data_ + size_};
        for (int i = 0; i < ncomp_; ++i) comp_[i].dc_pred = 0;
        eobrun_ = 0;
        int until_restart = restart_interval_ ? restart_interval_ : 0x7FFFFFFF;
        auto restart_if_due = [&]() -> bool {
            if (--until_restart > 0) return true;
            // byte-align, expect RSTn, reset predictors
            br.align_and_reset();
            const uint8_t* p = br.p;
            while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
                if (p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF) return true;       // some other marker: the scan is over; leave the rest zero
                ++p;
            }
            if (p + 1 >= br.end) return true;
            br.p = p + 2;
            for (int i = 0; i < ncomp_; ++i) comp_[i].dc_pred = 0;
            eobrun_ = 0;
            until_restart = restart_interval_;
            return true;
        };
        auto one_block = [&](Component& c, int bx, int by) -> bool {
            if (bx >= c.bw || by >= c.bh) return false;
            int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (!progressive_) return block_sequential(br, c, blk);
            return ss_ == 0 ? block_dc_progressive(br, c, blk) : block_ac_progressive(br, c, blk);
        };
        bool ok = true;
        if (scan_n_ == 1) {                                                            // not interleaved: the component's own block rows
            Component& c = comp_[scan_comp_[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int by = 0; by < h && ok; ++by)
                for (int bx = 0; bx < w && ok; ++bx) { ok = one_block(c, bx, by); if (ok) restart_if_due(); }
        } else {
            for (int my = 0; my < mcuy_ && ok; ++my)
                for (int mx = 0; mx < mcux_ && ok; ++mx) {
                    for (int k = 0; k < scan_n_ && ok; ++k) {
                        Component& c = comp_[scan_comp_[k]];
                        for (int v = 0; v < c.v && ok; ++v)
                            for (int h = 0; h < c.h && ok; ++h) ok = one_block(c, mx * c.h + h, my * c.v + v);
                    }
                    if (ok) restart_if_due();
                }
        }
        if (!ok) return false;
        ++scans_;
        // the next marker: scan forward from where the bit reader stopped consuming bytes
        const uint8_t* p = br.p;
        while (p + 1 < data_ + size_ && !(p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) ++p;
        after = (size_t)(p - data_);
        return true;
    }

    void finish() {
        for (int i = 0; i < ncomp_; ++i) {
            Component& c = comp_[i];
            c.pix.assign((size_t)c.bw * 8 * c.bh * 8, 0);
            const uint16_t* q = quant_[c.tq];
            for (int by = 0; by < c.bh; ++by)
                for (int bx = 0; bx < c.bw; ++bx)
                    idct_block(&c.coef[((size_t)by * c.bw + bx) * 64], q, &c.pix[((size_t)by * 8) * (c.bw * 8) + (size_t)bx * 8], c.bw * 8);
            c.coef.clear(); c.coef.shrink_to_fit();
        }
    }

    bool output(RgbImage& img) {
        img.width = width_; img.height = height_;
        img.rgb.assign((size_t)width_ * height_ * 3, 0);
        const bool rgb_direct = ncomp_ == 3 && ((comp_[0].id == 'R' && comp_[1].id == 'G' && comp_[2].id == 'B') || (adobe_transform_ == 0 && !jfif_));
        std::vector<uint8_t> line[3];
        struct Up { int hs, vs, ystep, ypos, w_lores; const uint8_t *line0, *line1; } up[3];
        for (int k = 0; k < ncomp_; ++k) {
            const Component& c = comp_[k];
            up[k] = {hmax_ / c.h, vmax_ / c.v, (vmax_ / c.v) >> 1, 0, (width_ + hmax_ / c.h - 1) / (hmax_ / c.h), c.pix.data(), c.pix.data()};
            line[k].assign((size_t)width_ + 8, 0);
        }
        for (int j = 0; j < height_; ++j) {
            const uint8_t* row[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < ncomp_; ++k) {
                Up& u = up[k];
                const bool bottom = u.ystep >= (u.vs >> 1);
                const uint8_t* near_row = bottom ? u.line1 : u.line0;
                const uint8_t* far_row = bottom ? u.line0 : u.line1;
                uint8_t* o = line[k].data();
                const int w = u.w_lores;
                if (u.hs == 1 && u.vs == 1) row[k] = near_row;
                else {
                    if (u.hs == 1 && u.vs == 2) { for (int i = 0; i < w; ++i) o[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2); }
                    else if (u.hs == 2 && u.vs == 1) {
                        if (w == 1) o[0] = o[1] = near_row[0];
                        else {
                            o[0] = near_row[0]; o[1] = (uint8_t)((near_row[0] * 3 + near_row[1] + 2) >> 2);
                            int i = 1;
                            for (; i < w - 1; ++i) { const int n = 3 * near_row[i] + 2; o[2 * i] = (uint8_t)((n + near_row[i - 1]) >> 2); o[2 * i + 1] = (uint8_t)((n + near_row[i + 1]) >> 2); }
                            o[2 * i] = (uint8_t)((near_row[w - 2] * 3 + near_row[w - 1] + 2) >> 2); o[2 * i + 1] = near_row[w - 1];
                        }
                    } else if (u.hs == 2 && u.vs == 2) {
                        if (w == 1) o[0] = o[1] = (uint8_t)((3 * near_row[0] + far_row[0] + 2) >> 2);
                        else {
                            int t1 = 3 * near_row[0] + far_row[0];
                            o[0] = (uint8_t)((t1 + 2) >> 2);
                            for (int i = 1; i < w; ++i) {
                                const int t0 = t1;
                                t1 = 3 * near_row[i] + far_row[i];
                                o[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                                o[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
                            }
                            o[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
                        }
                    } else {
                        for (int i = 0; i < w; ++i) for (int r = 0; r < u.hs; ++r) if (i * u.hs + r < width_ + 8) o[i * u.hs + r] = near_row[i];
                    }
                    row[k] = o;
                }
                if (++u.ystep >= u.vs) {
                    u.ystep = 0;
                    u.line0 = u.line1;
                    if (++u.ypos < comp_[k].y) u.line1 += (size_t)comp_[k].bw * 8;
                }
            }
            uint8_t* out = &img.rgb[(size_t)j * width_ * 3];
            if (ncomp_ == 1) { for (int i = 0; i < width_; ++i) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = row[0][i]; }
            else if (rgb_direct) { for (int i = 0; i < width_; ++i) { out[3 * i] = row[0][i]; out[3 * i + 1] = row[1][i]; out[3 * i + 2] = row[2][i]; } }
            else {
                const int k_cr_r = (int)(1.40200f * 4096.0f + 0.5f) << 8, k_cr_g = (int)(0.71414f * 4096.0f + 0.5f) << 8;
                const int k_cb_g = (int)(0.34414f * 4096.0f + 0.5f) << 8, k_cb_b = (int)(1.77200f * 4096.0f + 0.5f) << 8;
                for (int i = 0; i < width_; ++i) {
                    const int yf = (row[0][i] << 20) + (1 << 19), cb = row[1][i] - 128, cr = row[2][i] - 128;
                    int r = yf + cr * k_cr_r;
                    int g = yf + cr * -k_cr_g + (int)((unsigned)(cb * -k_cb_g) & 0xFFFF0000u);
                    int b = yf + cb * k_cb_b;
                    r >>= 20; g >>= 20; b >>= 20;
                    out[3 * i] = clamp8(r); out[3 * i + 1] = clamp8(g); out[3 * i + 2] = clamp8(b);
                }
            }
        }
        return true;
    }
};

}  // namespace

bool decode_jpeg(const std::vector<uint8_t>& f, RgbImage& img) {
    Decoder d;
    return d.run(f, img);
}

}  // namespace dsrt
