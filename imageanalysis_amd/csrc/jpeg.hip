// Split JPEG decoder for the image ingest: the reference reads every
// survey frame with cv2.imread(file, ANYCOLOR | ANYDEPTH | IGNORE_ORIENTATION)
// (scripts/lib/image.py:99-104), i.e. libjpeg(-turbo) on one host core per call, ~0.15 s per
// 20 MP frame -- 25x what the detector kernels need.  Here the host does only what is
// inherently serial, the Huffman decode of the entropy-coded segment (iamx_jpeg_decode_coefficients,
// plain C++, no device involved, thread safe: the python layer runs it on worker threads with the
// GIL released), and the device does the rest from the quantised coefficients:
//   jpeg_idct_kernel    dequantise + the accurate integer inverse DCT ("islow", the IJG / libjpeg-
//                       turbo default, jidctint.c: 13-bit constants, two passes, descaling with
//                       rounding, range limit) -- one thread per 8x8 block
//   jpeg_color_kernel   chroma "fancy" (triangle filter) upsampling h2v1 / h2v2 exactly as
//                       jdsample.c does it, including the edge columns / rows, then YCbCr -> BGR
//                       with jdcolor.c's 16-bit fixed-point tables
// All integer arithmetic: the pixels are bit-identical to libjpeg-turbo's (Pillow / OpenCV) output,
// which tests/test_jpeg_gpu.py checks on 4:4:4, 4:2:2, 4:2:0 and grey files with restart markers
// and odd sizes.  Baseline / extended-sequential Huffman files with one interleaved scan (what
// cameras write) are handled; anything else (progressive, arithmetic, 12 bit, CMYK, 4:4:0) is
// reported as IAMX_EUNSUPPORTED and the caller reads the file the host way.
#include "iamx_common.h"

#include <cstring>
#include <vector>

namespace {

constexpr int MAX_COMP = 3;

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    // lookahead on LOOK bits: (length << 8) | symbol, 0 = longer code
    static constexpr int LOOK = 9;
    uint16_t fast[1 << LOOK];
    int32_t maxcode[18];        // largest code of length l (-1 if none), [17] sentinel
    int32_t valoffset[17];      // huffval index of the first code of length l minus that code
    uint8_t huffval[256];
    // AC tables only: code AND magnitude bits inside the lookahead -> (value << 8) | (run << 4) |
    // total bits (value -128 .. 127), 0 = take the two-step path
    int16_t fast_ac[1 << LOOK];

    int eob_len = HuffTable::LOOK, eob_code = -1;   // the code of symbol 0x00 if it is <= LOOK bits long

    void build_fast_ac()
    {
        eob_len = LOOK;
        eob_code = -1;
        for (int i = 0; i < (1 << LOOK); ++i)
            if (fast[i] && (fast[i] & 255) == 0) {
                eob_len = fast[i] >> 8;
                eob_code = i >> (LOOK - eob_len);
                break;
            }
        for (int i = 0; i < (1 << LOOK); ++i) {
            fast_ac[i] = 0;
            const int f = fast[i];
            if (!f) continue;
            const int rs = f & 255, l = f >> 8, run = rs >> 4, sz = rs & 15;
            if (sz == 0 || l + sz > LOOK) continue;
            int v = ((i << l) & ((1 << LOOK) - 1)) >> (LOOK - sz);      // the sz bits behind the code
            if (v < (1 << (sz - 1))) v += 1 - (1 << sz);                // (jdhuff.c HUFF_EXTEND)
            if (v >= -128 && v <= 127) fast_ac[i] = (int16_t)(v * 256 + run * 16 + l + sz);
        }
    }

    bool build(const uint8_t *bits /* [1..16] at [0..15] */, const uint8_t *vals, int nvals)
    {
        int total = 0;
        for (int l = 0; l < 16; ++l) total += bits[l];
        if (total > 256 || total != nvals) return false;
        for (int i = 0; i < nvals; ++i) huffval[i] = vals[i];
        for (int i = 0; i < (1 << LOOK); ++i) fast[i] = 0;
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            const int n = bits[l - 1];
            valoffset[l] = k - code;
            if (n) {
                if (code + n > (1 << l)) return false;
                for (int i = 0; i < n; ++i, ++k, ++code) {
                    if (l <= LOOK) {
                        const int shift = LOOK - l;
                        for (int f = 0; f < (1 << shift); ++f)
                            fast[(code << shift) | f] = (uint16_t)((l << 8) | huffval[k]);
                    }
                }
                maxcode[l] = code - 1;
            } else {
                maxcode[l] = -1;
            }
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        build_fast_ac();
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;          // padded to whole MCUs
};

struct Header {
    int width = 0, height = 0, ncomp = 0, restart = 0;
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    int adobe_transform = -1;                // APP14 "Adobe" colour transform flag (-1: no marker)
    Component comp[MAX_COMP];
    uint16_t quant[4][64];                   // natural order
    bool have_quant[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    const uint8_t *scan = nullptr;           // entropy-coded data
    size_t scan_len = 0;
};

inline int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

// markers up to the first SOS; IAMX_OK / IAMX_EINVAL (broken file) / IAMX_EUNSUPPORTED
int parse(const uint8_t *data, size_t len, Header &H)
{
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return iamx::fail(IAMX_EINVAL, "jpeg: no SOI");
    size_t p = 2;
    bool have_sof = false;
    while (p + 4 <= len) {
        if (data[p] != 0xFF) return iamx::fail(IAMX_EINVAL, "jpeg: marker expected at %zu", p);
        while (p < len && data[p] == 0xFF) ++p;              // fill bytes
        if (p >= len) break;
        const int m = data[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (p + 2 > len) break;
        const int seg = be16(data + p);
        if (seg < 2 || p + seg > len) return iamx::fail(IAMX_EINVAL, "jpeg: bad segment length");
        const uint8_t *s = data + p + 2;
        const int n = seg - 2;
        if (m == 0xDB) {                                     // DQT
            int q = 0;
            while (q < n) {
                const int pq = s[q] >> 4, tq = s[q] & 15;
                ++q;
                if (tq > 3 || q + (pq ? 128 : 64) > n) return iamx::fail(IAMX_EINVAL, "jpeg: bad DQT");
                for (int i = 0; i < 64; ++i) {
                    H.quant[tq][kZigzag[i]] = pq ? (uint16_t)be16(s + q + 2 * i) : s[q + i];
                }
                q += pq ? 128 : 64;
                H.have_quant[tq] = true;
            }
        } else if (m == 0xC4) {                              // DHT
            int q = 0;
            while (q + 17 <= n) {
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3) return iamx::fail(IAMX_EINVAL, "jpeg: bad DHT");
                int total = 0;
                for (int l = 0; l < 16; ++l) total += s[q + 1 + l];
                if (q + 17 + total > n) return iamx::fail(IAMX_EINVAL, "jpeg: bad DHT length");
                HuffTable &T = tc ? H.ac[th] : H.dc[th];
                if (!T.build(s + q + 1, s + q + 17, total)) return iamx::fail(IAMX_EINVAL, "jpeg: bad Huffman table");
                q += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {                 // SOF0 / SOF1: sequential Huffman
            if (n < 6) return iamx::fail(IAMX_EINVAL, "jpeg: bad SOF");
            if (s[0] != 8) return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: %d-bit samples", s[0]);
            H.height = be16(s + 1);
            H.width = be16(s + 3);
            H.ncomp = s[5];
            if (H.ncomp != 1 && H.ncomp != 3) return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: %d components", H.ncomp);
            if (n < 6 + 3 * H.ncomp || H.width < 1 || H.height < 1) return iamx::fail(IAMX_EINVAL, "jpeg: bad SOF");
            for (int c = 0; c < H.ncomp; ++c) {
                H.comp[c].id = s[6 + 3 * c];
                H.comp[c].h = s[7 + 3 * c] >> 4;
                H.comp[c].v = s[7 + 3 * c] & 15;
                H.comp[c].tq = s[8 + 3 * c];
                if (H.comp[c].h < 1 || H.comp[c].h > 2 || H.comp[c].v < 1 || H.comp[c].v > 2 || H.comp[c].tq > 3)
                    return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: sampling factors %dx%d", H.comp[c].h, H.comp[c].v);
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: SOF%d (progressive / lossless / arithmetic)", m - 0xC0);
        } else if (m == 0xEE && n >= 12 && std::memcmp(s, "Adobe", 5) == 0) {
            H.adobe_transform = s[11];
        } else if (m == 0xDD) {                              // DRI
            if (n < 2) return iamx::fail(IAMX_EINVAL, "jpeg: bad DRI");
            H.restart = be16(s);
        } else if (m == 0xDA) {                              // SOS
            if (!have_sof) return iamx::fail(IAMX_EINVAL, "jpeg: SOS before SOF");
            if (n < 1 || s[0] != H.ncomp || n < 1 + 2 * H.ncomp + 3)
                return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: scan does not hold all components");
            for (int c = 0; c < H.ncomp; ++c) {
                if (s[1 + 2 * c] != H.comp[c].id) return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: component order");
                H.comp[c].td = s[2 + 2 * c] >> 4;
                H.comp[c].ta = s[2 + 2 * c] & 15;
                if (H.comp[c].td > 3 || H.comp[c].ta > 3 || !H.dc[H.comp[c].td].present ||
                    !H.ac[H.comp[c].ta].present || !H.have_quant[H.comp[c].tq])
                    return iamx::fail(IAMX_EINVAL, "jpeg: missing table");
            }
            const uint8_t *t = s + 1 + 2 * H.ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: spectral selection");
            H.scan = data + p + seg;
            H.scan_len = len - (p + seg);
            // geometry
            H.hmax = H.vmax = 1;
            for (int c = 0; c < H.ncomp; ++c) {
                H.hmax = H.comp[c].h > H.hmax ? H.comp[c].h : H.hmax;
                H.vmax = H.comp[c].v > H.vmax ? H.comp[c].v : H.vmax;
            }
            if (H.ncomp == 1) { H.comp[0].h = H.comp[0].v = 1; H.hmax = H.vmax = 1; }   // (a grey scan is never interleaved)
            if (H.ncomp == 3) {
                // YCbCr only (libjpeg's colour space guess: an Adobe marker with transform 0, or
                // component ids 'R' 'G' 'B' without a JFIF marker, mean RGB)
                if (H.adobe_transform == 0 ||
                    (H.comp[0].id == 'R' && H.comp[1].id == 'G' && H.comp[2].id == 'B'))
                    return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: RGB colour space");
                // luma at full resolution, both chroma planes alike: 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2)
                if (H.comp[0].h != H.hmax || H.comp[0].v != H.vmax || H.comp[1].h != H.comp[2].h ||
                    H.comp[1].v != H.comp[2].v || H.comp[1].h != 1 || H.comp[1].v != 1 ||
                    (H.hmax == 1 && H.vmax == 2))
                    return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: subsampling %dx%d / %dx%d", H.comp[0].h,
                                      H.comp[0].v, H.comp[1].h, H.comp[1].v);
            }
            H.mcus_x = (H.width + 8 * H.hmax - 1) / (8 * H.hmax);
            H.mcus_y = (H.height + 8 * H.vmax - 1) / (8 * H.vmax);
            for (int c = 0; c < H.ncomp; ++c) {
                H.comp[c].blocks_w = H.mcus_x * H.comp[c].h;
                H.comp[c].blocks_h = H.mcus_y * H.comp[c].v;
            }
            return IAMX_OK;
        }
        p += seg;
    }
    return iamx::fail(IAMX_EINVAL, "jpeg: no scan");
}

// bit reader over the entropy-coded segment (FF00 unstuffing; a marker ends the data: zeros)
struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int bits = 0;
    bool hit_marker = false;

    // afterwards at least 32 valid bits: a Huffman code (<= 16) and its magnitude bits (<= 15)
    inline void fill()
    {
        if (bits >= 32) return;
        if (!hit_marker && p + 8 <= end) {
            // eight bytes at once when none of them is 0xFF (no stuffing, no marker)
            uint64_t w;
            std::memcpy(&w, p, 8);
            w = __builtin_bswap64(w);
            const uint64_t t = ~w;
            if (!((t - 0x0101010101010101ull) & ~t & 0x8080808080808080ull)) {
                const int nb = (64 - bits) >> 3;                     // 1 .. 8 whole bytes fit
                const uint64_t take = nb == 8 ? w : (w & ~((1ull << (64 - 8 * nb)) - 1));
                acc |= take >> bits;
                bits += 8 * nb;
                p += nb;
                return;
            }
        }
        while (bits <= 56) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else {
                        hit_marker = true;      // RSTn / EOI / garbage: feed zeros
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            acc |= (uint64_t)b << (56 - bits);
            bits += 8;
        }
    }
    inline int peek(int n) { return (int)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; bits -= n; }
    inline int get(int n)
    {
        if (n == 0) return 0;
        const int v = (int)(acc >> (64 - n));
        acc <<= n;
        bits -= n;
        return v;
    }
    void reset_at_restart()
    {
        // byte align is implicit: drop the buffered bits, find the RSTn marker
        acc = 0; bits = 0;
        if (hit_marker) {
            while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
            if (p + 1 < end) p += 2;
            hit_marker = false;
        } else {
            // the marker was not reached through the bit buffer yet: scan forward for it
            while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
                if (p[0] == 0xFF && p[1] == 0x00) p += 2; else ++p;
            }
            if (p + 1 < end) p += 2;
        }
    }
};

inline int decode_symbol(BitReader &br, const HuffTable &T)
{
    br.fill();
    const int look = br.peek(HuffTable::LOOK);
    const int f = T.fast[look];
    if (f) {
        br.skip(f >> 8);
        return f & 255;
    }
    int l = HuffTable::LOOK + 1;
    int code = br.peek(l);
    while (l <= 16 && code > T.maxcode[l]) {
        ++l;
        code = br.peek(l);
    }
    if (l > 16) { br.skip(16); return 0; }                   // corrupt: libjpeg warns and uses 0
    br.skip(l);
    return T.huffval[(code + T.valoffset[l]) & 255];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

}  // namespace

extern "C" int iamx_jpeg_info(const uint8_t *data, int64_t len, int32_t *info /* [16] */)
{
    IAMX_REQUIRE(data && info && len > 0, "null pointer");
    Header H;
    const int rc = parse(data, (size_t)len, H);
    if (rc != IAMX_OK) return rc;
    info[0] = H.width; info[1] = H.height; info[2] = H.ncomp;
    info[3] = H.hmax; info[4] = H.vmax;
    int64_t off = 0;
    for (int c = 0; c < MAX_COMP; ++c) {
        const bool on = c < H.ncomp;
        info[5 + 2 * c] = on ? H.comp[c].blocks_w : 0;
        info[6 + 2 * c] = on ? H.comp[c].blocks_h : 0;
        if (on) off += (int64_t)H.comp[c].blocks_w * H.comp[c].blocks_h;
    }
    info[11] = (int32_t)off;                                 // total blocks
    info[12] = H.restart;
    info[13] = info[14] = info[15] = 0;
    return IAMX_OK;
}

extern "C" int iamx_jpeg_decode_coefficients(const uint8_t *data, int64_t len, int16_t *coef,
                                             int64_t coef_blocks, uint16_t *quant /* [3][64] */)
{
    IAMX_REQUIRE(data && coef && quant && len > 0, "null pointer");
    Header H;
    const int rc = parse(data, (size_t)len, H);
    if (rc != IAMX_OK) return rc;
    int64_t base[MAX_COMP], total = 0;
    for (int c = 0; c < H.ncomp; ++c) {
        base[c] = total;
        total += (int64_t)H.comp[c].blocks_w * H.comp[c].blocks_h;
    }
    IAMX_REQUIRE(coef_blocks >= total, "coefficient buffer too small (iamx_jpeg_info)");
    for (int c = 0; c < H.ncomp; ++c)
        for (int i = 0; i < 64; ++i) quant[c * 64 + i] = H.quant[H.comp[c].tq][i];
    BitReader br;
    br.p = H.scan;
    br.end = H.scan + H.scan_len;
    int pred[MAX_COMP] = {0, 0, 0};
    int to_restart = H.restart;
    for (int my = 0; my < H.mcus_y; ++my) {
        for (int mx = 0; mx < H.mcus_x; ++mx) {
            if (H.restart && to_restart == 0) {
                br.reset_at_restart();
                pred[0] = pred[1] = pred[2] = 0;
                to_restart = H.restart;
            }
            for (int c = 0; c < H.ncomp; ++c) {
                const Component &C = H.comp[c];
                const HuffTable &DC = H.dc[C.td], &AC = H.ac[C.ta];
                for (int by = 0; by < C.v; ++by)
                    for (int bx = 0; bx < C.h; ++bx) {
                        int16_t *blk = coef + 64 * (base[c] + (int64_t)(my * C.v + by) * C.blocks_w + mx * C.h + bx);
                        std::memset(blk, 0, 128);
                        int s = decode_symbol(br, DC);
                        if (s > 15) s = 0;                               // corrupt table entry (jdhuff.c warns)
                        if (s) {
                            const int r = br.get(s);
                            s = extend(r, s);
                        }
                        pred[c] += s;
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            br.fill();
                            const int look = br.peek(HuffTable::LOOK);
                            if (look >> (HuffTable::LOOK - AC.eob_len) == AC.eob_code) {   // end of block
                                br.skip(AC.eob_len);
                                break;
                            }
                            const int fa = AC.fast_ac[look];
                            if (fa) {                                    // code + value in one look-up
                                k += (fa >> 4) & 15;
                                br.skip(fa & 15);
                                if (k < 64) blk[kZigzag[k]] = (int16_t)(fa >> 8);
                                ++k;
                                continue;
                            }
                            const int rs = decode_symbol(br, AC);
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz) {
                                k += r;
                                const int v = extend(br.get(sz), sz);
                                if (k < 64) blk[kZigzag[k]] = (int16_t)v;
                                ++k;
                            } else {
                                if (r != 15) break;                      // EOB
                                k += 16;
                            }
                        }
                    }
            }
            if (H.restart) --to_restart;
        }
    }
    return IAMX_OK;
}

// ======================================= device side ===========================================
namespace {

__device__ __forceinline__ int range_limit_idct(int x)
{
    // IDCT_range_limit[x & RANGE_MASK] of jdmaster.c prepare_range_limit_table()
    const int i = x & 1023;
    return i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896));
}

// jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2
#define IAMX_FIX_0_298631336 2446
#define IAMX_FIX_0_390180644 3196
#define IAMX_FIX_0_541196100 4433
#define IAMX_FIX_0_765366865 6270
#define IAMX_FIX_0_899976223 7373
#define IAMX_FIX_1_175875602 9633
#define IAMX_FIX_1_501321110 12299
#define IAMX_FIX_1_847759065 15137
#define IAMX_FIX_1_961570560 16069
#define IAMX_FIX_2_053119869 16819
#define IAMX_FIX_2_562915447 20995
#define IAMX_FIX_3_072711026 25172

__device__ __forceinline__ void idct_1d(const int in[8], int out[8], int shift)
{
    // even part
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * IAMX_FIX_0_541196100;
    int tmp2 = z1 + z3 * (-IAMX_FIX_1_847759065);
    int tmp3 = z1 + z2 * IAMX_FIX_0_765366865;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13;
    int tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    // odd part
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    int z5 = (z3 + z4) * IAMX_FIX_1_175875602;
    tmp0 *= IAMX_FIX_0_298631336; tmp1 *= IAMX_FIX_2_053119869;
    tmp2 *= IAMX_FIX_3_072711026; tmp3 *= IAMX_FIX_1_501321110;
    z1 *= -IAMX_FIX_0_899976223; z2 *= -IAMX_FIX_2_562915447;
    z3 *= -IAMX_FIX_1_961570560; z4 *= -IAMX_FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    out[0] = (tmp10 + tmp3 + rnd) >> shift; out[7] = (tmp10 - tmp3 + rnd) >> shift;
    out[1] = (tmp11 + tmp2 + rnd) >> shift; out[6] = (tmp11 - tmp2 + rnd) >> shift;
    out[2] = (tmp12 + tmp1 + rnd) >> shift; out[5] = (tmp12 - tmp1 + rnd) >> shift;
    out[3] = (tmp13 + tmp0 + rnd) >> shift; out[4] = (tmp13 - tmp0 + rnd) >> shift;
}

// one thread per 8x8 block of any component: coefficients (natural order) * quant -> 8x8 samples
// at (block_y * 8, block_x * 8) of the component's plane [blocks_h * 8][blocks_w * 8]
struct JpegPlanes {
    int64_t first_block[MAX_COMP + 1];   // blocks of the components back to back
    int blocks_w[MAX_COMP];
    int64_t plane_off[MAX_COMP];         // byte offset of the component's plane in `planes`
};

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t *__restrict__ coef,
                                                        const uint16_t *__restrict__ quant,
                                                        JpegPlanes P, int ncomp,
                                                        uint8_t *__restrict__ planes)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= P.first_block[ncomp]) return;
    int c = 0;
    if (ncomp > 1 && b >= P.first_block[1]) c = b >= P.first_block[2] ? 2 : 1;
    const int64_t lb = b - P.first_block[c];
    const int bw = P.blocks_w[c];
    const int by = (int)(lb / bw), bx = (int)(lb - (int64_t)by * bw);
    const int16_t *src = coef + b * 64;
    const uint16_t *q = quant + c * 64;
    int ws[64];
    // 128 bytes of coefficients per thread: eight 16-byte loads
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    int cf[64];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint4 v = s4[r];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cf[r * 8 + 2 * e] = (int)(int16_t)(w[e] & 0xffff) * (int)q[r * 8 + 2 * e];
            cf[r * 8 + 2 * e + 1] = (int)(int16_t)(w[e] >> 16) * (int)q[r * 8 + 2 * e + 1];
        }
    }
    // pass 1: columns, results scaled up by 2^PASS1_BITS
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = cf[r * 8 + col];
        idct_1d(in, out, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r * 8 + col] = out[r];
    }
    // pass 2: rows, descale by CONST_BITS + PASS1_BITS + 3, range limit
    const int pitch = bw * 8;
    uint8_t *dst = planes + P.plane_off[c] + ((int64_t)by * 8) * pitch + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int in[8], out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) in[k] = ws[r * 8 + k];
        idct_1d(in, out, 13 + 2 + 3);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= (unsigned)range_limit_idct(out[k]) << (8 * k);
            hi |= (unsigned)range_limit_idct(out[4 + k]) << (8 * k);
        }
        *reinterpret_cast<uint2 *>(dst + (int64_t)r * pitch) = make_uint2(lo, hi);
    }
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// jdsample.c fancy upsampling of one chroma plane at output pixel (x, y):
//   MODE 0: none (4:4:4), 1: h2v1 (4:2:2), 2: h2v2 (4:2:0)
// dw / dh = downsampled_width / downsampled_height of the component (the REAL sample columns /
// rows: the edge cases use them, not the padded plane)
template <int MODE>
__device__ __forceinline__ int chroma_at(const uint8_t *__restrict__ pl, int pitch, int dw, int dh,
                                         int x, int y)
{
    if (MODE == 0) return pl[(int64_t)y * pitch + x];
    if (MODE == 1) {
        const uint8_t *row = pl + (int64_t)y * pitch;
        const int i = x >> 1;
        const int v = row[i];
        if (x & 1) {
            if (i == dw - 1) return v;                        // last column: plain copy
            return (v * 3 + row[i + 1] + 2) >> 2;
        }
        if (i == 0) return v;                                 // first column: plain copy
        return (v * 3 + row[i - 1] + 1) >> 2;
    }
    // h2v2: vertical 3:1 of the nearer / farther input row, then horizontal 3:1
    const int r = y >> 1;
    int rn = (y & 1) ? r + 1 : r - 1;                          // the farther row
    rn = rn < 0 ? 0 : (rn > dh - 1 ? dh - 1 : rn);             // (edge rows are replicated)
    const uint8_t *r0 = pl + (int64_t)r * pitch, *r1 = pl + (int64_t)rn * pitch;
    const int i = x >> 1;
    const int cur = r0[i] * 3 + r1[i];
    if (x & 1) {
        if (i == dw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + (r0[i + 1] * 3 + r1[i + 1]) + 7) >> 4;
    }
    if (i == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[i - 1] * 3 + r1[i - 1]) + 8) >> 4;
}

template <int MODE>
__global__ __launch_bounds__(256) void jpeg_color_kernel(const uint8_t *__restrict__ planes,
                                                         JpegPlanes P, int width, int height,
                                                         int dw, int dh, uint8_t *__restrict__ bgr)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    const int yv = planes[P.plane_off[0] + (int64_t)y * (P.blocks_w[0] * 8) + x];
    const int cpitch = P.blocks_w[1] * 8;
    const int cb = chroma_at<MODE>(planes + P.plane_off[1], cpitch, dw, dh, x, y) - 128;
    const int cr = chroma_at<MODE>(planes + P.plane_off[2], cpitch, dw, dh, x, y) - 128;
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF 32768
    const int r = yv + ((91881 * cr + 32768) >> 16);
    const int g = yv + ((-22554 * cb + 32768 + (-46802) * cr) >> 16);
    const int b = yv + ((116130 * cb + 32768) >> 16);
    uint8_t *o = bgr + ((int64_t)y * width + x) * 3;
    o[0] = (uint8_t)clamp255(b);
    o[1] = (uint8_t)clamp255(g);
    o[2] = (uint8_t)clamp255(r);
}

__global__ __launch_bounds__(256) void jpeg_gray_kernel(const uint8_t *__restrict__ planes, int pitch,
                                                        int width, uint8_t *__restrict__ bgr)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    const uint8_t v = planes[(int64_t)y * pitch + x];
    uint8_t *o = bgr + ((int64_t)y * width + x) * 3;
    o[0] = o[1] = o[2] = v;
}

}  // namespace

extern "C" int64_t iamx_jpeg_workspace_bytes(const int32_t *info)
{
    if (!info) return 0;
    int64_t n = 0;
    for (int c = 0; c < info[2] && c < MAX_COMP; ++c) n += (int64_t)info[5 + 2 * c] * info[6 + 2 * c] * 64;
    return (n + 255) / 256 * 256;
}

extern "C" int iamx_jpeg_reconstruct(const int16_t *coef, const uint16_t *quant, const int32_t *info,
                                     void *workspace, int64_t workspace_bytes, uint8_t *bgr,
                                     void *stream)
{
    IAMX_REQUIRE(coef && quant && info && workspace && bgr, "null pointer");
    const int width = info[0], height = info[1], ncomp = info[2], hmax = info[3], vmax = info[4];
    IAMX_REQUIRE(width > 0 && height > 0 && (ncomp == 1 || ncomp == 3), "bad image description");
    IAMX_REQUIRE(workspace_bytes >= iamx_jpeg_workspace_bytes(info), "workspace too small");
    JpegPlanes P;
    int64_t blocks = 0, bytes = 0;
    for (int c = 0; c < MAX_COMP; ++c) {
        P.first_block[c] = blocks;
        P.blocks_w[c] = c < ncomp ? info[5 + 2 * c] : 0;
        P.plane_off[c] = bytes;
        if (c < ncomp) {
            blocks += (int64_t)info[5 + 2 * c] * info[6 + 2 * c];
            bytes += (int64_t)info[5 + 2 * c] * info[6 + 2 * c] * 64;
        }
    }
    for (int c = ncomp; c <= MAX_COMP; ++c) P.first_block[c] = blocks;
    hipStream_t st = iamx::as_stream(stream);
    uint8_t *planes = static_cast<uint8_t *>(workspace);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, st, coef,
                       quant, P, ncomp, planes);
    const dim3 grid((width + 255) / 256, height);
    if (ncomp == 1) {
        hipLaunchKernelGGL(jpeg_gray_kernel, grid, dim3(256), 0, st, planes, P.blocks_w[0] * 8, width, bgr);
    } else {
        // downsampled_width / height of the chroma planes: ceil(image * 1 / max)
        const int dw = (width + hmax - 1) / hmax, dh = (height + vmax - 1) / vmax;
        if (hmax == 1 && vmax == 1)
            hipLaunchKernelGGL(jpeg_color_kernel<0>, grid, dim3(256), 0, st, planes, P, width, height, dw, dh, bgr);
        else if (hmax == 2 && vmax == 1)
            hipLaunchKernelGGL(jpeg_color_kernel<1>, grid, dim3(256), 0, st, planes, P, width, height, dw, dh, bgr);
        else if (hmax == 2 && vmax == 2)
            hipLaunchKernelGGL(jpeg_color_kernel<2>, grid, dim3(256), 0, st, planes, P, width, height, dw, dh, bgr);
        else
            return iamx::fail(IAMX_EUNSUPPORTED, "jpeg: subsampling %dx%d", hmax, vmax);
    }
    return iamx::check_launch("iamx_jpeg_reconstruct");
}
