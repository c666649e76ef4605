// Host-side (no device code): the reference's descriptor cache file straight from the detector's
// uint8 descriptors.
//
// The reference writes <image>.desc as gzip(np.save(des_list)) with des_list float32 [N, 128]
// (scripts/lib/image.py:205-217) -- 25 MB per 50 k-keypoint frame through zlib, ~0.5 s of one
// core even at level 1, which is what bounded fresh detection at ~34 frames/s once the JPEG decode
// had moved to the device.  SIFT descriptors are integers 0..255, so the float32 array is a
// sequence of only 256 different 4-byte patterns [00 00 b2 b3]: this encoder writes a valid gzip
// member whose payload IS the reference's byte stream (.npy header + the float32 array) without
// ever materialising it -- one dynamic-Huffman DEFLATE block of literals (the code is built from
// the histogram of the 256 values; a zero byte costs 1-2 bits), 4 table look-ups per descriptor
// value, CRC-32 by slicing over the same patterns.  ~25 ms per frame instead of ~500, files of the
// size zlib level 1 gives.  Any gzip reader (gzip.open + np.load in the reference) gets the bytes
// np.save would have written.
#include "iamx_common.h"
#include <mutex>

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

struct BitWriter {
    uint8_t *p, *end;
    uint64_t acc = 0;
    int n = 0;
    bool overflow = false;
    inline void put(uint32_t bits, int len)          // LSB first
    {
        acc |= (uint64_t)bits << n;
        n += len;
        while (n >= 8) {
            if (p < end) *p++ = (uint8_t)acc; else overflow = true;
            acc >>= 8;
            n -= 8;
        }
    }
    inline void flush()
    {
        if (n > 0) {
            if (p < end) *p++ = (uint8_t)acc; else overflow = true;
        }
        acc = 0;
        n = 0;
    }
};

inline uint32_t reverse_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Huffman code lengths (<= 15) of the symbols with freq > 0; at least two symbols get a code
void code_lengths(const uint64_t *freq, int n_sym, uint8_t *len)
{
    std::vector<uint64_t> f(freq, freq + n_sym);
    int used = 0;
    for (int s = 0; s < n_sym; ++s) used += f[s] > 0;
    for (int s = 0; used < 2 && s < n_sym; ++s)      // a complete code needs two leaves
        if (f[s] == 0) { f[s] = 1; ++used; }
    for (int shift = 0;; ++shift) {
        // plain Huffman on (scaled) frequencies
        struct Node { uint64_t w; int left, right; };
        std::vector<Node> nodes;
        std::vector<int> heap;
        auto cmp = [&](int a, int b) { return nodes[a].w > nodes[b].w || (nodes[a].w == nodes[b].w && a > b); };
        std::vector<int> leaf_of(n_sym, -1);
        for (int s = 0; s < n_sym; ++s)
            if (f[s] > 0) {
                leaf_of[s] = (int)nodes.size();
                nodes.push_back(Node{std::max<uint64_t>(1, f[s] >> shift), -1, -1});
                heap.push_back(leaf_of[s]);
            }
        std::make_heap(heap.begin(), heap.end(), cmp);
        while (heap.size() > 1) {
            std::pop_heap(heap.begin(), heap.end(), cmp); const int a = heap.back(); heap.pop_back();
            std::pop_heap(heap.begin(), heap.end(), cmp); const int b = heap.back(); heap.pop_back();
            nodes.push_back(Node{nodes[a].w + nodes[b].w, a, b});
            heap.push_back((int)nodes.size() - 1);
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        // depths
        std::vector<int> depth(nodes.size(), 0);
        int maxd = 0;
        for (int i = (int)nodes.size() - 1; i >= 0; --i) {
            if (nodes[i].left >= 0) {
                depth[nodes[i].left] = depth[i] + 1;
                depth[nodes[i].right] = depth[i] + 1;
            } else {
                maxd = std::max(maxd, depth[i]);
            }
        }
        if (maxd <= 15) {
            for (int s = 0; s < n_sym; ++s) len[s] = leaf_of[s] >= 0 ? (uint8_t)depth[leaf_of[s]] : 0;
            return;
        }
    }
}

void canonical_codes(const uint8_t *len, int n_sym, uint16_t *code /* bit reversed, ready to put() */)
{
    int bl_count[16] = {0};
    for (int s = 0; s < n_sym; ++s) bl_count[len[s]]++;
    bl_count[0] = 0;
    int next[16] = {0}, c = 0;
    for (int b = 1; b <= 15; ++b) {
        c = (c + bl_count[b - 1]) << 1;
        next[b] = c;
    }
    for (int s = 0; s < n_sym; ++s)
        code[s] = len[s] ? (uint16_t)reverse_bits((uint32_t)next[len[s]]++, len[s]) : 0;
}

uint32_t g_crc[16][256];
std::once_flag g_crc_once;

// (called with the GIL released from many writer threads: built exactly once, with the ordering
//  std::call_once gives between the table stores and every later reader)
void crc_tables_build()
{
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        g_crc[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = g_crc[0][i];
        for (int t = 1; t < 16; ++t) {
            c = g_crc[0][c & 0xff] ^ (c >> 8);
            g_crc[t][i] = c;
        }
    }
}

void crc_tables() { std::call_once(g_crc_once, crc_tables_build); }

}  // namespace

extern "C" int64_t iamx_gzip_f32_from_u8_bound(int64_t n_header, int64_t n_values)
{
    return 64 + 1024 + 2 * n_header + 8 * n_values;
}

// One gzip member whose payload is `header` followed by the float32 (little endian) values of
// `values` (uint8).  Returns the number of bytes written to out, or a negative error code.
extern "C" int64_t iamx_gzip_f32_from_u8(const uint8_t *header, int64_t n_header, const uint8_t *values,
                                         int64_t n_values, uint8_t *out, int64_t out_cap)
{
    if ((!header && n_header) || (!values && n_values) || !out || n_header < 0 || n_values < 0)
        return iamx::fail(IAMX_EINVAL, "iamx_gzip_f32_from_u8: null pointer");
    if (out_cap < iamx_gzip_f32_from_u8_bound(n_header, n_values))
        return iamx::fail(IAMX_EINVAL, "iamx_gzip_f32_from_u8: output buffer too small");
    crc_tables();
    // the 4 bytes of float32(v), v = 0..255
    uint8_t pat[256][4];
    uint32_t word[256];
    for (int v = 0; v < 256; ++v) {
        const float f = (float)v;
        std::memcpy(pat[v], &f, 4);
        std::memcpy(&word[v], &f, 4);
    }
    // histogram of the values -> literal frequencies
    uint64_t hist[256] = {0};
    {
        // four interleaved counters: runs of equal values (zeros) do not serialise on one slot
        std::vector<uint32_t> h4(4 * 256, 0);
        uint32_t *h = h4.data();
        int64_t i = 0;
        for (; i + 4 <= n_values && i < (int64_t)0xFFFFFFF0; i += 4) {
            h[values[i]]++; h[256 + values[i + 1]]++; h[512 + values[i + 2]]++; h[768 + values[i + 3]]++;
        }
        for (int v = 0; v < 256; ++v) hist[v] = (uint64_t)h[v] + h[256 + v] + h[512 + v] + h[768 + v];
        for (; i < n_values; ++i) hist[values[i]]++;
    }
    uint64_t freq[257] = {0};
    for (int64_t i = 0; i < n_header; ++i) freq[header[i]]++;
    for (int v = 0; v < 256; ++v)
        if (hist[v])
            for (int b = 0; b < 4; ++b) freq[pat[v][b]] += hist[v];
    freq[256] = 1;                                       // end of block
    uint8_t len[257];
    uint16_t code[257];
    code_lengths(freq, 257, len);
    canonical_codes(len, 257, code);
    // gzip header
    uint8_t *p = out;
    const uint8_t gz[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    std::memcpy(p, gz, 10);
    p += 10;
    BitWriter bw;
    bw.p = p;
    bw.end = out + out_cap - 8;
    bw.put(1, 1);                                        // BFINAL
    bw.put(2, 2);                                        // BTYPE = dynamic Huffman
    bw.put(0, 5);                                        // HLIT: 257 literal / length codes
    bw.put(0, 5);                                        // HDIST: 1 distance code (of length 0: literals only)
    bw.put(15, 4);                                       // HCLEN: all 19 code length codes
    // code-length alphabet: symbols 0..15 with 4-bit codes (a complete code), 16..18 unused;
    // transmitted in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    bw.put(0, 3); bw.put(0, 3); bw.put(0, 3);
    for (int i = 0; i < 16; ++i) bw.put(4, 3);
    for (int s = 0; s < 257; ++s) bw.put(reverse_bits(len[s], 4), 4);      // literal / length code lengths
    bw.put(reverse_bits(0, 4), 4);                                            // the one distance code: length 0
    // data: header bytes, then 4 literals per value (one combined bit string per value)
    for (int64_t i = 0; i < n_header; ++i) bw.put(code[header[i]], len[header[i]]);
    uint64_t vbits[256];
    int vlen[256];
    for (int v = 0; v < 256; ++v) {
        uint64_t b = 0;
        int l = 0;
        if (hist[v])
            for (int k = 0; k < 4; ++k) {
                b |= (uint64_t)code[pat[v][k]] << l;
                l += len[pat[v][k]];
            }
        vbits[v] = b;                                    // <= 60 bits
        vlen[v] = l;
    }
    uint32_t crc = 0xFFFFFFFFu;
    for (int64_t i = 0; i < n_header; ++i) crc = g_crc[0][(crc ^ header[i]) & 0xff] ^ (crc >> 8);
    // CRC-32 sixteen bytes (four values) per step.  A pattern's two low bytes are zero and
    // table[.][0] = 0, so the words that are not mixed with the running CRC cost ONE look-up
    // each in a per-position table of the value: u1/u2/u3[v] = slices of bytes 2, 3 of word 1/2/3.
    uint32_t u1[256], u2[256], u3[256];
    for (int v = 0; v < 256; ++v) {
        u1[v] = g_crc[9][pat[v][2]] ^ g_crc[8][pat[v][3]];
        u2[v] = g_crc[5][pat[v][2]] ^ g_crc[4][pat[v][3]];
        u3[v] = g_crc[1][pat[v][2]] ^ g_crc[0][pat[v][3]];
    }
    // bits: a 64-bit accumulator stored whole (the buffer bound leaves > 8 bytes of slack past
    // the longest possible stream: 60 bits per value), advanced by the full bytes it holds
    uint8_t *q = bw.p;
    uint64_t acc = bw.acc;
    int nb = bw.n;                                       // < 8
    auto emit = [&](uint64_t b, int l) {                 // l <= 56
        acc |= b << nb;
        nb += l;
        std::memcpy(q, &acc, 8);
        q += nb >> 3;
        acc >>= nb & ~7;
        nb &= 7;
    };
    auto emit_value = [&](int v) {
        const int l = vlen[v];
        if (l <= 56) {
            emit(vbits[v], l);
        } else {
            emit(vbits[v] & 0xFFFFFFFFull, 32);
            emit(vbits[v] >> 32, l - 32);
        }
    };
    int64_t i = 0;
    for (; i + 4 <= n_values; i += 4) {
        const int v0 = values[i], v1 = values[i + 1], v2 = values[i + 2], v3 = values[i + 3];
        // two values per accumulator step where their bits fit (nearly always: ~20 bits a value)
        const int l01 = vlen[v0] + vlen[v1], l23 = vlen[v2] + vlen[v3];
        if (l01 <= 56) {
            emit(vbits[v0] | (vbits[v1] << vlen[v0]), l01);
        } else {
            emit_value(v0);
            emit_value(v1);
        }
        if (l23 <= 56) {
            emit(vbits[v2] | (vbits[v3] << vlen[v2]), l23);
        } else {
            emit_value(v2);
            emit_value(v3);
        }
        const uint32_t x = crc ^ word[v0];
        crc = g_crc[15][x & 0xff] ^ g_crc[14][(x >> 8) & 0xff] ^ g_crc[13][(x >> 16) & 0xff] ^ g_crc[12][x >> 24]
              ^ u1[v1] ^ u2[v2] ^ u3[v3];
    }
    for (; i < n_values; ++i) {
        const int v = values[i];
        emit_value(v);
        const uint32_t x = crc ^ word[v];
        crc = g_crc[3][x & 0xff] ^ g_crc[2][(x >> 8) & 0xff] ^ g_crc[1][(x >> 16) & 0xff] ^ g_crc[0][x >> 24];
    }
    bw.p = q;
    bw.acc = acc;
    bw.n = nb;
    bw.put(code[256], len[256]);
    bw.flush();
    if (bw.overflow) return iamx::fail(IAMX_EINVAL, "iamx_gzip_f32_from_u8: output buffer too small");
    p = bw.p;
    crc ^= 0xFFFFFFFFu;
    const uint32_t isize = (uint32_t)((uint64_t)(n_header + 4 * n_values) & 0xFFFFFFFFu);
    for (int k = 0; k < 4; ++k) *p++ = (uint8_t)(crc >> (8 * k));
    for (int k = 0; k < 4; ++k) *p++ = (uint8_t)(isize >> (8 * k));
    return (int64_t)(p - out);
}

// =====================================================================================
// The rest of the cache writers' byte work, each as ONE call without the interpreter lock.
// Measured on the 256-thread host of the GPU box (tools/detect_stages.py): with the members of a
// .feat file compressed by eight python futures, the float32 conversion by eight more and the
// pickle records assembled by numpy field assignments, a thread that wakes up needs 1-4 ms to get
// the lock back and a fresh detection costs 11.5 ms of serialised host time per frame whatever the
// number of workers.
// =====================================================================================
#include <zlib.h>

#include <atomic>
#include <thread>

namespace {

struct Chunk {
    const uint8_t *src;
    size_t len;
    uint8_t *dst;            // its own region of the output (deflateBound of the chunk)
    size_t cap, out;
    int rc;
};

void deflate_chunk(Chunk &c, int level, int strategy)
{
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    c.rc = deflateInit2(&zs, level, Z_DEFLATED, 31, 8, strategy);     // 31: gzip container
    if (c.rc != Z_OK) return;
    zs.next_in = const_cast<Bytef *>(c.src);
    zs.avail_in = (uInt)c.len;
    zs.next_out = c.dst;
    zs.avail_out = (uInt)c.cap;
    const int r = deflate(&zs, Z_FINISH);
    c.out = c.cap - zs.avail_out;
    c.rc = r == Z_STREAM_END ? Z_OK : (r == Z_OK ? Z_BUF_ERROR : r);
    deflateEnd(&zs);
}

inline size_t member_bound(size_t len) { return len + (len >> 10) + 64; }   // >= deflateBound + gzip wrapper

}  // namespace

// upper bound of iamx_gzip_members' output for `total` input bytes in `n_members` members
extern "C" int64_t iamx_gzip_members_bound(int64_t total, int64_t n_members)
{
    if (total < 0 || n_members < 0) return 0;
    return total + (total >> 10) + 64 * (n_members + 1);
}

// gzip members (each at most member_bytes of input, never spanning two buffers; zlib `level` and
// `strategy`: 0 default, 1 filtered, 2 Huffman only, 3 RLE, 4 fixed) of the buffers
// bufs[0..n_bufs), compressed on up to `threads` threads and written back to back: a multi-member
// gzip stream that decompresses to the concatenation of the buffers.  Returns the number of bytes
// written or a negative error code.
extern "C" int64_t iamx_gzip_members(const uint8_t *const *bufs, const int64_t *lens, int n_bufs,
                                     int64_t member_bytes, int level, int strategy, int threads,
                                     uint8_t *out, int64_t out_cap)
{
    if (!bufs || !lens || n_bufs < 0 || !out || member_bytes < 1 || member_bytes > (1ll << 30) || level < 0 ||
        level > 9 || strategy < 0 || strategy > 4)
        return iamx::fail(IAMX_EINVAL, "iamx_gzip_members: bad argument");
    std::vector<Chunk> chunks;
    size_t need = 0;
    for (int b = 0; b < n_bufs; ++b) {
        if (lens[b] < 0 || (lens[b] > 0 && !bufs[b])) return iamx::fail(IAMX_EINVAL, "iamx_gzip_members: bad buffer");
        for (int64_t o = 0; o < lens[b]; o += member_bytes) {
            const size_t n = (size_t)std::min<int64_t>(member_bytes, lens[b] - o);
            chunks.push_back(Chunk{bufs[b] + o, n, nullptr, member_bound(n), 0, Z_OK});
            need += member_bound(n);
        }
    }
    if (chunks.empty()) {
        chunks.push_back(Chunk{reinterpret_cast<const uint8_t *>(""), 0, nullptr, member_bound(0), 0, Z_OK});
        need = member_bound(0);
    }
    if ((size_t)out_cap < need) return iamx::fail(IAMX_EINVAL, "iamx_gzip_members: output buffer too small");
    size_t off = 0;
    for (Chunk &c : chunks) {
        c.dst = out + off;
        off += c.cap;
    }
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), chunks.size()));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < chunks.size(); i = next.fetch_add(1)) deflate_chunk(chunks[i], level, strategy);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    size_t w = 0;
    for (Chunk &c : chunks) {
        if (c.rc != Z_OK) return iamx::fail(IAMX_EINVAL, "iamx_gzip_members: zlib error %d", c.rc);
        if (out + w != c.dst) std::memmove(out + w, c.dst, c.out);
        w += c.out;
    }
    return (int64_t)w;
}

// =====================================================================================
// DEFLATE for streams of fixed-width records (the .feat pickle: 58 bytes per keypoint,
// keypoints.py _REC).  Such a stream has one kind of redundancy -- a byte equals the byte one
// record earlier (the pickle opcodes, the zero tails of float32 values widened to float64, class_id
// = -1, exponent bytes) -- and zlib finds it through hash chains over every 3-byte window: 50-70 ms
// of one core per 50 k-keypoint frame at level 4, the largest host cost of a fresh detection (more
// than the JPEG's Huffman decode).  This encoder looks at distance `record_bytes` ONLY: one compare
// pass makes the tokens (literal | match of length >= 4 at that distance), a dynamic Huffman code is
// built from their histogram, one more pass writes the bits.  Valid gzip members (any reader
// inflates them to the same payload); smaller than zlib's level 4 output and ~5x faster.
// =====================================================================================
namespace {

inline uint32_t crc32_bytes(const uint8_t *p, size_t n)
{
    uint32_t crc = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t a, b;
        std::memcpy(&a, p, 4);
        std::memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = g_crc[7][a & 0xff] ^ g_crc[6][(a >> 8) & 0xff] ^ g_crc[5][(a >> 16) & 0xff] ^ g_crc[4][a >> 24] ^
              g_crc[3][b & 0xff] ^ g_crc[2][(b >> 8) & 0xff] ^ g_crc[1][(b >> 16) & 0xff] ^ g_crc[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = g_crc[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFu;
}

// length symbol / extra bits of a match length 3..258 (RFC 1951 3.2.5)
struct LenCode { uint16_t sym; uint8_t extra_bits; uint16_t extra; };
inline LenCode length_code(int len)
{
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                      35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t ebits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                      3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    int k = 28;
    while (base[k] > len) --k;
    return LenCode{(uint16_t)(257 + k), ebits[k], (uint16_t)(len - base[k])};
}

struct DistCode { int sym, extra_bits, extra; };
inline DistCode distance_code(int dist)              // 1..32768
{
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513,
                                      769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t ebits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
                                      9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int k = 29;
    while (base[k] > dist) --k;
    return DistCode{k, ebits[k], dist - base[k]};
}

// one gzip member of src[0, len) with matches at distance D only; returns bytes written or -1
int64_t record_member(const uint8_t *src, size_t len, int D, uint8_t *out, size_t cap)
{
    constexpr int MIN_MATCH = 4, MAX_MATCH = 258;
    // tokens: < 256 literal, >= 256: match of length (tok - 256 + MIN_MATCH - ... ) -- kept as
    // 0x8000 | length
    std::vector<uint16_t> tok;
    tok.reserve(len / 2 + 16);
    uint64_t freq[286] = {0};
    size_t n_match = 0;
    {
        size_t i = 0;
        const size_t first = std::min<size_t>((size_t)D, len);
        for (; i < first; ++i) { tok.push_back(src[i]); freq[src[i]]++; }
        while (i < len) {
            // run of bytes equal to the ones D earlier, 8 at a time
            size_t L = 0;
            const size_t maxl = std::min<size_t>((size_t)MAX_MATCH, len - i);
            while (L + 8 <= maxl) {
                uint64_t a, b;
                std::memcpy(&a, src + i + L, 8);
                std::memcpy(&b, src + i + L - D, 8);
                const uint64_t x = a ^ b;
                if (x) { L += (size_t)(__builtin_ctzll(x) >> 3); goto counted; }
                L += 8;
            }
            while (L < maxl && src[i + L] == src[i + L - D]) ++L;
        counted:
            if (L >= (size_t)MIN_MATCH) {
                tok.push_back((uint16_t)(0x8000u | L));
                freq[length_code((int)L).sym]++;
                ++n_match;
                i += L;
            } else {
                // (the bytes up to the mismatch become literals too: a short run does not pay)
                const size_t lit = L + 1 <= len - i ? L + 1 : len - i;
                for (size_t k = 0; k < lit; ++k) { tok.push_back(src[i + k]); freq[src[i + k]]++; }
                i += lit;
            }
        }
    }
    freq[256] = 1;
    uint8_t llen[286];
    uint16_t lcode[286];
    code_lengths(freq, 286, llen);
    canonical_codes(llen, 286, lcode);
    const DistCode dc = distance_code(D);
    if (cap < 18 + 400) return -1;
    uint8_t *p = out;
    const uint8_t gz[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    std::memcpy(p, gz, 10);
    p += 10;
    BitWriter bw;
    bw.p = p;
    bw.end = out + cap - 8;
    const int n_dist = n_match ? dc.sym + 1 : 1;
    bw.put(1, 1);                                        // BFINAL
    bw.put(2, 2);                                        // BTYPE = dynamic Huffman
    bw.put(286 - 257, 5);                                // HLIT
    bw.put((uint32_t)(n_dist - 1), 5);                   // HDIST
    bw.put(15, 4);                                       // HCLEN: all 19 code length codes
    // code-length alphabet: symbols 0..15 with 4-bit codes (complete), 16..18 unused; order
    // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    bw.put(0, 3); bw.put(0, 3); bw.put(0, 3);
    for (int k = 0; k < 16; ++k) bw.put(4, 3);
    for (int sy = 0; sy < 286; ++sy) bw.put(reverse_bits(llen[sy], 4), 4);
    // distance code lengths: the one distance in use has a 1-bit code ("0"), the others none
    for (int d = 0; d < n_dist; ++d) bw.put(reverse_bits((n_match && d == dc.sym) ? 1u : 0u, 4), 4);
    // a 64-bit accumulator stored whole (slack past the end is guaranteed by the bound)
    uint8_t *q = bw.p;
    uint8_t *const qend = out + cap - 16;
    uint64_t acc = bw.acc;
    int nb = bw.n;
    auto emit = [&](uint64_t b, int l) {                 // l <= 56
        acc |= b << nb;
        nb += l;
        std::memcpy(q, &acc, 8);
        q += nb >> 3;
        acc >>= nb & ~7;
        nb &= 7;
    };
    // per match length: the whole bit string (length code + extra, distance code "0" + extra)
    uint64_t mbits[MAX_MATCH + 1];
    uint8_t mlen[MAX_MATCH + 1];
    for (int L = MIN_MATCH; L <= MAX_MATCH; ++L) {
        const LenCode lc = length_code(L);
        uint64_t b = lcode[lc.sym];
        int l = llen[lc.sym];
        b |= (uint64_t)lc.extra << l;
        l += lc.extra_bits;
        l += 1;                                          // the distance code: one 0 bit
        b |= (uint64_t)dc.extra << l;
        l += dc.extra_bits;
        mbits[L] = b;
        mlen[L] = (uint8_t)l;                            // <= 15 + 5 + 1 + 13
    }
    for (size_t k = 0; k < tok.size(); ++k) {
        if (q >= qend) return -1;
        const uint16_t t = tok[k];
        if (t & 0x8000u) emit(mbits[t & 0x7FFF], mlen[t & 0x7FFF]);
        else emit(lcode[t], llen[t]);
    }
    emit(lcode[256], llen[256]);
    bw.p = q;
    bw.acc = acc;
    bw.n = nb;
    bw.flush();
    if (bw.overflow) return -1;
    p = bw.p;
    const uint32_t crc = crc32_bytes(src, len);
    const uint32_t isize = (uint32_t)(len & 0xFFFFFFFFu);
    for (int k = 0; k < 4; ++k) *p++ = (uint8_t)(crc >> (8 * k));
    for (int k = 0; k < 4; ++k) *p++ = (uint8_t)(isize >> (8 * k));
    return (int64_t)(p - out);
}

}  // namespace

// upper bound of iamx_gzip_records' output: the bound the member encoder itself works to -- its
// code lengths are depth-limited by rescaling the histogram, so "less than 9 bits per byte" of an
// optimal Huffman code is not guaranteed; 15 bits per literal is (2 bytes per input byte), plus
// the code table and the gzip wrapper of every member
extern "C" int64_t iamx_gzip_records_bound(int64_t total, int64_t n_members)
{
    if (total < 0 || n_members < 0) return 0;
    return 2 * total + 1024 * (n_members + 1);
}

// gzip members of the buffers bufs[0..n_bufs) (as iamx_gzip_members: each member at most
// member_bytes of input, never spanning two buffers, compressed on up to `threads` threads, written
// back to back) with the record encoder above: matches at distance record_bytes only.  Returns the
// number of bytes written or a negative error code; out_cap >= iamx_gzip_records_bound().
extern "C" int64_t iamx_gzip_records(const uint8_t *const *bufs, const int64_t *lens, int n_bufs,
                                     int64_t member_bytes, int record_bytes, int threads, uint8_t *out,
                                     int64_t out_cap)
{
    if (!bufs || !lens || n_bufs < 0 || !out || member_bytes < 1 || member_bytes > (1ll << 30) ||
        record_bytes < 1 || record_bytes > 32768)
        return iamx::fail(IAMX_EINVAL, "iamx_gzip_records: bad argument");
    crc_tables();
    struct Part { const uint8_t *src; size_t len; uint8_t *dst; size_t cap; int64_t out; };
    std::vector<Part> parts;
    // (a literal costs at most 15 bits: twice the input + the 150-byte code table is a safe bound;
    //  the parts are compacted afterwards)
    auto bound = [](size_t n) { return 2 * n + 1024; };
    size_t need = 0;
    for (int b = 0; b < n_bufs; ++b) {
        if (lens[b] < 0 || (lens[b] > 0 && !bufs[b])) return iamx::fail(IAMX_EINVAL, "iamx_gzip_records: bad buffer");
        for (int64_t o = 0; o < lens[b]; o += member_bytes) {
            const size_t n = (size_t)std::min<int64_t>(member_bytes, lens[b] - o);
            parts.push_back(Part{bufs[b] + o, n, nullptr, bound(n), 0});
            need += bound(n);
        }
    }
    if (parts.empty()) {
        parts.push_back(Part{reinterpret_cast<const uint8_t *>(""), 0, nullptr, bound(0), 0});
        need = bound(0);
    }
    // the members are encoded into a scratch area of the worst-case size and copied together
    std::vector<uint8_t> scratch(need);
    size_t off = 0;
    for (Part &c : parts) {
        c.dst = scratch.data() + off;
        off += c.cap;
    }
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), parts.size()));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < parts.size(); i = next.fetch_add(1))
            parts[i].out = record_member(parts[i].src, parts[i].len, record_bytes, parts[i].dst, parts[i].cap);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    size_t w = 0;
    for (Part &c : parts) {
        if (c.out < 0) return iamx::fail(IAMX_EINVAL, "iamx_gzip_records: member overflow");
        if (w + (size_t)c.out > (size_t)out_cap) return iamx::fail(IAMX_EINVAL, "iamx_gzip_records: output buffer too small");
        std::memcpy(out + w, c.dst, (size_t)c.out);
        w += (size_t)c.out;
    }
    return (int64_t)w;
}

// dst[i] = (float)src[i]: the reference's float32 des_list from the detector's uint8 descriptors
extern "C" int iamx_u8_to_f32(const uint8_t *src, float *dst, int64_t n, int threads)
{
    if (n < 0 || (n > 0 && (!src || !dst))) return iamx::fail(IAMX_EINVAL, "iamx_u8_to_f32: null pointer");
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(threads, 1), n >> 18));
    auto part = [&](int t) {
        const int64_t a = n * t / nt, b = n * (t + 1) / nt;
        for (int64_t i = a; i < b; ++i) dst[i] = (float)src[i];
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(part, t);
    part(0);
    for (std::thread &t : pool) t.join();
    return IAMX_OK;
}

// Many images' float32 descriptors (the reference's des_list: integer valued 0..255) -> uint8,
// back to back in dst: round-half-even + clamp like the device pack kernels' own conversion
// (csrc/match_knn2*.hip load_value).  srcs HOST [n] pointers, rows HOST [n] elements per image.
// The threads split the total evenly (an image may be shared by two threads).
extern "C" int iamx_f32_to_u8_many(const float *const *srcs, const int64_t *counts, int n, uint8_t *dst,
                                   int threads)
{
    if (n < 0 || (n > 0 && (!srcs || !counts || !dst))) return iamx::fail(IAMX_EINVAL, "iamx_f32_to_u8_many: null pointer");
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (counts[i] < 0 || (counts[i] > 0 && !srcs[i])) return iamx::fail(IAMX_EINVAL, "iamx_f32_to_u8_many: bad image");
        off[(size_t)i + 1] = off[(size_t)i] + counts[i];
    }
    const int64_t total = off[(size_t)n];
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(threads, 1), total >> 18));
    auto part = [&](int t) {
        const int64_t a = total * t / nt, b = total * (t + 1) / nt;
        int img = (int)(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;
        for (int64_t e = a; e < b;) {
            while (off[(size_t)img + 1] <= e) ++img;
            const int64_t stop = std::min(b, off[(size_t)img + 1]);
            const float *s = srcs[img] + (e - off[(size_t)img]);
            uint8_t *d = dst + e;
            for (int64_t i = 0; i < stop - e; ++i) {
                const float v = rintf(s[i]);
                d[i] = (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
            }
            e = stop;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(part, t);
    part(0);
    for (std::thread &t : pool) t.join();
    return IAMX_OK;
}

// n host byte blocks back to back in dst (the uint8 descriptor arrays of a group of images into ONE
// page-locked staging buffer: a numpy loop copies 5 GB/s on one core, the upload that waits for it
// 25).  srcs HOST [n] pointers, counts HOST [n] bytes per block.  The threads split the total evenly.
extern "C" int iamx_u8_gather_many(const uint8_t *const *srcs, const int64_t *counts, int n, uint8_t *dst,
                                   int threads)
{
    if (n < 0 || (n > 0 && (!srcs || !counts || !dst))) return iamx::fail(IAMX_EINVAL, "iamx_u8_gather_many: null pointer");
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (counts[i] < 0 || (counts[i] > 0 && !srcs[i])) return iamx::fail(IAMX_EINVAL, "iamx_u8_gather_many: bad block");
        off[(size_t)i + 1] = off[(size_t)i] + counts[i];
    }
    const int64_t total = off[(size_t)n];
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(threads, 1), total >> 20));
    auto part = [&](int t) {
        const int64_t a = total * t / nt, b = total * (t + 1) / nt;
        int img = (int)(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;
        for (int64_t e = a; e < b;) {
            while (off[(size_t)img + 1] <= e) ++img;
            const int64_t stop = std::min(b, off[(size_t)img + 1]);
            std::memcpy(dst + e, srcs[img] + (e - off[(size_t)img]), (size_t)(stop - e));
            e = stop;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(part, t);
    part(0);
    for (std::thread &t : pool) t.join();
    return IAMX_OK;
}

// The .feat pickle's fixed-width records (imageanalysis_amd/keypoints.py _REC: protocol-2 opcodes
// "( G x G y TUPLE2 G size G angle G response J octave J class_id t", BINFLOAT big endian,
// BININT little endian) from the keypoint columns: out [n][58].
extern "C" int iamx_feat_records(const float *x, const float *y, const float *size, const float *angle,
                                 const float *response, const int32_t *octave, const int32_t *class_id,
                                 int64_t n, uint8_t *out)
{
    if (n < 0 || (n > 0 && (!x || !y || !size || !angle || !response || !octave || !class_id || !out)))
        return iamx::fail(IAMX_EINVAL, "iamx_feat_records: null pointer");
    auto put_f64 = [](uint8_t *p, float v) {
        const double d = (double)v;
        uint64_t u;
        std::memcpy(&u, &d, 8);
        u = __builtin_bswap64(u);
        std::memcpy(p, &u, 8);
    };
    for (int64_t i = 0; i < n; ++i) {
        uint8_t *p = out + 58 * i;
        p[0] = '(';
        p[1] = 'G';  put_f64(p + 2, x[i]);
        p[10] = 'G'; put_f64(p + 11, y[i]);
        p[19] = 0x86;
        p[20] = 'G'; put_f64(p + 21, size[i]);
        p[29] = 'G'; put_f64(p + 30, angle[i]);
        p[38] = 'G'; put_f64(p + 39, response[i]);
        p[47] = 'J'; std::memcpy(p + 48, &octave[i], 4);
        p[52] = 'J'; std::memcpy(p + 53, &class_id[i], 4);
        p[57] = 't';
    }
    return IAMX_OK;
}

// The pair lists of the .match pickles (Image.save_matches, scripts/lib/image.py:222-233: a dict
// {other image: [[i, j], ...]}): list k = rows [off[k], off[k+1]) of `pairs` as the protocol-2
// stream  ] ( { ] ( <int> i <int> j e }* e   (no memo entries; BININT2 'M' when every index of the
// list fits 16 bits, else BININT 'J'), an empty list as  ] .  out_off[k] .. out_off[k+1] = the
// bytes of list k in out (out_cap >= sum over lists of 3 + 13 rows).  Returns bytes written.
extern "C" int64_t iamx_pickle_pair_lists(const int32_t *pairs, const int64_t *off, int64_t n_lists,
                                          uint8_t *out, int64_t out_cap, int64_t *out_off)
{
    if (n_lists < 0 || (n_lists > 0 && (!off || !out || !out_off)))
        return iamx::fail(IAMX_EINVAL, "iamx_pickle_pair_lists: null pointer");
    uint8_t *p = out;
    for (int64_t k = 0; k < n_lists; ++k) {
        const int64_t a = off[k], b = off[k + 1];
        out_off[k] = (int64_t)(p - out);
        if (b < a || (b > a && !pairs)) return iamx::fail(IAMX_EINVAL, "iamx_pickle_pair_lists: bad offsets");
        if ((int64_t)(p - out) + 3 + 13 * (b - a) > out_cap)
            return iamx::fail(IAMX_EINVAL, "iamx_pickle_pair_lists: output buffer too small");
        if (b == a) {
            *p++ = ']';
            continue;
        }
        bool narrow = true;
        for (int64_t r = 2 * a; r < 2 * b; ++r) narrow &= (uint32_t)pairs[r] < 65536u;
        *p++ = ']';
        *p++ = '(';
        if (narrow) {
            for (int64_t r = a; r < b; ++r) {
                const uint16_t i = (uint16_t)pairs[2 * r], j = (uint16_t)pairs[2 * r + 1];
                p[0] = ']'; p[1] = '('; p[2] = 'M';
                std::memcpy(p + 3, &i, 2);
                p[5] = 'M';
                std::memcpy(p + 6, &j, 2);
                p[8] = 'e';
                p += 9;
            }
        } else {
            for (int64_t r = a; r < b; ++r) {
                p[0] = ']'; p[1] = '('; p[2] = 'J';
                std::memcpy(p + 3, &pairs[2 * r], 4);
                p[7] = 'J';
                std::memcpy(p + 8, &pairs[2 * r + 1], 4);
                p[12] = 'e';
                p += 13;
            }
        }
        *p++ = 'e';
    }
    if (n_lists > 0) out_off[n_lists] = (int64_t)(p - out);
    return (int64_t)(p - out);
}
