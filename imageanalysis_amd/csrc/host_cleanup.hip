// Host-side (no device code) match consolidation -- scripts/lib/match_cleanup.py:246-301
// link_matches(): union of the pair-wise matches into per-feature chains.  The reference does it
// with python dicts keyed by "%d-%d" strings, one full pass over all matches per iteration until
// a pass no longer shrinks the list.  The rules are order dependent and are kept verbatim:
//   * a match joins the chain of the FIRST of its points (in stored order) that was seen before;
//   * joining appends only the points whose IMAGE is not yet in that chain, and only appended
//     points become look-up keys of the chain;
//   * a match none of whose points was seen starts a new chain and registers all its points.
// at 2812-10 k images this serial python is the wall-clock bottleneck after
// GPU matching; here it is flat arrays + one hash map, O(points) per pass.
#include "iamx_common.h"
#include <algorithm>
#include <atomic>
#include <chrono>

#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace {

// A fixed-size array in anonymous memory with MADV_HUGEPAGE (where the kernel grants transparent
// huge pages: a first touch per 2 MiB instead of per 4 KiB).  The linking pass of a 512-frame
// survey touches ~1 GB of fresh arrays once; as std::vectors their page faults cost more than the
// pass itself.  Falls back to plain pages silently; contents are zero after construction.
// Whether MADV_HUGEPAGE pays on THIS host, decided once per process by touching 32 MiB each way.
// With free huge pages a first touch maps 2 MiB at a time (10x fewer faults: measured in round 4 on
// the GPU box and in the build container).  On a host whose memory is fragmented the same advice
// makes every fault wait for the kernel's direct compaction (defrag = madvise): round 5's build
// container took 0.9 s to copy 144 MB into such a mapping, 0.1 s into plain pages, and ran the
// linking pass 2.3x slower on top.  IAMX_THP=0 / 1 overrides the probe.
inline double seconds_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline double touch_probe(bool huge_advice)
{
    const size_t huge = (size_t)2 << 20, bytes = (size_t)32 << 20;
    void *m = mmap(nullptr, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return 1e9;
    char *p = reinterpret_cast<char *>(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
    (void)madvise(p, bytes, huge_advice ? MADV_HUGEPAGE : MADV_NOHUGEPAGE);
    const double t0 = seconds_now();
    for (size_t o = 0; o < bytes; o += 4096) p[o] = 1;
    const double dt = seconds_now() - t0;
    munmap(m, bytes + huge);
    return dt;
}

inline bool thp_pays()
{
    static const bool v = []() {
        if (const char *e = getenv("IAMX_THP")) return e[0] == '1';
        const double plain = touch_probe(false), huge = touch_probe(true);
        return huge <= 1.5 * plain;
    }();
    return v;
}

template <class T>
struct HugeBuf {
    T *p = nullptr;
    size_t n = 0, bytes = 0;
    explicit HugeBuf(size_t count) : n(count)
    {
        const size_t huge = (size_t)2 << 20;
        bytes = ((count * sizeof(T) + huge - 1) / huge + 1) * huge;
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        if (thp_pays()) (void)madvise(m, bytes, MADV_HUGEPAGE);
        p = static_cast<T *>(m);
    }
    ~HugeBuf() { if (p) munmap(p, bytes); }
    HugeBuf(const HugeBuf &) = delete;
    HugeBuf &operator=(const HugeBuf &) = delete;
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *data() { return p; }
};

// open-addressing map (img, kp) -> chain index, cleared per pass
struct PointMap {
    std::vector<uint64_t> key;          // 0 = empty, else ((img << 32) | kp) + 1
    std::vector<int32_t> val;
    uint64_t mask;

    explicit PointMap(size_t n_points)
    {
        size_t cap = 16;
        while (cap < 2 * n_points + 16) cap <<= 1;
        key.assign(cap, 0);
        val.resize(cap);
        mask = cap - 1;
    }
    void clear() { std::memset(key.data(), 0, key.size() * sizeof(uint64_t)); }
    static uint64_t mix(uint64_t k)
    {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
        return k;
    }
    static uint64_t code(int32_t img, int32_t kp) { return (((uint64_t)(uint32_t)img << 32) | (uint32_t)kp) + 1; }
    // slot of the key, or of the empty slot where it would go
    size_t slot(uint64_t c) const
    {
        size_t h = (size_t)(mix(c) & mask);
        while (key[h] != 0 && key[h] != c) h = (h + 1) & mask;
        return h;
    }
};

}  // namespace

// 1 when transparent huge pages speed up first-touch on this host right now (probed once per
// process; what HugeBuf and matchpairs.empty_huge() go by), else 0
extern "C" int iamx_thp_pays(void) { return thp_pays() ? 1 : 0; }

namespace {
// what fills the first pass's input: either the caller's flat chains, or blocks of pair matches
struct LinkInput {
    // form A: chains [ptr[i], ptr[i+1]) of (img[], kp[])
    const int32_t *img = nullptr, *kp = nullptr;
    const int64_t *ptr = nullptr;
    // form B: n_blocks blocks of pair matches; block b = counts[b] rows [kp of image ij[2b],
    // kp of image ij[2b+1]] (int32 [counts[b]][2], the arrays behind the images' match lists)
    const int32_t *const *blocks = nullptr;
    const int64_t *counts = nullptr;
    const int32_t *ij = nullptr;
    int64_t n_blocks = 0;
};

int64_t link_impl(const LinkInput &in, int64_t n_matches, int64_t n_pts, int32_t *out_img,
                  int32_t *out_kp, int64_t *out_ptr, int32_t *n_passes)
{
    try {
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // current pass input: flat points + offsets (rewritten in place per pass)
    HugeBuf<int32_t> c_img((size_t)n_pts), c_kp((size_t)n_pts);
    HugeBuf<int64_t> c_ptr((size_t)n_matches + 1);
    if (in.blocks) {
        // pair matches: chain m = points 2m, 2m + 1; written straight into the work arrays (the
        // python side used to concatenate the blocks, repeat (i, j) per match and build the
        // offsets 0, 2, 4, ...: three fresh arrays of the size of these, then copied here)
        int64_t m = 0;
        for (int64_t b = 0; b < in.n_blocks; ++b) {
            const int32_t i = in.ij[2 * b], j = in.ij[2 * b + 1];
            const int32_t *blk = in.blocks[b];
            for (int64_t k = 0; k < in.counts[b]; ++k, ++m) {
                c_img[(size_t)(2 * m)] = i;
                c_img[(size_t)(2 * m + 1)] = j;
                c_kp[(size_t)(2 * m)] = blk[2 * k];
                c_kp[(size_t)(2 * m + 1)] = blk[2 * k + 1];
            }
        }
        for (int64_t q = 0; q <= n_matches; ++q) c_ptr[(size_t)q] = 2 * q;
    } else {
        std::memcpy(c_img.data(), in.img, (size_t)n_pts * sizeof(int32_t));
        std::memcpy(c_kp.data(), in.kp, (size_t)n_pts * sizeof(int32_t));
        std::memcpy(c_ptr.data(), in.ptr, (size_t)(n_matches + 1) * sizeof(int64_t));
    }
    const int32_t *img = c_img.data(), *kp = c_kp.data();   // (the scans below read the first pass's input)
    // chains under construction: every appended point is a NODE (image, keypoint, chain) written
    // sequentially in append order; a pass ends with a stable counting sort of the nodes by chain
    // (the chains in creation order, their points in insertion order).  Rounds 2-4 linked a
    // chain's nodes through next pointers and rebuilt the flat form by walking every list: one
    // DEPENDENT cache miss per point and pass (12 M on a 512-frame survey's first pass, a second
    // of the stage); the scatter's misses are independent of each other and overlap.
    HugeBuf<int32_t> node_img((size_t)n_pts), node_kp((size_t)n_pts), node_chain((size_t)n_pts);
    int64_t n_chains = 0;
    // the images of a chain's first INL points, side by side (one cache line): the "image already
    // in the chain?" test of a join reads it instead of the chain's scattered nodes; the rare
    // longer chain keeps the rest of its images on an overflow list
    constexpr int INL = 15;
    struct ChainImgs { int32_t n; int32_t img[INL]; };
    static_assert(sizeof(ChainImgs) == 64, "one cache line per chain");
    HugeBuf<ChainImgs> cimg((size_t)n_matches + 1);
    HugeBuf<int32_t> ovf_head((size_t)n_matches + 1), ovf_next((size_t)n_pts);
    HugeBuf<int64_t> cursor((size_t)n_matches + 2);
    // (image, keypoint) -> chain: a DENSE table over the keypoints that occur (image i owns
    // [base[i], base[i] + max keypoint index of i + 1)) -- one 4-byte access per look-up where
    // the hash map of rounds 1-3 paid two cache misses (key, value) and the mixing; the entries
    // of the matches a few steps ahead are prefetched, since the walk order is known: the pass
    // is bound by memory latency (6.4 M points on a 128-frame survey, several passes).
    // A negative index or more than 2^31 table entries fall back to the hash map.
    int32_t n_img = 0;
    bool dense = getenv("IAMX_LINK_HASH") == nullptr;      // (A/B switch: the hash map of rounds 1-3)
    for (int64_t j = 0; j < n_pts; ++j) {
        if (img[j] < 0 || kp[j] < 0) { dense = false; break; }
        if (img[j] >= n_img) n_img = img[j] + 1;
    }
    std::vector<int64_t> base;
    size_t table_n = 0;
    if (dense) {
        base.assign((size_t)n_img + 1, 0);
        for (int64_t j = 0; j < n_pts; ++j)
            if (kp[j] + 1 > base[(size_t)img[j] + 1]) base[(size_t)img[j] + 1] = kp[j] + 1;
        for (int32_t i = 0; i < n_img; ++i) base[(size_t)i + 1] += base[(size_t)i];
        if (base[(size_t)n_img] >= (1LL << 31)) dense = false;
        else table_n = (size_t)base[(size_t)n_img];
    }
    HugeBuf<int32_t> table(dense ? table_n : 1);
    PointMap map(dense ? 0 : (size_t)n_pts);
    constexpr int64_t AHEAD = 48;                         // points of look-ahead for the prefetch
    int passes = 0;
    int64_t n_cur = n_matches;
    // The reference repeats the pass until one changes nothing.  A pass can only join chains that
    // SHARE a point, and a chain it builds shares a point with another one only when a join
    // appends a point that is already registered to a different chain (every other point of a
    // pass is registered once).  A pass that never did that leaves pairwise disjoint chains: the
    // next one would copy them unchanged -- it is counted, not run.  (IAMX_LINK_VERIFY=1 runs it.)
    const bool verify = getenv("IAMX_LINK_VERIFY") != nullptr;
    const bool timing = getenv("IAMX_LINK_TIMING") != nullptr;        // per-pass seconds on stderr
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (timing) fprintf(stderr, "iamx_link_matches setup: %.3f s\n", now() - t_enter);
    // chains of a pass's OUTPUT that share a point with another one (ids = creation order = the
    // next pass's input order): the two ends of every join that appended a point already
    // registered elsewhere.  When they are few the next pass only has to look at them (below).
    std::vector<int32_t> touched;
    bool touched_complete = true;
    const bool full_only = verify || getenv("IAMX_LINK_FULL") != nullptr;    // (A/B: every pass the full walk)
    while (true) {
        ++passes;
        const double t_pass = now();
        bool shared = false;
        // ---- incremental pass: a chain that shares no point with any other chain is copied by a
        // full pass exactly as it is (nothing it looks up is registered, nothing it registers is
        // looked up by anybody else), so only the chains named in `touched` are walked -- in input
        // order, against a small hash map over THEIR points -- and the result is assembled by one
        // sequential copy.  Same output as the full walk (IAMX_LINK_FULL=1 / IAMX_LINK_VERIFY=1 run
        // that one; tests compare); the late passes of a survey merge a handful of chains and
        // cost a full walk over ~10^8 points each.
        if (!full_only && passes > 1 && touched_complete && (int64_t)touched.size() * 8 < n_cur) {
            std::sort(touched.begin(), touched.end());
            touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
            const size_t nt_ = touched.size();
            struct Root { std::vector<int32_t> ex_img, ex_kp, imgs; };
            std::vector<int32_t> root_of(nt_, -1);           // per touched chain: its root (itself or the chain it joined)
            std::vector<uint8_t> is_root(nt_, 0);
            std::vector<Root> roots(nt_);
            size_t pts_t = 0;
            for (size_t k = 0; k < nt_; ++k) pts_t += (size_t)(c_ptr[(size_t)touched[k] + 1] - c_ptr[(size_t)touched[k]]);
            PointMap H(pts_t);
            std::vector<int32_t> next_touched;               // (root positions among `touched`, mapped below)
            for (size_t k = 0; k < nt_; ++k) {
                const int64_t b = c_ptr[(size_t)touched[k]], e = c_ptr[(size_t)touched[k] + 1];
                int32_t index = -1;
                for (int64_t j = b; j < e; ++j) {
                    const size_t h = H.slot(PointMap::code(c_img[j], c_kp[j]));
                    if (H.key[h] != 0) { index = H.val[h]; break; }
                }
                if (index < 0) {
                    is_root[k] = 1;
                    root_of[k] = (int32_t)k;
                    Root &R = roots[k];
                    for (int64_t j = b; j < e; ++j) {
                        const uint64_t c = PointMap::code(c_img[j], c_kp[j]);
                        const size_t h = H.slot(c);
                        H.key[h] = c;
                        H.val[h] = (int32_t)k;
                        R.imgs.push_back(c_img[j]);
                    }
                } else {
                    root_of[k] = index;
                    Root &R = roots[(size_t)index];
                    for (int64_t j = b; j < e; ++j) {
                        const int32_t pi = c_img[j], pk = c_kp[j];
                        bool have = false;
                        for (int32_t im : R.imgs) if (im == pi) { have = true; break; }
                        if (have) continue;
                        const uint64_t c = PointMap::code(pi, pk);
                        const size_t h = H.slot(c);
                        if (H.key[h] != 0 && H.val[h] != index) {
                            shared = true;
                            next_touched.push_back(index);
                            next_touched.push_back(H.val[h]);
                        }
                        R.ex_img.push_back(pi);
                        R.ex_kp.push_back(pk);
                        R.imgs.push_back(pi);
                        H.key[h] = c;
                        H.val[h] = index;
                    }
                }
            }
            // assemble: every input chain in order -- untouched ones and roots (with what joined
            // them) stay, the touched chains that joined a root are gone
            std::vector<int32_t> out_pos(nt_, -1);
            int64_t o = 0, n_new = 0;
            size_t tk = 0;
            int32_t *d_img = node_img.data(), *d_kp = node_kp.data();
            for (int64_t i = 0; i < n_cur; ++i) {
                const int64_t b = c_ptr[(size_t)i], e = c_ptr[(size_t)i + 1];
                const bool is_t = tk < nt_ && touched[tk] == (int32_t)i;
                if (is_t && !is_root[tk]) { ++tk; continue; }
                cursor[(size_t)n_new] = o;
                std::memcpy(d_img + o, c_img.data() + b, (size_t)(e - b) * sizeof(int32_t));
                std::memcpy(d_kp + o, c_kp.data() + b, (size_t)(e - b) * sizeof(int32_t));
                o += e - b;
                if (is_t) {
                    const Root &R = roots[tk];
                    std::memcpy(d_img + o, R.ex_img.data(), R.ex_img.size() * sizeof(int32_t));
                    std::memcpy(d_kp + o, R.ex_kp.data(), R.ex_kp.size() * sizeof(int32_t));
                    o += (int64_t)R.ex_img.size();
                    out_pos[tk] = (int32_t)n_new;
                    ++tk;
                }
                ++n_new;
            }
            cursor[(size_t)n_new] = o;
            std::memcpy(c_img.data(), d_img, (size_t)o * sizeof(int32_t));
            std::memcpy(c_kp.data(), d_kp, (size_t)o * sizeof(int32_t));
            std::memcpy(c_ptr.data(), cursor.data(), (size_t)(n_new + 1) * sizeof(int64_t));
            touched.clear();
            for (int32_t r : next_touched) touched.push_back(out_pos[(size_t)r]);
            if (timing)
                fprintf(stderr, "iamx_link_matches pass %d: %lld -> %lld chains, incremental over %zu chains, %.3f s\n",
                        passes, (long long)n_cur, (long long)n_new, nt_, now() - t_pass);
            const bool done_i = n_new == n_cur;
            n_cur = n_new;
            if (done_i) break;
            if (!shared) {
                ++passes;                                   // the pass that would change nothing
                break;
            }
            continue;
        }
        touched.clear();
        touched_complete = true;
        if (dense) std::memset(table.data(), 0xff, table_n * sizeof(int32_t));      // every entry -1
        else map.clear();
        n_chains = 0;
        int32_t n_nodes = 0;
        auto append = [&](int32_t chain, int32_t pi, int32_t pk) {
            const int32_t nd = n_nodes++;
            node_img[nd] = pi; node_kp[nd] = pk; node_chain[nd] = chain;
            ChainImgs &ci = cimg[(size_t)chain];
            if (ci.n < INL) {
                ci.img[ci.n] = pi;
            } else {                                        // overflow list, newest first
                ovf_next[nd] = ci.n == INL ? -1 : ovf_head[(size_t)chain];
                ovf_head[(size_t)chain] = nd;
            }
            ++ci.n;
        };
        auto in_chain = [&](int32_t chain, int32_t pi) -> bool {
            const ChainImgs &ci = cimg[(size_t)chain];
            const int32_t k = ci.n < INL ? ci.n : INL;
            for (int32_t t = 0; t < k; ++t)
                if (ci.img[t] == pi) return true;
            if (ci.n > INL)                                 // a long chain: the rest through its nodes
                for (int32_t nd = ovf_head[(size_t)chain]; nd >= 0; nd = ovf_next[nd])
                    if (node_img[nd] == pi) return true;
            return false;
        };
        auto lookup = [&](int32_t pi, int32_t pk) -> int32_t {
            if (dense) return table[(size_t)(base[(size_t)pi] + pk)];
            const size_t h = map.slot(PointMap::code(pi, pk));
            return map.key[h] != 0 ? map.val[h] : -1;
        };
        auto store = [&](int32_t pi, int32_t pk, int32_t chain) {
            if (dense) { table[(size_t)(base[(size_t)pi] + pk)] = chain; return; }
            const uint64_t c = PointMap::code(pi, pk);
            const size_t h = map.slot(c);
            map.key[h] = c;
            map.val[h] = chain;
        };
        const int64_t n_pts_cur = c_ptr[(size_t)n_cur];
        for (int64_t m = 0; m < n_cur; ++m) {
            const int64_t b = c_ptr[m], e = c_ptr[m + 1];
            if (dense) {
                for (int64_t j = b + AHEAD; j < e + AHEAD && j < n_pts_cur; ++j)
                    __builtin_prefetch(&table[(size_t)(base[(size_t)c_img[j]] + c_kp[j])], 1, 1);
                // ... and, half way, the chain a point will most likely join (a hint: the entry
                // may still change before its turn)
                for (int64_t j = b + AHEAD / 2; j < e + AHEAD / 2 && j < n_pts_cur; ++j) {
                    const int32_t guess = table[(size_t)(base[(size_t)c_img[j]] + c_kp[j])];
                    if (guess >= 0) __builtin_prefetch(&cimg[(size_t)guess], 1, 1);
                }
            }
            int32_t index = -1;
            for (int64_t j = b; j < e; ++j) {               // first point seen before decides
                index = lookup(c_img[j], c_kp[j]);
                if (index >= 0) break;
            }
            if (index < 0) {                                // new chain: register every point
                index = (int32_t)n_chains++;
                cimg[(size_t)index].n = 0;
                for (int64_t j = b; j < e; ++j) {
                    store(c_img[j], c_kp[j], index);
                    append(index, c_img[j], c_kp[j]);
                }
            } else {                                        // join: only images new to the chain
                for (int64_t j = b; j < e; ++j) {
                    if (!in_chain(index, c_img[j])) {
                        const int32_t owner = lookup(c_img[j], c_kp[j]);
                        if (owner >= 0 && owner != index) {
                            shared = true;
                            if (touched_complete) {
                                if ((int64_t)touched.size() * 4 < n_cur + 1024) {
                                    touched.push_back(index);
                                    touched.push_back(owner);
                                } else {
                                    touched_complete = false;   // (too many to be worth it)
                                    touched.clear();
                                }
                            }
                        }
                        append(index, c_img[j], c_kp[j]);
                        store(c_img[j], c_kp[j], index);
                    }
                }
            }
        }
        // next pass input = the chains in creation order, points in insertion order: a stable
        // counting sort of the nodes by chain (the counts are the chains' lengths)
        const double t_walk = now();
        const int64_t n_new = n_chains;
        int64_t o = 0;
        for (int64_t i = 0; i < n_new; ++i) {
            c_ptr[(size_t)i] = o;
            cursor[(size_t)i] = o;
            o += cimg[(size_t)i].n;
        }
        c_ptr[(size_t)n_new] = o;
        constexpr int32_t SCATTER_AHEAD = 24;
        for (int32_t nd = 0; nd < n_nodes; ++nd) {
            if (nd + SCATTER_AHEAD < n_nodes) __builtin_prefetch(&cursor[(size_t)node_chain[nd + SCATTER_AHEAD]], 1, 1);
            const int64_t dst = cursor[(size_t)node_chain[nd]]++;
            c_img[(size_t)dst] = node_img[nd];
            c_kp[(size_t)dst] = node_kp[nd];
        }
        if (timing)
            fprintf(stderr, "iamx_link_matches pass %d: %lld -> %lld chains, %d nodes, walk %.3f s, regroup %.3f s\n",
                    passes, (long long)n_cur, (long long)n_new, n_nodes, t_walk - t_pass, now() - t_walk);
        const bool done = n_new == n_cur;
        n_cur = n_new;
        if (done) break;
        if (!shared && !verify) {
            ++passes;                                       // the pass that would change nothing
            break;
        }
    }
    const int64_t total = c_ptr[(size_t)n_cur];
    if (timing) fprintf(stderr, "iamx_link_matches total: %.3f s\n", now() - t_enter);
    std::memcpy(out_img, c_img.data(), (size_t)total * sizeof(int32_t));
    std::memcpy(out_kp, c_kp.data(), (size_t)total * sizeof(int32_t));
    std::memcpy(out_ptr, c_ptr.data(), (size_t)(n_cur + 1) * sizeof(int64_t));
    if (n_passes) *n_passes = passes;
    return n_cur;
    } catch (const std::bad_alloc &) {
        iamx::fail(IAMX_ENOMEM, "iamx_link_matches: out of memory");
        return IAMX_ENOMEM;
    }
}
}  // namespace

// in : n_matches chains, chain i = points [ptr[i], ptr[i+1]) of (img[], kp[])
// out: linked chains in the reference's order (NOT yet sorted by length), same flat layout;
//      out arrays must hold ptr[n_matches] points / n_matches + 1 offsets.
// returns the number of chains (>= 0) or a negative error code; *n_passes = passes executed.
extern "C" int64_t iamx_link_matches(const int32_t *img, const int32_t *kp, const int64_t *ptr,
                                     int64_t n_matches, int32_t *out_img, int32_t *out_kp,
                                     int64_t *out_ptr, int32_t *n_passes)
{
    if (n_matches < 0 || !ptr || !out_ptr || (n_matches > 0 && (!img || !kp || !out_img || !out_kp))) {
        iamx::fail(IAMX_EINVAL, "iamx_link_matches: null pointer or negative count");
        return IAMX_EINVAL;
    }
    for (int64_t i = 0; i < n_matches; ++i)
        if (ptr[i + 1] < ptr[i]) {
            iamx::fail(IAMX_EINVAL, "iamx_link_matches: offsets not monotone");
            return IAMX_EINVAL;
        }
    const int64_t n_pts = n_matches ? ptr[n_matches] : 0;
    if (n_matches >= (1LL << 31) || n_pts >= (1LL << 31)) {
        iamx::fail(IAMX_EINVAL, "iamx_link_matches: more than 2^31 matches / points");
        return IAMX_EINVAL;
    }
    LinkInput in;
    in.img = img; in.kp = kp; in.ptr = ptr;
    return link_impl(in, n_matches, n_pts, out_img, out_kp, out_ptr, n_passes);
}

// iamx_link_matches for the pair matches of make_match_structure() without the flat copies:
// blocks[b] = int32 [counts[b]][2] (keypoint of image ij[2b], keypoint of image ij[2b + 1]), in
// the reference's order of pairs; out arrays must hold 2 * sum(counts) points.
extern "C" int64_t iamx_link_pair_blocks(const int32_t *const *blocks, const int64_t *counts,
                                         const int32_t *ij, int64_t n_blocks, int32_t *out_img,
                                         int32_t *out_kp, int64_t *out_ptr, int32_t *n_passes)
{
    if (n_blocks < 0 || !out_ptr || (n_blocks > 0 && (!blocks || !counts || !ij || !out_img || !out_kp))) {
        iamx::fail(IAMX_EINVAL, "iamx_link_pair_blocks: null pointer or negative count");
        return IAMX_EINVAL;
    }
    int64_t n_matches = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        if (counts[b] < 0 || (counts[b] > 0 && !blocks[b])) {
            iamx::fail(IAMX_EINVAL, "iamx_link_pair_blocks: negative count or null block");
            return IAMX_EINVAL;
        }
        n_matches += counts[b];
    }
    if (2 * n_matches >= (1LL << 31)) {
        iamx::fail(IAMX_EINVAL, "iamx_link_pair_blocks: more than 2^31 points");
        return IAMX_EINVAL;
    }
    LinkInput in;
    in.blocks = blocks; in.counts = counts; in.ij = ij; in.n_blocks = n_blocks;
    return link_impl(in, n_matches, 2 * n_matches, out_img, out_kp, out_ptr, n_passes);
}

// One group level of scripts/lib/groups.py:59-118 compute() (HOST): pick the seed feature (the
// unused chain with the most unplaced images, > 2, none of its images placed before), then
// sweep all chains repeatedly, adding every unused chain that is connected to the growing group
// by the reference's counting rules, until a sweep adds nothing.
//   img / ptr        the chains (image index of every point)
//   level            [n_matches] in/out: -1 = unused, else the group level it was added at
//   placed_images    [n_images] 0/1: images placed by EARLIER levels
//   placed_matches   [n_images] out: features placed per image at this level
// returns the seed chain index, or -1 when no seed exists (placed_matches is zeroed either way).
extern "C" int64_t iamx_group_level(const int32_t *img, const int64_t *ptr, int64_t n_matches,
                                    int n_images, int32_t *level, const uint8_t *placed_images,
                                    int group_level, int use_single_pairs, int max_wanted,
                                    int min_connections, int32_t *placed_matches)
{
    if (n_matches < 0 || n_images <= 0 || !ptr || !level || !placed_images || !placed_matches ||
        (n_matches > 0 && !img)) {
        iamx::fail(IAMX_EINVAL, "iamx_group_level: null pointer or bad count");
        return IAMX_EINVAL;
    }
    std::memset(placed_matches, 0, (size_t)n_images * sizeof(int32_t));
    int max_connections = 2;
    int64_t seed = -1;
    for (int64_t i = 0; i < n_matches; ++i) {
        if (level[i] >= 0) continue;
        int count = 0;
        bool connected = false;
        for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) {
            if (placed_images[img[j]]) connected = true; else ++count;
        }
        if (!connected && count > max_connections) { max_connections = count; seed = i; }
    }
    if (seed < 0) return -1;
    auto add = [&](int64_t i) {
        for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) ++placed_matches[img[j]];
        level[i] = group_level;
    };
    const int seed_image = img[ptr[seed] + 1];          // match[3]: the SECOND point of the chain
    add(seed);
    bool still_working = true;
    while (still_working) {
        still_working = false;
        for (int64_t i = 0; i < n_matches; ++i) {
            if (level[i] >= 0) continue;
            const int64_t len = ptr[i + 1] - ptr[i];
            if (!(use_single_pairs || len > 2)) continue;
            int placed_count = 0, placed_need_count = 0, unplaced_count = 0;
            bool seed_connection = false;
            for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) {
                const int im = img[j];
                if (placed_images[im]) continue;        // placed in a previous grouping
                if (im == seed_image) seed_connection = true;
                const int pm = placed_matches[im];
                if (pm >= max_wanted) ++placed_count;
                else if (pm >= min_connections) { ++placed_count; ++placed_need_count; }
                else if (pm > 0) ++placed_need_count;
                else ++unplaced_count;
            }
            if (placed_count > 1 || (use_single_pairs && placed_count > 0) || seed_connection) {
                if (placed_need_count > 0 || unplaced_count > 0) {
                    add(i);
                    still_working = true;
                }
            }
        }
    }
    return seed;
}

// The per-image index of find_matches' ledger of pairs WITHOUT matches (matchpairs.QuietLedger):
// m pairs (qi, qj) with their processing positions seq, ASCENDING in seq; every pair belongs to
// both of its images.  One counting sort by image (O(m + n_images)) writes, per image k, its
// partners and their seq in processing order at [bounds[k], bounds[k + 1]).  The numpy form (three
// gathers of 2 m entries through a radix argsort) was 1.0 s of the 2.0 s that writing the .match
// files of the 2812-image all-pairs survey took.
extern "C" int iamx_ledger_index(const int64_t *qi, const int64_t *qj, const int64_t *seq, int64_t m,
                                 int64_t n_images, int64_t *other, int64_t *seq_out, int64_t *bounds)
{
    if (m < 0 || n_images < 0 || !bounds || (m > 0 && (!qi || !qj || !seq || !other || !seq_out)))
        return iamx::fail(IAMX_EINVAL, "iamx_ledger_index: null pointer or negative count");
    for (int64_t k = 0; k <= n_images; ++k) bounds[k] = 0;
    for (int64_t e = 0; e < m; ++e) {
        if (qi[e] < 0 || qi[e] >= n_images || qj[e] < 0 || qj[e] >= n_images)
            return iamx::fail(IAMX_EINVAL, "iamx_ledger_index: image index out of range");
        ++bounds[qi[e] + 1];
        ++bounds[qj[e] + 1];
    }
    for (int64_t k = 0; k < n_images; ++k) bounds[k + 1] += bounds[k];
    std::vector<int64_t> fill(bounds, bounds + n_images);
    for (int64_t e = 0; e < m; ++e) {           // (the order of the numpy form: i's entry, then j's)
        int64_t p = fill[(size_t)qi[e]]++;
        other[p] = qj[e]; seq_out[p] = seq[e];
        p = fill[(size_t)qj[e]]++;
        other[p] = qi[e]; seq_out[p] = seq[e];
    }
    return IAMX_OK;
}

// ---- a round of find_matches: the bulk work between the device and the python lists -----------
// A round that holds pairs WITH matches hands back millions of (query, train) rows in one packed,
// page-locked buffer that the next round reuses.  What python did with them per round -- one copy,
// one column-swapped copy (the reversed direction's list), the mean / standard deviation of every
// pair's triangulated heights through three temporaries of the size of the round -- cost 70-100 ms
// in numpy, most of it first-touch page faults of fresh arrays on ONE thread, while the device
// idled (profiles/r4_fm_timeline_c2.txt: the first 24 rounds of the 2812-image survey took 102 ms
// each against 29 ms of sweep).  Here: a few threads, every byte touched once.
namespace {
template <class F>
void run_threads(int64_t n, int threads, int64_t grain, F body)          // body(lo, hi) on [0, n)
{
    int t = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    if ((int64_t)t > (n + grain - 1) / grain) t = (int)((n + grain - 1) / grain);
    if (t <= 1) { body((int64_t)0, n); return; }
    std::vector<std::thread> pool;
    const int64_t per = (n + t - 1) / t;
    for (int k = 1; k < t; ++k) {
        const int64_t lo = per * k, hi = lo + per < n ? lo + per : n;
        if (lo < hi) pool.emplace_back([=] { body(lo, hi); });
    }
    body((int64_t)0, per < n ? per : n);
    for (auto &th : pool) th.join();
}

// numpy's float64 add.reduce (loops_utils.h.src DOUBLE_pairwise_sum): straight below 8 terms,
// eight interleaved partial sums up to 128, halves (multiples of 8) above
double pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}
}  // namespace

// fwd[k] = src[k], rev[k] = (src[k][1], src[k][0]) for n rows of two int32
extern "C" int iamx_pairs_fwd_rev(const int32_t *src, int64_t n, int32_t *fwd, int32_t *rev, int threads)
{
    if (n < 0 || (n > 0 && (!src || !fwd || !rev)))
        return iamx::fail(IAMX_EINVAL, "iamx_pairs_fwd_rev: null pointer or negative count");
    run_threads(n, threads, (int64_t)1 << 16, [=](int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            const int32_t a = src[2 * k], b = src[2 * k + 1];
            fwd[2 * k] = a; fwd[2 * k + 1] = b;
            rev[2 * k] = b; rev[2 * k + 1] = a;
        }
    });
    return IAMX_OK;
}

// per segment s (z[starts[s] .. + counts[s]), counts >= 1): mean = np.add.reduceat(z, starts) / c
// and std = sqrt(np.add.reduceat((z - mean)**2, starts) / c) -- numpy's own summation order (the
// first term, then the pairwise sum of the rest), so the values are the ones the numpy form gave
extern "C" int iamx_segment_mean_std(const double *z, const int64_t *starts, const int64_t *counts,
                                     int64_t n_seg, int64_t n_z, double *mean, double *std, int threads)
{
    if (n_seg < 0 || n_z < 0 || (n_seg > 0 && (!z || !starts || !counts || !mean || !std)))
        return iamx::fail(IAMX_EINVAL, "iamx_segment_mean_std: null pointer or negative count");
    for (int64_t s = 0; s < n_seg; ++s)
        if (counts[s] < 1 || starts[s] < 0 || starts[s] + counts[s] > n_z)
            return iamx::fail(IAMX_EINVAL, "iamx_segment_mean_std: segment out of range or empty");
    run_threads(n_seg, threads, 64, [=](int64_t lo, int64_t hi) {
        std::vector<double> d2;
        for (int64_t s = lo; s < hi; ++s) {
            const double *a = z + starts[s];
            const int64_t c = counts[s];
            const double m = (a[0] + pairwise_sum(a + 1, c - 1)) / (double)c;
            d2.resize((size_t)c);
            for (int64_t i = 0; i < c; ++i) {
                const double d = a[i] - m;
                d2[(size_t)i] = d * d;
            }
            mean[s] = m;
            std[s] = sqrt((d2[0] + pairwise_sum(d2.data() + 1, c - 1)) / (double)c);
        }
    });
    return IAMX_OK;
}

// Reads one byte of every 4 KiB page of [p, p + bytes) on `threads` threads: the first touch of a
// fresh page-locked (or plain) buffer costs microseconds per page, and a round's landing buffers
// are hundreds of megabytes (find_matches pays it once per buffer set, 0.45 s each on one thread:
// matcher._prewarm_pools does it beside the schedule bookkeeping instead of inside round 0-2).
extern "C" int iamx_touch_pages(const void *p, int64_t bytes, int threads)
{
    if (bytes < 0 || (bytes > 0 && !p))
        return iamx::fail(IAMX_EINVAL, "iamx_touch_pages: null pointer or negative size");
    const volatile unsigned char *b = static_cast<const volatile unsigned char *>(p);
    const int64_t pages = (bytes + 4095) / 4096;
    run_threads(pages, threads, 1024, [=](int64_t lo, int64_t hi) {
        unsigned acc = 0;
        for (int64_t k = lo; k < hi; ++k) {
            const int64_t at = k * 4096;
            acc += b[at < bytes ? at : bytes - 1];
        }
        (void)acc;
    });
    return IAMX_OK;
}

// ---- consolidation helpers (HOST) ----------------------------------------------------------
// first[k] = the smallest j with key[j] == key[k]: what np.unique(key, return_index=True,
// return_inverse=True) delivers as first[inverse] for merge_duplicates (match_cleanup.py:19-104:
// keypoints of one image with the same "%.2f-%.2f" pixel collapse onto the first one), by one
// pass over an open-addressing table instead of a stable argsort.
extern "C" int iamx_first_occurrence(const int64_t *key, int64_t n, int64_t *first)
{
    if (n < 0 || (n > 0 && (!key || !first)))
        return iamx::fail(IAMX_EINVAL, "iamx_first_occurrence: null pointer or negative count");
    size_t cap = 16;
    while (cap < (size_t)(2 * n + 16)) cap <<= 1;
    std::vector<int64_t> slot(cap, -1);
    const size_t mask = cap - 1;
    for (int64_t k = 0; k < n; ++k) {
        uint64_t h = (uint64_t)key[k];
        h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
        size_t at = (size_t)(h & mask);
        while (slot[at] >= 0 && key[slot[at]] != key[k]) at = (at + 1) & mask;
        if (slot[at] < 0) slot[at] = k;
        first[k] = slot[at];
    }
    return IAMX_OK;
}

// key2[k] = round-half-even(100 * xy[k]) computed exactly (matcher.kp_key2: two keypoints have the
// same "%.2f-%.2f" % kp.pt string, scripts/lib/matcher.py:166-167 / match_cleanup.py:36-38, iff
// their key pairs are equal): x * 2^40 is an exact integer for every float32 in [2^-16, 2^14),
// smaller values print as 0.00 either way.  IAMX_EINVAL for a coordinate outside [0, 16384).
namespace {
inline bool key_of(float x, int32_t &key)
{
    if (!(x >= 0.f) || x >= 16384.f) return false;
    const int64_t m = (int64_t)((double)x * 1099511627776.0) * 100;
    int64_t q = m >> 40;
    const int64_t rem = m & ((1ll << 40) - 1), half = 1ll << 39;
    q += (rem > half) | ((rem == half) & ((q & 1) == 1));
    key = (int32_t)q;
    return true;
}
}  // namespace

extern "C" int iamx_kp_key2(const float *xy, int64_t n, int32_t *key2)
{
    if (n < 0 || (n > 0 && (!xy || !key2))) return iamx::fail(IAMX_EINVAL, "iamx_kp_key2: null pointer or negative count");
    for (int64_t k = 0; k < 2 * n; ++k)
        if (!key_of(xy[k], key2[k])) return iamx::fail(IAMX_EINVAL, "iamx_kp_key2: keypoint coordinates outside [0, 16384)");
    return IAMX_OK;
}

// merge_duplicates' per-image index (scripts/lib/match_cleanup.py:19-60): remap[k] = the first USED
// keypoint (lowest index) of the image whose pixel has the same "%.2f-%.2f" key as keypoint k, k
// itself for an unused keypoint.  Images image_lo .. image_hi - 1 of the flat arrays (keypoints of
// image i at [kp_base[i], kp_base[i + 1]): xy float32 [..][2], used uint8, remap int32 holding the
// index INSIDE the image); identity[i] = 1 when image i maps every keypoint onto itself.
extern "C" int iamx_kp_dup_remap(const float *xy, const uint8_t *used, const int64_t *kp_base,
                                 int32_t n_images, int32_t *remap, uint8_t *identity, int threads)
{
    if (n_images < 0 || !kp_base || (n_images > 0 && (!xy || !used || !remap || !identity)))
        return iamx::fail(IAMX_EINVAL, "iamx_kp_dup_remap: null pointer or negative count");
    std::atomic<int32_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        std::vector<uint64_t> keys;
        std::vector<int32_t> slot;
        for (int32_t i = next.fetch_add(1); i < n_images; i = next.fetch_add(1)) {
            const int64_t b = kp_base[i], n = kp_base[i + 1] - b;
            size_t cap = 16;
            while (cap < (size_t)(2 * n + 16)) cap <<= 1;
            keys.assign(cap, 0);
            slot.assign(cap, -1);
            const size_t mask = cap - 1;
            bool same = true;
            for (int64_t k = 0; k < n; ++k) {
                int32_t r = (int32_t)k;
                if (used[b + k]) {
                    int32_t kx, ky;
                    if (!key_of(xy[2 * (b + k)], kx) || !key_of(xy[2 * (b + k) + 1], ky)) { bad.store(1); break; }
                    const uint64_t code = (((uint64_t)(uint32_t)kx << 32) | (uint32_t)ky) + 1;
                    uint64_t h = code;
                    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
                    size_t at = (size_t)(h & mask);
                    while (keys[at] != 0 && keys[at] != code) at = (at + 1) & mask;
                    if (keys[at] == 0) { keys[at] = code; slot[at] = (int32_t)k; }
                    r = slot[at];
                    same = same && r == (int32_t)k;
                }
                remap[b + k] = r;
            }
            identity[i] = same ? 1 : 0;
        }
    };
    const int nt = std::max(1, std::min(std::max(threads, 1), (int)n_images));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (bad.load()) return iamx::fail(IAMX_EINVAL, "iamx_kp_dup_remap: keypoint coordinates outside [0, 16384)");
    return IAMX_OK;
}

// One pass over the images' match lists for the three per-list loops in front of link_matches
// (scripts/lib/match_cleanup.py:19-188 + lib/project.py:331-350): list b = int32 [cnt[b]][2],
// column 0 keypoints of image ia[b], column 1 of image ib[b]; keypoints of image i live at
// [kp_base[i], kp_base[i + 1]) of the flat per-keypoint arrays.
//   mode & 1  compute_kp_usage:   used[kp_base[ia] + a] = used[kp_base[ib] + b] = 1
//   mode & 2  merge_duplicates:   a <- remap[kp_base[ia] + a], b <- remap[kp_base[ib] + b], IN PLACE
//   mode & 4  check_for_pair_dups / check_for_1vn_dups: dup_pairs[b] = rows that repeat an earlier
//             row of the list, dup_first[b] = rows whose column 0 repeats an earlier row's
// A python loop over the lists did each of these with two or three numpy calls per list: 64 k lists
// of ~2 k rows on a 2048-frame survey, 6 s of the consolidation stage in np.sort / np.unique.
// Lists are independent; `threads` of them are scanned at a time.
extern "C" int iamx_match_lists_scan(int32_t *const *lists, const int64_t *cnt, const int32_t *ia,
                                     const int32_t *ib, int64_t n_lists, const int64_t *kp_base,
                                     int32_t n_images, uint8_t *used, const int32_t *remap, int mode,
                                     int32_t *dup_pairs, int32_t *dup_first, int threads)
{
    if (n_lists < 0 || !kp_base || n_images < 0 ||
        (n_lists > 0 && (!lists || !cnt || !ia || !ib)) || ((mode & 1) && !used) ||
        ((mode & 2) && !remap) || ((mode & 4) && (!dup_pairs || !dup_first)))
        return iamx::fail(IAMX_EINVAL, "iamx_match_lists_scan: null pointer or negative count");
    for (int64_t b = 0; b < n_lists; ++b)
        if (cnt[b] < 0 || (cnt[b] > 0 && !lists[b]) || ia[b] < 0 || ia[b] >= n_images || ib[b] < 0 ||
            ib[b] >= n_images)
            return iamx::fail(IAMX_EINVAL, "iamx_match_lists_scan: bad list %lld", (long long)b);
    std::atomic<int64_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        std::vector<uint64_t> slot;                       // open addressing, 0 = empty (codes are + 1)
        std::vector<uint32_t> slot1;
        for (int64_t b = next.fetch_add(1); b < n_lists; b = next.fetch_add(1)) {
            int32_t *p = lists[b];
            const int64_t n = cnt[b];
            const int64_t ba = kp_base[ia[b]], bb = kp_base[ib[b]];
            const int64_t na = kp_base[ia[b] + 1] - ba, nb = kp_base[ib[b] + 1] - bb;
            if (mode & 3) {
                // (the per-keypoint tables of a survey are hundreds of MB and a list's rows point all
                //  over two images' parts of them: the lines of the rows a few steps ahead are asked
                //  for now -- an index outside its image is only prefetched, never dereferenced)
                constexpr int64_t PF = 12;
                for (int64_t k = 0; k < n; ++k) {
                    if (k + PF < n) {
                        const int64_t pa = ba + p[2 * (k + PF)], pc = bb + p[2 * (k + PF) + 1];
                        if (mode & 1) { __builtin_prefetch(used + pa, 1, 1); __builtin_prefetch(used + pc, 1, 1); }
                        if (mode & 2) { __builtin_prefetch(remap + pa, 0, 1); __builtin_prefetch(remap + pc, 0, 1); }
                    }
                    const int64_t a = p[2 * k], c = p[2 * k + 1];
                    if (a < 0 || a >= na || c < 0 || c >= nb) { bad.store(1); break; }
                    if (mode & 1) { used[ba + a] = 1; used[bb + c] = 1; }
                    if (mode & 2) { p[2 * k] = remap[ba + a]; p[2 * k + 1] = remap[bb + c]; }
                }
            }
            if (mode & 4) {
                size_t cap = 16;
                while (cap < (size_t)(2 * n + 16)) cap <<= 1;
                slot.assign(cap, 0);
                slot1.assign(cap, 0);
                const size_t mask = cap - 1;
                int32_t dp = 0, d1 = 0;
                for (int64_t k = 0; k < n; ++k) {
                    const uint64_t code = (((uint64_t)(uint32_t)p[2 * k] << 32) | (uint32_t)p[2 * k + 1]) + 1;
                    uint64_t h = code;
                    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33;
                    size_t at = (size_t)(h & mask);
                    while (slot[at] != 0 && slot[at] != code) at = (at + 1) & mask;
                    if (slot[at] == code) ++dp; else slot[at] = code;
                    const uint32_t c1 = (uint32_t)p[2 * k] + 1u;
                    uint64_t g = c1;
                    g *= 0x9E3779B97F4A7C15ULL; g ^= g >> 29;
                    size_t a1 = (size_t)(g & mask);
                    while (slot1[a1] != 0 && slot1[a1] != c1) a1 = (a1 + 1) & mask;
                    if (slot1[a1] == c1) ++d1; else slot1[a1] = c1;
                }
                dup_pairs[b] = dp;
                dup_first[b] = d1;
            }
        }
    };
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(threads, 1), n_lists));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (bad.load()) return iamx::fail(IAMX_EINVAL, "iamx_match_lists_scan: keypoint index out of range");
    return IAMX_OK;
}

// uv[k] = (double) kp.pt of member k of the chains: keypoint f_kp[k] of image f_img[k], read from
// the images' own keypoint arrays (xy[i] = float32 [n_kp[i]][2]) -- the "replace keypoint indices
// with uv coordinates" step of scripts/lib/match_cleanup.py:277-287.  A negative index counts from
// the end of the image's list like python's kp_list[m[1]]; an index outside the list is
// IAMX_EINVAL with *bad_member = the first offending member (the reference raises IndexError).
// The numpy form (a concatenation of every image's positions, five index passes and a gather over
// all members) was a third of link_matches' time on a survey of thousands of frames.
extern "C" int iamx_chain_members_uv(const int32_t *f_img, const int32_t *f_kp, int64_t n_members,
                                     const float *const *xy, const int64_t *n_kp, int32_t n_images,
                                     double *uv, int64_t *bad_member, int threads)
{
    if (n_members < 0 || n_images < 0 || (n_members > 0 && (!f_img || !f_kp || !xy || !n_kp || !uv)))
        return iamx::fail(IAMX_EINVAL, "iamx_chain_members_uv: null pointer or negative count");
    std::atomic<int64_t> first_bad{-1};
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(threads, 1), n_members >> 16));
    auto work = [&](int t) {
        const int64_t lo = n_members * t / nt, hi = n_members * (t + 1) / nt;
        constexpr int64_t AHEAD = 16;
        for (int64_t k = lo; k < hi; ++k) {
            if (k + AHEAD < hi) {
                const int32_t ia = f_img[k + AHEAD];
                if (ia >= 0 && ia < n_images && xy[ia]) {
                    int64_t ka = f_kp[k + AHEAD];
                    if (ka < 0) ka += n_kp[ia];
                    if (ka >= 0 && ka < n_kp[ia]) __builtin_prefetch(xy[ia] + 2 * ka, 0, 1);
                }
            }
            const int32_t i = f_img[k];
            int64_t kp = f_kp[k];
            if (i < 0 || i >= n_images) { kp = -1; } else { if (kp < 0) kp += n_kp[i]; if (kp >= n_kp[i]) kp = -1; }
            if (kp < 0) {
                int64_t cur = first_bad.load();
                while ((cur < 0 || k < cur) && !first_bad.compare_exchange_weak(cur, k)) {}
                uv[2 * k] = uv[2 * k + 1] = 0.0;
                continue;
            }
            uv[2 * k] = (double)xy[i][2 * kp];
            uv[2 * k + 1] = (double)xy[i][2 * kp + 1];
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread &t : pool) t.join();
    if (bad_member) *bad_member = first_bad.load();
    if (first_bad.load() >= 0) return iamx::fail(IAMX_EINVAL, "iamx_chain_members_uv: keypoint index out of range");
    return IAMX_OK;
}

// The chains of iamx_link_matches, longest first, chains of equal length in their given order
// (list.sort(key=len, reverse=True) of match_cleanup.py:291-292 is stable): a counting sort of
// the chains by length, then the members copied chain by chain on `threads` threads.  out_ptr
// [n_chains + 1]; out_img / out_kp sized like the input.
extern "C" int iamx_chains_longest_first(const int32_t *img, const int32_t *kp, const int64_t *ptr,
                                         int64_t n_chains, int32_t *out_img, int32_t *out_kp,
                                         int64_t *out_ptr, int threads)
{
    if (n_chains < 0 || !ptr || !out_ptr || (n_chains > 0 && (!img || !kp || !out_img || !out_kp)))
        return iamx::fail(IAMX_EINVAL, "iamx_chains_longest_first: null pointer or negative count");
    int64_t longest = 0;
    for (int64_t c = 0; c < n_chains; ++c) {
        const int64_t len = ptr[c + 1] - ptr[c];
        if (len < 0) return iamx::fail(IAMX_EINVAL, "iamx_chains_longest_first: offsets not monotone");
        if (len > longest) longest = len;
    }
    // chains with length L start at rank[L] in the output order (longer lengths first)
    std::vector<int64_t> rank((size_t)longest + 2, 0);
    for (int64_t c = 0; c < n_chains; ++c) ++rank[(size_t)(ptr[c + 1] - ptr[c])];
    int64_t at = 0;
    for (int64_t len = longest; len >= 0; --len) {
        const int64_t cnt = rank[(size_t)len];
        rank[(size_t)len] = at;
        at += cnt;
    }
    std::vector<int64_t> order((size_t)n_chains);
    for (int64_t c = 0; c < n_chains; ++c) order[(size_t)rank[(size_t)(ptr[c + 1] - ptr[c])]++] = c;
    out_ptr[0] = 0;
    for (int64_t r = 0; r < n_chains; ++r) {
        const int64_t c = order[(size_t)r];
        out_ptr[r + 1] = out_ptr[r] + (ptr[c + 1] - ptr[c]);
    }
    const int64_t *ord = order.data();
    run_threads(n_chains, threads, (int64_t)1 << 14, [=](int64_t lo, int64_t hi) {
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t c = ord[r], len = ptr[c + 1] - ptr[c], src = ptr[c], dst = out_ptr[r];
            for (int64_t t = 0; t < len; ++t) {
                out_img[dst + t] = img[src + t];
                out_kp[dst + t] = kp[src + t];
            }
        }
    });
    return IAMX_OK;
}


// ---------------------------------------------------------------------------------------------
// The yaw-error feedback of the reference's pair loop as a prefix computation over the schedule
// (HOST code) -- scripts/lib/matcher.py:987-993: after every pair
//     yaw1 = smart.update_yaw_error_estimate(i1, i2); i1.set_aircraft_yaw_error_estimate(yaw1)
//     yaw2 = smart.update_yaw_error_estimate(i2, i1); i2.set_aircraft_yaw_error_estimate(yaw2)
// where update_yaw_error_estimate (scripts/lib/smart.py:251-283) returns 0 for a pair without
// matches / without a similarity fit, else writes the pair's entry (values rounded through
// "%.1f", the weight read back with getInt) under /smart/<image>/yaw_pairs and returns the
// weighted average over the image's entries in the tree's child order (sorted by partner name)
// that are at least 0.5 m apart and at most 30 degrees off.  The NEXT pair an image takes part in
// triangulates with the pose that estimate gives (lib/image.py:434-457), so find_matches needs,
// for every pair with matches, the estimate of both images as they stand when the loop reaches
// the pair.  State per image: entries sorted by partner rank, current value, touched flag.
// smart.PoseFeedback drives this; the python form of the same replay stays beside it for partner
// names outside the project.
// ---------------------------------------------------------------------------------------------
namespace {

struct YawEntry {
    int32_t partner;     // rank of the partner's name among the project's image names
    double err;          // yaw error, rounded to 0.1
    double w;            // weight: trunc of the value rounded to 0.1 (what getInt returns)
    double dist;         // pair distance, rounded to 0.1
};

struct YawFeedback {
    int n_images;
    std::vector<int32_t> rank;                 // image index -> rank of its name
    std::vector<std::vector<YawEntry>> entries;
    std::vector<double> value;
    std::vector<uint8_t> touched;
};

// float("%.1f" % x): both sides print the correctly rounded decimal and read it back exactly
inline double round_tenth(double x)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%.1f", x);
    return strtod(buf, nullptr);
}

#pragma clang fp contract(off)
double yaw_average(const std::vector<YawEntry> &es)
{
    // total / count as python forms them: total a float sum of separately rounded products,
    // count an exact integer sum, one division
    double total = 0.0;
    __int128 count = 0;
    for (const YawEntry &e : es) {
        if (e.dist >= 0.5 && std::fabs(e.err) <= 30.0) {
            total = total + e.err * e.w;
            count += (__int128)e.w;
        }
    }
    if (count > 0) return total / (double)count;
    return 0.0;
}

int yaw_record(YawFeedback *fb, int x, int y, const double *v)
{
    // v = (yaw_error, dist, relative course, weight)
    const double w = std::trunc(round_tenth(v[3]));
    if (!(std::fabs(w) < 1e37)) return -1;     // python: int(inf) / int(nan) raises
    YawEntry e{fb->rank[(size_t)y], round_tenth(v[0]), w, round_tenth(v[1])};
    std::vector<YawEntry> &es = fb->entries[(size_t)x];
    auto at = std::lower_bound(es.begin(), es.end(), e.partner,
                               [](const YawEntry &a, int32_t p) { return a.partner < p; });
    if (at != es.end() && at->partner == e.partner) *at = e; else es.insert(at, e);
    fb->value[(size_t)x] = yaw_average(es);
    return 0;
}

}  // namespace

extern "C" void *iamx_yaw_feedback_new(int n_images, const int32_t *name_rank)
{
    if (n_images <= 0 || !name_rank) return nullptr;
    YawFeedback *fb = new (std::nothrow) YawFeedback();
    if (!fb) return nullptr;
    fb->n_images = n_images;
    fb->rank.assign(name_rank, name_rank + n_images);
    fb->entries.resize((size_t)n_images);
    fb->value.assign((size_t)n_images, 0.0);
    fb->touched.assign((size_t)n_images, 0);
    return fb;
}

extern "C" void iamx_yaw_feedback_free(void *h) { delete static_cast<YawFeedback *>(h); }

// entries an image's yaw_pairs node holds before the call (an earlier find_matches run)
extern "C" int iamx_yaw_feedback_seed(void *h, int image, int n, const int32_t *partner_image,
                                      const double *err, const double *weight, const double *dist)
{
    YawFeedback *fb = static_cast<YawFeedback *>(h);
    if (!fb || image < 0 || image >= fb->n_images || n < 0 || (n && (!partner_image || !err || !weight || !dist)))
        return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_seed: bad argument");
    std::vector<YawEntry> &es = fb->entries[(size_t)image];
    es.clear();
    for (int k = 0; k < n; ++k) {
        if (partner_image[k] < 0 || partner_image[k] >= fb->n_images)
            return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_seed: partner index out of range");
        es.push_back(YawEntry{fb->rank[(size_t)partner_image[k]], err[k], weight[k], dist[k]});
    }
    std::sort(es.begin(), es.end(), [](const YawEntry &a, const YawEntry &b) { return a.partner < b.partner; });
    return IAMX_OK;
}

// One round, pairs in schedule order: pi / pj [n] image indices, quiet [n] 1 = no matches; the
// h pairs with matches are the rows hit_rows [h] (ascending) with yv_f / yv_r [h][4] (yaw_error,
// dist, relative course, weight per direction) and ok [h][2] (1 = that direction has a fit).
// Out per pair with matches: e1 / e2 = the estimate of pair.i1 / pair.i2 BEFORE the pair,
// fresh1 / fresh2 = 1 while that image has not been part of any pair of the call yet (its stored
// camera pose stands).
extern "C" int iamx_yaw_feedback_feed(void *h_, int64_t n, const int32_t *pi, const int32_t *pj,
                                      const uint8_t *quiet, int64_t h, const int64_t *hit_rows,
                                      const double *yv_f, const double *yv_r, const uint8_t *ok,
                                      double *e1, double *e2, uint8_t *fresh1, uint8_t *fresh2)
{
    YawFeedback *fb = static_cast<YawFeedback *>(h_);
    if (!fb || n < 0 || h < 0 || (n && (!pi || !pj || !quiet)) ||
        (h && (!hit_rows || !yv_f || !yv_r || !ok || !e1 || !e2 || !fresh1 || !fresh2)))
        return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: bad argument");
    int64_t t = 0;
    for (int64_t k = 0; k < n; ++k) {
        const int x = pi[k], y = pj[k];
        if (x < 0 || x >= fb->n_images || y < 0 || y >= fb->n_images)
            return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: image index out of range");
        if (t < h && hit_rows[t] == k) {
            if (quiet[k]) return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: a quiet pair among the hits");
            e1[t] = fb->value[(size_t)x];
            e2[t] = fb->value[(size_t)y];
            fresh1[t] = !fb->touched[(size_t)x];
            fresh2[t] = !fb->touched[(size_t)y];
            // image 1, then image 2 (matcher.py:990-993)
            if (ok[2 * t]) {
                if (yaw_record(fb, x, y, yv_f + 4 * t))
                    return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: cannot convert float infinity to integer");
            } else {
                fb->value[(size_t)x] = 0.0;
            }
            if (ok[2 * t + 1]) {
                if (yaw_record(fb, y, x, yv_r + 4 * t))
                    return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: cannot convert float infinity to integer");
            } else {
                fb->value[(size_t)y] = 0.0;
            }
            ++t;
        } else if (quiet[k]) {
            fb->value[(size_t)x] = 0.0;
            fb->value[(size_t)y] = 0.0;
        } else {
            return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: a pair with matches is missing from hit_rows");
        }
        fb->touched[(size_t)x] = 1;
        fb->touched[(size_t)y] = 1;
    }
    if (t != h) return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_feed: hit_rows not ascending rows of the round");
    return IAMX_OK;
}

extern "C" int iamx_yaw_feedback_state(void *h, double *value, uint8_t *touched)
{
    YawFeedback *fb = static_cast<YawFeedback *>(h);
    if (!fb || !value || !touched) return iamx::fail(IAMX_EINVAL, "iamx_yaw_feedback_state: bad argument");
    std::copy(fb->value.begin(), fb->value.end(), value);
    std::copy(fb->touched.begin(), fb->touched.end(), touched);
    return IAMX_OK;
}
