"""MI355X-path stand-in for the reference's scripts/lib/match_cleanup.py (SURVEY.md 8f ranks
1-2): the steps process.py runs between pair matching and bundle adjustment
(scripts/process.py:305-331).

    merge_duplicates(proj)          match_cleanup.py:19-104   keypoints with the same "%.2f-%.2f"
                                                               pixel collapse onto the first used one
    check_for_pair_dups(proj)       :117-146                   repeated [i, j] pairs dropped
    check_for_1vn_dups(proj)        :158-188                   report only
    make_match_structure(proj)      :190-220                   [None, -1, [i, kp_i], [j, kp_j]] for j > i
    link_matches(proj, direct)      :223-301                   chains per feature, kp -> uv, longest first
    triangulate_smart(proj, matches):303-347                   match[0] = mean ground intersection

Same function names, arguments, in-place effects and result structures (plain python lists, so
the `matches_grouped` pickle stays readable by the reference's tools).  What changes is how the
work is done: keys are exact integers instead of formatted strings (matcher.kp_key2), per-pair
work is numpy, the chain linking is one native routine over flat arrays
(csrc/host_cleanup.hip, same order-dependent rules), and the per-feature ray / ground
intersection runs on the GPU (csrc/triangulate.hip)."""
import contextlib
import gc

import os

import numpy as np

from . import _deps
from .matcher import _kp_xy, kp_key2
from .matchpairs import MatchPairs, empty_huge

CAM2BODY = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=float)     # lib/image.py:50-52


def _log(*a):
    _deps.logger().log(*a)


def _qlog(*a):
    _deps.logger().qlog(*a)


@contextlib.contextmanager
def _no_gc():
    """building millions of small lists: the cyclic collector would rescan them over and over"""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _index_by_name(proj):
    """name -> index of the FIRST image with that name (findImageByName / findIndexByName)."""
    index = {}
    for i, im in enumerate(proj.image_list):
        index.setdefault(im.name, i)
    return index


def _pairs(matches):
    return np.asarray(matches, np.int64).reshape(-1, 2)


# --------------------------------------------------------------------------------------
# duplicates -- match_cleanup.py:19-188 (+ project.py:331-350 compute_kp_usage)
# --------------------------------------------------------------------------------------
SCAN_THREADS = max(1, min(16, os.cpu_count() or 8))    # (the GPU boxes grant sixteen cores)


def _scan_lists(proj, index, mode, wanted=None, used=None, remap=None, base=None):
    """iamx_match_lists_scan over the match lists that are array backed (what find_matches leaves:
    matchpairs.MatchPairs in their array form).  -> (entries, dup_pairs, dup_first, rest): entries =
    [(i, key, j, matches)] of the lists scanned, rest = the same for the lists that are plain
    python lists (the caller's per-list numpy path).  wanted(i, j): scan this pair at all?"""
    import ctypes
    from ._lib import check, lib
    # the four scans of a consolidation walk the same lists: their tables (which lists are array
    # backed, the pointers, counts and image pairs) are made once and kept on the project while
    # no list object or backing array has been replaced
    # (three comprehensions and three tuple hashes: 48 ms for the 131 k lists of a 2048-frame survey,
    #  75 ms as an explicit loop with its arithmetic -- per scan, four scans a stage.  The KEYS are
    #  part of it: a renamed / re-keyed entry that keeps its list object must not find the old
    #  (key, partner) table; so are the backing arrays: an edited MatchPairs leaves its array form)
    lists = [m for i1 in proj.image_list for m in i1.match_list.values()]
    sig_n = len(lists)
    sig_h = (hash(tuple(map(id, lists))),
             hash(tuple([id(getattr(m, '_a', None)) for m in lists])),
             hash(tuple([k for i1 in proj.image_list for k in i1.match_list])))
    del lists
    cached = getattr(proj, '_iamx_scan', None) if wanted is None else None
    if cached is not None and cached[0] == (sig_n, sig_h, len(proj.image_list)):
        entries, rest, arrays, tables = cached[1:]
    else:
        entries, rest, arrays, tables = [], [], [], None
        for i, i1 in enumerate(proj.image_list):
            for key, matches in i1.match_list.items():
                j = index.get(key)
                if j is None or len(matches) == 0:
                    continue
                if wanted is not None and not wanted(i, j):
                    continue
                a = matches._a if isinstance(matches, MatchPairs) else None
                if a is not None and a.dtype == np.int32 and a.flags.c_contiguous and a.flags.writeable:
                    entries.append((i, key, j, matches))
                    arrays.append(a)
                else:
                    rest.append((i, key, j, matches))
    n = len(entries)
    if mode == 4 and wanted is None:
        # check_for_pair_dups and check_for_1vn_dups ask for the same scan back to back (0.4 s each on
        # a 2048-frame survey): the second takes the first's counts while no list object, backing
        # array or length has changed (a list check_for_pair_dups repaired is a new object)
        kept = getattr(proj, '_iamx_scan4', None)
        if kept is not None and kept[0] == (sig_n, sig_h, len(proj.image_list)) and len(kept[1]) == n:
            return entries, kept[1], kept[2], rest
    dup_pairs, dup_first = np.zeros(n, np.int32), np.zeros(n, np.int32)
    if n:
        if base is None:
            base = _kp_base(proj)
        if tables is None:
            tables = ((ctypes.c_void_p * n)(*[a.__array_interface__['data'][0] for a in arrays]),
                      np.fromiter((len(a) for a in arrays), np.int64, n),
                      np.fromiter((e[0] for e in entries), np.int32, n),
                      np.fromiter((e[2] for e in entries), np.int32, n))
        ptrs, cnt, ia, ib = tables
        if wanted is None:
            proj._iamx_scan = ((sig_n, sig_h, len(proj.image_list)), entries, rest, arrays, tables)
        P = lambda x: None if x is None else x.ctypes.data_as(ctypes.c_void_p)
        rc = lib().iamx_match_lists_scan(ptrs, P(cnt), P(ia), P(ib), n, P(base), len(proj.image_list),
                                         P(used), P(remap), mode, P(dup_pairs), P(dup_first), SCAN_THREADS)
        if rc != 0:
            msg = (lib().iamx_last_error() or b'?').decode()
            if 'out of range' in msg:
                raise IndexError("index out of range in a match list (%s)" % msg)
            check(rc, 'iamx_match_lists_scan')
        if mode & 2:
            for _i, _k, _j, m in entries:
                m._pk = None                      # (the pickled form was made from the old pairs)
            proj._iamx_scan4 = None               # (the pairs changed in place)
    if mode == 4 and wanted is None:
        proj._iamx_scan4 = ((sig_n, sig_h, len(proj.image_list)), dup_pairs, dup_first)
    return entries, dup_pairs, dup_first, rest


def _kp_base(proj):
    base = np.zeros(len(proj.image_list) + 1, np.int64)
    np.cumsum([len(im.kp_list) for im in proj.image_list], out=base[1:])
    return base


def compute_kp_usage(proj):
    index = _index_by_name(proj)
    base = _kp_base(proj)
    used = np.zeros(int(base[-1]), np.uint8)
    _e, _dp, _d1, rest = _scan_lists(proj, index, 1, used=used, base=base)
    for i, im in enumerate(proj.image_list):
        im.kp_used = used[base[i]:base[i + 1]].view(np.bool_)
    for i, _key, j, matches in rest:
        p = _pairs(matches)
        proj.image_list[i].kp_used[p[:, 0]] = True
        proj.image_list[j].kp_used[p[:, 1]] = True


def _first_occurrence(key):
    """index of the first element equal to key[k], for every k (one hashing pass in libiamx; the
    numpy form -- a stable argsort inside np.unique -- without the library)"""
    import ctypes
    from . import _lib
    try:
        first = np.empty(len(key), np.int64)
        _lib.check(_lib.lib().iamx_first_occurrence(key.ctypes.data_as(ctypes.c_void_p), len(key),
                                                    first.ctypes.data_as(ctypes.c_void_p)),
                   'iamx_first_occurrence')
        return first
    except (OSError, _lib.IamxError):
        _uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        return first[inv]


def merge_duplicates(proj):
    compute_kp_usage(proj)
    _log("Indexing features by unique uv coordinates:")
    import ctypes
    from ._lib import check, lib
    base = _kp_base(proj)
    n_img = len(proj.image_list)
    # every image's table in ONE native call (iamx_kp_dup_remap: exact "%.2f" keys + first
    # occurrence among the used keypoints, a few images at a time on threads); the numpy form --
    # kp_key2 + a hashing pass per image -- was 4 ms per 37 k-keypoint frame, 17 s of a 4186-frame
    # survey's consolidation stage
    xy = [np.ascontiguousarray(_kp_xy(im), np.float32).reshape(-1, 2) for im in proj.image_list]
    xy_all = np.concatenate(xy) if xy else np.zeros((0, 2), np.float32)
    used = np.concatenate([np.asarray(im.kp_used, np.uint8) for im in proj.image_list]) if n_img \
        else np.zeros(0, np.uint8)
    flat = np.empty(int(base[-1]), np.int32)
    identity = np.zeros(max(n_img, 1), np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib().iamx_kp_dup_remap(P(xy_all), P(used), P(base), n_img, P(flat), P(identity), SCAN_THREADS)
    if rc != 0:
        if b'outside' in (lib().iamx_last_error() or b''):
            raise ValueError("keypoint coordinates outside [0, 16384)")
        check(rc, 'iamx_kp_dup_remap')
    remaps = []
    for i, im in enumerate(proj.image_list):
        remap = flat[base[i]:base[i + 1]]
        im.kp_remap_index = remap             # (the reference keeps a {"x-y": index} dict here)
        remaps.append(remap)
    _log("Merging keypoints with duplicate uv coordinates:")
    index = _index_by_name(proj)
    # (an image without two used keypoints on one pixel maps every index onto itself)
    same = [bool(v) for v in identity[:n_img].tolist()]
    if all(same):
        return
    # the array-backed lists in one native pass (in place), the rest list by list
    _e, _dp, _d1, rest = _scan_lists(proj, index, 2, wanted=lambda i, j: not (same[i] and same[j]),
                                     remap=flat, base=base)
    for i, _key, j, matches in rest:
        p = _pairs(matches)
        matches[:] = np.stack([remaps[i][p[:, 0]], remaps[j][p[:, 1]]], 1).tolist()


def _drop_pair_dups(proj, i1, key, j, matches):
    p = _pairs(matches)
    code = (p[:, 0] << 32) | p[:, 1]
    _u, first = np.unique(code, return_index=True)
    count = len(p) - len(first)
    if count > 0:
        print('Match:', i1.name, 'vs', proj.image_list[j].name, 'matches:', len(matches),
              'dups:', count)
        kept = p[np.sort(first)]
        i1.match_list[key] = MatchPairs(kept) if isinstance(matches, MatchPairs) else kept.tolist()


def check_for_pair_dups(proj):
    _log("Checking for pair duplicates (there never should be any):")
    index = _index_by_name(proj)
    for i1 in proj.image_list:
        for key in list(i1.match_list):
            if index.get(key) is not None and len(i1.match_list[key]) == 0:
                i1.match_list[key] = []
    entries, dup_pairs, _d1, rest = _scan_lists(proj, index, 4)
    # (the usual case -- no pair twice in any list -- ends here)
    for k in np.nonzero(dup_pairs)[0].tolist():
        i, key, j, matches = entries[k]
        _drop_pair_dups(proj, proj.image_list[i], key, j, matches)
    for i, key, j, matches in rest:
        p = _pairs(matches)
        srt = np.sort((p[:, 0] << 32) | p[:, 1])
        if (srt[1:] == srt[:-1]).any():
            _drop_pair_dups(proj, proj.image_list[i], key, j, matches)


def check_for_1vn_dups(proj):
    _log("Testing for 1 vs. n keypoint duplicates (there never should be any):")
    index = _index_by_name(proj)
    entries, _dp, dup_first, rest = _scan_lists(proj, index, 4)
    for k in np.nonzero(dup_first)[0].tolist():
        i, _key, _j, matches = entries[k]
        _qlog('Match:', i, 'vs', len(matches) - 1, 'matches:', len(matches), 'dups:', int(dup_first[k]))
    for i, _key, _j, matches in rest:
        p = _pairs(matches)
        count = len(p) - len(np.unique(p[:, 0]))
        if count > 0:
            _qlog('Match:', i, 'vs', len(matches) - 1, 'matches:', len(matches), 'dups:', count)


# --------------------------------------------------------------------------------------
# unified structure + chains -- match_cleanup.py:190-301
# --------------------------------------------------------------------------------------
def make_match_structure(proj):
    _log("Constructing unified match structure:")
    index = _index_by_name(proj)
    # the pairs (i < j) with matches, in the reference's order, and their [n, 2] keypoint arrays
    ij, blocks = [], []
    for i, img in enumerate(proj.image_list):
        for key, matches in img.match_list.items():
            j = index.get(key)
            if j is None or j <= i or len(matches) == 0:
                continue
            ij.append((i, j))
            blocks.append(np.asarray(matches, np.int32).reshape(-1, 2))
    counts = np.fromiter((len(b) for b in blocks), np.int64, len(blocks))
    n = int(counts.sum())
    blocks = [np.ascontiguousarray(b, np.int32) for b in blocks]
    ij_arr = np.asarray(ij, np.int32).reshape(-1, 2)
    # the reference's [[None, -1, [i, a], [j, b]], ...] (match_cleanup.py:190-215), as a list
    # that is only built when somebody other than link_matches() looks into it -- and so are the
    # flat (image, keypoint) arrays behind it: link_matches() hands the blocks themselves to
    # libiamx (round 4 concatenated them, repeated (i, j) per match and built the offsets
    # 0, 2, 4, ...: three fresh arrays of 8 bytes per match each)
    matches_direct = DirectMatches.from_blocks(ij_arr, counts, blocks)
    # link_matches() normally receives this very object: the block form goes with it.  The blocks
    # ALIAS the live match lists (no copy of millions of pairs): a fingerprint of every block --
    # length, first, middle and last row -- goes along, and link_matches() falls back to the
    # flattened copy when a list was edited in between (an edit that keeps all four is not seen:
    # the match lists are not to be changed between the two calls, as in process.py:305-317)
    proj._iamx_direct = (matches_direct, n, ij_arr, counts, blocks, _blocks_fingerprint(blocks))
    if n:
        _log("Total feature pairs in image set:", n)
        _log("Keypoint average instances = %.1f (should be 2.0 here)" % 2.0)
    return matches_direct


def _blocks_fingerprint(blocks):
    fp = np.empty((len(blocks), 7), np.int64)
    for k, b in enumerate(blocks):
        m = len(b)
        fp[k] = (m, b[0, 0], b[0, 1], b[m // 2, 0], b[m // 2, 1], b[m - 1, 0], b[m - 1, 1]) if m else \
            (0, 0, 0, 0, 0, 0, 0)
    return fp


class DirectMatches(object):
    """make_match_structure()'s result: a sequence of [None, -1, [i, a], [j, b]] lists -- one
    python list of four objects per pair match, millions on a survey -- backed by the interleaved
    (image, image) and (keypoint, keypoint) arrays it was computed as.  The lists are created
    (all of them, once) when an element is asked for; len() and link_matches() do not need them."""

    def __init__(self, img, kp):
        self._img = np.asarray(img).reshape(-1, 2)
        self._kp = np.asarray(kp).reshape(-1, 2)
        self._rows = None
        self._blocks = None
        self._n = len(self._img)

    @classmethod
    def from_blocks(cls, ij, counts, blocks):
        """pair (ij[b, 0], ij[b, 1]) contributes the rows of blocks[b] ([n, 2] keypoint indices);
        the flat arrays are made when somebody asks for them"""
        self = object.__new__(cls)
        self._img = self._kp = self._rows = None
        self._blocks = (np.asarray(ij, np.int32).reshape(-1, 2), np.asarray(counts, np.int64), blocks)
        self._n = int(self._blocks[1].sum())
        return self

    def arrays(self):
        """(img, kp): int32 [n, 2] each"""
        if self._img is None:
            ij, counts, blocks = self._blocks
            self._kp = np.concatenate(blocks).reshape(-1, 2) if blocks else np.zeros((0, 2), np.int32)
            self._img = np.repeat(ij, counts, axis=0) if blocks else np.zeros((0, 2), np.int32)
        return self._img, self._kp

    def rows(self):
        if self._rows is None:
            img, kp = self.arrays()
            with _no_gc():
                self._rows = [[None, -1, [i, a], [j, b]] for (i, j), (a, b)
                              in zip(img.tolist(), kp.tolist())]
        return self._rows

    def __len__(self):
        return self._n if self._rows is None else len(self._rows)

    def __getitem__(self, k):
        return self.rows()[k]

    def __iter__(self):
        return iter(self.rows())

    def __eq__(self, other):
        return self.rows() == (other.rows() if isinstance(other, DirectMatches) else other)

    __hash__ = None

    def __reduce_ex__(self, protocol):
        return (list, (self.rows(),))

    def untouched(self):
        return self._rows is None


def _flatten(matches):
    n = len(matches)
    ptr = np.zeros(n + 1, np.int64)
    if n:
        np.cumsum([len(m) - 2 for m in matches], out=ptr[1:])
    flat = np.array([p for m in matches for p in m[2:]], np.int64).reshape(-1, 2)
    return (np.ascontiguousarray(flat[:, 0], np.int32), np.ascontiguousarray(flat[:, 1], np.int32),
            ptr)


class Chains(object):
    """link_matches()'s result -- the `matches_grouped` of scripts/process.py:305-405 -- as a
    sequence of `[ned | None, group, [image, [u, v]], ...]` lists backed by flat arrays: ptr int64
    [n + 1], img int32 [total], uv float64 [total, 2] (chain c owns [ptr[c], ptr[c + 1])), ned
    float64 [n, 3] + has_ned, group int32 [n].  A survey of hundreds of frames links millions of
    chains (2.2 M at 512 frames of 20 MP): as python lists they are tens of millions of objects that
    every consumer re-flattened, and at BASELINE configs[4]'s 10 k frames they do not fit.  This
    package's own consumers (triangulate_smart, groups.compute, Optimizer.setup / refit) read and
    write the arrays; `pickle.dump` writes the reference's plain list of lists WITHOUT keeping it;
    anybody who indexes or iterates gets real lists -- all of them, built once, and from then on
    they are the truth (`untouched()` is False and every consumer takes its list path), because
    the caller may edit what it was handed."""

    def __init__(self, img, uv, ptr):
        self.img = np.ascontiguousarray(img, np.int32)
        self.uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        self.ptr = np.ascontiguousarray(ptr, np.int64)
        n = len(self.ptr) - 1
        self.ned = np.zeros((n, 3), np.float64)
        self.has_ned = np.zeros(n, bool)
        self.group = np.full(n, -1, np.int32)
        self._rows = None

    @classmethod
    def from_lists(cls, rows):
        """a loaded `matches_grouped` pickle (plain lists) as the array-backed form"""
        ptr = np.zeros(len(rows) + 1, np.int64)
        if len(rows):
            np.cumsum([len(m) - 2 for m in rows], out=ptr[1:])
        flat = [p for m in rows for p in m[2:]]
        self = cls(np.fromiter((p[0] for p in flat), np.int32, len(flat)),
                   np.array([p[1] for p in flat], np.float64).reshape(-1, 2), ptr)
        for c, m in enumerate(rows):
            if m[0] is not None:
                self.ned[c] = m[0]
                self.has_ned[c] = True
            self.group[c] = m[1]
        return self

    def untouched(self):
        return self._rows is None

    def _build(self):
        with _no_gc():
            pts = [[i, p] for i, p in zip(self.img.tolist(), self.uv.tolist())]
            lo, hi = self.ptr[:-1].tolist(), self.ptr[1:].tolist()
            ned = self.ned.tolist()
            return [[ned[c] if h else None, g] + pts[a:b]
                    for c, (a, b, h, g) in enumerate(zip(lo, hi, self.has_ned.tolist(), self.group.tolist()))]

    def rows(self):
        if self._rows is None:
            self._rows = self._build()
            if len(self._rows) > 100000 and hasattr(gc, 'freeze'):
                gc.freeze()                    # (see link_matches)
        return self._rows

    def __len__(self):
        return len(self.ptr) - 1 if self._rows is None else len(self._rows)

    def __getitem__(self, k):
        return self.rows()[k]

    def __iter__(self):
        return iter(self.rows())

    def __eq__(self, other):
        return self.rows() == (other.rows() if isinstance(other, Chains) else other)

    __hash__ = None

    def __reduce_ex__(self, protocol):
        # the matches_grouped pickle: a plain list of lists (not kept when the arrays are the truth)
        return (list, (self._rows if self._rows is not None else self._build(),))

    # list methods the reference's scripts use on matches_grouped
    def __delitem__(self, k):
        del self.rows()[k]

    def __setitem__(self, k, v):
        self.rows()[k] = v

    def append(self, v):
        self.rows().append(v)

    def sort(self, *a, **k):
        self.rows().sort(*a, **k)


def link_matches(proj, matches_direct):
    """Chains of [img, kp] per feature by the reference's order-dependent rules (native code),
    keypoint indices replaced by [u, v], longest chains first (stable)."""
    from ._lib import c_void_p, lib
    _log("Linking common matches together into chains:")
    n = len(matches_direct)
    cached = getattr(proj, '_iamx_direct', None)
    passes = np.zeros(1, np.int32)
    P = lambda a: c_void_p(a.ctypes.data)
    if cached is not None and cached[0] is matches_direct and cached[1] == n \
            and matches_direct.untouched() and np.array_equal(_blocks_fingerprint(cached[4]), cached[5]):
        # untouched output of make_match_structure(): the pair blocks as they lie
        import ctypes
        ij, counts, blocks = cached[2:5]
        o_img, o_kp = empty_huge(2 * n, np.int32), empty_huge(2 * n, np.int32)
        o_ptr = np.zeros(n + 1, np.int64)
        ptrs = (ctypes.c_void_p * max(len(blocks), 1))(*[b.ctypes.data for b in blocks])
        n_chain = int(lib().iamx_link_pair_blocks(ptrs, P(counts), P(np.ascontiguousarray(ij)), len(blocks),
                                                  P(o_img), P(o_kp), P(o_ptr), P(passes)))
    else:
        img, kp, ptr = _flatten(matches_direct)
        o_img, o_kp = np.empty_like(img), np.empty_like(kp)
        o_ptr = np.zeros(n + 1, np.int64)
        n_chain = int(lib().iamx_link_matches(P(img), P(kp), P(ptr), n, P(o_img), P(o_kp), P(o_ptr),
                                              P(passes)))
    # (the consolidation's scan tables pin every backing array they were made from, and the lists
    #  check_for_pair_dups replaced: the stage ends here)
    for attr in ('_iamx_scan', '_iamx_scan4'):
        try:
            delattr(proj, attr)
        except AttributeError:
            pass
    if n_chain < 0:
        raise RuntimeError("iamx_link_matches failed (%d): %s"
                           % (n_chain, (lib().iamx_last_error() or b'?').decode()))
    _log("Iterations: %d (%d -> %d)" % (int(passes[0]), n, n_chain))
    _log('Replacing keypoint indices with uv coordinates:')
    o_ptr = o_ptr[:n_chain + 1]
    total = int(o_ptr[-1])
    _log("Sorting matches by longest chain first.")
    # list.sort(key=len, reverse=True) is stable: a counting sort of the chains by length and the
    # members copied chain by chain, in libiamx (the numpy form -- argsort, repeat, two gathers
    # over all members -- was a third of this function)
    f_img, f_kp = empty_huge(total, np.int32), empty_huge(total, np.int32)
    new_ptr = np.zeros(n_chain + 1, np.int64)
    rc = lib().iamx_chains_longest_first(P(o_img), P(o_kp), P(o_ptr), n_chain, P(f_img), P(f_kp),
                                         P(new_ptr), min(SCAN_THREADS, 8))
    if rc != 0:
        raise RuntimeError("iamx_chains_longest_first failed (%d): %s"
                           % (rc, (lib().iamx_last_error() or b'?').decode()))
    # kp.pt of every chain member (python floats of the float32 values, like list(kp.pt)), in the
    # final order: one native pass over the members that reads the images' own position arrays
    # (iamx_chain_members_uv).  The reference's kp_list[m[1]] raises IndexError for a keypoint
    # index its image does not have -- a .match file that is stale against its .feat --, negative
    # indices wrap inside the image: both are kept.
    import ctypes
    n_img = len(proj.image_list)
    xy_of, n_kp = [None] * n_img, np.zeros(max(n_img, 1), np.int64)
    uv = empty_huge((total, 2), np.float64) if total else np.zeros((0, 2), np.float64)
    if total:
        # positions of every image that occurs (a missing one shows up as an out-of-range index)
        occurs = np.zeros(n_img, bool)
        occurs[f_img] = True
        for i in np.nonzero(occurs)[0].tolist():
            xy_of[i] = np.ascontiguousarray(_kp_xy(proj.image_list[i]), np.float32).reshape(-1, 2)
            n_kp[i] = len(xy_of[i])
        ptrs = (ctypes.c_void_p * max(n_img, 1))(*[(a.ctypes.data if a is not None and len(a) else None) for a in xy_of])
        bad = np.full(1, -1, np.int64)
        rc = lib().iamx_chain_members_uv(P(f_img), P(f_kp), total, ptrs, P(n_kp), n_img, P(uv), P(bad), SCAN_THREADS)
        if rc != 0:
            k = int(bad[0])
            if k >= 0:
                raise IndexError("list index out of range: keypoint %d of %s (%d keypoints); is its "
                                 ".match file stale against the .feat?"
                                 % (int(f_kp[k]), proj.image_list[int(f_img[k])].name, int(n_kp[int(f_img[k])])))
            raise RuntimeError("iamx_chain_members_uv failed (%d): %s"
                               % (rc, (lib().iamx_last_error() or b'?').decode()))
    out = Chains(f_img, uv, new_ptr)
    if n_chain:
        _log("Total unique features in image set:", n_chain)
        _log("Keypoint average instances:", "%.2f" % (total / float(n_chain)))
    return out


# --------------------------------------------------------------------------------------
# initial triangulation -- match_cleanup.py:303-347
# --------------------------------------------------------------------------------------
def _base_elevations(proj):
    """per image: /smart/<name>/tri_surface_m if present, else the SRTM ground under the camera
    (lib.srtm, only inside the reference environment); never above 1 m below the camera."""
    smart = _deps.smart()
    if smart is not None:
        smart.load(proj.analysis_dir)
        smart_node = smart.smart_node
    else:
        smart_node = _deps.getNode("/smart", True)
    srtm = None
    base = np.zeros(len(proj.image_list))
    for i, image in enumerate(proj.image_list):
        image_node = smart_node.getChild(image.name, True)
        ned, _ypr, _quat = image.get_camera_pose()
        if image_node.hasChild("tri_surface_m"):
            b = image_node.getFloat("tri_surface_m")
        else:
            if srtm is None:
                srtm = _deps.srtm()
                if srtm is None:
                    raise RuntimeError("no /smart/%s/tri_surface_m estimate and no lib.srtm to "
                                       "look the ground elevation up" % image.name)
            b = srtm.ned_interp([ned[0], ned[1]])[0]
        if -ned[2] - 1 < b:
            b = -ned[2] - 1
        image.base_elev = b
        base[i] = b
    return base


def triangulate_smart(proj, matches):
    import torch
    from . import kernels
    from .kernels import _ptr, check, lib, stream_ptr
    cam = _deps.camera()
    IK = np.linalg.inv(cam.get_K(optimized=False))
    _log("Looking up [smart] base elevation for each image location...")
    base = _base_elevations(proj)
    _log("Estimating initial projection for each feature...")
    n_img = len(proj.image_list)
    M = np.zeros((n_img, 9))
    ned = np.zeros((n_img, 3))
    for i, image in enumerate(proj.image_list):
        cam2body = image.get_cam2body() if hasattr(image, 'get_cam2body') else CAM2BODY
        M[i] = image.get_body2ned().dot(cam2body).dot(IK).ravel()
        ned[i] = image.get_camera_pose()[0]
    n = len(matches)
    if n == 0:
        return
    fast = isinstance(matches, Chains) and matches.untouched()
    # (millions of small lists are alive here -- the matches_grouped contract --: every
    #  generation-2 pass of the cyclic collector walks them all again, and the list-building
    #  calls below trigger many)
    if fast:
        ptr, obs_img, obs_uv = matches.ptr, matches.img, matches.uv
    else:
      with _no_gc():
        ptr = np.zeros(n + 1, np.int64)
        np.cumsum([len(m) - 2 for m in matches], out=ptr[1:])
        flat = [p for m in matches for p in m[2:]]
        obs_img = np.fromiter((p[0] for p in flat), np.int32, len(flat))
        obs_uv = np.array([p[1] for p in flat], np.float64).reshape(-1, 2)
        del flat
    dev = kernels.require_gpu()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_M, d_ned, d_base, d_img, d_uv, d_ptr = (t(a) for a in (M, ned, base, obs_img, obs_uv, ptr))
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    n_sky = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().iamx_triangulate_ground(_ptr(d_M), _ptr(d_ned), _ptr(d_base), n_img, _ptr(d_img),
                                        _ptr(d_uv), _ptr(d_ptr), n, _ptr(out), _ptr(n_sky),
                                        stream_ptr()), 'iamx_triangulate_ground')
    for _ in range(int(n_sky.item())):
        _log('vector projected above horizon.')
    if fast and matches.untouched():
        matches.ned[:] = out.cpu().numpy()
        matches.has_ned[:] = True
        return
    with _no_gc():
        res = out.cpu().numpy().tolist()
        for m, p in zip(matches, res):
            m[0] = p
