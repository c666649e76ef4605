"""Array-backed match lists.

The reference keeps `image.match_list[other_name]` as a python list of `[i, j]` lists
(scripts/lib/matcher.py:978-979; pickled as they are by image.py:261-268) -- tens of millions of
two-element lists on a dense survey.  `MatchPairs` holds the same pairs as ONE int32 [n, 2]
array (what the device hands back) and behaves like that list for every reader: len(),
indexing, iteration, slicing, comparison with lists, in-place edits (the first element access
turns it into the real list of lists, once), `np.asarray()` without a copy while untouched.  It
pickles as a plain list of lists, so `.match` files written from it load in the reference
(and anywhere else) without this module.

`dump_match_dict()` writes the `.match` pickle of a whole {name: pairs} dictionary straight
from the arrays (a hand-assembled protocol-2 stream: the objects the file loads to are the ones
`pickle.dump()` of the equivalent dict of lists would give)."""
import pickle
import struct
from collections.abc import MutableSequence

import os

import numpy as np


class MatchPairs(MutableSequence):
    __slots__ = ('_a', '_l', '_pk')

    def __init__(self, pairs=()):
        self._pk = None         # the pickled form of the array (find_matches fills it in while the
        if isinstance(pairs, np.ndarray):       # GPU works; dropped by any edit)
            self._a = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
            self._l = None
        else:
            self._a = None
            self._l = [list(p) for p in pairs]

    @classmethod
    def of_array(cls, a):
        """a contiguous int32 [n, 2] array as it is (no checks, no copy: find_matches makes two
        of these per image pair with matches)"""
        m = object.__new__(cls)
        m._a, m._l, m._pk = a, None, None
        return m

    def pickled(self):
        """bytes of this list inside a .match pickle (cached while the pairs are untouched: the
        cache entry names the array it was made from, prepickle() fills it in from another thread)"""
        pk = self._pk
        if pk is None or self._a is None or pk[0] is not self._a:
            return _pairs_bytes(self.array())
        return pk[1]

    # ---- the two representations
    def array(self):
        """int32 [n, 2] (a view while the pairs have not been edited as a list)"""
        if self._a is not None:
            return self._a
        return np.asarray(self._l, np.int32).reshape(-1, 2)

    def _as_list(self):
        if self._l is None:
            self._l = self._a.tolist()
            self._a = None
            self._pk = None
        return self._l

    def tolist(self):
        return self._a.tolist() if self._a is not None else [list(p) for p in self._l]

    def __array__(self, dtype=None, copy=None):
        a = self.array()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    # ---- list protocol
    def __len__(self):
        return len(self._a) if self._a is not None else len(self._l)

    # Element access turns the object into the real list of lists, once: a reader that keeps or
    # edits a pair it was handed (`pair[0] = ...`, `matches[k] is matches[k]`) then sees plain
    # list semantics.  len(), np.asarray(), comparison and pickling never need the lists.
    def __getitem__(self, k):
        return self._as_list()[k]

    def __iter__(self):
        return iter(self._as_list())

    def __setitem__(self, k, value):
        if isinstance(k, slice) and k == slice(None) and isinstance(value, (np.ndarray, MatchPairs)):
            # a full-slice assignment of an array (match_cleanup.merge_duplicates) resets the
            # object to its array form whatever it was before: storing the array's rows as the
            # elements of the list form would hand out ndarrays where `pair == [a, b]` is expected
            self._a = np.ascontiguousarray(np.asarray(value), np.int32).reshape(-1, 2).copy()
            self._l = None
            self._pk = None
            return
        if isinstance(value, np.ndarray):
            value = value.tolist()
        self._as_list()[k] = value

    def __delitem__(self, k):
        del self._as_list()[k]

    def insert(self, k, value):
        self._as_list().insert(k, value)

    def __eq__(self, other):
        if isinstance(other, MatchPairs):
            a, b = self.array(), other.array()
            return a.shape == b.shape and bool((a == b).all())
        if isinstance(other, list):
            return self.tolist() == other
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __add__(self, other):
        return self.tolist() + list(other)

    def __repr__(self):
        return repr(self.tolist())

    def __reduce_ex__(self, protocol):
        return (list, (self.tolist(),))


_HUGE = 2 << 20
_libc = None


_thp = None


def _thp_pays():
    global _thp
    if _thp is None:
        try:
            from . import _lib
            _thp = bool(_lib.lib().iamx_thp_pays())
        except Exception:                 # noqa: BLE001  (no library: plain pages)
            _thp = False
    return _thp


def empty_huge(shape, dtype):
    """np.empty() for arrays of many megabytes that are written once: anonymous memory with
    MADV_HUGEPAGE, so that the first touch maps 2 MiB at a time where the kernel grants
    transparent huge pages (a fresh 4 KiB page costs microseconds of kernel time, and a heavy
    round of find_matches writes ~10^5 of them: 0.45 s per round on the GPU box, 1.6 s per
    268 MB in the build container; 10x less with huge pages).  Plain np.empty() below 8 MiB or
    when mmap / madvise are not to be had."""
    import ctypes
    import mmap
    global _libc
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.ndim(shape) == 0 else tuple(int(x) for x in shape)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n < (8 << 20):
        return np.empty(shape, dtype)
    try:
        m = mmap.mmap(-1, n + _HUGE)
        raw = np.frombuffer(m, np.uint8)
        off = (-raw.ctypes.data) % _HUGE
        if _libc is None:
            _libc = ctypes.CDLL(None, use_errno=True)
        # (only where huge pages pay on this host today: libiamx probes once per process -- on a
        #  fragmented host the advice makes every first touch wait for compaction, 9x slower)
        if _thp_pays():
            _libc.madvise(ctypes.c_void_p(raw.ctypes.data + off), ctypes.c_size_t(n), 14)   # MADV_HUGEPAGE
        return raw[off:off + n].view(dtype).reshape(shape)
    except (OSError, ValueError, AttributeError):
        return np.empty(shape, dtype)


def _int_opcode(max_value, min_value):
    """(opcode, numpy dtype) of the narrowest fixed-width pickle integer that holds the range"""
    if min_value >= 0 and max_value < 65536:
        return b'M', '<u2'          # BININT2
    return b'J', '<i4'              # BININT (signed)


def _pairs_bytes(a):
    """pickle of a list of [i, j] lists: ] ( {] ( int int e}* e   -- no memo entries"""
    n = len(a)
    if n == 0:
        return b']'
    op, dt = _int_opcode(int(a.max()), int(a.min()))
    rec = np.empty(n, np.dtype([('h', 'S3'), ('i', dt), ('m', 'S1'), ('j', dt), ('t', 'S1')]))
    rec['h'] = b'](' + op
    rec['i'] = a[:, 0]
    rec['m'] = op
    rec['j'] = a[:, 1]
    rec['t'] = b'e'
    return b'](' + rec.tobytes() + b'e'


def prepickle(match_lists):
    """fill in the cached .match bytes of several MatchPairs with one pass (find_matches: the
    hits of a round, while the GPU works on the next one; safe to run on another thread)"""
    todo = [(m, m._a) for m in match_lists if isinstance(m, MatchPairs) and m._a is not None and m._pk is None]
    for (m, a), b in zip(todo, _pairs_bytes_many([a for _m, a in todo])):
        m._pk = (a, b)


def _pairs_bytes_many(arrays):
    """_pairs_bytes() of several int32 [n, 2] arrays in one call into libiamx (an image has ~100
    partners with ~200 matches each: the numpy calls, not the bytes, were the cost): memoryviews
    into ONE buffer"""
    import ctypes
    from . import _lib
    if not arrays:
        return []
    lens = np.fromiter((len(a) for a in arrays), np.int64, len(arrays))
    off = np.zeros(len(arrays) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    full = [np.ascontiguousarray(a, np.int32).reshape(-1, 2) for a in arrays if len(a)]
    if len(full) > 1:
        cat = empty_huge((total, 2), np.int32)
        np.concatenate(full, out=cat)
    else:
        cat = full[0] if full else np.zeros((0, 2), np.int32)
    cap = 3 * len(arrays) + 13 * total
    out = empty_huge(cap, np.uint8)
    out_off = np.zeros(len(arrays) + 1, np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n = _lib.lib().iamx_pickle_pair_lists(p(cat), p(off), len(arrays), p(out), cap, p(out_off))
    if n < 0:
        _lib.check(int(n), 'iamx_pickle_pair_lists')
    view = memoryview(out)
    oo = out_off.tolist()
    return [view[oo[k]:oo[k + 1]] for k in range(len(arrays))]


_key_bytes = {}         # image name -> its BINUNICODE record (every image lists every partner)


def _key_record(name):
    head = _key_bytes.get(name)
    if head is None:
        key = name.encode('utf-8', 'surrogatepass')
        if len(_key_bytes) > 1 << 16:
            _key_bytes.clear()
        head = _key_bytes[name] = b'X' + struct.pack('<I', len(key)) + key
    return head


class QuietLedger(object):
    """The image pairs of one find_matches() call that ended without matches -- on an all-pairs
    schedule 95-99 % of several million pairs -- as index arrays instead of two dictionary
    entries and two empty lists per pair.  Every image's MatchDict refers to the ledger and
    turns its share into real `{other_name: []}` entries the first time somebody reads it.
    `seq` is the position of the pair in the order find_matches processed the pairs (the
    reference assigns match_list entries in that order, scripts/lib/matcher.py:978-979)."""

    def __init__(self, names, capacity=0):
        """capacity: the number of pairs the call will process at most (the three columns then
        never move: rounds append into them as they finish, nothing is joined at the end)"""
        self.names = names
        self._cols = np.empty((3, max(int(capacity), 1024)), np.int64)      # i, j, seq
        self._n = 0
        self._index = None

    def add(self, i, j, seq):
        k = len(i)
        if k:
            if self._n + k > self._cols.shape[1]:
                grown = np.empty((3, max(2 * self._cols.shape[1], self._n + k)), np.int64)
                grown[:, :self._n] = self._cols[:, :self._n]
                self._cols = grown
            self._cols[0, self._n:self._n + k] = i
            self._cols[1, self._n:self._n + k] = j
            self._cols[2, self._n:self._n + k] = seq
            self._n += k
            self._index = None

    def __len__(self):
        return self._n

    def entry_records(self):
        """per image: the bytes of `name: []` inside a .match pickle (key record + EMPTY_LIST);
        as one uint8 [n_images, L] table when all records have the same length (image names of
        one camera do), else a list of bytes"""
        rec = getattr(self, '_records', None)
        if rec is None:
            rec = [_key_record(n) + b']' for n in self.names]
            if rec and len(set(map(len, rec))) == 1:
                rec = np.frombuffer(b''.join(rec), np.uint8).reshape(len(rec), -1)
            self._records = rec
        return rec

    def partners_of(self, k):
        """(partner image indices, seq) of image k's quiet pairs, in processing order"""
        if self._index is None:
            qi, qj, sq = (self._cols[r, :self._n] for r in range(3))
            if len(sq) > 1 and np.any(np.diff(sq) < 0):
                o = np.argsort(sq, kind='stable')
                qi, qj, sq = qi[o], qj[o], sq[o]
            # both directions of every pair, in seq order, grouped by image with that order kept
            # inside every image: one counting sort in libiamx (iamx_ledger_index), the numpy form
            # (a stable radix argsort + three gathers) without the library
            m = len(sq)
            n = len(self.names)
            import ctypes
            from . import _lib
            try:
                L = _lib.lib()
                qi, qj, sq = (np.ascontiguousarray(a, np.int64) for a in (qi, qj, sq))
                other, seq = np.empty(2 * m, np.int64), np.empty(2 * m, np.int64)
                bounds = np.empty(n + 1, np.int64)
                P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
                _lib.check(L.iamx_ledger_index(P(qi), P(qj), P(sq), m, n, P(other), P(seq), P(bounds)),
                           'iamx_ledger_index')
            except (OSError, _lib.IamxError):
                img, other, seq = (np.empty(2 * m, np.int64) for _ in range(3))
                img[0::2], img[1::2] = qi, qj
                other[0::2], other[1::2] = qj, qi
                seq[0::2], seq[1::2] = sq, sq
                order = np.argsort(img.astype(np.uint16) if n <= 65535 else img, kind='stable')
                img, other, seq = img[order], other[order], seq[order]
                bounds = np.searchsorted(img, np.arange(n + 1))
            self._index = (other, seq, bounds)
        other, seq, bounds = self._index
        lo, hi = bounds[k], bounds[k + 1]
        return other[lo:hi], seq[lo:hi]


class MatchDict(dict):
    """`image.match_list` while / after find_matches: a dict {other image name: match list} whose
    entries for pairs WITHOUT matches live in a QuietLedger until the dict is first read.  The
    order of the keys, once read, is the reference's: entries that were there before the call
    keep their place, new ones follow in the order the pairs were processed."""

    def __init__(self, *a, **k):
        dict.__init__(self, *a, **k)
        self._ledger = None
        self._me = -1
        self._seq = None        # name -> seq of the entries set (by find_matches) during the call

    def attach(self, ledger, me):
        if self._ledger is not None and self._ledger is not ledger:
            self._materialize()
        self._ledger, self._me = ledger, me
        if self._seq is None:
            self._seq = {}

    def set_in_order(self, name, value, seq):
        """find_matches' assignment of a pair WITH matches (no read, nothing materialised)"""
        if self._seq is not None and not dict.__contains__(self, name):
            self._seq[name] = seq
        dict.__setitem__(self, name, value)

    def _pending(self):
        if self._ledger is None:
            return None
        other, seq = self._ledger.partners_of(self._me)
        return (other, seq) if len(other) else None

    def entries(self):
        """[(name, value)] in the final key order WITHOUT turning the quiet pairs into dictionary
        entries (the .match writer): value is the shared EMPTY for them"""
        pend = self._pending()
        items = list(dict.items(self))
        if pend is None:
            return items
        other, seq = pend
        names = self._ledger.names
        sq = self._seq or {}
        have = dict.__contains__
        # a quiet pair whose key was there before the call keeps its place but NOT its value: the
        # reference assigns match_list[name] = [] unconditionally (lib/matcher.py:978-979), so a
        # stale one-sided entry (an interrupted save) does not survive the retry
        redone = set(names[o] for o in other.tolist() if have(self, names[o]) and names[o] not in sq)
        old = [(k, EMPTY if k in redone else v) for k, v in items if k not in sq]
        new = [(sq[k], k, v) for k, v in items if k in sq]
        quiet = [(s, names[o], EMPTY) for o, s in zip(other.tolist(), seq.tolist())
                 if not have(self, names[o])]
        merged = sorted(new + quiet, key=lambda t: t[0]) if new else quiet
        return old + [(k, v) for _s, k, v in merged]

    def pickle_body(self):
        """the `name value` records of the .match pickle in the final key order, or None when the
        plain walk over entries() is needed (entries from before the call).  The quiet pairs come
        out of precomputed per-image records, a slice of them between two pairs with matches."""
        pend = self._pending()
        if pend is None or len(self._seq or {}) != dict.__len__(self):
            return None
        other, seq = pend
        rec = self._ledger.entry_records()
        if isinstance(rec, np.ndarray):
            table = rec[other]                            # [n_quiet, L]: a slice of it is a run
            run = lambda a, b: (table[a:b].tobytes(),) if b > a else ()
        else:
            quiet = list(map(rec.__getitem__, other.tolist()))
            run = lambda a, b: quiet[a:b]
        direct = sorted((s, k) for k, s in self._seq.items())
        cut = np.searchsorted(seq, np.array([s for s, _k in direct], np.int64)).tolist() if direct else []
        values = [dict.__getitem__(self, name) for _s, name in direct]
        arr_bytes = {t: v._pk[1] for t, v in enumerate(values)
                     if isinstance(v, MatchPairs) and v._pk is not None and v._a is not None
                     and v._pk[0] is v._a}
        arr_at = [t for t, v in enumerate(values) if isinstance(v, MatchPairs) and t not in arr_bytes]
        arr_bytes.update(zip(arr_at, _pairs_bytes_many([values[t].array() for t in arr_at])))
        out, prev = [], 0
        for t, ((s_, name), pos) in enumerate(zip(direct, cut)):
            out.extend(run(prev, pos))
            prev = pos
            pairs = values[t]
            out.append(_key_record(name))
            if t in arr_bytes:
                out.append(arr_bytes[t])
            elif isinstance(pairs, list) and not pairs:
                out.append(b']')
            else:
                body = pickle.dumps(pairs, 2)
                out.append(body[2:-1])
        out.extend(run(prev, len(other)))
        return out

    def _materialize(self):
        if self._ledger is None:
            return
        ent = self.entries()
        self._ledger, self._seq = None, None
        dict.clear(self)
        for k, v in ent:
            dict.__setitem__(self, k, [] if v is EMPTY else v)

    # ---- every read sees real entries
    def _read(name):                                   # noqa: N805
        base = getattr(dict, name)

        def method(self, *a, **k):
            self._materialize()
            return base(self, *a, **k)
        method.__name__ = name
        return method

    for _n in ('__getitem__', '__contains__', '__iter__', '__len__', '__eq__', '__ne__', '__repr__',
               'get', 'keys', 'values', 'items', 'copy', 'pop', 'popitem', 'setdefault', 'update',
               '__delitem__', '__setitem__', 'clear', '__reversed__', '__or__', '__ior__', '__ror__'):
        locals()[_n] = _read(_n)
    del _n, _read
    __hash__ = None

    def __bool__(self):
        return dict.__len__(self) > 0 or self._pending() is not None

    def __reduce_ex__(self, protocol):
        self._materialize()
        return (dict, (dict(self),))


class _Empty(list):
    """marker of MatchDict.entries() for a pair without matches (read only)"""
    __slots__ = ()


EMPTY = _Empty()


def dumps_match_dict(match_list):
    """bytes of the `.match` file of {name: MatchPairs | list}; values that are not MatchPairs
    (or hold anything but integer pairs) go through pickle itself."""
    return b''.join(_match_dict_pieces(match_list))


def _match_dict_pieces(match_list):
    """... as the list of byte pieces that follow each other (the file writer hands them to
    writelines(): the pair lists of a survey are a gigabyte that need not be joined first)"""
    out = [b'\x80\x02}']
    if match_list:
        out.append(b'(')
        # (a MatchDict hands out its quiet pairs without creating dictionary entries for them)
        if isinstance(match_list, MatchDict):
            body = match_list.pickle_body()
            if body is not None:
                out.extend(body)
                items = ()
            else:
                items = match_list.entries()
        else:
            items = match_list.items()
        for name, pairs in items:
            if not isinstance(name, str):
                return [pickle.dumps(dict(match_list))]
            out.append(_key_record(name))
            if isinstance(pairs, MatchPairs):
                out.append(pairs.pickled())
            elif isinstance(pairs, list) and not pairs:
                out.append(b']')
            else:
                body = pickle.dumps(pairs, 2)
                assert body[:2] == b'\x80\x02' and body[-1:] == b'.'
                # a nested stream's memo indices start at 0 again; they stay inside this value
                # (pickle memo entries may be overwritten), and nothing outside refers to them
                out.append(body[2:-1])
        out.append(b'u')
    out.append(b'.')
    return out


def dump_match_dict(match_list, fp):
    fp.write(dumps_match_dict(match_list))
