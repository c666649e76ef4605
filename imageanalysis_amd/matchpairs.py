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

import numpy as np


class MatchPairs(MutableSequence):
    __slots__ = ('_a', '_l')

    def __init__(self, pairs=()):
        if isinstance(pairs, np.ndarray):
            self._a = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
            self._l = None
        else:
            self._a = None
            self._l = [list(p) for p in pairs]

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


_key_bytes = {}         # image name -> its BINUNICODE record (every image lists every partner)


def dumps_match_dict(match_list):
    """bytes of the `.match` file of {name: MatchPairs | list}; values that are not MatchPairs
    (or hold anything but integer pairs) go through pickle itself."""
    out = [b'\x80\x02}']
    if match_list:
        out.append(b'(')
        for name, pairs in match_list.items():
            head = _key_bytes.get(name)
            if head is None:
                if not isinstance(name, str):
                    return pickle.dumps(match_list)
                key = name.encode('utf-8', 'surrogatepass')
                if len(_key_bytes) > 1 << 16:
                    _key_bytes.clear()
                head = _key_bytes[name] = b'X' + struct.pack('<I', len(key)) + key
            out.append(head)
            if isinstance(pairs, MatchPairs):
                out.append(_pairs_bytes(pairs.array()))
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
    return b''.join(out)


def dump_match_dict(match_list, fp):
    fp.write(dumps_match_dict(match_list))
