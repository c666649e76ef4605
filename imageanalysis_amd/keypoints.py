"""Array-backed keypoint lists.

The reference keeps `image.kp_list` as a python list of cv2.KeyPoint objects (image.py:313-346)
and caches it as a pickled list of `(pt, size, angle, response, octave, class_id)` tuples
(image.py:187-203).  At ~50 k keypoints per frame building those objects costs more than the
GPU detector.  `KeyPointList` holds the columns as arrays; it is a sequence of keypoint objects
for every reader (the objects are created -- all of them, once -- the first time one is asked
for, so edits of a keypoint persist like in a list), `len()` and the package's own readers
(`xy()`: the [N,2] float32 array of kp.pt) never create them.

`feat_bytes()` / `from_feat_bytes()`: the `.feat` pickle written straight from / parsed straight
into the columns.  The stream is a plain protocol-2 pickle of the list of tuples (any
`pickle.load` reads it); the parser recognises that fixed-width layout and falls back to
`pickle.loads` for files written by anybody else."""
import pickle
from collections.abc import Sequence

import numpy as np

_COLS = ('x', 'y', 'size', 'angle', 'response')

# one keypoint of the .feat pickle:  ( G x G y TUPLE2  G size G angle G response  J octave
# J class_id  TUPLE          (BINFLOAT is big-endian, BININT little-endian signed)
_REC = np.dtype([('m', 'S1'), ('gx', 'S1'), ('x', '>f8'), ('gy', 'S1'), ('y', '>f8'), ('t2', 'S1'),
                 ('gs', 'S1'), ('size', '>f8'), ('ga', 'S1'), ('angle', '>f8'),
                 ('gr', 'S1'), ('response', '>f8'), ('jo', 'S1'), ('octave', '<i4'),
                 ('jc', 'S1'), ('class_id', '<i4'), ('t', 'S1')])
_OPS = (('m', b'('), ('gx', b'G'), ('gy', b'G'), ('t2', b'\x86'), ('gs', b'G'), ('ga', b'G'),
        ('gr', b'G'), ('jo', b'J'), ('jc', b'J'), ('t', b't'))
_HEAD, _TAIL, _EMPTY = b'\x80\x02](', b'e.', b'\x80\x02].'
assert _REC.itemsize == 58


class KeyPointList(Sequence):
    __slots__ = ('x', 'y', 'size', 'angle', 'response', 'octave', 'class_id', '_objs', '_xy',
                 '__weakref__')

    def __init__(self, x, y, size, angle, response, octave, class_id=None):
        """columns; the float members are rounded to float32 like cv2.KeyPoint's"""
        n = len(x)
        for name, col in zip(_COLS, (x, y, size, angle, response)):
            setattr(self, name, np.ascontiguousarray(np.asarray(col, np.float64).astype(np.float32)))
        self.octave = np.ascontiguousarray(octave, np.int32).reshape(n)
        self.class_id = np.full(n, -1, np.int32) if class_id is None \
            else np.ascontiguousarray(class_id, np.int32).reshape(n)
        self._objs = None
        self._xy = None

    # ---- array side
    def xy(self):
        """[N,2] float32 of kp.pt (as the keypoints were when this was first asked for: the
        same caching rule as matcher._kp_xy applies to a plain list)"""
        if self._xy is None:
            if self._objs is not None:
                self._xy = np.array([kp.pt for kp in self._objs], np.float32).reshape(-1, 2)
            else:
                self._xy = np.ascontiguousarray(np.stack([self.x, self.y], 1))
        return self._xy

    # ---- list side
    def objects(self):
        if self._objs is None:
            from .image import make_keypoints
            self._objs = make_keypoints(self.x, self.y, self.size, self.angle, self.response,
                                        self.octave, self.class_id)
        return self._objs

    def __len__(self):
        return len(self.x)

    def __getitem__(self, k):
        return self.objects()[k]

    def __iter__(self):
        return iter(self.objects())

    def __reduce_ex__(self, protocol):
        return (list, (self.objects(),))

    # ---- the .feat file
    def feat_bytes(self, as_view=False):
        """the .feat pickle (bytes; as_view: a memoryview of the buffer it was written into)"""
        if self._objs is not None:                       # the objects are the truth once they exist
            return pickle.dumps([(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id)
                                 for kp in self._objs])
        n = len(self)
        if n == 0:
            return _EMPTY
        # (libiamx writes the records: numpy field assignments into the packed record type cost
        #  11 ms per 50 k keypoints; head and tail go into the same buffer)
        import ctypes
        from . import _lib
        h = len(_HEAD)
        out = np.empty(h + n * _REC.itemsize + len(_TAIL), np.uint8)
        out[:h] = np.frombuffer(_HEAD, np.uint8)
        out[h + n * _REC.itemsize:] = np.frombuffer(_TAIL, np.uint8)
        try:
            L = _lib.lib()
        except (OSError, _lib.IamxError):
            # no libiamx.so (a CPU-only tool re-saving a .feat): the same bytes from numpy -- this
            # is file formatting, not the compute path, which has no fallback
            rec = out[h:h + n * _REC.itemsize].view(_REC)
            for name, op in _OPS:
                rec[name] = op
            for name in _COLS:
                rec[name] = getattr(self, name)
            rec['octave'] = self.octave
            rec['class_id'] = self.class_id
            return memoryview(out) if as_view else out.tobytes()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(L.iamx_feat_records(p(self.x), p(self.y), p(self.size), p(self.angle),
                                       p(self.response), p(self.octave), p(self.class_id), n,
                                       ctypes.c_void_p(out.ctypes.data + h)),
                   'iamx_feat_records')
        return memoryview(out) if as_view else out.tobytes()

    @staticmethod
    def from_feat_bytes(blob):
        """KeyPointList of a decompressed .feat file (a plain list of keypoint objects for an
        empty one, like the reference)"""
        blob = bytes(blob) if not isinstance(blob, bytes) else blob
        n, rem = divmod(len(blob) - len(_HEAD) - len(_TAIL), _REC.itemsize)
        if n > 0 and rem == 0 and blob.startswith(_HEAD) and blob.endswith(_TAIL):
            rec = np.frombuffer(blob, _REC, n, len(_HEAD))
            if all(bool((rec[name] == op).all()) for name, op in _OPS):
                return KeyPointList(rec['x'], rec['y'], rec['size'], rec['angle'], rec['response'],
                                    rec['octave'], rec['class_id'])
        feature_list = pickle.loads(blob)
        if not len(feature_list):
            return []
        pt, size, angle, response, octave, class_id = zip(*feature_list)
        x, y = zip(*pt)
        return KeyPointList(x, y, size, angle, response, octave, class_id)
