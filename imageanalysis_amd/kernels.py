"""Thin python bindings over the C ABI: torch tensors are device-buffer containers only,
every number is produced by a hand-written HIP kernel in libiamx.so (csrc/*.hip)."""
import os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, require_gpu, stream_ptr

I8, I32, I64, F64, U8 = torch.int8, torch.int32, torch.int64, torch.float64, torch.uint8


def _ptr(t):
    return _lib.c_void_p(t.data_ptr()) if t is not None else _lib.c_void_p(0)


def _dev(x, dtype):
    """numpy / tensor -> contiguous device tensor of `dtype` (a copy, not a compute step)."""
    dev = require_gpu()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    x = np.ascontiguousarray(x)
    if not x.flags.writeable:
        import warnings
        with warnings.catch_warnings():       # a read-only source (bytes of a decoder) is only read
            warnings.simplefilter('ignore', UserWarning)
            return torch.from_numpy(x).to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(x).to(device=dev, dtype=dtype).contiguous()


# --------------------------------------------------------------------------------------
# descriptor store
# --------------------------------------------------------------------------------------
# Upload staging of DescriptorStore.set_images: two page-locked host buffers (a group of images is
# gathered into one while the copy out of the other runs), one device buffer the pack kernels read
# (re-used in stream order) and their scratch.  One set per device, kept for the process.
STAGE_BYTES = 64 << 20
STAGE_TAIL = 64 << 10            # int64 offsets of up to 8191 images per group
_stages = {}
_stage_lock = threading.Lock()


def _upload_stage(dev, nbytes):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _stages.get(key)
    if st is None or st['bytes'] < nbytes:
        rows = (nbytes + 127) // 128
        # (+ STAGE_TAIL bytes behind the rows: the group's row offsets travel with the same copy -- a
        #  pageable `tensor.to(device)` of their own would make the host wait for the device per group)
        st = _stages[key] = dict(bytes=rows * 128,
                                 pin=[torch.empty(rows * 128 + STAGE_TAIL, dtype=U8).pin_memory() for _ in range(2)],
                                 dev=torch.empty(rows * 128 + STAGE_TAIL, dtype=U8, device=dev),
                                 scratch=torch.empty(3 * rows, dtype=I32, device=dev),
                                 ev=[torch.cuda.Event(), torch.cuda.Event()], used=[False, False], next=0)
    return st


def prewarm_upload_stage():
    """Allocate the upload staging of the current device on a helper thread (page-locking its two
    64 MB host buffers takes 0.1-0.2 s: matcher.configure() starts it, find_matches finds it done)."""
    if not torch.cuda.is_available():
        return None
    idx = torch.cuda.current_device()

    def _go():
        try:
            with torch.cuda.device(idx), _stage_lock:
                _upload_stage(torch.device('cuda', idx), STAGE_BYTES)
        except Exception:                     # noqa: BLE001  (best effort: set_images allocates what is missing)
            pass
    th = threading.Thread(target=_go, name='iamx-stage', daemon=True)
    th.start()
    return th


class DescriptorStore(object):
    """All images' SIFT descriptors packed back to back in HBM (include/iamx.h, 'Descriptor
    store'): int8 rows (value-128), each image zero-padded to 128 rows, + two int32 norms
    per row.  288 GB of HBM hold > 10^9 descriptors, so a whole survey stays resident."""

    def __init__(self, counts, train_layout=True, reserve_rows=0, reserve_images=0):
        """train_layout=False: the parity-partitioned copy `desc2` (a third of the arena) is not
        kept.  Only the ONE-direction fast sweep reads it; a store that serves batches holding
        both directions of every pair (find_matches: the symmetric sweep reads `desc3`, the exact
        stage `desc`) does without, and a batch the symmetric sweep refuses (an image of < 2 rows)
        then takes the general kernel, which reads `desc` only.
        reserve_rows / reserve_images: capacity beyond `counts` -- try_extend() appends images
        without re-allocating (find_matches meeting undetected images registers ~250 per round:
        rebuilding a 40 GB arena each time was a quarter of that call form's time)."""
        dev = require_gpu()
        L = lib()
        self.has_train_layout = bool(train_layout)
        self.counts = [int(c) for c in counts]
        reserve_rows, reserve_images = int(reserve_rows), int(reserve_images)
        pads = [int(L.iamx_desc_padded_rows(c)) for c in self.counts]
        offs = np.zeros(len(pads) + 1, np.int64)
        np.cumsum(pads, out=offs[1:])
        if offs[-1] >= 2 ** 31:
            raise ValueError("descriptor store limited to 2^31 rows")
        self.offsets = offs
        total = max(int(offs[-1]), reserve_rows, 1)
        if total >= 2 ** 31:
            total = max(int(offs[-1]), 1)
        self.desc = torch.empty((total, 128), dtype=I8, device=dev)
        self.norm_q = torch.empty(total, dtype=I32, device=dev)
        self.norm_t = torch.empty(total, dtype=I32, device=dev)
        self.key_t = torch.empty(total, dtype=I32, device=dev)      # scratch of iamx_knn2sym_exact
        self.img_off = torch.from_numpy(offs[:-1].astype(np.int32)).to(dev)
        self.img_n = torch.tensor(self.counts, dtype=I32, device=dev) if self.counts else \
            torch.zeros(0, dtype=I32, device=dev)
        # train-side layout of the fast kernel (parity partitioned, include/iamx.h "desc2")
        caps = [int(L.iamx_desc2_rows_cap(c)) for c in self.counts]
        offs2 = np.zeros(len(caps) + 1, np.int64)
        np.cumsum(caps, out=offs2[1:])
        if offs2[-1] >= 2 ** 31:
            raise ValueError("descriptor store limited to 2^31 rows")
        self.offsets2 = offs2
        total2 = max(int(offs2[-1]), min(reserve_rows, 2 ** 31 - 1), 1) if train_layout else 1
        self.desc2 = torch.empty((total2, 128), dtype=I8, device=dev)
        self.norm2 = torch.empty(total2, dtype=I32, device=dev)
        self.cinit = torch.empty(total2, dtype=I32, device=dev)
        self.perm = torch.empty(total2, dtype=I32, device=dev)
        self.meta = torch.zeros((max(len(caps), reserve_images, 1), 4), dtype=I32, device=dev)
        self.img_off2 = torch.from_numpy(offs2[:-1].astype(np.int32)).to(dev)
        # sorted layout of the symmetric sweep (rows ordered by |a-128|^2, include/iamx.h "desc3")
        caps3 = [int(L.iamx_desc3_rows_cap(c)) for c in self.counts]
        offs3 = np.zeros(len(caps3) + 1, np.int64)
        np.cumsum(caps3, out=offs3[1:])
        self.caps3 = np.asarray(caps3, np.int64)
        self.offsets3 = offs3
        total3 = max(int(offs3[-1]), min(reserve_rows, 2 ** 31 - 1), 1)
        self.desc3 = torch.empty((total3, 128), dtype=I8, device=dev)
        self.sn2 = torch.empty(total3, dtype=I32, device=dev)
        self.sct = torch.empty(total3, dtype=I32, device=dev)
        self.sperm = torch.empty(total3, dtype=I32, device=dev)
        self.sinv = torch.empty(total3, dtype=I32, device=dev)
        self.img_off3 = torch.from_numpy(offs3[:-1].astype(np.int32)).to(dev)

    def __len__(self):
        return len(self.counts)

    @property
    def rows_used(self):
        """rows of the original-order layout that belong to images (the buffers may be larger)"""
        return int(self.offsets[-1])

    def try_extend(self, new_counts):
        """Append images of new_counts rows each behind the existing ones if every layout's
        buffers have room: the big buffers stay, the small per-image tables are rebuilt (batches
        made earlier keep their own references; the rows of old images do not move).  -> False
        (and nothing changed) when the capacity does not suffice."""
        L = lib()
        dev = self.desc.device
        counts = self.counts + [int(c) for c in new_counts]
        offs = np.zeros(len(counts) + 1, np.int64)
        np.cumsum([int(L.iamx_desc_padded_rows(c)) for c in counts], out=offs[1:])
        offs2 = np.zeros(len(counts) + 1, np.int64)
        np.cumsum([int(L.iamx_desc2_rows_cap(c)) for c in counts], out=offs2[1:])
        caps3 = [int(L.iamx_desc3_rows_cap(c)) for c in counts]
        offs3 = np.zeros(len(counts) + 1, np.int64)
        np.cumsum(caps3, out=offs3[1:])
        if offs[-1] > self.desc.shape[0] or offs3[-1] > self.desc3.shape[0]:
            return False
        if self.has_train_layout and (offs2[-1] > self.desc2.shape[0] or len(counts) > self.meta.shape[0]):
            return False
        if max(offs[-1], offs2[-1], offs3[-1]) >= 2 ** 31:
            return False
        self.counts = counts
        self.offsets, self.offsets2, self.offsets3 = offs, offs2, offs3
        self.caps3 = np.asarray(caps3, np.int64)
        self.img_off = torch.from_numpy(offs[:-1].astype(np.int32)).to(dev)
        self.img_n = torch.tensor(counts, dtype=I32, device=dev)
        self.img_off2 = torch.from_numpy(offs2[:-1].astype(np.int32)).to(dev)
        self.img_off3 = torch.from_numpy(offs3[:-1].astype(np.int32)).to(dev)
        return True

    def ensure_train_layout(self, chunk_rows=4 << 20):
        """Builds the parity-partitioned copy `desc2` of a store made without it, from the
        original-order rows already on the device (iamx_desc_unpack_u8 -> iamx_desc2_pack_batch_u8,
        a few million rows at a time).  Enqueued on the current stream; find_matches asks for it the
        first time a round is routed to the one-direction sweep."""
        if self.has_train_layout:
            return
        dev = self.desc.device
        L, sp = lib(), stream_ptr()
        total2 = max(int(self.offsets2[-1]), 1)
        self.desc2 = torch.empty((total2, 128), dtype=I8, device=dev)
        self.norm2 = torch.empty(total2, dtype=I32, device=dev)
        self.cinit = torch.empty(total2, dtype=I32, device=dev)
        self.perm = torch.empty(total2, dtype=I32, device=dev)
        counts = np.asarray(self.counts, np.int64)
        keep = []
        first = 0
        while first < len(counts):
            k, rows = 0, 0
            while first + k < len(counts) and (k == 0 or rows + counts[first + k] <= chunk_rows):
                rows += int(counts[first + k])
                k += 1
            if rows:
                off = np.zeros(k + 1, np.int64)
                np.cumsum(counts[first:first + k], out=off[1:])
                d_off = torch.from_numpy(off).to(dev)
                u8 = torch.empty((rows, 128), dtype=U8, device=dev)
                mx = int(counts[first:first + k].max())
                check(L.iamx_desc_unpack_u8(_ptr(self.desc), _ptr(self.img_off[first:]), _ptr(d_off), k, mx,
                                            _ptr(u8), sp), 'iamx_desc_unpack_u8')
                scratch = torch.empty(3 * rows, dtype=I32, device=dev)
                check(L.iamx_desc2_pack_batch_u8(_ptr(u8), _ptr(d_off), _ptr(self.img_off2[first:]), k, rows, mx,
                                                 _ptr(self.desc2), _ptr(self.norm2), _ptr(self.cinit),
                                                 _ptr(self.perm), _ptr(self.meta[first]), _ptr(scratch), sp),
                      'iamx_desc2_pack_batch_u8')
                keep.append((d_off, u8, scratch))
            first += k
        torch.cuda.current_stream().synchronize()          # (the temporaries above)
        del keep
        self.has_train_layout = True

    def set_image(self, i, des, sync=True):
        """Pack descriptors of image i.  `des`: [n,128] float32 (cv2/reference layout, integer
        valued) or uint8, numpy or device tensor.  With sync=False the kernels are only
        enqueued and the temporaries they read are returned: keep them alive until the stream
        has been synchronised."""
        n = self.counts[i]
        if n == 0:
            return ()
        if isinstance(des, np.ndarray) and des.dtype != np.uint8 and des.dtype != np.float32:
            des = des.astype(np.float32)
        is_u8 = (des.dtype == np.uint8) if isinstance(des, np.ndarray) else (des.dtype == U8)
        src = _dev(des, U8 if is_u8 else torch.float32)
        if tuple(src.shape) != (n, 128):
            raise ValueError("image %d: expected descriptors of shape (%d,128), got %s"
                             % (i, n, tuple(src.shape)))
        o = int(self.offsets[i])
        fn = lib().iamx_desc_pack_u8 if is_u8 else lib().iamx_desc_pack_f32
        check(fn(_ptr(src), n, _ptr(self.desc[o:]), _ptr(self.norm_q[o:]), _ptr(self.norm_t[o:]),
                 stream_ptr()), 'iamx_desc_pack')
        scratch = None
        if self.has_train_layout:
            o2 = int(self.offsets2[i])
            fn2 = lib().iamx_desc2_pack_u8 if is_u8 else lib().iamx_desc2_pack_f32
            scratch = torch.empty(3 * n, dtype=I32, device=src.device)
            check(fn2(_ptr(src), n, _ptr(self.desc2[o2:]), _ptr(self.norm2[o2:]), _ptr(self.cinit[o2:]),
                      _ptr(self.perm[o2:]), _ptr(self.meta[i]), _ptr(scratch), stream_ptr()),
                  'iamx_desc2_pack')
        o3 = int(self.offsets3[i])
        fn3 = lib().iamx_desc3_pack_u8 if is_u8 else lib().iamx_desc3_pack_f32
        scratch3 = torch.empty(3 * n, dtype=I32, device=src.device)
        check(fn3(_ptr(src), n, _ptr(self.desc3[o3:]), _ptr(self.sn2[o3:]), _ptr(self.sct[o3:]),
                  _ptr(self.sperm[o3:]), _ptr(self.sinv[o3:]), _ptr(scratch3), stream_ptr()),
              'iamx_desc3_pack')
        # the source buffer must outlive the enqueued kernels
        if sync:
            torch.cuda.current_stream().synchronize()
            return ()
        return (src, scratch, scratch3)

    def set_images(self, first, arrays):
        """Pack the descriptors of the consecutive images first .. first + len(arrays) - 1: host
        arrays ([n,128] float32 or uint8) go up in GROUPS of <= STAGE_BYTES -- gathered (float32:
        converted) into one of two page-locked staging buffers by libiamx's threads, copied to the
        device asynchronously and packed by the batched kernels while the threads fill the other
        buffer.  (Until round 6 the whole list was concatenated into ONE pageable host block first:
        2.4 GB copied by a numpy loop and uploaded synchronously for 512 frames of 37 k keypoints --
        0.37 s of a 2.3 s match stage -- and 47 GB of host AND device temporaries at 10 000 frames.)
        Enqueues on the current stream and returns the temporaries the kernels read (keep them
        until the stream is synchronised)."""
        import ctypes
        k = len(arrays)
        if k == 0:
            return ()
        counts = [self.counts[first + i] for i in range(k)]
        for i, a in enumerate(arrays):
            if not isinstance(a, np.ndarray) or a.dtype not in (np.float32, np.uint8) \
                    or tuple(a.shape) != (counts[i], 128):
                raise ValueError("set_images: image %d is not a host [n,128] float32 / uint8 array" % (first + i))
        arrays = [np.ascontiguousarray(a) for a in arrays]
        if int(sum(counts)) == 0:
            return ()
        dev = self.desc.device
        L, sp = lib(), stream_ptr()
        threads = min(16, os.cpu_count() or 1)
        keep = []
        with _stage_lock:
            self._set_images_staged(first, arrays, counts, keep, L, sp, threads, dev)
        return tuple(keep)

    def _set_images_staged(self, first, arrays, counts, keep, L, sp, threads, dev):
        import ctypes
        k = len(arrays)
        stage = _upload_stage(dev, max(STAGE_BYTES, 128 * int(max(counts))))
        keep.append(stage)
        g0 = 0
        while g0 < k:
            g1, rows = g0, 0
            while g1 < k and g1 - g0 < STAGE_TAIL // 8 - 1 and \
                    (g1 == g0 or (rows + counts[g1]) * 128 <= stage['bytes']):
                rows += counts[g1]
                g1 += 1
            n = g1 - g0
            if rows == 0:
                g0 = g1
                continue
            slot = stage['next'] = (stage['next'] + 1) % 2
            pin, ev = stage['pin'][slot], stage['ev'][slot]
            if stage['used'][slot]:
                ev.synchronize()                     # the copy that read this buffer last has run
            off = np.zeros(n + 1, np.int64)
            np.cumsum(counts[g0:g1], out=off[1:])
            grp = arrays[g0:g1]
            dst = ctypes.c_void_p(pin.data_ptr())
            srcs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in grp])
            if all(a.dtype == np.float32 for a in grp):
                cnt = (ctypes.c_int64 * n)(*[a.size for a in grp])
                check(L.iamx_f32_to_u8_many(srcs, cnt, n, dst, threads), 'iamx_f32_to_u8_many')
            elif all(a.dtype == np.uint8 for a in grp):
                cnt = (ctypes.c_int64 * n)(*[a.size for a in grp])
                check(L.iamx_u8_gather_many(srcs, cnt, n, dst, threads), 'iamx_u8_gather_many')
            else:
                host = pin.numpy()[:rows * 128].reshape(rows, 128)
                for i, a in enumerate(grp):
                    host[off[i]:off[i + 1]] = a if a.dtype == np.uint8 else \
                        np.clip(np.rint(a), 0, 255).astype(np.uint8)
            tail = stage['bytes']
            pin[tail:tail + 8 * (n + 1)].view(torch.int64).copy_(torch.from_numpy(off))
            # (two copies: the rows of the group, and the offsets behind the buffer's row area)
            stage['dev'][:rows * 128].copy_(pin[:rows * 128], non_blocking=True)
            stage['dev'][tail:tail + 8 * (n + 1)].copy_(pin[tail:tail + 8 * (n + 1)], non_blocking=True)
            src = stage['dev'][:rows * 128].view(rows, 128)
            ev.record()
            stage['used'][slot] = True
            # original-order store: image by image (one launch each: images are padded to 128 rows)
            for i in range(n):
                if counts[g0 + i]:
                    o = int(self.offsets[first + g0 + i])
                    check(L.iamx_desc_pack_u8(_ptr(src[int(off[i]):]), counts[g0 + i], _ptr(self.desc[o:]),
                                              _ptr(self.norm_q[o:]), _ptr(self.norm_t[o:]), sp), 'iamx_desc_pack_u8')
            src_off = stage['dev'][tail:tail + 8 * (n + 1)].view(torch.int64)
            mx = int(max(counts[g0:g1]))
            scratch = stage['scratch'][:3 * rows]
            if self.has_train_layout:
                check(L.iamx_desc2_pack_batch_u8(_ptr(src), _ptr(src_off), _ptr(self.img_off2[first + g0:]), n, rows, mx,
                                                 _ptr(self.desc2), _ptr(self.norm2), _ptr(self.cinit),
                                                 _ptr(self.perm), _ptr(self.meta[first + g0]), _ptr(scratch), sp),
                      'iamx_desc2_pack_batch_u8')
            check(L.iamx_desc3_pack_batch_u8(_ptr(src), _ptr(src_off), _ptr(self.img_off3[first + g0:]), n, rows, mx,
                                             _ptr(self.desc3), _ptr(self.sn2), _ptr(self.sct), _ptr(self.sperm),
                                             _ptr(self.sinv), _ptr(scratch), sp), 'iamx_desc3_pack_batch_u8')
            g0 = g1

    @classmethod
    def from_arrays(cls, arrays):
        st = cls([a.shape[0] for a in arrays])
        for i, a in enumerate(arrays):
            st.set_image(i, a)
        return st


def wg_per_pair(nq):
    return (int(nq) + 255) // 256


SYM_ROWS_PER_WG = {0: 256, 1: 512, 2: 1024}       # iamx_knn2sym_rows_per_wg(form)


def sym_tables(pairs, counts, caps3):
    """Host-side launch tables of the symmetric sweep (include/iamx.h, iamx_knn2sym_sweep) for a
    batch of ORDERED pairs [P,2], or None when the batch does not hold both directions of every
    image pair exactly once.  Unordered pair u = (B image, A image): B = the query image of the
    first of its two ordered pairs; the list is sorted by A so that consecutive workgroups
    re-read the same streamed rows from L2.  osrc[p] = (u, role) with role 0 when the query image
    of ordered pair p is B (its bounds come from the lane-local column direction), 1 when it is A."""
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    P = len(pairs)
    if P == 0 or P % 2 or (pairs[:, 0] == pairs[:, 1]).any():
        return None
    counts = np.asarray(counts, np.int64)
    caps3 = np.asarray(caps3, np.int64)
    n_img = len(counts)
    key = pairs[:, 0].astype(np.int64) * n_img + pairs[:, 1]
    mkey = pairs[:, 1].astype(np.int64) * n_img + pairs[:, 0]
    order = np.argsort(key, kind='stable')
    skey = key[order]
    if (skey[1:] == skey[:-1]).any():
        return None                                   # an ordered pair listed twice
    where = np.minimum(np.searchsorted(skey, mkey), P - 1)
    if (skey[where] != mkey).any():
        return None                                   # a pair without its mirror
    mirror = order[where]
    first = np.nonzero(np.arange(P) < mirror)[0]      # one ordered pair per unordered pair
    first = first[np.argsort(pairs[first, 1], kind='stable')]
    up = pairs[first]                                 # (B, A)
    n_u = len(up)
    nb = counts[up[:, 0]]
    mn = int(min(nb.min(), counts[up[:, 1]].min()))
    if mn < 2:
        return None
    form = 0 if mn < 2048 else (2 if mn >= 4096 else 1)
    rows_wg = SYM_ROWS_PER_WG[form]
    nwg = (nb + rows_wg - 1) // rows_wg
    wg = np.zeros(n_u + 1, np.int64)
    np.cumsum(nwg, out=wg[1:])
    if wg[-1] >= 2 ** 31:
        return None
    col_off = np.zeros(n_u + 1, np.int64)
    np.cumsum(caps3[up[:, 0]], out=col_off[1:])
    rowp_off = np.zeros(n_u + 1, np.int64)
    np.cumsum(nwg * caps3[up[:, 1]], out=rowp_off[1:])
    osrc = np.zeros((P, 2), np.int32)
    osrc[first, 0] = np.arange(n_u)
    osrc[mirror[first], 0] = np.arange(n_u)
    osrc[mirror[first], 1] = 1
    return dict(upairs=up, form=form, wg=wg, col_off=col_off, rowp_off=rowp_off, osrc=osrc)


def knn2_pairs(store, pairs):
    """Exact 2-NN for a batch of ordered (query image, train image) pairs.

    Returns (idx, d2, out_off): idx/d2 int32 device tensors [sum n_q, 2], out_off a numpy
    int64 [n_pairs+1] of each pair's first output row."""
    dev = require_gpu()
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    counts = np.asarray(store.counts, np.int64)
    P = len(pairs)
    nq = counts[pairs[:, 0]] if P else np.zeros(0, np.int64)
    if P and counts[pairs[:, 1]].min() < 2:
        raise ValueError("train image with fewer than 2 descriptors "
                         "(the reference returns no matches there, lib/matcher.py:205-210)")
    out_off = np.zeros(P + 1, np.int64)
    np.cumsum(nq, out=out_off[1:])
    wg = np.zeros(P + 1, np.int64)
    np.cumsum((nq + 255) // 256, out=wg[1:])
    if wg[-1] >= 2 ** 31:
        raise ValueError("batch too large: split the pair list")
    total = int(out_off[-1])
    idx = torch.empty((max(total, 1), 2), dtype=I32, device=dev)
    d2 = torch.empty((max(total, 1), 2), dtype=I32, device=dev)
    if P == 0 or total == 0:
        return idx[:0], d2[:0], out_off
    d_pairs = torch.from_numpy(pairs).to(dev)
    d_wg = torch.from_numpy(wg.astype(np.int32)).to(dev)
    d_out = torch.from_numpy(out_off[:-1].copy()).to(dev)
    check(lib().iamx_knn2_l2_pairs(_ptr(store.desc), _ptr(store.norm_q), _ptr(store.norm_t),
                                   _ptr(store.img_off), _ptr(store.img_n), _ptr(d_pairs),
                                   _ptr(d_wg), _ptr(d_out), P, int(wg[-1]), _ptr(idx), _ptr(d2),
                                   stream_ptr()), 'iamx_knn2_l2_pairs')
    torch.cuda.current_stream().synchronize()      # launch tables are temporaries
    return idx[:total], d2[:total], out_off


def knn2(des_q, des_t):
    """Single pair convenience (packs both sides): returns device (idx[nq,2], d2[nq,2])."""
    dev = require_gpu()
    st = DescriptorStore.from_arrays([des_q, des_t])
    nq, nt = st.counts
    idx = torch.empty((max(nq, 1), 2), dtype=I32, device=dev)
    d2 = torch.empty((max(nq, 1), 2), dtype=I32, device=dev)
    o = int(st.offsets[1])
    check(lib().iamx_knn2_l2_u8(_ptr(st.desc), _ptr(st.norm_q), nq, _ptr(st.desc[o:]),
                                _ptr(st.norm_t[o:]), nt, _ptr(idx), _ptr(d2), stream_ptr()),
          'iamx_knn2_l2_u8')
    torch.cuda.current_stream().synchronize()
    return idx[:nq], d2[:nq]


# --------------------------------------------------------------------------------------
# metric filter + compaction
# --------------------------------------------------------------------------------------
def match_metric(d2, seg_off, thresh):
    """d2 [n,2] int32 device; seg_off numpy/tensor int64 [n_seg+1].  Returns device tensors
    (metric f64 [n], keep u8 [n], seg_count i32 [n_seg]) and the number of d1==0 rows."""
    dev = require_gpu()
    n = d2.shape[0]
    seg = _dev(seg_off, I64)
    n_seg = seg.numel() - 1
    metric = torch.empty(max(n, 1), dtype=F64, device=dev)
    keep = torch.empty(max(n, 1), dtype=U8, device=dev)
    cnt = torch.zeros(max(n_seg, 1), dtype=I32, device=dev)
    zd = torch.zeros(1, dtype=I32, device=dev)
    check(lib().iamx_match_metric(_ptr(d2), _ptr(seg), n_seg, float(thresh), _ptr(metric),
                                  _ptr(keep), _ptr(cnt), _ptr(zd), stream_ptr()),
          'iamx_match_metric')
    return metric[:n], keep[:n], cnt[:n_seg], zd


def exclusive_scan(counts):
    dev = require_gpu()
    n = counts.numel()
    out = torch.empty(n + 1, dtype=I64, device=dev)
    check(lib().iamx_exclusive_scan_i32(_ptr(counts), n, _ptr(out), stream_ptr()),
          'iamx_exclusive_scan_i32')
    return out


def match_compact(idx, metric, keep, seg_off, surv_off, total):
    dev = require_gpu()
    seg = _dev(seg_off, I64)
    n_seg = seg.numel() - 1
    sq = torch.empty(max(total, 1), dtype=I32, device=dev)
    stt = torch.empty(max(total, 1), dtype=I32, device=dev)
    sm = torch.empty(max(total, 1), dtype=F64, device=dev)
    check(lib().iamx_match_compact(_ptr(idx), 2, _ptr(metric), _ptr(keep), _ptr(seg), _ptr(surv_off),
                                   n_seg, _ptr(sq), _ptr(stt), _ptr(sm), stream_ptr()),
          'iamx_match_compact')
    return sq[:total], stt[:total], sm[:total]


# --------------------------------------------------------------------------------------
# bundle adjustment
# --------------------------------------------------------------------------------------
def ba_residual(cams, pts, cam_idx, pt_idx, uv, calib, out=None):
    """All arguments device tensors (f64 / i32).  Returns r [2*n_obs] f64."""
    dev = require_gpu()
    n_obs = cam_idx.numel()
    r = out if out is not None else torch.empty(max(2 * n_obs, 1), dtype=F64, device=dev)
    check(lib().iamx_ba_residual(_ptr(cams), cams.numel() // 7, _ptr(pts), pts.numel() // 3,
                                 _ptr(cam_idx), _ptr(pt_idx), _ptr(uv), n_obs, _ptr(calib),
                                 _ptr(r), stream_ptr()), 'iamx_ba_residual')
    return r[:2 * n_obs]


def ba_residual_jac(cams, pts, cam_idx, pt_idx, uv, calib, with_calib=False):
    dev = require_gpu()
    n_obs = cam_idx.numel()
    r = torch.empty(max(2 * n_obs, 1), dtype=F64, device=dev)
    Jc = torch.empty((max(n_obs, 1), 2, 7), dtype=F64, device=dev)
    Jp = torch.empty((max(n_obs, 1), 2, 3), dtype=F64, device=dev)
    Jk = torch.empty((max(n_obs, 1), 2, 8), dtype=F64, device=dev) if with_calib else None
    check(lib().iamx_ba_residual_jac(_ptr(cams), cams.numel() // 7, _ptr(pts), pts.numel() // 3,
                                     _ptr(cam_idx), _ptr(pt_idx), _ptr(uv), n_obs, _ptr(calib),
                                     _ptr(r), _ptr(Jc), _ptr(Jp), _ptr(Jk), stream_ptr()),
          'iamx_ba_residual_jac')
    return r[:2 * n_obs], Jc[:n_obs], Jp[:n_obs], (Jk[:n_obs] if with_calib else None)


# --------------------------------------------------------------------------------------
# a reusable, allocation-free batch of ordered pairs: knn2 -> metric -> scan -> compaction
# --------------------------------------------------------------------------------------
class PairWorkspace(object):
    """Output buffers for up to `max_rows` query rows / `max_pairs` ordered pairs."""

    def __init__(self, max_rows, max_pairs):
        dev = require_gpu()
        r, p = max(int(max_rows), 1), max(int(max_pairs), 1)
        self.max_rows, self.max_pairs = r, p
        self.idx = torch.empty((r, 2), dtype=I32, device=dev)
        self.d2 = torch.empty((r, 2), dtype=I32, device=dev)
        self.metric = torch.empty(r, dtype=F64, device=dev)
        self.keep = torch.empty(max(r, 8), dtype=U8, device=dev)   # (the symmetric form's bit map: whole words)
        self.seg_count = torch.zeros(p, dtype=I32, device=dev)
        self.surv_off = torch.zeros(p + 1, dtype=I64, device=dev)
        self.surv_q = torch.empty(r, dtype=I32, device=dev)
        self.surv_t = torch.empty(r, dtype=I32, device=dev)
        self.surv_metric = torch.empty(r, dtype=F64, device=dev)
        # [0] a zero second distance was met (ZeroDivisionError of matcher.py:255), [1] rows the
        # bound form could not resolve exactly (must stay 0: find_matches refuses the batch)
        self.flags = torch.zeros(2, dtype=I32, device=dev)
        self.zero_div = self.flags[0:1]
        self.tile = torch.empty(r, dtype=I32, device=dev)
        self.unresolved = self.flags[1:2]
        self.surv_cnt = torch.zeros(p, dtype=I32, device=dev)
        # symmetric sweep: bounds per row (allocated on first use, grown when a batch needs more)
        self.col = self.rowp = self.colmask = self.nar = None
        self.cand_keep = torch.empty(r, dtype=U8, device=dev)
        # exact stage of the symmetric form: two task lists in one buffer (wave tasks from entry
        # 0, workgroup tasks from entry n_pairs) and their two counters
        self.task_total = torch.zeros(2, dtype=I32, device=dev)
        self.tasks = torch.empty((2 * p + r // 32 + 2, 2), dtype=I32, device=dev)

    def ensure_sym(self, col_rows, rowp_rows):
        dev = self.d2.device
        if self.col is None or self.col.shape[0] < col_rows:
            self.col = torch.empty((max(col_rows, 1), 2), dtype=I32, device=dev)
            # which of its eight train-row groups can hold a query's best / second (narrow exact stage)
            self.colmask = torch.empty(max(col_rows, 1), dtype=U8, device=dev)
        if self.nar is None:
            # items / tasks / partial results of the narrow exact stage, sized for the workspace's
            # row and pair capacity (its control words start at 0; the stage leaves them 0)
            self.nar = torch.zeros(max(int(lib().iamx_knn2sym_narrow_bytes(self.max_rows, self.max_pairs)),
                                       256), dtype=U8, device=dev)
        if self.rowp is None or self.rowp.shape[0] < rowp_rows:
            self.rowp = torch.empty((max(rowp_rows, 1), 2), dtype=I32, device=dev)

    def survivor_counts(self, n_pairs):
        """host copies of (first, count) per pair only"""
        first = self.surv_off[:n_pairs].cpu().numpy()
        count = self.surv_cnt[:n_pairs].cpu().numpy().astype(np.int64)
        return first, count

    def survivors(self, n_pairs):
        """host copies: (first, count) per pair and the survivor arrays they index
        (pair p owns [first[p], first[p] + count[p]); query rows ascending)."""
        first, count = self.survivor_counts(n_pairs)
        total = int(self.surv_off[n_pairs].item())
        return (first, count, self.surv_q[:total].cpu().numpy(),
                self.surv_t[:total].cpu().numpy(), self.surv_metric[:total].cpu().numpy())


class UploadArena(object):
    """Small host tables -> device without stalling the stream.  A pageable `tensor.to(device)`
    blocks the host until everything already enqueued has run -- in find_matches' software
    pipeline that is the previous round's 4 ms of kernels, per table, a dozen tables per round.
    The arena packs all tables of a round into ONE page-locked buffer and issues ONE asynchronous
    copy into a device buffer of a small ring (a slot is reused after the event behind its copy);
    the tensors it hands out are views of that device buffer, valid in stream order."""

    def __init__(self, nbytes=8 << 20, slots=4):
        dev = require_gpu()
        self.dev = dev
        self.slots = [dict(pin=torch.empty(nbytes, dtype=U8).pin_memory(),
                           dev=torch.empty(nbytes, dtype=U8, device=dev),
                           ev=torch.cuda.Event(), used=False) for _ in range(slots)]
        self.cur, self.off, self.k = None, 0, 0

    def begin(self):
        if self.cur is not None:
            self.commit()
        self.k = (self.k + 1) % len(self.slots)
        self.cur = self.slots[self.k]
        if self.cur['used']:
            self.cur['ev'].synchronize()
        self.off = 0

    def put(self, a):
        """numpy array -> device tensor (same dtype / shape) behind the arena's next commit()"""
        a = np.ascontiguousarray(a)
        nb = a.nbytes
        off = (self.off + 15) & ~15
        if self.cur is None or off + nb > self.cur['pin'].numel():
            return torch.from_numpy(a).to(self.dev)            # (does not fit: the plain way)
        if nb:
            self.cur['pin'].numpy()[off:off + nb] = a.reshape(-1).view(np.uint8)
        self.off = off + nb
        t = self.cur['dev'][off:off + nb].view(_TORCH_OF[a.dtype.str]) if nb else \
            torch.empty(0, dtype=_TORCH_OF[a.dtype.str], device=self.dev)
        return t.view(a.shape) if a.ndim != 1 else t

    def commit(self):
        c = self.cur
        if c is None:
            return
        if self.off:
            c['dev'][:self.off].copy_(c['pin'][:self.off], non_blocking=True)
        c['ev'].record()
        c['used'] = True
        self.cur = None


_TORCH_OF = {'<i4': torch.int32, '<i8': torch.int64, '<f4': torch.float32, '<f8': torch.float64,
             '|u1': torch.uint8, '|i1': torch.int8}


class PairBatch(object):
    """Device-side launch tables of one batch of ordered (query image, train image) pairs.
    `run()` only enqueues kernels on the current stream (no host sync, no allocation)."""

    def __init__(self, store, pairs, sym=None, arena=None):
        """sym: None = use the symmetric sweep (one MFMA pass for both directions of an image
        pair) when the batch holds both directions of every pair, False = never (one sweep per
        ordered pair), True = require it.  arena: an UploadArena between begin() and commit() --
        the launch tables go up with its one asynchronous copy."""
        dev = require_gpu()
        up = arena.put if arena is not None else (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        self._up = up
        self.store = store
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        counts = np.asarray(store.counts, np.int64)
        self.pairs = pairs
        self.n_pairs = P = len(pairs)
        nq = counts[pairs[:, 0]] if P else np.zeros(0, np.int64)
        if P and counts[pairs[:, 1]].min() < 2:
            raise ValueError("train image with fewer than 2 descriptors")
        self.out_off = np.zeros(P + 1, np.int64)
        np.cumsum(nq, out=self.out_off[1:])
        wg = np.zeros(P + 1, np.int64)
        np.cumsum((nq + 255) // 256, out=wg[1:])
        if wg[-1] >= 2 ** 31:
            raise ValueError("batch too large: split the pair list")
        self.rows = int(self.out_off[-1])
        self.max_query_rows = int(nq.max()) if P else 0
        self.total_wg = int(wg[-1])
        self.d_pairs = up(pairs)
        self.d_wg = up(wg.astype(np.int32))
        self.d_out = up(self.out_off.copy())                           # also the metric seg_off
        # query rows per workgroup of the fast sweep (256 / 512 / 1024 by image size): bigger =
        # fewer LDS reads and barriers per MFMA
        self.fast_rows = 256 if not (P and nq.min() >= 2048) else (1024 if nq.min() >= 4096 else 512)
        wgf = np.zeros(P + 1, np.int64)
        np.cumsum((nq + self.fast_rows - 1) // self.fast_rows, out=wgf[1:])
        self.total_wg_fast = int(wgf[-1])
        self.d_wg_fast = up(wgf.astype(np.int32))
        self.sym = False
        if sym is not False:
            self._setup_sym(dev)
            if sym and not self.sym:
                raise ValueError("symmetric sweep needs both directions of every pair (i != j) "
                                 "in the batch")

    def _setup_sym(self, dev):
        t = sym_tables(self.pairs, self.store.counts, self.store.caps3)
        if t is None:
            return
        assert SYM_ROWS_PER_WG[t['form']] == int(lib().iamx_knn2sym_rows_per_wg(t['form']))
        self.sym = True
        self.sym_form = t['form']
        self.n_u, self.sym_total_wg = len(t['upairs']), int(t['wg'][-1])
        self.sym_col_rows, self.sym_rowp_rows = int(t['col_off'][-1]), int(t['rowp_off'][-1])
        up = self._up
        self.d_upairs = up(np.ascontiguousarray(t['upairs']))
        self.d_sym_wg = up(t['wg'].astype(np.int32))
        self.d_col_off = up(t['col_off'][:-1].copy())
        self.d_rowp_off = up(t['rowp_off'][:-1].copy())
        self.d_osrc = up(t['osrc'])

    def run_sym_sweep(self, ws):
        st = self.store
        ws.ensure_sym(self.sym_col_rows, self.sym_rowp_rows)
        check(lib().iamx_knn2sym_sweep(_ptr(st.desc3), _ptr(st.sn2), _ptr(st.sct), _ptr(st.img_off3),
                                       _ptr(st.img_n), _ptr(self.d_upairs), _ptr(self.d_sym_wg),
                                       _ptr(self.d_col_off), _ptr(self.d_rowp_off), self.n_u,
                                       self.sym_total_wg, self.sym_form, _ptr(ws.col), _ptr(ws.rowp),
                                       _ptr(ws.colmask), stream_ptr()), 'iamx_knn2sym_sweep')

    def run_sym_filter(self, ws, thresh):
        """candidate rows from the sweep's bounds, exact top-2 + metric test for those rows,
        survivors compacted in place (pair p: surv_*[surv_off[p] .. + surv_cnt[p]))"""
        L, s, st = lib(), stream_ptr(), self.store
        check(L.iamx_knn2sym_candidates(_ptr(st.sn2), _ptr(st.sperm), _ptr(st.img_off3), _ptr(st.img_n),
                                        _ptr(self.d_pairs), _ptr(self.d_osrc), _ptr(self.d_sym_wg),
                                        _ptr(self.d_col_off), _ptr(self.d_rowp_off), _ptr(self.d_out),
                                        _ptr(ws.col), _ptr(ws.rowp), self.n_pairs, float(thresh),
                                        _ptr(ws.keep), _ptr(ws.seg_count), _ptr(ws.surv_q),
                                        _ptr(ws.task_total), _ptr(ws.tasks), _ptr(ws.d2),
                                        _ptr(ws.colmask), _ptr(ws.nar), ws.max_rows, self.max_query_rows,
                                        self.sym_form, s),
              'iamx_knn2sym_candidates')
        check(L.iamx_knn2sym_exact(_ptr(st.desc), _ptr(st.norm_q), _ptr(st.norm_t), _ptr(st.key_t),
                                   st.rows_used, _ptr(st.img_off),
                                   _ptr(st.img_n), _ptr(self.d_pairs), _ptr(self.d_out),
                                   _ptr(ws.seg_count), _ptr(ws.task_total), _ptr(ws.tasks),
                                   _ptr(ws.surv_q), self.n_pairs, float(thresh), _ptr(ws.d2),
                                   _ptr(ws.surv_t), _ptr(ws.surv_metric), _ptr(ws.cand_keep),
                                   _ptr(ws.surv_cnt), _ptr(ws.zero_div), _ptr(st.desc3), _ptr(st.sn2),
                                   _ptr(st.sct), _ptr(st.sperm), _ptr(st.img_off3), _ptr(self.d_osrc),
                                   _ptr(ws.nar), ws.max_rows, self.sym_form, s), 'iamx_knn2sym_exact')
        # pair p's survivors sit at the start of its own row range
        ws.surv_off[:self.n_pairs + 1].copy_(self.d_out, non_blocking=True)

    def run_knn2(self, ws):
        st = self.store
        check(lib().iamx_knn2_l2_pairs(_ptr(st.desc), _ptr(st.norm_q), _ptr(st.norm_t),
                                       _ptr(st.img_off), _ptr(st.img_n), _ptr(self.d_pairs),
                                       _ptr(self.d_wg), _ptr(self.d_out), self.n_pairs,
                                       self.total_wg, _ptr(ws.idx), _ptr(ws.d2), stream_ptr()),
              'iamx_knn2_l2_pairs')

    def run_filter(self, ws, thresh):
        L, s = lib(), stream_ptr()
        check(L.iamx_match_metric(_ptr(ws.d2), _ptr(self.d_out), self.n_pairs, float(thresh),
                                  _ptr(ws.metric), _ptr(ws.keep), _ptr(ws.seg_count),
                                  _ptr(ws.zero_div), s), 'iamx_match_metric')
        check(L.iamx_exclusive_scan_i32(_ptr(ws.seg_count), self.n_pairs, _ptr(ws.surv_off), s),
              'iamx_exclusive_scan_i32')
        check(L.iamx_match_compact(_ptr(ws.idx), 2, _ptr(ws.metric), _ptr(ws.keep),
                                   _ptr(self.d_out), _ptr(ws.surv_off), self.n_pairs,
                                   _ptr(ws.surv_q), _ptr(ws.surv_t), _ptr(ws.surv_metric), s),
              'iamx_match_compact')

    # ---- fast form: distances + tile in the sweep, train index only for the survivors
    def run_knn2_fast(self, ws, exact_second=False):
        st = self.store
        if self.sym and not exact_second:
            return self.run_sym_sweep(ws)
        if not st.has_train_layout:
            # (no parity-partitioned copy in this store: the general kernel, exact as well)
            return self.run_knn2(ws)
        if exact_second and self.fast_rows == 1024:          # the exact-second form stops at 512
            self._use_rows(512)
        check(lib().iamx_knn2v2_pairs(_ptr(st.desc), _ptr(st.norm_q), _ptr(st.img_off),
                                      _ptr(st.img_n), _ptr(st.desc2), _ptr(st.cinit),
                                      _ptr(st.img_off2), _ptr(st.meta), _ptr(self.d_pairs),
                                      _ptr(self.d_wg_fast), _ptr(self.d_out), self.n_pairs,
                                      self.total_wg_fast, self.fast_rows, 1 if exact_second else 0,
                                      _ptr(ws.d2), _ptr(ws.tile), stream_ptr()),
              'iamx_knn2v2_pairs')

    def _use_rows(self, rows):
        nq = np.asarray(self.store.counts, np.int64)[self.pairs[:, 0]]
        wgf = np.zeros(self.n_pairs + 1, np.int64)
        np.cumsum((nq + rows - 1) // rows, out=wgf[1:])
        self.fast_rows, self.total_wg_fast = rows, int(wgf[-1])
        self.d_wg_fast = torch.from_numpy(wgf.astype(np.int32)).to(self.d_pairs.device)

    def run_filter_fast(self, ws, thresh, exact_second=False):
        """threshold + compaction + train rows.  With the bound form of the sweep the first
        threshold keeps a superset; iamx_knn2v2_finish makes it exact (include/iamx.h)."""
        if self.sym and not exact_second:
            return self.run_sym_filter(ws, thresh)
        if not self.store.has_train_layout:
            self.run_filter(ws, thresh)
            ws.surv_cnt[:self.n_pairs].copy_(ws.seg_count[:self.n_pairs])
            return
        L, s, st = lib(), stream_ptr(), self.store
        check(L.iamx_match_metric(_ptr(ws.d2), _ptr(self.d_out), self.n_pairs, float(thresh),
                                  _ptr(ws.metric), _ptr(ws.keep), _ptr(ws.seg_count),
                                  _ptr(ws.zero_div), s), 'iamx_match_metric')
        check(L.iamx_exclusive_scan_i32(_ptr(ws.seg_count), self.n_pairs, _ptr(ws.surv_off), s),
              'iamx_exclusive_scan_i32')
        check(L.iamx_match_compact(_ptr(ws.tile), 1, _ptr(ws.metric), _ptr(ws.keep),
                                   _ptr(self.d_out), _ptr(ws.surv_off), self.n_pairs,
                                   _ptr(ws.surv_q), _ptr(ws.surv_t), _ptr(ws.surv_metric), s),
              'iamx_match_compact')
        if exact_second:
            check(L.iamx_knn2v2_resolve(_ptr(st.desc), _ptr(st.norm_q), _ptr(st.img_off),
                                        _ptr(st.desc2), _ptr(st.norm2), _ptr(st.perm),
                                        _ptr(st.img_off2), _ptr(self.d_pairs), _ptr(self.d_out),
                                        _ptr(ws.d2), _ptr(ws.surv_off), _ptr(ws.surv_q),
                                        _ptr(ws.surv_t), self.n_pairs, _ptr(ws.unresolved), s),
                  'iamx_knn2v2_resolve')
            ws.surv_cnt[:self.n_pairs].copy_(ws.seg_count[:self.n_pairs])
        else:
            check(L.iamx_knn2v2_finish(_ptr(st.desc), _ptr(st.norm_q), _ptr(st.img_off),
                                       _ptr(st.desc2), _ptr(st.norm2), _ptr(st.perm),
                                       _ptr(st.img_off2), _ptr(self.d_pairs), _ptr(self.d_out),
                                       _ptr(ws.d2), float(thresh), _ptr(ws.surv_off),
                                       _ptr(ws.surv_q), _ptr(ws.surv_t), _ptr(ws.surv_metric),
                                       _ptr(ws.surv_cnt), self.n_pairs, _ptr(ws.zero_div),
                                       _ptr(ws.unresolved), s), 'iamx_knn2v2_finish')

    def run(self, ws, thresh, fast=True):
        """enqueue top-2 + metric threshold + survivor compaction (+ index resolve).
        fast: True = the shipped form (symmetric sweep when the batch holds both directions of
        every pair, else the one-direction bound form), 'exact' = exact-second fast sweep,
        False = the general kernel that tracks indices in the sweep."""
        if self.rows > ws.max_rows or self.n_pairs > ws.max_pairs:
            raise ValueError("workspace too small for this batch")
        if self.n_pairs == 0 or self.rows == 0:
            return
        if fast:
            exact = fast == 'exact'
            self.run_knn2_fast(ws, exact_second=exact)
            self.run_filter_fast(ws, thresh, exact_second=exact)
        else:
            self.run_knn2(ws)
            self.run_filter(ws, thresh)
            ws.surv_cnt[:self.n_pairs].copy_(ws.seg_count[:self.n_pairs])


# --------------------------------------------------------------------------------------
# SIFT
# --------------------------------------------------------------------------------------
_sift_ws = {}
_sift_out = {}
_prep_ws = {}

# Detector slots: the workspace / output / staging caches above are keyed by (device, slot).  Slot 0
# belongs to whoever calls without asking (one thread at a time, as before); worker threads that
# detect concurrently take one of DETECT_SLOTS numbered slots each (image.py runs whole detections
# on its prefetch workers): a slot is ~1 GB of workspace for a 3 MP detect image (8 of them keep the GPU fed), so the number of
# detections in flight is bounded by the slots, not by the number of threads.
DETECT_SLOTS = 8
_slot_tls = threading.local()
_slot_free = None
_slot_lock = threading.Lock()


def _slot_key(dev):
    return (dev.index, getattr(_slot_tls, 'slot', 0))


def wait_stream(polite=None):
    """Wait for the current stream.  The HIP runtime spins on the host while it waits
    (hipDeviceScheduleAuto: one core at 100 % per waiting thread, measured with
    torch.cuda.Event(blocking=True) as well); a thread that holds a detector slot -- or says
    polite=True -- polls an event with short sleeps instead: +0.2 ms of latency, no core burnt.
    (Two dozen prefetch workers waiting for uploads and detections cost more CPU than the JPEG
    Huffman decoding and the cache compression together on a host with a 16-core quota.)"""
    if polite is None:
        polite = getattr(_slot_tls, 'slot', 0) != 0 or getattr(_slot_tls, 'polite', False)
    if not polite:
        torch.cuda.current_stream().synchronize()
        return
    import time
    ev = torch.cuda.Event()
    ev.record()
    pause = 0.0001
    while not ev.query():
        time.sleep(pause)
        pause = min(pause * 1.5, 0.001)


class polite_waits(object):
    """with polite_waits(): waits of this thread inside the package's calls sleep instead of spin"""

    def __enter__(self):
        self.prev = getattr(_slot_tls, 'polite', False)
        _slot_tls.polite = True
        return self

    def __exit__(self, *exc):
        _slot_tls.polite = self.prev
        return False


class detector_slot(object):
    """with detector_slot(): ... -- this thread's sift_detect / equalize_resize calls use a
    private set of device buffers (blocks while all DETECT_SLOTS are taken)"""

    def __enter__(self):
        global _slot_free
        with _slot_lock:
            if _slot_free is None:
                import queue
                _slot_free = queue.Queue()
                for k in range(1, DETECT_SLOTS + 1):
                    _slot_free.put(k)
        self.slot = _slot_free.get()
        self.prev = getattr(_slot_tls, 'slot', 0)
        _slot_tls.slot = self.slot
        return self

    def __exit__(self, *exc):
        _slot_tls.slot = self.prev
        _slot_free.put(self.slot)
        return False


class OverlappedSweeps(object):
    """Runs a sequence of PairBatch launches with the small threshold / compaction / finish
    kernels of launch k on a second stream, beside the sweep of launch k+1 (two workspaces,
    events both ways).  The sweep is ~95 % of a launch and fills the machine; the filter kernels
    are latency bound and fit into its tail -- 6 % more pairs/s at BASELINE configs[1]."""

    def __init__(self, max_rows, max_pairs, first_workspace=None):
        self.ws = [first_workspace or PairWorkspace(max_rows, max_pairs),
                   PairWorkspace(max_rows, max_pairs)]
        self.side = torch.cuda.Stream()
        self.swept = [torch.cuda.Event(), torch.cuda.Event()]
        self.filtered = [torch.cuda.Event(), torch.cuda.Event()]

    def run(self, batches, thresh, after_filter=None, sweep_events=None):
        """after_filter(batch, workspace) is called with the side stream current (enqueue only);
        sweep_events: optional [(start, stop)] timing events, one pair per batch.  On return
        the current stream has waited for everything."""
        main = torch.cuda.current_stream()
        for k, b in enumerate(batches):
            w = self.ws[k & 1]
            if k >= 2:
                main.wait_event(self.filtered[k & 1])        # that workspace is free again
            if sweep_events is not None:
                sweep_events[k][0].record()
            b.run_knn2_fast(w)
            if sweep_events is not None:
                sweep_events[k][1].record()
            self.swept[k & 1].record(main)
            self.side.wait_event(self.swept[k & 1])
            with torch.cuda.stream(self.side):
                b.run_filter_fast(w, thresh)
                if after_filter is not None:
                    after_filter(b, w)
                self.filtered[k & 1].record(self.side)
        main.wait_stream(self.side)


def sift_detect(image, cap=200000, contrast_threshold=0.04, edge_threshold=10.0, sigma=1.6,
                order='opencv'):
    """cv2.SIFT_create().detectAndCompute(image, None): image [h,w,3] BGR or [h,w] gray uint8
    (numpy or device tensor).  Returns numpy (kp [N,5] float32: x, y, size, angle, response;
    octave [N] int32 (cv2 packing); desc [N,128] uint8), duplicates removed, in OpenCV's output
    order (KeyPointsFilter::removeDuplicatedSorted) -- or, order='canonical', in the pyramid-local
    (octave, layer, y, x, angle) order."""
    dev = require_gpu()
    img = _dev(image, U8)
    if img.dim() == 2:
        h, w, ch = img.shape[0], img.shape[1], 1
    else:
        h, w, ch = img.shape
    need = int(lib().iamx_sift_workspace_bytes(h, w))
    ws = _sift_ws.get(_slot_key(dev))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=U8, device=dev)
        _sift_ws[_slot_key(dev)] = ws
    ob = _sift_out.get(_slot_key(dev))                  # (allocations cost ~0.3 ms each per frame)
    if ob is None or ob[0].shape[0] < cap:
        ob = (torch.empty((cap, 8), dtype=torch.float32, device=dev),
              torch.empty((cap, 128), dtype=U8, device=dev), torch.zeros(1, dtype=I32, device=dev))
        _sift_out[_slot_key(dev)] = ob
    kp, desc, n = ob[0][:cap], ob[1][:cap], ob[2]
    check(lib().iamx_sift_detect(_ptr(img), h, w, ch, contrast_threshold, edge_threshold, sigma,
                                 _ptr(ws), need, _ptr(kp), _ptr(desc), cap, _ptr(n), stream_ptr()),
          'iamx_sift_detect')
    # removeDuplicatedSorted on the device (the kernels append in a nondeterministic order, with
    # duplicates): iamx_sift_sort, then one pinned download
    sw = _sift_sort_ws.get(_slot_key(dev))
    need_s = int(lib().iamx_sift_sort_workspace_bytes(cap))
    if sw is None or sw[0].numel() < need_s or sw[1].shape[0] < cap:
        sw = (torch.empty(need_s, dtype=U8, device=dev),
              torch.empty((cap, 8), dtype=torch.float32, device=dev),
              torch.empty((cap, 128), dtype=U8, device=dev),
              torch.zeros(2, dtype=I32, device=dev))
        _sift_sort_ws[_slot_key(dev)] = sw
    if order not in ('opencv', 'canonical'):
        raise ValueError("sift_detect: order must be 'opencv' or 'canonical'")
    check(lib().iamx_sift_sort(_ptr(kp), _ptr(desc), _ptr(n), cap, 1 if order == 'opencv' else 0,
                               w, h, _ptr(sw[0]), need_s, _ptr(sw[1]), _ptr(sw[2]), _ptr(sw[3]),
                               stream_ptr()), 'iamx_sift_sort')
    sw[3][1:2].copy_(n, non_blocking=True)                # [kept, appended]
    hn = _sift_pinned_count(dev)
    hn.copy_(sw[3], non_blocking=True)
    wait_stream()
    cnt, found = int(hn[0]), int(hn[1])
    if found > cap:
        raise _lib.IamxError("sift_detect: %d keypoints exceed the capacity %d" % (found, cap))
    sift_detect.last_removed = found - cnt               # duplicates dropped (diagnosis / tests)
    hk, hd = _sift_pinned(dev, cnt)
    hk[:cnt].copy_(sw[1][:cnt], non_blocking=True)
    hd[:cnt].copy_(sw[2][:cnt], non_blocking=True)
    wait_stream()
    k = hk[:cnt].numpy()
    return k[:, :5].copy(), k[:, 5].copy().view(np.int32), hd[:cnt].numpy().copy()


_sift_sort_ws = {}
_sift_pin = {}


def _sift_pinned_count(dev):
    cur = _sift_pin.get((_slot_key(dev), 'n'))
    if cur is None:
        cur = _sift_pin[(_slot_key(dev), 'n')] = torch.zeros(2, dtype=I32).pin_memory()
    return cur


def _sift_pinned(dev, n):
    """page-locked staging for the keypoint / descriptor download (grown geometrically)"""
    cur = _sift_pin.get(_slot_key(dev))
    if cur is None or cur[0].shape[0] < n:
        m = max(1 << 16, 1 << int(n - 1).bit_length())
        cur = (torch.empty((m, 8), dtype=torch.float32).pin_memory(),
               torch.empty((m, 128), dtype=U8).pin_memory())
        _sift_pin[_slot_key(dev)] = cur
    return cur


# ---- split JPEG decoder (csrc/jpeg.hip): Huffman decode on the host, the rest on the device --------
class JpegCoefficients(object):
    """A JPEG after the host half of the decoder: info (int32 [16], iamx_jpeg_info), the
    quantised coefficients in a page-locked buffer (int16 [blocks, 64]) and the quantisation
    tables (uint16 [3, 64]).  release() hands the buffer back to the pool."""
    __slots__ = ('info', 'coef', 'quant', '_slot')

    def release(self):
        if self._slot is not None:
            with _jpeg_lock:
                _jpeg_pool.append(self._slot)
            self._slot = None
            self.coef = None


_jpeg_pool = []          # free page-locked coefficient buffers (torch int16, flat)
_jpeg_lock = __import__('threading').Lock()


def _jpeg_buffer(n_values):
    with _jpeg_lock:
        for k, b in enumerate(_jpeg_pool):
            if b.numel() >= n_values:
                return _jpeg_pool.pop(k)
        if len(_jpeg_pool) >= 8:             # (a survey's frames are all the same size)
            _jpeg_pool.pop(0)
    return torch.empty(int(n_values), dtype=torch.int16).pin_memory()


def jpeg_host_decode(source):
    """The host half: file name or bytes -> JpegCoefficients, or None for a file the split
    decoder does not handle (progressive, arithmetic, CMYK, ... -- the caller reads those the
    host way).  No device involved; the C call releases the GIL (worker threads decode in
    parallel).  Raises IamxError for a broken file."""
    import ctypes
    if isinstance(source, (bytes, bytearray, memoryview)):
        data = bytes(source)
    else:
        with open(source, 'rb') as fp:
            data = fp.read()
    raw = np.frombuffer(data, np.uint8)
    L = lib()
    jc = JpegCoefficients()
    jc.info = np.zeros(16, np.int32)
    jc._slot = None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = L.iamx_jpeg_info(p(raw), len(raw), p(jc.info))
    if rc == -4:
        return None
    check(rc, 'iamx_jpeg_info')
    blocks = int(jc.info[11])
    jc._slot = _jpeg_buffer(blocks * 64)
    jc.coef = jc._slot[:blocks * 64].view(blocks, 64)
    jc.quant = np.zeros((3, 64), np.uint16)
    rc = L.iamx_jpeg_decode_coefficients(p(raw), len(raw), ctypes.c_void_p(jc.coef.data_ptr()), blocks,
                                         p(jc.quant))
    if rc != 0:
        jc.release()
        if rc == -4:
            return None
        check(rc, 'iamx_jpeg_decode_coefficients')
    return jc


def jpeg_reconstruct(jc, release=True):
    """The device half: JpegCoefficients -> BGR uint8 [h, w, 3] in HBM, bit-identical to
    libjpeg-turbo's output (dequantisation, islow IDCT, fancy upsampling, YCbCr -> BGR)."""
    import ctypes
    dev = require_gpu()
    L = lib()
    info = jc.info
    w, h = int(info[0]), int(info[1])
    # (the 384-byte table first: a pageable copy waits for what is in front of it on the stream)
    d_quant = torch.from_numpy(jc.quant.astype(np.int16)).to(dev)        # (bit pattern; read as uint16)
    d_coef = torch.empty(jc.coef.shape, dtype=torch.int16, device=dev)
    d_coef.copy_(jc.coef, non_blocking=True)
    need = int(L.iamx_jpeg_workspace_bytes(info.ctypes.data_as(ctypes.c_void_p)))
    ws = torch.empty(need, dtype=U8, device=dev)
    out = torch.empty((h, w, 3), dtype=U8, device=dev)
    check(L.iamx_jpeg_reconstruct(_ptr(d_coef), _ptr(d_quant), info.ctypes.data_as(ctypes.c_void_p),
                                  _ptr(ws), need, _ptr(out), stream_ptr()), 'iamx_jpeg_reconstruct')
    if release:
        wait_stream()                                    # the page-locked buffer has been read
        jc.release()
    return out


def jpeg_decode(source):
    """file name or bytes -> BGR uint8 [h, w, 3] on the device, or None for an unsupported file"""
    jc = jpeg_host_decode(source)
    return None if jc is None else jpeg_reconstruct(jc)


def equalize_resize(bgr, scale, equalize=True, clip_limit=3.0):
    """Image.load_rgb(equalize=True) + cv2.resize(fx=fy=scale) on the device.
    bgr [h,w,3] uint8 (numpy or device) -> device tensor [round(h*s), round(w*s), 3] uint8."""
    dev = require_gpu()
    img = _dev(bgr, U8)
    h, w, ch = img.shape
    if ch != 3:
        raise ValueError("expected a BGR image")
    import ctypes
    oh, ow = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().iamx_image_resized_dims(h, w, float(scale), ctypes.byref(oh), ctypes.byref(ow)),
          'iamx_image_resized_dims')
    need = int(lib().iamx_image_prep_workspace_bytes(h, w))
    ws = _prep_ws.get(_slot_key(dev))
    if ws is None or ws.numel() < need:
        ws = _prep_ws[_slot_key(dev)] = torch.empty(need, dtype=U8, device=dev)
    out = torch.empty((oh.value, ow.value, 3), dtype=U8, device=dev)
    check(lib().iamx_image_equalize_resize(_ptr(img), h, w, 1 if equalize else 0, float(clip_limit),
                                           float(scale), _ptr(ws), need, _ptr(out), stream_ptr()),
          'iamx_image_equalize_resize')
    if not isinstance(bgr, torch.Tensor):
        torch.cuda.current_stream().synchronize()      # (the upload's staging copy has been read)
    return out
