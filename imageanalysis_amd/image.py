"""MI355X-native stand-in for the feature part of the reference's scripts/lib/image.py:

    Image.load_rgb(equalize)                       image.py:99-121   (decode + CLAHE on HSV-V)
    Image.detect_features(scale, use_cache=True)   image.py:287-350  (resize + SIFT + cache)
    load_/save_features, _descriptors, _matches    image.py:140-228  (same on-disk formats)

JPEG decoding stays on the host (Pillow; the reference uses cv2.imread); CLAHE, resize, the
SIFT pyramid, keypoints and descriptors run on the GPU (csrc/image_prep.hip, csrc/sift.hip).
Cache files are byte-compatible with the reference's: `.feat` = gzip(level 6) pickle of
[((x, y), size, angle, response, octave, class_id), ...] in FULL-RES pixels, `.desc` =
gzip(level 6) of np.save(float32 [N,128]), `.match` = pickle {other_name: [[i, j], ...]}.

Use `install(lib.image)` to give the reference's own Image class these methods (drop-in), or
the stand-alone `Image` class below.
"""
import gzip
import io
import os
import pickle
import sys

import numpy as np

from . import _deps, cacheio
from .matchpairs import dump_match_dict
from .hostlib.image_pose import PoseImage
from .keypoints import KeyPointList

try:                                     # inside the reference environment keep cv2's type
    from cv2 import KeyPoint as _CvKeyPoint
except Exception:                        # noqa: BLE001
    _CvKeyPoint = None


class KeyPoint(object):
    """the fields of cv2.KeyPoint the pipeline reads and caches (image.py:192-198)"""
    __slots__ = ('pt', 'size', 'angle', 'response', 'octave', 'class_id')

    def __init__(self, x=0.0, y=0.0, size=0.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        f32 = np.float32                 # cv2.KeyPoint stores float32 members
        self.pt = (float(f32(x)), float(f32(y)))
        self.size = float(f32(size))
        self.angle = float(f32(angle))
        self.response = float(f32(response))
        self.octave = int(octave)
        self.class_id = int(class_id)


def make_keypoint(x, y, size, angle, response, octave, class_id=-1):
    if _CvKeyPoint is not None:
        return _CvKeyPoint(x=float(x), y=float(y), size=float(size), angle=float(angle),
                           response=float(response), octave=int(octave), class_id=int(class_id))
    return KeyPoint(x, y, size, angle, response, octave, class_id)


def make_keypoints(x, y, size, angle, response, octave, class_id=None):
    """column arrays -> list of keypoints; the float32 rounding cv2.KeyPoint applies to its
    members is done once per column instead of five times per object (50 k keypoints per
    frame: the per-object constructor costs more than the whole GPU detector)"""
    n = len(x)
    if class_id is None:
        class_id = np.full(n, -1, np.int64)
    if _CvKeyPoint is not None:
        return [make_keypoint(*t) for t in zip(np.asarray(x).tolist(), np.asarray(y).tolist(),
                                               np.asarray(size).tolist(), np.asarray(angle).tolist(),
                                               np.asarray(response).tolist(),
                                               np.asarray(octave).tolist(),
                                               np.asarray(class_id).tolist())]
    f32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64).tolist()
    new, cls = KeyPoint.__new__, KeyPoint
    out = []
    for px, py, s, a, r, o, c in zip(f32(x), f32(y), f32(size), f32(angle), f32(response),
                                     np.asarray(octave, np.int64).tolist(),
                                     np.asarray(class_id, np.int64).tolist()):
        k = new(cls)
        k.pt = (px, py)
        k.size = s
        k.angle = a
        k.response = r
        k.octave = o
        k.class_id = c
        out.append(k)
    return out


ASYNC_CACHE_WRITES = True      # cache files are written by background threads (cacheio.wait())
USE_DESC_SIDECAR = True        # <image>.desc.u8.npy: raw uint8 descriptors beside the reference's .desc
USE_DEVICE_JPEG = True         # split decoder: Huffman on the host, IDCT / upsampling / colour on the GPU
SIDECAR_MARGIN_S = 30.0        # a .desc / .feat newer than the sidecar by more than this is not ours
DES_LIST_U8 = False            # True: image.des_list stays uint8 [N,128] (the same integer values; 6 MB
                               # instead of 25 MB per 50 k-keypoint frame -- 47 GB instead of 190 GB of host
                               # memory for 10 000 frames).  The cache files are unchanged (float32 .desc);
                               # only code that needs des_list.dtype == float32 would notice.
WRITE_REFERENCE_DESC = True    # False: skip the 25 MB float32 gzip (only this package reads the cache then)
# zlib level of the float32 .desc (the reference passes compresslevel=6, image.py:213; every level
# decompresses to the same bytes).  On integer-valued float32 descriptors level 6 runs at 9 MB/s
# per core for a 0.32 ratio, level 1 at 50 MB/s for 0.38: the file is what a fresh detection costs
# IAMX_REFERENCE_GZIP=1: both cache files at the reference's own zlib setting (level 6, default
# strategy) -- byte sizes like the reference's, at ~4x the host time per fresh detection
_REF_GZIP = os.environ.get('IAMX_REFERENCE_GZIP') == '1'
DESC_GZIP_LEVEL = 6 if _REF_GZIP else 1
# decoded / cache-loaded images held ahead of the detector (one worker thread each; a 20 MP JPEG
# takes ~0.2 s of one core to decode against ~6 ms on the GPU, 60 MB per decoded frame)
# zlib setting of the .feat members.  The reference asks for compresslevel=6 (image.py:201); on a
# 50 k-keypoint .feat that is 270 ms of one core for 23.3 bytes per keypoint, the largest single host
# cost of a fresh detection (tools/detect_stages.py: the GPU box's 16-core quota was spent on it).
# Level 4 with Z_FILTERED (the records are mostly float bits: few matches, Huffman does the work) is
# 72 ms for 24.9 bytes per keypoint -- the same bytes for every reader.  (6, 0) = the reference's.
# Round 5: the .feat members no longer go through zlib at all.  The pickle is a stream of 58-byte
# records whose only redundancy is "same byte as one record earlier"; libiamx's record encoder
# (iamx_gzip_records: matches at distance 58 only, dynamic Huffman code) writes valid gzip members
# of the REFERENCE's size (28.5 bytes per keypoint, level 6 gives 28.4) in 18 ms of one core where
# level 4 / Z_FILTERED took 99 ms -- more than the Huffman decode of the frame's JPEG (43 ms).
FEAT_GZIP_LEVEL, FEAT_GZIP_STRATEGY = (6, 0) if _REF_GZIP else (4, ('records', 58))
PREFETCH_DEPTH = min(24, max(6, (os.cpu_count() or 8) // 4))


def _log(*a):
    _deps.logger().log(*a)


def _qlog(*a):
    _deps.logger().qlog(*a)


# --------------------------------------------------------------------------------------
# cache I/O -- image.py:140-228
# --------------------------------------------------------------------------------------
def load_features(self):
    cacheio.wait(self.features_file)
    if os.path.exists(self.features_file):
        try:
            with gzip.open(self.features_file, "rb") as fp:
                blob = fp.read()
            # (an array-backed sequence of keypoints: keypoints.py)
            self.kp_list = KeyPointList.from_feat_bytes(blob)
            return True
        except Exception:                 # noqa: BLE001  (the reference prints and carries on)
            print(self.features_file + ":\n" + "  feature load error: "
                  + str(sys.exc_info()[0]) + ": " + str(sys.exc_info()[1]))
    return False


def _sidecar(desc_file):
    """<image>.desc.u8.npy next to the reference's <image>.desc: the same descriptors as raw
    uint8 (SIFT values are integers 0..255): 1/4 of the bytes and no gzip on either side --
    6 MB and a few milliseconds instead of 25 MB float32 through zlib.
    The reference's own readers keep using the .desc file."""
    return desc_file + '.u8.npy'


def _load_sidecar(self):
    """float32 [N,128] from the uint8 sidecar if it is there and not older than the .desc"""
    side = _sidecar(self.desc_file)
    cacheio.wait(side)
    try:
        if not os.path.exists(side):
            return None
        # a .desc / .feat that is newer was written by someone else (the reference
        # re-detecting); our own background gzip of the same arrays finishes seconds after the
        # sidecar (SIDECAR_MARGIN_S covers the slowest writer queue measured, 20 MP frames)
        t_side = os.path.getmtime(side)
        for other in (self.desc_file, self.features_file):
            if os.path.exists(other) and os.path.getmtime(other) > t_side + SIDECAR_MARGIN_S:
                return None
        u8 = np.load(side)
        if u8.dtype != np.uint8 or u8.ndim != 2 or u8.shape[1] != 128:
            return None
        if self.kp_list is not None and len(self.kp_list) and len(self.kp_list) != len(u8):
            return None                                   # not the descriptors of these keypoints
        return u8 if DES_LIST_U8 else u8.astype(np.float32)
    except Exception:                     # noqa: BLE001  (fall back to the .desc file)
        return None


def load_descriptors(self):
    if USE_DESC_SIDECAR and self.des_list is None:
        des = _load_sidecar(self)
        if des is not None:
            self.des_list = des
            return True
    cacheio.wait(self.desc_file)
    if os.path.exists(self.desc_file):
        if self.des_list is None:
            try:
                with gzip.open(self.desc_file, 'rb') as fp:
                    self.des_list = np.load(fp)
                return True
            except Exception:             # noqa: BLE001
                print(self.desc_file + ":\n" + "  desc load error: " + str(sys.exc_info()[1]))
    return False


def load_matches(self):
    try:
        with open(self.match_file, "rb") as fp:
            self.match_list = pickle.load(fp)
        self.matches_clean = True
    except Exception:                     # noqa: BLE001
        print(self.match_file + ":\n" + "  matches load error: "
              + str(sys.exc_info()[0]) + ": " + str(sys.exc_info()[1]))


def save_features(self):
    """same bytes as the reference's gzip.open(..., compresslevel=6) + pickle.dump once
    decompressed; written in the background as a multi-member gzip stream (cacheio)"""
    kps = self.kp_list
    if isinstance(kps, KeyPointList):
        payload = lambda: kps.feat_bytes(as_view=True)   # straight from the columns (in the writer thread)
    else:
        feature_list = [(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id)
                        for kp in kps]
        payload = lambda: pickle.dumps(feature_list)
    # (3 MB at the reference's level 6 is 0.3 s of one core: eight members, eight cores)
    cacheio.write_gzip(self.features_file, payload,
                       background=ASYNC_CACHE_WRITES,
                       on_error=lambda e: print("save_features(): I/O error: %s" % e),
                       member_bytes=384 << 10, level=FEAT_GZIP_LEVEL, strategy=FEAT_GZIP_STRATEGY)


def _npy_bytes(arr):
    """the bytes np.save() writes, as (header, the array's own memory): 25 MB of descriptors are
    not copied (with the GIL held) just to be compressed"""
    arr = np.asarray(arr)
    if arr.dtype.hasobject or not arr.flags.c_contiguous or arr.size == 0:
        buf = io.BytesIO()
        np.save(buf, arr)
        return buf.getbuffer()
    head = io.BytesIO()
    np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(arr))
    return [head.getvalue(), memoryview(arr.reshape(-1).view(np.uint8))]


def _desc_gzip_from_u8(u8):
    """the bytes of the reference's <image>.desc -- gzip(np.save(float32 [N,128])) -- straight
    from the uint8 descriptors (libiamx iamx_gzip_f32_from_u8: the float32 array is 256 different
    4-byte patterns; ~10x faster than zlib level 1 on it, GIL released)"""
    import ctypes
    from . import _lib
    u8 = np.ascontiguousarray(u8, np.uint8)
    head = io.BytesIO()
    np.lib.format.write_array_header_1_0(head, dict(descr='<f4', fortran_order=False, shape=u8.shape))
    header = np.frombuffer(head.getvalue(), np.uint8)
    L = _lib.lib()
    cap = int(L.iamx_gzip_f32_from_u8_bound(len(header), u8.size))
    out = np.empty(cap, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n = L.iamx_gzip_f32_from_u8(p(header), len(header), p(u8) if u8.size else None, u8.size, p(out), cap)
    if n <= 0:
        _lib.check(int(n), 'iamx_gzip_f32_from_u8')
    return memoryview(out)[:int(n)]


def save_descriptors(self):
    des = self.des_list                    # the array as it is now (a later flush drops only the name)
    as_u8 = None
    if des is not None and len(des):
        u8 = np.asarray(des)
        known = getattr(self, '_iamx_des_u8', None)       # (float32 array, its uint8 original)
        if known is not None and known[0] is des:
            as_u8 = known[1]                              # straight from the detector
        elif u8.dtype == np.uint8:
            as_u8 = u8
        else:
            cand = np.clip(np.rint(u8), 0, 255).astype(np.uint8)
            if np.array_equal(cand, u8):                  # integer valued 0..255 only
                as_u8 = cand
    if WRITE_REFERENCE_DESC:
        on_error = lambda e: print(self.desc_file + ": error saving file: " + str(e))
        if as_u8 is not None and as_u8.ndim == 2 and np.asarray(des).dtype in (np.float32, np.uint8):
            # the reference's file without the float32 bytes ever going through zlib
            cacheio.write_raw(self.desc_file, lambda: _desc_gzip_from_u8(as_u8),
                              background=ASYNC_CACHE_WRITES, on_error=on_error)
        else:
            cacheio.write_gzip(self.desc_file, lambda: _npy_bytes(des), background=ASYNC_CACHE_WRITES,
                               on_error=on_error, level=DESC_GZIP_LEVEL)
    if USE_DESC_SIDECAR and as_u8 is not None:
        side = _sidecar(self.desc_file)
        cacheio.write_raw(side, lambda: _npy_bytes(as_u8), background=ASYNC_CACHE_WRITES,
                          on_error=lambda e: print(side + ": error saving file: " + str(e)))
        return
    # no sidecar written for these descriptors (non-integer values, none at all, sidecars
    # off): one left over from an earlier detection must not answer the next load
    side = _sidecar(self.desc_file)
    cacheio.wait(side)
    try:
        os.remove(side)
    except OSError:
        pass


def save_matches(self):
    try:
        with open(self.match_file, 'wb') as fp:
            # same objects on load as pickle.dump(self.match_list, fp) of lists of [i, j] lists
            # (image.py:261-268), written straight from the arrays behind the match lists
            dump_match_dict(self.match_list, fp)
        self.matches_clean = True
    except IOError:
        print(self.match_file + ": error saving file: " + str(sys.exc_info()[1]))


# --------------------------------------------------------------------------------------
# decode / equalise -- image.py:99-121
# --------------------------------------------------------------------------------------
def _decode_bgr(path, writable=True):
    """host-side JPEG decode (libjpeg through Pillow), EXIF orientation ignored like cv2.imread
    with IMREAD_ANYCOLOR | IMREAD_ANYDEPTH | IMREAD_IGNORE_ORIENTATION (image.py:101-104).
    The decoder's RGB rows are packed as BGR in ONE pass (three 60 MB passes for convert +
    asarray + channel flip cost as much as the decode itself); writable=False hands the packed
    buffer over as it is (the detector only uploads it)."""
    from PIL import Image as PILImage
    with PILImage.open(path) as im:
        if im.mode != 'RGB':
            im = im.convert('RGB')
        else:
            im.load()
        w, h = im.size
        buf = im.tobytes('raw', 'BGR')
    if writable:
        return np.frombuffer(bytearray(buf), np.uint8).reshape(h, w, 3)
    return np.frombuffer(buf, np.uint8).reshape(h, w, 3)


def load_rgb(self, equalize=False):
    """BGR uint8 [h,w,3] like cv2.imread; with equalize=True CLAHE(3.0, 8x8) on the HSV value
    channel (on the device).  Records width/height in the image's property node."""
    try:
        bgr = _decode_bgr(self.image_file)
        if equalize:
            from . import kernels
            bgr = kernels.equalize_resize(bgr, 1.0, equalize=True).cpu().numpy()
        h, w = bgr.shape[:2]
        self.node.setInt('height', h)
        self.node.setInt('width', w)
        return bgr
    except Exception:                     # noqa: BLE001
        print(str(self.image_file) + ":\n" + "  rgb load error: " + str(sys.exc_info()[1]))
        return None


# --------------------------------------------------------------------------------------
# detect -- image.py:287-350
# --------------------------------------------------------------------------------------
def features_from_bgr(bgr, scale, equalize=True, keep_u8=False, slot=False):
    """full-res BGR -> (kp_list in full-res pixels, des_list float32 [N,128]); everything
    after the decode runs on the GPU.  slot: take a detector slot for the device part (worker
    threads that detect concurrently; the host part below runs outside the slot)."""
    from . import kernels
    if slot:
        with kernels.detector_slot():
            scaled = kernels.equalize_resize(bgr, scale, equalize=equalize)
            kp, octave, desc = kernels.sift_detect(scaled)
    else:
        scaled = kernels.equalize_resize(bgr, scale, equalize=equalize)
        kp, octave, desc = kernels.sift_detect(scaled)
    # kp.pt = (kp.pt[0]/scale, kp.pt[1]/scale): keypoints are cached in FULL-RES pixels (:344-346)
    kp = np.asarray(kp)
    # python-float division like `kp.pt[0] / scale` on a cv2.KeyPoint, then float32 members
    kp_list = KeyPointList(kp[:, 0].astype(np.float64) / scale, kp[:, 1].astype(np.float64) / scale,
                           kp[:, 2], kp[:, 3], kp[:, 4], octave)
    des = np.ascontiguousarray(desc, np.uint8) if DES_LIST_U8 else _to_float32(desc)
    if keep_u8:
        return kp_list, des, desc
    return kp_list, des


def _to_float32(u8):
    """uint8 [N,128] -> float32 [N,128] (the reference's des_list dtype) in one call into libiamx
    (threads of its own: 25 MB per frame is 2.5 ms of one core)"""
    import ctypes
    from . import _lib
    u8 = np.ascontiguousarray(u8, np.uint8)
    out = np.empty(u8.shape, np.float32)
    _lib.check(_lib.lib().iamx_u8_to_f32(u8.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                         u8.size, 4), 'iamx_u8_to_f32')
    return out


def _prefetch_job(self):
    """worker-thread half of detect_features: the parts that release the GIL -- gunzip of both
    cache files, or the JPEG decode when there is no cache"""
    try:
        cacheio.wait(self.features_file)
        cacheio.wait(self.desc_file)
        if os.path.exists(self.features_file) and USE_DESC_SIDECAR:
            des = _load_sidecar(self)
            if des is not None:
                with gzip.open(self.features_file, 'rb') as fp:
                    feat = fp.read()
                return ('cache', feat, des)
        if os.path.exists(self.features_file) and os.path.exists(self.desc_file):
            with gzip.open(self.features_file, 'rb') as fp:
                feat = fp.read()
            with gzip.open(self.desc_file, 'rb') as fp:
                desc = fp.read()
            return ('cache', feat, desc)
    except Exception:                     # noqa: BLE001  (fall through to a fresh detection)
        pass
    try:
        # the host half of the split decoder (Huffman only, GIL released); files it does not
        # handle are decoded whole
        if USE_DEVICE_JPEG:
            from . import kernels
            jc = kernels.jpeg_host_decode(self.image_file)
            if jc is not None:
                # ... and the device half right behind it, on this worker's own stream: the 80 MB of
                # coefficients go up and become pixels while the detector works on an earlier
                # frame; the main thread receives a finished frame in HBM
                import torch
                # the current HIP device is per host thread: a fresh worker starts on device 0,
                # not on the device prefetch() was called under (a rank with LOCAL_RANK != 0)
                dev = getattr(self, '_iamx_prefetch_device', None)
                with torch.cuda.device(dev if dev is not None else torch.cuda.current_device()), \
                        torch.cuda.stream(_worker_stream()), kernels.polite_waits():
                    bgr = kernels.jpeg_reconstruct(jc)       # (waits for this stream only)
                    scale = getattr(self, '_iamx_prefetch_scale', None)
                    if scale is not None:
                        # the caller said which scale it will ask for: the whole detection runs
                        # here, in one of the detector slots, and the loop that calls
                        # detect_features() only collects results
                        feats = features_from_bgr(bgr, scale, equalize=True, keep_u8=True, slot=True)
                        return ('features', float(scale), feats, int(bgr.shape[0]), int(bgr.shape[1]))
                return ('bgr', bgr)
        return ('bgr', _decode_bgr(self.image_file, writable=False))
    except Exception:                     # noqa: BLE001  (detect_features repeats it and reports)
        return None


_worker_streams = __import__('threading').local()


def _worker_stream():
    """one HIP stream per prefetch worker thread AND device (uploads + JPEG reconstruction off the
    detector's stream); created on the thread's current device"""
    import torch
    streams = getattr(_worker_streams, 'streams', None)
    if streams is None:
        streams = _worker_streams.streams = {}
    dev = torch.cuda.current_device()
    st = streams.get(dev)
    if st is None:
        # high priority: a detection is ~40 short dependent kernels; beside find_matches' sweeps (long
        # kernels that fill every CU) each of them would otherwise queue for the compute units
        # the sweep's workgroups give back
        prio = int(os.environ.get('IAMX_WORKER_PRIO', '-1'))
        st = streams[dev] = torch.cuda.Stream(device=dev, priority=prio)
    return st


def prefetch(images, depth=None, scale=None):
    """Start decoding / cache-loading `images` (in this order) on worker threads; each image's
    next detect_features() picks its result up.  With `scale` (the argument detect_features()
    is going to be called with) images without cache files are DETECTED on the workers as well
    (kernels.DETECT_SLOTS at a time).  Returns the cacheio.Prefetch (close() it)."""
    todo = [im for im in images if getattr(im, 'image_file', None) or
            os.path.exists(getattr(im, 'features_file', '') or '')]
    dev = None
    try:
        import torch
        if torch.cuda.is_available():
            dev = torch.cuda.current_device()          # the workers run their device half THERE
    except ImportError:                                # pragma: no cover
        pass
    for im in todo:
        im._iamx_prefetch_scale = scale
        im._iamx_prefetch_device = dev
    pf = cacheio.Prefetch(_prefetch_job, todo, PREFETCH_DEPTH if depth is None else depth)
    for im in todo:
        im._iamx_prefetch = pf
    return pf


def detect_features(self, scale, use_cache=True):
    pf = getattr(self, '_iamx_prefetch', None)
    pre = pf.take(self) if pf is not None and pf.pending(self) else None
    if use_cache and pre is not None and pre[0] in ('bgr', 'coef', 'features'):
        use_cache = False          # the worker looked a moment ago: no cache files (stat() is not free)
    if pre is not None and pre[0] == 'features' and pre[1] != float(scale):
        pre = None                 # detected at another scale than the one asked for now: start over
    if use_cache:
        if pre is not None and pre[0] == 'cache':
            try:
                self.kp_list = KeyPointList.from_feat_bytes(pre[1])
                self.des_list = pre[2] if isinstance(pre[2], np.ndarray) else np.load(io.BytesIO(pre[2]))
                _qlog("Loaded features/descriptors from cache:", self.name)
                return
            except Exception:             # noqa: BLE001
                pass
        success = True
        if not self.load_features():
            success = False
        if not self.load_descriptors():
            success = False
        if success:
            _qlog("Loaded features/descriptors from cache:", self.name)
            return
    _qlog("Detecting features/descriptors for:", self.name)
    detector_node = _deps.getNode('/config/detector', True)
    if detector_node.getString('detector') not in ('SIFT', ''):
        _log("Detector", detector_node.getString('detector'),
             "is not on the MI355X path (SIFT only)")
        quit()
    feats = None
    if pre is not None and pre[0] == 'features':
        feats, h, w = pre[2], pre[3], pre[4]
        bgr = None
    elif pre is not None and pre[0] == 'bgr':
        bgr = pre[1]
        if hasattr(bgr, 'record_stream'):
            import torch                # (allocated on a worker's stream, used on this one)
            bgr.record_stream(torch.cuda.current_stream())
    else:
        # Huffman decode on the host (already done by the prefetch worker if there is one), the
        # rest of the JPEG decode on the device: the frame never exists in host memory
        bgr = None
        if USE_DEVICE_JPEG:
            from . import kernels
            jc = pre[1] if pre is not None and pre[0] == 'coef' else kernels.jpeg_host_decode(self.image_file)
            if jc is not None:
                bgr = kernels.jpeg_reconstruct(jc)
        if bgr is None:
            bgr = _decode_bgr(self.image_file, writable=False)
    if feats is None:
        h, w = int(bgr.shape[0]), int(bgr.shape[1])
    self.node.setInt('height', h)
    self.node.setInt('width', w)
    cam_w, cam_h = _deps.camera().get_image_params()
    if w != cam_w or h != cam_h:
        _log("Error: image dimensions", w, h, "do not match camera config",
             cam_w, cam_h, "cannot continue safely.")
        _log("Please track down and fix the camera config vs. image size issue.")
        quit()
    if feats is None:
        feats = features_from_bgr(bgr, scale, equalize=True, keep_u8=True)
    self.kp_list, self.des_list, u8 = feats
    self._iamx_des_u8 = (self.des_list, u8)
    self.num_features = len(self.kp_list)
    self.save_features()
    self.save_descriptors()
    self._iamx_des_u8 = None       # (the writers hold what they need; a cache flush must free it)


_METHODS = dict(load_features=load_features, load_descriptors=load_descriptors,
                load_matches=load_matches, save_features=save_features,
                save_descriptors=save_descriptors, save_matches=save_matches,
                load_rgb=load_rgb, detect_features=detect_features)


def install(ref_image_module):
    """Give the reference's lib.image.Image the GPU feature methods (drop-in)."""
    for name, fn in _METHODS.items():
        setattr(ref_image_module.Image, name, fn)


class Image(PoseImage):
    """Stand-alone image record: poses (hostlib.image_pose) + the feature methods above, with
    the reference's file layout under <analysis_dir>/{meta,cache} (image.py:76-97)."""

    def __init__(self, analysis_dir=None, image_base=None):
        super(Image, self).__init__(image_base if image_base is not None else 'unnamed')
        self.image_file = None
        if image_base and analysis_dir:
            project_dir = _deps.getNode('/config/directories', True).getString('project_dir')
            for d in (project_dir, os.path.join(project_dir, 'images')):
                for ext in ('.JPG', '.jpg'):
                    p = os.path.join(d, image_base + ext)
                    if os.path.isfile(p):
                        self.image_file = p
            self.features_file = os.path.join(analysis_dir, 'cache', image_base + ".feat")
            self.desc_file = os.path.join(analysis_dir, 'cache', image_base + ".desc")
            self.match_file = os.path.join(analysis_dir, 'meta', image_base + ".match")

    def get_size(self):
        return self.node.getInt('width'), self.node.getInt('height')


for _n, _f in _METHODS.items():
    setattr(Image, _n, _f)
