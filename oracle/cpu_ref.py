"""ORACLE / TEST INFRASTRUCTURE ONLY -- ctypes wrapper over oracle/cpu_ref.c
(liboracle_cpu.so, built by oracle/Makefile or __graft_entry__.build())."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle_cpu.so')
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < max(os.path.getmtime(os.path.join(_HERE, f))
                                        for f in ('cpu_ref.c', 'sift_ref.c', 'knn2_simd.c', 'jpeg_ref.c', 'Makefile')):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle_cpu.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_knn2_l2_u8.restype = ctypes.c_int
        _lib.oracle_ba_residual.restype = ctypes.c_int
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knn2_l2_u8(q, t, nthreads=0):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.empty((q.shape[0], 2), np.int32)
    d2 = np.empty((q.shape[0], 2), np.int32)
    rc = lib().oracle_knn2_l2_u8(_p(q), ctypes.c_int(q.shape[0]), _p(t), ctypes.c_int(t.shape[0]),
                                 _p(idx), _p(d2), ctypes.c_int(nthreads))
    if rc != 0:
        raise ValueError("oracle_knn2_l2_u8 rc=%d" % rc)
    return idx, d2


def knn2_l2_u8_batch(images, pairs, nthreads=0):
    """images [n_img, n_rows, 128] u8, pairs [n_pairs, 2] (query, train) -> idx, d2
    [n_pairs, n_rows, 2]; one OpenMP region over all pairs (the CPU-baseline form)."""
    images = np.ascontiguousarray(images, np.uint8)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    n_rows = images.shape[1]
    idx = np.empty((len(pairs), n_rows, 2), np.int32)
    d2 = np.empty((len(pairs), n_rows, 2), np.int32)
    L = lib()
    L.oracle_knn2_l2_u8_batch.restype = ctypes.c_int
    rc = L.oracle_knn2_l2_u8_batch(_p(images), ctypes.c_int(n_rows), _p(pairs),
                                   ctypes.c_int(len(pairs)), _p(idx), _p(d2),
                                   ctypes.c_int(nthreads))
    if rc != 0:
        raise ValueError("oracle_knn2_l2_u8_batch rc=%d" % rc)
    return idx, d2


def knn2_simd_available():
    L = lib()
    L.oracle_knn2_simd_available.restype = ctypes.c_int
    return bool(L.oracle_knn2_simd_available())


def knn2_l2_u8_batch_simd(images, pairs, nthreads=0):
    """knn2_l2_u8_batch with the AVX-512 VNNI kernel of oracle/knn2_simd.c (same results);
    raises where the host has no AVX-512 VNNI"""
    images = np.ascontiguousarray(images, np.uint8)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    n_rows = images.shape[1]
    idx = np.empty((len(pairs), n_rows, 2), np.int32)
    d2 = np.empty((len(pairs), n_rows, 2), np.int32)
    L = lib()
    L.oracle_knn2_l2_u8_batch_simd.restype = ctypes.c_int
    rc = L.oracle_knn2_l2_u8_batch_simd(_p(images), ctypes.c_int(n_rows), _p(pairs),
                                        ctypes.c_int(len(pairs)), _p(idx), _p(d2), ctypes.c_int(nthreads))
    if rc != 0:
        raise ValueError("oracle_knn2_l2_u8_batch_simd rc=%d" % rc)
    return idx, d2


def ba_residual(cams, pts, cam_idx, pt_idx, uv, intr, dist, nthreads=0):
    cams = np.ascontiguousarray(cams, np.float64).reshape(-1, 7)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    cam_idx = np.ascontiguousarray(cam_idx, np.int32)
    pt_idx = np.ascontiguousarray(pt_idx, np.int32)
    uv = np.ascontiguousarray(uv, np.float64)
    intr = np.ascontiguousarray(intr, np.float64)
    dist = np.ascontiguousarray(dist, np.float64)
    r = np.empty(2 * cam_idx.size, np.float64)
    lib().oracle_ba_residual(_p(cams), ctypes.c_int(cams.shape[0]), _p(pts),
                             ctypes.c_int(pts.shape[0]), _p(cam_idx), _p(pt_idx), _p(uv),
                             ctypes.c_int64(cam_idx.size), _p(intr), _p(dist), _p(r),
                             ctypes.c_int(nthreads))
    return r


def num_threads():
    return lib().oracle_num_threads()


def host_cores():
    """cores this process may really use: the CPU affinity mask and the cgroup quota (cpu.max),
    not the number of hardware threads the machine shows"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return n


def set_num_threads(n):
    """OpenMP threads of the oracle's loops from now on (bench.py: host_cores())"""
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))
    return num_threads()


# ---- oracle/sift_ref.c: the hot loops of sift_oracle.py
def sift_blur(img, taps, nthreads=0):
    img = np.ascontiguousarray(img, np.float32)
    taps = np.ascontiguousarray(taps, np.float32)
    out = np.empty_like(img)
    rc = lib().oracle_sift_blur(_p(img), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]),
                                _p(taps), ctypes.c_int(len(taps) // 2), _p(out), ctypes.c_int(nthreads))
    if rc != 0:
        raise ValueError("oracle_sift_blur rc=%d" % rc)
    return out


def _ptr_array(arrays):
    return (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


def sift_keypoints(dogs, gauss, octave, cand, sigma0, nthreads=0):
    """dogs / gauss: the float32 levels of one octave; cand int32 [n,3] (layer, r, c) ->
    keypoints float64 [m,6] in the candidates' order"""
    dogs = [np.ascontiguousarray(d, np.float32) for d in dogs]
    gauss = [np.ascontiguousarray(g, np.float32) for g in gauss]
    cand = np.ascontiguousarray(cand, np.int32).reshape(-1, 3)
    h, w = dogs[0].shape
    cap = max(4 * len(cand), 16)
    L = lib()
    L.oracle_sift_keypoints.restype = ctypes.c_int
    while True:
        kps = np.empty((cap, 6), np.float64)
        n = L.oracle_sift_keypoints(_ptr_array(dogs), _ptr_array(gauss), ctypes.c_int(h),
                                    ctypes.c_int(w), ctypes.c_int(octave), _p(cand),
                                    ctypes.c_int(len(cand)), ctypes.c_double(sigma0), _p(kps),
                                    ctypes.c_int(cap), ctypes.c_int(nthreads))
        if n < 0:
            raise ValueError("oracle_sift_keypoints rc=%d" % n)
        if n <= cap:
            return kps[:n]
        cap = n


def sift_descriptors(img, par, nthreads=0):
    """img: one Gaussian level; par float32 [n,4] (ptx, pty, ori, scl) -> uint8 [n,128]"""
    img = np.ascontiguousarray(img, np.float32)
    par = np.ascontiguousarray(par, np.float32).reshape(-1, 4)
    desc = np.empty((len(par), 128), np.uint8)
    rc = lib().oracle_sift_descriptors(_p(img), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]),
                                       _p(par), ctypes.c_int(len(par)), _p(desc), ctypes.c_int(nthreads))
    if rc != 0:
        raise ValueError("oracle_sift_descriptors rc=%d" % rc)
    return desc


def sift_detect(gray, cap=None, nthreads=0):
    """the whole detector + descriptor in C (oracle/sift_ref.c oracle_sift_detect): gray uint8
    [h,w] -> (kps float64 [n,6], desc uint8 [n,128]), duplicates removed, OpenCV's output order"""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = int(cap or max(h * w // 8, 4096))
    kps = np.empty((cap, 6), np.float64)
    desc = np.empty((cap, 128), np.uint8)
    L = lib()
    L.oracle_sift_detect.restype = ctypes.c_int
    n = L.oracle_sift_detect(_p(gray), ctypes.c_int(h), ctypes.c_int(w), _p(kps), _p(desc),
                             ctypes.c_int(cap), ctypes.c_int(nthreads))
    if n < 0:
        raise ValueError("oracle_sift_detect rc=%d" % n)
    return kps[:n].copy(), desc[:n].copy()


def jpeg_reconstruct(coef, quant, info):
    """oracle/jpeg_ref.c: quantised coefficients (as iamx_jpeg_decode_coefficients writes them)
    -> BGR uint8 [h, w, 3] the way libjpeg-turbo's defaults reconstruct them"""
    coef = np.ascontiguousarray(coef, np.int16)
    quant = np.ascontiguousarray(quant, np.uint16)
    info = np.ascontiguousarray(info, np.int32)
    out = np.empty((int(info[1]), int(info[0]), 3), np.uint8)
    rc = lib().oracle_jpeg_reconstruct(_p(coef), _p(quant), _p(info), _p(out))
    if rc != 0:
        raise ValueError("oracle_jpeg_reconstruct rc=%d" % rc)
    return out
