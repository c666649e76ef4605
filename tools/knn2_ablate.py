"""Timing ablation of the knn2 kernel (GPU box only): python tools/knn2_ablate.py [variants]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from imageanalysis_amd import kernels
from imageanalysis_amd.kernels import _ptr, lib, stream_ptr
variants = [int(v) for v in sys.argv[1:]] or [0, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23]
n_img = 64
rng = np.random.default_rng(0)
imgs = rng.integers(0, 256, (n_img, 4096, 128), dtype=np.uint8)
store = kernels.DescriptorStore([4096] * n_img)
L = lib()
raw = torch.from_numpy(imgs).cuda()
kernels.check(L.iamx_desc_pack_u8(_ptr(raw), n_img * 4096, _ptr(store.desc), _ptr(store.norm_q), _ptr(store.norm_t), stream_ptr()))
pairs = np.array([(i, j) for j in range(n_img) for i in range(n_img) if i != j], np.int32)
b = kernels.PairBatch(store, pairs)
ws = kernels.PairWorkspace(b.rows, b.n_pairs)
QB = {16: 128, 17: 128, 18: 384, 19: 512, 20: 512, 21: 512, 22: 256, 23: 256}
ref = None
fn = L.iamxdbg_knn2_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
for v in variants:
    ts = []
    qb = QB.get(v, 256)
    wg = np.zeros(b.n_pairs + 1, np.int64); wg[1:] = np.cumsum(np.full(b.n_pairs, (4096 + qb - 1) // qb))
    b.d_wg = torch.from_numpy(wg.astype(np.int32)).cuda(); b.total_wg = int(wg[-1])
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        kernels.check(fn(v, _ptr(store.desc), _ptr(store.norm_q), _ptr(store.norm_t), _ptr(store.img_off), _ptr(store.img_n),
                         _ptr(b.d_pairs), _ptr(b.d_wg), _ptr(b.d_out), b.n_pairs, b.total_wg, _ptr(ws.idx), _ptr(ws.d2), stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts[1:])
    if v == 0: ref = (ws.idx.clone(), ws.d2.clone())
    elif v >= 10 and ref is not None:
        assert torch.equal(ref[0], ws.idx) and torch.equal(ref[1], ws.d2), "variant %d wrong results" % v
    print("variant %d: %.3f ms for %d ordered pairs -> %.3f us/ordered pair, %.1f alg TFLOP/s (x2 actual i8 ops)" % (
        v, t, b.n_pairs, t * 1e3 / b.n_pairs, b.n_pairs / 2 * 4.295e9 / (t * 1e-3) / 1e12))
