"""Timing ablation of the symmetric sweep (GPU box only).  Needs the ablation build:
    IAMX_ABLATE=1 bash imageanalysis_amd/csrc/build.sh
    IAMX_LIB=imageanalysis_amd/libiamx_ablate.so python tools/knn2sym_ablate.py [variants]
variant bits: 1 no column direction, 2 no row direction, 4 row direction without the cross-lane
butterfly, 8 no MFMA (0 = the shipped kernel)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from imageanalysis_amd import kernels
from imageanalysis_amd.kernels import _ptr, lib, stream_ptr
import os
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 8, 11]
n_img = int(os.environ.get('ABL_IMAGES', '64'))
reps = int(os.environ.get('ABL_REPS', '4'))
max_pairs = int(os.environ.get('ABL_PAIRS', '1000000'))
rng = np.random.default_rng(0)


def sift_like(n):
    g = rng.gamma(0.6, 1.0, size=(n, 128))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g = np.minimum(g, 0.2)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return np.clip(np.rint(g * 512.0), 0, 255).astype(np.uint8)


store = kernels.DescriptorStore.from_arrays([sift_like(4096) for _ in range(n_img)])
L = lib()
und = np.array([(i, j) for j in range(n_img) for i in range(j)], np.int32)[-max_pairs:]
pairs = np.concatenate([und, und[:, ::-1]])
b = kernels.PairBatch(store, pairs, sym=True)
ws = kernels.PairWorkspace(b.rows, b.n_pairs)
ws.ensure_sym(b.sym_col_rows, b.sym_rowp_rows)
fn = L.iamxdbg_knn2sym_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
st = store
ref = None
ref300 = None
names = {600: '8 query blocks per wave, one wave per SIMD, B operand in AGPRs, PIPE 6', 601: '... PIPE 5', 602: '... PIPE 0', 603: '... PIPE 4', 604: '... PIPE 7', 605: '... PIPE 5, 256-row chunks, merge on 4 waves', 606: '... PIPE 5, merge on 4 waves', 500: 'cross-chunk pipeline, barrier at step 4 / 4', 501: '... 4 / 0', 502: '... 0 / 0', 503: '... 5 / 1', 504: '... 2 / 2', 505: '... 4 / 1', 506: '... 4 / 0, PIPE 5', 507: '... 4 / 0, PIPE 7', 508: '... 5 / 2', 509: '... 3 / 0', 330: 'MFMA + staging, one LDS operand read per chunk', 331: '... and no staging after chunk 1', 332: '... and no barrier / merge: MFMA issue only', 320: 'PIPE 0 with the MFMA bursts at raised wave priority', 400: 'no staging after chunk 1 (timing only)', 401: '... and no chunk barrier', 402: '... and no row merge', 403: 'MFMA only, no staging after chunk 1', 404: 'MFMA + staging only (new form)', 310: '256-row chunks, merge on 4 waves, PIPE 5', 311: '128-row chunks, merge on 4 waves, PIPE 5', 300: 'fused butterfly + group Cq floor, PIPE 6', 301: '... PIPE 0', 302: '... PIPE 4', 303: '... PIPE 5', 200: 'merge on waves 0-3', 201: 'sleep 2 for waves 4-7', 202: 'sleep 5 for waves 4-7', 203: 'merge on waves 0-3 + sleep 3', 204: 'merge on all 8 waves', 16: 'shipped schedule without the per-chunk row merge', 48: '... and without the chunk barrier (timing only)', 100: 'pipelined, 4 VALU per MFMA', 101: 'pipelined, 6 VALU per MFMA', 102: 'pipelined, 8 VALU per MFMA', 0: 'shipped', 1: 'no column direction', 2: 'no row direction', 3: 'MFMA + staging only',
         4: 'row direction without butterfly', 5: 'row min tree only', 8: 'no MFMA',
         11: 'staging + barriers only'}
for v in variants:
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
          kernels.check(fn(v, _ptr(st.desc3), _ptr(st.sn2), _ptr(st.sct), _ptr(st.img_off3), _ptr(st.img_n),
                           _ptr(b.d_upairs), _ptr(b.d_sym_wg), _ptr(b.d_col_off), _ptr(b.d_rowp_off), b.n_u,
                         b.sym_total_wg, _ptr(ws.col), _ptr(ws.rowp), stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    t = min(ts[1:])
    if v == 0:
        ref = (ws.col[:b.sym_col_rows].clone(), ws.rowp[:b.sym_rowp_rows].clone())
    elif v == 300:
        ref300 = (ws.col[:b.sym_col_rows].clone(), ws.rowp[:b.sym_rowp_rows].clone())
    elif v >= 600:                              # other lane -> row map (32-row Cq groups): own bounds, checked
        pass                                    # through tests/test_match_sym_gpu.py once it ships
    elif v >= 500:                              # cross-chunk pipeline: the arithmetic of variant 300
        assert ref300 is not None, 'run variant 300 first'
        assert torch.equal(ref300[0], ws.col[:b.sym_col_rows]), 'variant %d: column bounds differ' % v
        assert torch.equal(ref300[1][:, :2], ws.rowp[:b.sym_rowp_rows, :3]), 'variant %d: row bounds differ' % v
    elif 100 <= v < 300 and ref is not None:    # alternative schedules of the same arithmetic
        # (300+: group-shared Cq floor -- different, equally valid bounds)
        assert torch.equal(ref[0], ws.col[:b.sym_col_rows]), 'variant %d: column bounds differ' % v
        assert torch.equal(ref[1][:, :2], ws.rowp[:b.sym_rowp_rows, :3]), 'variant %d: row bounds differ' % v
    print("sym variant %2d (%s): %.3f ms for %d image pairs -> %.3f us / unordered pair"
          % (v, names.get(v, '?'), t, b.n_u, t * 1e3 / b.n_u))
