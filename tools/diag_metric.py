import numpy as np, torch, sys
sys.path.insert(0, '.')
from imageanalysis_amd import kernels
n = 128*255*255+1
d = np.arange(n, dtype=np.int64)
d2 = np.stack([d, np.maximum(d,1)], 1).astype(np.int32)
seg = np.array([0, n], np.int64)
m, keep, cnt, zd = kernels.match_metric(torch.from_numpy(d2).cuda(), seg, 202.5)
m = m.cpu().numpy()
f = np.sqrt(d2.astype(np.float32)).astype(np.float64)
ref = f[:,0]*(f[:,0]/f[:,1])
bad = np.nonzero(m != ref)[0]
print("sqrt-only mismatches:", len(bad), bad[:10], m[bad[:5]], ref[bad[:5]])
# division: perfect squares
rng = np.random.default_rng(0)
a = rng.integers(1, 2885, 1<<20); b = np.maximum(a, rng.integers(1, 2885, 1<<20))
d2 = np.stack([a*a, b*b], 1).astype(np.int32)
m, keep, cnt, zd = kernels.match_metric(torch.from_numpy(d2).cuda(), np.array([0, len(a)], np.int64), 202.5)
m = m.cpu().numpy()
ref = a.astype(np.float64)*(a.astype(np.float64)/b.astype(np.float64))
bad = np.nonzero(m != ref)[0]
print("div/mul mismatches:", len(bad), bad[:10])
if len(bad):
    i = bad[0]; print(a[i], b[i], repr(m[i]), repr(ref[i]), repr(a[i]/b[i]))
    q = a.astype(np.float64)/b.astype(np.float64)
    # is it fma contraction? d0*ratio with ratio computed... check alternative: a*a/b
    alt = np.array([float(np.float64(x)*np.float64(x)/np.float64(y)) for x,y in zip(a[bad[:5]], b[bad[:5]])])
    print(m[bad[:5]], ref[bad[:5]], alt)
