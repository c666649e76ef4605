import ctypes, time, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from imageanalysis_amd import kernels
L = kernels.lib()
arrs = [np.random.randint(0, 255, (37000, 128), dtype=np.uint8) for _ in range(13)]
n = len(arrs); tot = sum(a.size for a in arrs)
t0 = time.perf_counter(); pin = torch.empty(tot, dtype=torch.uint8).pin_memory(); print('pin alloc %.1f ms for %d MB' % ((time.perf_counter() - t0) * 1e3, tot >> 20))
srcs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs]); cnt = (ctypes.c_int64 * n)(*[a.size for a in arrs])
for th in (1, 4, 8, 16, 32):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); L.iamx_u8_gather_many(srcs, cnt, n, ctypes.c_void_p(pin.data_ptr()), th); ts.append(time.perf_counter() - t0)
    print('gather %2d threads: %.2f ms (%.1f GB/s)' % (th, min(ts) * 1e3, tot / min(ts) / 1e9))
dev = torch.empty(tot, dtype=torch.uint8, device='cuda')
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(pin, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('H2D pinned: %.2f ms (%.1f GB/s)' % (dt * 1e3, tot / dt / 1e9))
big = np.concatenate(arrs)
torch.cuda.synchronize(); t0 = time.perf_counter(); x = torch.from_numpy(big).to('cuda'); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('H2D pageable: %.2f ms (%.1f GB/s)' % (dt * 1e3, tot / dt / 1e9))
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
