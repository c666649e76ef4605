#!/usr/bin/env python3
"""A/B of the BA residual kernel on BASELINE configs[3] (2812 cameras / ~272 k points / 1.96 M
observations): the one-chain-per-workgroup form (IAMX_BA_RESIDUAL=lds) against the persistent
pipelined walk at several grid sizes (IAMX_BA_RESIDUAL_WGS), back to back (125 MB: Infinity-Cache
resident) and over six problem copies in rotation (750 MB: every launch finds its inputs evicted),
beside the 16 B/lane grid-stride copy over a rotating set of the same size (iamx_hbm_copy16)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_amd import _lib, ba_solver, synth  # noqa: E402

HBM = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    p = synth.make_ba_problem()
    C, P, O = len(p['cams0']), len(p['pts0']), len(p['cam_idx'])
    K = p['K']
    calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    n_rot = 6
    rot = []
    for _ in range(n_rot):
        q = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib)
        q.set_x(x0)
        rot.append(q)
    launch = [q.bound_launchers()[0] for q in rot]
    ref = None
    print("configs[3]: %d cameras, %d points, %d observations; 64 B/obs = %.1f MB per evaluation" % (C, P, O, 64e-6 * O))
    grids = [int(a) for a in sys.argv[1:]] or [128, 192, 256, 320, 384, 512, 768, 1024, 2048]
    for form, wgs in [('lds', 0)] + [('pipe', g) for g in grids]:
        os.environ['IAMX_BA_RESIDUAL'] = form
        if wgs:
            os.environ['IAMX_BA_RESIDUAL_WGS'] = str(wgs)
        t_hot = timed(launch[0], 100)

        def rot_res():
            for f in launch:
                f()
        t_cold = timed(rot_res, 20) / n_rot
        r = rot[0].r.clone() if hasattr(rot[0], 'r') else None
        same = ''
        if r is not None:
            if ref is None:
                ref = r
            same = ' identical to lds form: %s' % bool(torch.equal(r, ref))
        print("%-4s wgs %4d: hot %6.2f us (%.3f of 8 TB/s), rotating %6.2f us (%.3f)%s"
              % (form, wgs, t_hot * 1e6, 64.0 * O / t_hot / 1e9 / HBM, t_cold * 1e6,
                 64.0 * O / t_cold / 1e9 / HBM, same))
    # the yardsticks: 125 MB moved per launch, six buffers in rotation (and back to back): a copy
    # (62.5 MB read + 62.5 MB written) and the residual's own mix (94 MB read + 31 MB written)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in (1, 3):
        n_out = int(125.5e6 / (k + 1)) // 16
        srcs = [torch.empty(2 * n_out * k, dtype=torch.float64, device='cuda').normal_() for _ in range(n_rot)]
        dsts = [torch.empty(2 * n_out, dtype=torch.float64, device='cuda') for _ in range(n_rot)]
        for wg in (1024, 2048, 8192):
            def one(a_=srcs[0], b_=dsts[0]):
                L.iamx_hbm_copy16(ctypes.c_void_p(a_.data_ptr()), ctypes.c_void_p(b_.data_ptr()), n_out, k, wg, st)

            def rot_copy():
                for a_, b_ in zip(srcs, dsts):
                    one(a_, b_)
            t_hot = timed(one, 100)
            t = timed(rot_copy, 20) / n_rot
            by = n_out * 16 * (k + 1)
            print("stream %d:1 read:write, %5d workgroups: hot %6.2f us (%.3f), rotating %6.2f us = %.0f GB/s "
                  "(%.3f of 8 TB/s)" % (k, wg, t_hot * 1e6, by / t_hot / 1e9 / HBM, t * 1e6, by / t / 1e9,
                                        by / t / 1e9 / HBM))
        del srcs, dsts
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
