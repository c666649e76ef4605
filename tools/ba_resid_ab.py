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
    for form, wgs in (('lds', 0), ('pipe', 256), ('pipe', 512), ('pipe', 768), ('pipe', 1024), ('pipe', 1536),
                      ('pipe', 2048), ('pipe', 3072)):
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
    # the yardstick: 62.5 MB read + 62.5 MB written per launch, six buffers in rotation
    n_el = int(62.5e6 // 16)
    srcs = [torch.empty(2 * n_el, dtype=torch.float64, device='cuda').normal_() for _ in range(n_rot)]
    dsts = [torch.empty(2 * n_el, dtype=torch.float64, device='cuda') for _ in range(n_rot)]
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for wg in (1024, 2048, 4096, 8192):
        def rot_copy():
            for a_, b_ in zip(srcs, dsts):
                L.iamx_hbm_copy16(ctypes.c_void_p(a_.data_ptr()), ctypes.c_void_p(b_.data_ptr()), n_el, wg, st)
        t = timed(rot_copy, 20) / n_rot
        print("copy16 %5d workgroups, rotating: %6.2f us = %.0f GB/s (%.3f of 8 TB/s)"
              % (wg, t * 1e6, 2 * n_el * 16 / t / 1e9, 2 * n_el * 16 / t / 1e9 / HBM))

    def rot_torch():
        for a_, b_ in zip(srcs, dsts):
            b_.copy_(a_)
    t = timed(rot_torch, 20) / n_rot
    print("torch copy_, rotating: %6.2f us = %.0f GB/s" % (t * 1e6, 2 * n_el * 16 / t / 1e9))


if __name__ == '__main__':
    main()
