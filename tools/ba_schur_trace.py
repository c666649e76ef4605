"""Per-outer-iteration trace of the device TRF at BASELINE configs[3]: cost, inner iterations, stop
code and wall clock of every Gauss-Newton solve, for the Schur solver's settings named on the
command line (eta qtol maxit triples) and for the LSMR formulation.
    python tools/ba_schur_trace.py 0.1:0.1:500 0.1:0:50 lsmr"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from imageanalysis_amd import ba_solver, synth
    p = synth.make_ba_problem()
    C, P, O = len(p['cams0']), len(p['pts0']), len(p['cam_idx'])
    K = p['K']
    calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    lb = np.full(x0.size, -np.inf)
    ub = np.full(x0.size, np.inf)
    for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
        lb[j:C * 7:7] = p['cams0'][:, j] - dlt
        ub[j:C * 7:7] = p['cams0'][:, j] + dlt
    prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib)
    for spec in sys.argv[1:] or ['0.1:0.1:500']:
        if spec == 'lsmr':
            prob.inner = 'lsmr'
        else:
            prob.inner = 'schur'
            eta, qtol, maxit = spec.split(':')
            prob.schur_eta, prob.schur_qtol, prob.schur_max_iter = float(eta), float(qtol), int(maxit)
        ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=2)
        costs = []
        del prob.inner_iterations[:], prob.inner_stops[:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, callback=lambda x, c: costs.append(c))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-16s outer %2d nfev %2d status %d  %.3f s = %.1f it/s  cost %.6e  rms %.4f px  inner %d"
              % (spec, res.iterations, res.nfev, res.status, dt, res.iterations / dt, res.cost,
                 np.sqrt(2 * res.cost / (2 * O)), sum(prob.inner_iterations)))
        print("   inner per solve:", prob.inner_iterations)
        print("   stops          :", prob.inner_stops)
        print("   cost trace     :", " ".join("%.4e" % c for c in costs))
        prob.profile = {}
        ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=6)
        tot = sum(prob.profile.values())
        print("   phases (first 5 outer iterations, synchronised): "
              + ", ".join("%s %.1f ms" % (k, 1e3 * v) for k, v in sorted(prob.profile.items(), key=lambda kv: -kv[1]))
              + " | total %.1f ms" % (1e3 * tot))
        prob.profile = None


if __name__ == '__main__':
    main()
