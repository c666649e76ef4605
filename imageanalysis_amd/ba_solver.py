"""Device-resident trust-region-reflective solver for the bundle adjustment
(`Optimizer.solver = 'device'`).

The reference calls scipy.optimize.least_squares(method='trf', jac_sparsity=A, x_scale='jac',
bounds, ftol=1e-4)  (scripts/lib/optimizer.py:491-501): SciPy finite-differences fun() into a
sparse J and runs `trf_bounds` with an LSMR subproblem solver.  SciPy refuses x_scale='jac'
for operator Jacobians (scipy/optimize/_lsq/least_squares.py), so to keep J in HBM the outer
iteration is restated here -- same sequence as scipy/optimize/_lsq/trf.py:205-400 (SciPy 1.15.3;
the reference pins 1.6.2, same algorithm): Coleman-Li scaling, x_scale='jac' column norms,
regularised Gauss-Newton step by LSMR, 2-D subspace trust-region solve, reflective step
selection, radius update, ftol/xtol/gtol tests.  The n-vectors of that logic (x, g, the
Coleman-Li vectors, candidate steps) live on the device too (_trf_device: torch elementwise ops
and reductions, the host sees a few dozen scalars per outer iteration and calls SciPy's scalar
helpers); _trf_host keeps the numpy form with SciPy's own vector helpers as the cross-check.
Everything that touches J or an m-vector runs through the K3/K4 kernels (ba_kernels.hip,
ba_linalg.hip):

    fun / jac           iamx_ba_residual, iamx_ba_residual_jac
    g = J^T f, ||J_j||  iamx_ba_jtv (square=0 / 1)
    LSMR                iamx_ba_jv / iamx_ba_jtv + iamx_vec_* (scipy/sparse/linalg/_isolve/lsmr.py)
    quadratic models    Gram matrices of J_h s_i  (iamx_ba_jv + iamx_vec_dot)

Multi-GPU: observations are sharded by point (dist.shard_observations_by_point); m-vectors are
rank-local, n-vectors replicated; J^T u and every m-dot are summed over ranks (RCCL all-reduce).
"""
import os

import numpy as np
import torch
from numpy.linalg import norm

from . import _lib, dist as _dist
from ._lib import check, lib, stream_ptr

F64, I32 = torch.float64, torch.int32


def _ptr(t):
    return _lib.c_void_p(t.data_ptr()) if t is not None else _lib.c_void_p(0)


class _Phase(object):
    """wall-clock per solver phase (only when DeviceBA.profile is a dict; synchronises)."""

    def __init__(self, prob, name):
        self.prob, self.name = prob, name

    def __enter__(self):
        if self.prob.profile is not None:
            import time
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.prob.profile is not None:
            import time
            torch.cuda.synchronize()
            d = self.prob.profile
            d[self.name] = d.get(self.name, 0.0) + time.perf_counter() - self.t0


class DeviceBA(object):
    """The BA problem of an Optimizer after setup(), resident in HBM."""

    def __init__(self, n_cameras, n_points, camera_indices, point_indices, points_2d,
                 with_calib, fixed_calib=None, rank=0, world=1):
        dev = _lib.require_gpu()
        self.dev = dev
        self.C, self.P = int(n_cameras), int(n_points)
        self.with_calib = bool(with_calib)
        self.fixed_calib = None if fixed_calib is None else np.asarray(fixed_calib, np.float64)
        self.n = self.C * 7 + self.P * 3 + (8 if with_calib else 0)
        cam = np.asarray(camera_indices, np.int64)
        pt = np.asarray(point_indices, np.int64)
        uv = np.asarray(points_2d, np.float64).reshape(-1, 2)
        self.rank, self.world = rank, world
        # Internal point order: by the first camera that observes a point (then by id).  The
        # reference numbers points by chain length, unrelated to where they are; with
        # camera-major observations that turns every 24-byte point access of the residual /
        # Jacobian / J.v kernels into a random gather over the whole point array.  Renumbered,
        # a camera's points sit in a few contiguous runs.  The permutation is private to this
        # class: host n-vectors keep the reference's order and cross through upload_n /
        # download_n.  It is computed from ALL observations, so every rank uses the same one.
        first_cam = np.full(self.P, self.C, np.int64)
        if cam.size:
            np.minimum.at(first_cam, pt, cam)
        old_of_new = np.lexsort((np.arange(self.P), first_cam))
        new_of_old = np.empty(self.P, np.int64)
        new_of_old[old_of_new] = np.arange(self.P)
        pt = new_of_old[pt]
        tail = np.arange(self.C * 7 + self.P * 3, self.n)
        h2i = np.concatenate([np.arange(self.C * 7),
                              (self.C * 7 + 3 * old_of_new[:, None] + np.arange(3)).ravel(), tail])
        i2h = np.concatenate([np.arange(self.C * 7),
                              (self.C * 7 + 3 * new_of_old[:, None] + np.arange(3)).ravel(), tail])
        # several ranks: observations AND the point part of the LSMR n-vectors are sharded by
        # point -- this rank owns the contiguous block [pt_lo, pt_hi) of the internal point order
        self.pt_lo, self.pt_hi = 0, self.P
        if world > 1:
            sel = _dist.shard_observations_by_point(pt, self.P, rank, world)
            self.pt_lo, self.pt_hi = _dist.point_range(pt, self.P, rank, world)
        else:
            sel = np.arange(cam.size)
        # Internal observation order: camera-major like the reference, but inside a camera by the
        # (renumbered) point id, so that the point gathers of a camera walk forward through
        # memory and the ut gathers of neighbouring points land next to each other.  local_obs[k]
        # = position of internal observation k in the reference's list (gather_residual()).
        sel = sel[np.lexsort((pt[sel], cam[sel]))]
        cam, pt, uv = cam[sel], pt[sel], uv[sel]
        self.local_obs = sel
        self.O = int(cam.size)
        self.m = 2 * self.O
        if self.O and np.any(np.diff(cam) < 0):
            raise ValueError("observations must be camera-major (optimizer.py:397-404)")
        cam_ptr = np.zeros(self.C + 1, np.int64)
        np.cumsum(np.bincount(cam, minlength=self.C), out=cam_ptr[1:])
        order = np.argsort(pt, kind='stable')
        pt_ptr = np.zeros(self.P + 1, np.int64)
        np.cumsum(np.bincount(pt, minlength=self.P), out=pt_ptr[1:])
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
        self.cam_idx, self.pt_idx = t(cam, I32), t(pt, I32)
        self.cam_ptr, self.pt_ptr, self.pt_obs = t(cam_ptr, I32), t(pt_ptr, I32), t(order, I32)
        # (camera, point) of the observation in every point-sorted slot (matrix-free LSMR adjoint)
        self.slot_cp = t(np.stack([cam[order], pt[order]], 1) if cam.size else np.zeros((1, 2)), I32)
        self.uv = t(uv, F64)
        self.idx_h2i, self.idx_i2h = t(h2i, torch.int64), t(i2h, torch.int64)
        z = lambda k: torch.zeros(max(int(k), 1), dtype=F64, device=dev)
        self.x = z(self.n)
        self.calib = z(9)
        self.r = z(self.m)
        self.Jc, self.Jp = z(self.O * 14), z(self.O * 6)
        self.Jk = z(self.O * 16) if with_calib else None
        self.scratch = z(2048)
        self.cam_rt = z(self.C * 12)
        self.out1 = z(4)
        self.tmp_n, self.tmp_n2, self.tmp_m, self.tmp_m2 = z(self.n), z(self.n), z(self.m), z(self.m)
        self.tmp_perm = z(self.n)
        self._tmp_many = None
        self.lsmr_ws = None
        self.schur_ws = None
        self.schur_last_itn = 0
        # normal-equation blocks of the current Jacobian (iamx_ba_accumulate): U [C][7][7],
        # V [P][3][3], g = (gc [C][7], gp [P][3]) -- this rank's observations only
        self.acc = None
        self._acc_valid = False
        # Gauss-Newton subproblem solver: 'schur' (points eliminated, preconditioned CG on the
        # reduced camera system; default) or 'lsmr' (SciPy's formulation, the reference's path)
        self.inner = os.environ.get('IAMX_BA_INNER', 'schur')
        self.schur_eta = float(os.environ.get('IAMX_BA_SCHUR_ETA', '0.1'))
        self.schur_qtol = float(os.environ.get('IAMX_BA_SCHUR_QTOL', '0.3'))
        self.schur_max_iter = int(os.environ.get('IAMX_BA_SCHUR_MAXIT', '500'))
        self.inner_stops = []
        self.inner_iterations = []        # inner iterations of every subproblem solved (tests)
        self._pin = None
        self._state_pin = None
        self.profile = None
        self.force_stepwise_lsmr = False
        self.fused_phase_iterations = 0     # iterations run through iamx_ba_lsmr_phase (tests)
        # fused single-rank LSMR: replay a captured HIP graph per 64-iteration chunk
        self.use_graph = os.environ.get('IAMX_BA_GRAPH', '1') != '0'
        self.host_logic = False          # True: the O(n) TRF vector logic in numpy (_trf_host)
        # several ranks, device-resident outer iteration, no calibration columns: the point part
        # of every n-vector stays on the rank that owns the points (zeros elsewhere), only the
        # camera part and scalars are reduced inside the outer loop (set by _trf_device)
        self.local_points = False
        self._calib_idx = None
        self._fixed_calib_up = False

    # ---- host <-> device staging -------------------------------------------------------
    # Every n-/m-vector crosses PCIe through ONE page-locked buffer allocated up front.
    # Pageable transfers make the runtime pin/unpin the numpy pages per call; the unmapping
    # evicts the process's GPU queues and stalls whatever is in flight for 60-90 ms
    # (measured: one such stall per LSMR solve, profiles/r1_ba_notes.txt).
    def _stage(self, k):
        if self._pin is None or self._pin.numel() < k:
            self._pin = torch.empty(max(int(k), 3 * self.n, self.m, 64), dtype=F64).pin_memory()
        return self._pin[:k]

    def upload(self, a, out=None):
        """host float64 array -> device vector (new tensor unless `out` is given)."""
        a = np.asarray(a, np.float64).ravel()
        st = self._stage(a.size)
        st.numpy()[:] = a
        if out is None:
            out = torch.empty(max(a.size, 1), dtype=F64, device=self.dev)
        out[:a.size].copy_(st, non_blocking=True)
        torch.cuda.current_stream().synchronize()          # the staging buffer is reused
        return out

    def download(self, t, k):
        """first k entries of a device vector -> new host float64 array."""
        st = self._stage(k)
        st.copy_(t[:k], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return st.numpy().copy()

    def upload_n(self, a, out=None):
        """host n-vector (reference order) -> device n-vector (internal point order)"""
        raw = self.upload(a, out=self.tmp_perm)
        if out is None:
            out = torch.empty(max(self.n, 1), dtype=F64, device=self.dev)
        if self.n:
            check(lib().iamx_vec_gather(self.n, _ptr(raw), _ptr(self.idx_h2i), _ptr(out),
                                        stream_ptr()), 'iamx_vec_gather')
        return out

    def download_n(self, t):
        """device n-vector (internal point order) -> host n-vector (reference order)"""
        if self.n:
            check(lib().iamx_vec_gather(self.n, _ptr(t), _ptr(self.idx_i2h), _ptr(self.tmp_perm),
                                        stream_ptr()), 'iamx_vec_gather')
        return self.download(self.tmp_perm, self.n)

    def upload_n_many(self, arrays):
        """upload_n of several host n-vectors (a scalar stands for a constant vector): one staging
        copy, one transfer, one wait"""
        n, out, full = self.n, [None] * len(arrays), []
        for k, a in enumerate(arrays):
            if np.ndim(a) == 0:
                out[k] = torch.full((max(n, 1),), float(a), dtype=F64, device=self.dev)
            else:
                full.append(k)
        if full and n:
            st = self._stage(len(full) * n)
            for j, k in enumerate(full):
                st.numpy()[j * n:(j + 1) * n] = np.asarray(arrays[k], np.float64).ravel()
            raw = self._many(len(full))
            raw[:len(full) * n].copy_(st, non_blocking=True)
            for j, k in enumerate(full):
                out[k] = torch.empty(n, dtype=F64, device=self.dev)
                check(lib().iamx_vec_gather(n, _ptr(raw[j * n:]), _ptr(self.idx_h2i), _ptr(out[k]),
                                            stream_ptr()), 'iamx_vec_gather')
            torch.cuda.current_stream().synchronize()      # the staging buffer is reused
        else:
            for k in full:
                out[k] = torch.empty(1, dtype=F64, device=self.dev)
        return out

    def download_n_many(self, vectors):
        """download_n of several device n-vectors: one transfer, one wait"""
        n, k = self.n, len(vectors)
        if not n:
            return [np.empty(0) for _ in vectors]
        raw = self._many(k)
        for j, t in enumerate(vectors):
            check(lib().iamx_vec_gather(n, _ptr(t), _ptr(self.idx_i2h), _ptr(raw[j * n:]), stream_ptr()),
                  'iamx_vec_gather')
        st = self._stage(k * n)
        st.copy_(raw[:k * n], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = st.numpy().copy()
        return [host[j * n:(j + 1) * n] for j in range(k)]

    def _many(self, k):
        if self._tmp_many is None or self._tmp_many.numel() < k * self.n:
            self._tmp_many = torch.empty(max(k * self.n, 1), dtype=F64, device=self.dev)
        return self._tmp_many

    def upload_m(self, a):
        """host m-vector in the order of this rank's slice of the reference's observation list
        (ascending reference index) -> device m-vector in the internal order"""
        a = np.asarray(a, np.float64).reshape(-1, 2)
        rank_of = np.argsort(np.argsort(self.local_obs, kind='stable'), kind='stable')
        return self.upload(a[rank_of].ravel())

    def download_m(self, t):
        """device m-vector (internal order) -> host, ascending reference index of this rank's slice"""
        a = self.download(t, self.m).reshape(-1, 2)
        out = np.empty_like(a)
        out[np.argsort(np.argsort(self.local_obs, kind='stable'), kind='stable')] = a
        return out.ravel()

    # ---- parameters ------------------------------------------------------------------
    def set_x(self, x):
        x = np.ascontiguousarray(x, np.float64)
        self.upload_n(x, out=self.x)
        if self.with_calib:
            c = x[self.C * 7 + self.P * 3:]
            cal = np.array([c[0], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]])
        else:
            cal = self.fixed_calib
        self.upload(cal, out=self.calib)

    def set_x_dev(self, x_dev):
        """device n-vector in the INTERNAL order -> current parameters (no host round trip)"""
        self.x[:self.n].copy_(x_dev[:self.n])
        if self.with_calib:
            if self._calib_idx is None:
                base = self.C * 7 + self.P * 3
                self._calib_idx = torch.tensor([base, base, base + 1, base + 2, base + 3, base + 4,
                                                base + 5, base + 6, base + 7], device=self.dev)
            self.calib.copy_(self.x[self._calib_idx])
        elif not self._fixed_calib_up:
            self.upload(self.fixed_calib, out=self.calib)
            self._fixed_calib_up = True

    def _cams_pts(self):
        return self.x[:self.C * 7], self.x[self.C * 7:self.C * 7 + self.P * 3]

    # ---- kernels ---------------------------------------------------------------------
    def residual(self, out=None):
        cams, pts = self._cams_pts()
        r = self.r if out is None else out
        if self.O:
            check(lib().iamx_ba_residual_prepared(_ptr(cams), self.C, _ptr(pts), self.P,
                                                  _ptr(self.cam_idx), _ptr(self.pt_idx),
                                                  _ptr(self.uv), self.O, _ptr(self.calib),
                                                  _ptr(self.cam_rt), _ptr(r), stream_ptr()),
                  'iamx_ba_residual_prepared')
        return r

    def bound_launchers(self):
        """(residual, residual_jac) as zero-argument callables with the ctypes arguments bound
        once -- for timing the kernels without the per-call python marshalling."""
        cams, pts = self._cams_pts()
        L = lib()
        a1 = (_ptr(cams), self.C, _ptr(pts), self.P, _ptr(self.cam_idx), _ptr(self.pt_idx),
              _ptr(self.uv), self.O, _ptr(self.calib), _ptr(self.cam_rt), _ptr(self.r), stream_ptr())
        a2 = (_ptr(cams), self.C, _ptr(pts), self.P, _ptr(self.cam_idx), _ptr(self.pt_idx),
              _ptr(self.uv), self.O, _ptr(self.calib), _ptr(self.r), _ptr(self.Jc), _ptr(self.Jp),
              _ptr(self.Jk), stream_ptr())
        f1, f2 = L.iamx_ba_residual_prepared, L.iamx_ba_residual_jac
        return (lambda: f1(*a1)), (lambda: f2(*a2))

    def residual_jac(self):
        cams, pts = self._cams_pts()
        self._acc_valid = False
        if self.O:
            check(lib().iamx_ba_residual_jac(_ptr(cams), self.C, _ptr(pts), self.P,
                                             _ptr(self.cam_idx), _ptr(self.pt_idx), _ptr(self.uv),
                                             self.O, _ptr(self.calib), _ptr(self.r), _ptr(self.Jc),
                                             _ptr(self.Jp), _ptr(self.Jk), stream_ptr()),
                  'iamx_ba_residual_jac')
        return self.r

    def jv(self, x_dev, y_dev):
        if self.O:
            check(lib().iamx_ba_jv(_ptr(self.Jc), _ptr(self.Jp), _ptr(self.Jk), _ptr(self.cam_idx),
                                   _ptr(self.pt_idx), self.O, self.C, self.P, _ptr(x_dev),
                                   _ptr(y_dev), stream_ptr()), 'iamx_ba_jv')

    def jtv(self, u_dev, out_dev, square=False):
        check(lib().iamx_ba_jtv(_ptr(self.Jc), _ptr(self.Jp), _ptr(self.Jk), _ptr(self.cam_ptr),
                                _ptr(self.pt_ptr), _ptr(self.pt_obs), self.O, self.C, self.P,
                                _ptr(u_dev), 1 if square else 0, _ptr(out_dev),
                                _ptr(self.scratch), stream_ptr()), 'iamx_ba_jtv')
        if self.world > 1:
            _dist.allreduce_sum_(out_dev[:self.n])

    def dot(self, a, b, n, reduce_ranks):
        if n == 0:
            v = torch.zeros(1, dtype=F64, device=self.dev)
        else:
            check(lib().iamx_vec_dot(n, _ptr(a), _ptr(b), _ptr(self.out1), _ptr(self.scratch),
                                     stream_ptr()), 'iamx_vec_dot')
            v = self.out1[:1]
        if reduce_ranks and self.world > 1:
            v = v.clone()
            _dist.allreduce_sum_(v)
        return float(v.item())

    def axpby(self, n, a, x, b, y):
        check(lib().iamx_vec_axpby(n, float(a), _ptr(x), float(b), _ptr(y), stream_ptr()),
              'iamx_vec_axpby')

    def mul2(self, n, a, b, out, c=None, d=None):
        check(lib().iamx_vec_mul2(n, _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(out), stream_ptr()),
              'iamx_vec_mul2')

    def lsmr_update(self, n, h, hbar, x, v, c_hbar, c_x, c_h):
        check(lib().iamx_vec_lsmr_update(n, _ptr(h), _ptr(hbar), _ptr(x), _ptr(v), float(c_hbar),
                                         float(c_x), float(c_h), stream_ptr()),
              'iamx_vec_lsmr_update')

    # ---- composite operations --------------------------------------------------------
    def cost_of_r(self, r):
        return 0.5 * self.dot(r, r, self.m, True)

    def grad(self):
        """J^T r -> host n-vector."""
        self.jtv(self.r, self.tmp_n)
        return self.download_n(self.tmp_n)

    def colnorm(self):
        """sqrt of the column sums of J.^2 -> host n-vector (scipy compute_jac_scale)."""
        self.jtv(self.r, self.tmp_n, square=True)
        return np.sqrt(self.download_n(self.tmp_n))

    def accumulate(self):
        """U, V, g_c, g_p of the Jacobian residual_jac() left in Jc / Jp, and of self.r
        (iamx_ba_accumulate: one pass, this rank's observations).  Valid until the next
        residual_jac()."""
        if self._acc_valid:
            return self.acc
        if self.acc is None:
            z = lambda k: torch.zeros(max(int(k), 1), dtype=F64, device=self.dev)
            self.acc = dict(U=z(self.C * 49), V=z(self.P * 9), g=z(self.C * 7 + self.P * 3))
        a = self.acc
        nc = self.C * 7
        check(lib().iamx_ba_accumulate(_ptr(self.Jc), _ptr(self.Jp), _ptr(self.r), _ptr(self.cam_ptr),
                                       _ptr(self.pt_ptr), _ptr(self.pt_obs), self.O, self.C, self.P,
                                       _ptr(a['U']), _ptr(a['V']), _ptr(a['g']),
                                       _lib.c_void_p(a['g'].data_ptr() + 8 * nc), stream_ptr()),
              'iamx_ba_accumulate')
        self._acc_valid = True
        return a

    def grad_dev(self):
        """J^T r as a device n-vector (internal order)"""
        if self.with_calib:
            self.jtv(self.r, self.tmp_n)
            return self.tmp_n[:self.n].clone()
        g = self.accumulate()['g'][:self.n].clone()
        if self.world > 1:
            # (the point entries are complete on the rank that owns the point and zero elsewhere)
            _dist.allreduce_sum_(g[:self.C * 7] if self.local_points else g)
        return g

    def colsq_dev(self):
        """column sums of J.^2 as a device n-vector (a view of the scratch vector: use it
        before the next operator application)"""
        if self.with_calib:
            self.jtv(self.r, self.tmp_n, square=True)
            return self.tmp_n[:self.n]
        a = self.accumulate()
        check(lib().iamx_ba_block_diag(_ptr(a['U']), _ptr(a['V']), self.C, self.P,
                                       _ptr(self.tmp_n), stream_ptr()), 'iamx_ba_block_diag')
        if self.world > 1:
            _dist.allreduce_sum_(self.tmp_n[:self.C * 7] if self.local_points else self.tmp_n[:self.n])
        return self.tmp_n[:self.n]

    def vec_ops(self):
        if getattr(self, '_vec_ops', None) is None:
            self._vec_ops = VecOps(self.dev)
        return self._vec_ops

    def jd(self, d_dev, s):
        """y = J diag(d) s for a device n-vector s (this rank's observations)"""
        V = self.vec_ops()
        V.mul(d_dev[:self.n], s[:self.n], out=self.tmp_n2[:self.n])
        y = torch.empty(max(self.m, 1), dtype=F64, device=self.dev)
        self.jv(self.tmp_n2, y)
        return y[:self.m]

    def q_pairs(self, *pairs):
        """queue inner products of observation-space vectors (this rank's share, summed over the
        ranks); the slot of the first (at most 8 per call)"""
        V = self.vec_ops()
        at = V.q_dots(*pairs, n=self.m, n_vectors=False) if self.m else V.q_zero(len(pairs))
        if self.world > 1:
            _dist.allreduce_sum_(V.out[at:at + len(pairs)])
        return at

    def q_gram(self, ys):
        """queue the upper triangle (row by row) of the Gram matrix of observation-space vectors;
        the slot of its first entry (k <= 3 vectors: at most 6 products)"""
        k = len(ys)
        return self.q_pairs(*[(ys[i], ys[j]) for i in range(k) for j in range(i, k)])

    def q_cost(self, r):
        """queue r.r over this rank's observations, summed over the ranks (cost = half of it)"""
        return self.q_gram([r[:self.m]])

    @staticmethod
    def gram_of(vals, at, k):
        G = np.zeros((k, k))
        it = iter(vals[at:])
        for i in range(k):
            for j in range(i, k):
                G[i, j] = G[j, i] = next(it)
        return G

    def gram_dev(self, d_dev, vectors):
        """gram() for device n-vectors (internal order); one host read for the whole matrix"""
        V = self.vec_ops()
        V.begin()
        at = self.q_gram([self.jd(d_dev, s) for s in vectors])
        return self.gram_of(V.fetch(), at, len(vectors))

    def gram(self, d_host, vectors, d_dev=None):
        """G[i][j] = (J diag(d) s_i) . (J diag(d) s_j), summed over ranks.  With `d_dev` (d
        already on the device) the scaling is applied there instead of on the host."""
        k = len(vectors)
        ys = []
        for s in vectors:
            if d_dev is None:
                vs = self.upload_n(d_host * s)
            else:
                vs = self.upload_n(s)
                self.mul2(self.n, d_dev, vs, vs)
            y = torch.empty(max(self.m, 1), dtype=F64, device=self.dev)
            self.jv(vs, y)
            ys.append(y)
        G = np.zeros((k, k))
        for i in range(k):
            for j in range(i, k):
                G[i, j] = G[j, i] = self.dot(ys[i], ys[j], self.m, True)
        return G


def _sym_ortho(a, b):
    """stable Givens rotation (scipy/sparse/linalg/_isolve/lsqr.py:_sym_ortho)."""
    if b == 0:
        return np.sign(a), 0.0, abs(a)
    if a == 0:
        return 0.0, np.sign(b), abs(b)
    if abs(b) > abs(a):
        tau = a / b
        s = np.sign(b) / np.sqrt(1 + tau * tau)
        c = s * tau
        r = b / s
    else:
        tau = b / a
        c = np.sign(a) / np.sqrt(1 + tau * tau)
        s = c * tau
        r = a / c
    return c, s, r


def lsmr_device(prob, d_dev, dreg_dev, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=None,
                to_host=True):
    """min || [J diag(d); diag(dreg)] x - [r; 0] ||  on the device; returns host x and stats.
    Follows scipy/sparse/linalg/_isolve/lsmr.py (Fong & Saunders 2011) step by step."""
    n, m = prob.n, prob.m
    dev = prob.dev
    z = lambda k: torch.zeros(max(k, 1), dtype=F64, device=dev)
    u1, u2, v, h, hbar, x = z(m), z(n), z(n), z(n), z(n), z(n)
    if maxiter is None:
        maxiter = n
    u1[:m].copy_(prob.r[:m])
    normb = np.sqrt(prob.dot(u1, u1, m, True))
    beta = normb

    def matvec_into_u(alpha):           # u = A v - alpha u
        prob.mul2(n, d_dev, v, prob.tmp_n)
        prob.jv(prob.tmp_n, prob.tmp_m)
        prob.axpby(m, 1.0, prob.tmp_m, -alpha, u1)
        prob.mul2(n, dreg_dev, v, prob.tmp_n2)
        prob.axpby(n, 1.0, prob.tmp_n2, -alpha, u2)

    def rmatvec_into_v(beta_):          # v = A^T u - beta v
        prob.jtv(u1, prob.tmp_n)
        prob.mul2(n, d_dev, prob.tmp_n, prob.tmp_n2, dreg_dev, u2)
        prob.axpby(n, 1.0, prob.tmp_n2, -beta_, v)

    def unorm():
        return np.sqrt(prob.dot(u1, u1, m, True) + prob.dot(u2, u2, n, False))

    if beta > 0:
        prob.axpby(m, 0.0, u1, 1.0 / beta, u1)
        rmatvec_into_v(0.0)
        alpha = np.sqrt(prob.dot(v, v, n, False))
    else:
        alpha = 0.0
    if alpha > 0:
        prob.axpby(n, 0.0, v, 1.0 / alpha, v)

    itn = 0
    zetabar = alpha * beta
    alphabar = alpha
    rho = rhobar = cbar = 1.0
    sbar = 0.0
    h.copy_(v)
    betadd, betad, rhodold, tautildeold, thetatilde, zeta, dsum = beta, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0
    normA2 = alpha * alpha
    maxrbar, minrbar = 0.0, 1e+100
    normA, condA, normx = np.sqrt(normA2), 1.0, 0.0
    istop = 0
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    normr = beta
    normar = alpha * beta
    result = (lambda t: prob.download_n(t)) if to_host else (lambda t: t[:n].clone())
    if normar == 0 or normb == 0:
        return result(x), istop, itn, normr, normar

    while itn < maxiter:
        itn += 1
        matvec_into_u(alpha)
        beta = unorm()
        if beta > 0:
            prob.axpby(m, 0.0, u1, 1.0 / beta, u1)
            prob.axpby(n, 0.0, u2, 1.0 / beta, u2)
            rmatvec_into_v(beta)
            alpha = np.sqrt(prob.dot(v, v, n, False))
            if alpha > 0:
                prob.axpby(n, 0.0, v, 1.0 / alpha, v)
        chat, shat, alphahat = _sym_ortho(alphabar, 0.0)
        rhoold = rho
        c, s, rho = _sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold, zetaold = rhobar, zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = _sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        prob.lsmr_update(n, h, hbar, x, v, -(thetabar * rho / (rhoold * rhobarold)),
                         zeta / (rho * rhobar), -(thetanew / rho))
        betaacute = chat * betadd
        betacheck = -shat * betadd
        betahat = c * betaacute
        betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = _sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        dsum += betacheck * betacheck
        normr = np.sqrt(dsum + (betad - taud) ** 2 + betadd * betadd)
        normA2 += beta * beta
        normA = np.sqrt(normA2)
        normA2 += alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)
        normar = abs(zetabar)
        normx = np.sqrt(prob.dot(x, x, n, False))
        test1 = normr / normb
        test2 = normar / (normA * normr) if (normA * normr) != 0 else np.inf
        test3 = 1.0 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= maxiter:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop > 0:
            break
    return result(x), istop, itn, normr, normar


# state block layout of iamx_ba_lsmr_iterate (csrc/ba_linalg.hip, enums S_* / R_*)
_S = {k: i for i, k in enumerate(
    'ALPHA BETA ZETABAR ALPHABAR RHO RHOBAR CBAR SBAR BETADD BETAD RHODOLD TAUTILDEOLD THETATILDE '
    'ZETA D NORMA2 MAXRBAR MINRBAR ITN NORMR NORMAR NORMA CONDA'.split())}
_R = {k: 2 * len(_S) + i for i, k in enumerate(
    'ATOL BTOL CTOL MAXITER NORMB ISTOP ITN NORMR NORMAR NORMA CONDA NORMX'.split())}


def lsmr_device_fused(prob, d_dev, dreg_dev, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=None,
                      chunk=64, to_host=True):
    """Same recurrence as lsmr_device with the scalars resident on the device.  Single rank:
    iterations are enqueued `chunk` at a time by iamx_ba_lsmr_iterate (3 launches each).
    Several ranks (observations sharded by point): every iteration is three
    iamx_ba_lsmr_phase calls with two RCCL all-reduces between them (|ut1|^2 and J^T ut1), all
    enqueued asynchronously.  Either way the host only reads the state block between chunks.
    No calibration columns."""
    n, m = prob.n, prob.m
    dev = prob.dev
    L = lib()
    multi = prob.world > 1
    if maxiter is None:
        maxiter = n
    chunk = max(2, int(os.environ.get('IAMX_LSMR_CHUNK', chunk)) & ~1)
    ws = prob.lsmr_ws
    if ws is None:
        z = lambda k: torch.zeros(max(int(k), 1), dtype=F64, device=dev)
        ws = prob.lsmr_ws = dict(u1=z(m), u2=z(n), vt=z(n), h=z(n), hbar=z(n), x=z(n),
                                 ctab=z(prob.C * 32), ptab=z(prob.P * 6),
                                 state=z(L.iamx_ba_lsmr_state_size()),
                                 part=z(L.iamx_ba_lsmr_partials_size(prob.C, prob.P)),
                                 xr=z(4), tbuf=z(n), eprod=z(3 * prob.O))
    u1, u2, vt, h, hbar, x = (ws[k] for k in ('u1', 'u2', 'vt', 'h', 'hbar', 'x'))
    if 'dreg' not in ws:
        ws['dreg'] = torch.zeros(max(n, 1), dtype=F64, device=dev)
    ws['dreg'][:n].copy_(dreg_dev[:n])           # a fixed address for the captured launches
    dreg_dev = ws['dreg']
    ph = _Phase(prob, 'lsmr:init')
    ph.__enter__()
    # the tables of the matrix-free operator at the current parameters (the point J was
    # evaluated at: residual_jac() ran on prob.x)
    cams, pts = prob._cams_pts()
    check(L.iamx_ba_lsmr_prepare(_ptr(cams), _ptr(pts), _ptr(d_dev), prob.C, prob.P,
                                 _ptr(ws['ctab']), _ptr(ws['ptab']), stream_ptr()),
          'iamx_ba_lsmr_prepare')
    u1[:m].copy_(prob.r[:m])
    u2.zero_(); hbar.zero_(); x.zero_()
    normb = np.sqrt(prob.dot(u1, u1, m, True))
    beta = normb
    alpha = 0.0
    if beta > 0:
        prob.jtv(u1, prob.tmp_n)                   # all-reduced over ranks inside
        prob.mul2(n, d_dev, prob.tmp_n, vt)
        prob.axpby(n, 0.0, vt, 1.0 / beta, vt)
        alpha = np.sqrt(prob.dot(vt, vt, n, False))
    if alpha > 0:
        prob.axpby(n, 1.0 / alpha, vt, 0.0, h)
    normar = alpha * beta
    result = (lambda t: prob.download_n(t)) if to_host else (lambda t: t[:n].clone())
    if normar == 0 or normb == 0:
        return result(x), 0, 0, beta, normar
    st = np.zeros(L.iamx_ba_lsmr_state_size())
    for k, val in dict(ALPHA=alpha, BETA=beta, ZETABAR=alpha * beta, ALPHABAR=alpha, RHO=1.0,
                       RHOBAR=1.0, CBAR=1.0, SBAR=0.0, BETADD=beta, RHODOLD=1.0,
                       NORMA2=alpha * alpha, MINRBAR=1e100, NORMR=beta, NORMAR=normar,
                       NORMA=abs(alpha), CONDA=1.0).items():
        st[_S[k]] = val
    for k, val in dict(ATOL=atol, BTOL=btol, CTOL=1.0 / conlim if conlim > 0 else 0.0,
                       MAXITER=float(maxiter), NORMB=normb).items():
        st[_R[k]] = val
    prob.upload(st, out=ws['state'])
    ph.__exit__()
    ph = _Phase(prob, 'lsmr:iterate')
    ph.__enter__()
    common = (_ptr(ws['ctab']), _ptr(ws['ptab']), _ptr(prob.calib), _ptr(prob.pt_idx),
              _ptr(prob.cam_ptr), _ptr(prob.pt_ptr), _ptr(prob.pt_obs), _ptr(prob.slot_cp), prob.O,
              prob.C, prob.P, _ptr(dreg_dev), _ptr(u1), _ptr(u2), _ptr(vt), _ptr(h), _ptr(hbar),
              _ptr(x), _ptr(ws['state']), _ptr(ws['part']))
    mcommon = common[:11] + (prob.pt_lo, prob.pt_hi) + common[11:]

    def enqueue_launches():
        check(L.iamx_ba_lsmr_iterate(*common, _ptr(ws['xr']), _ptr(ws['tbuf']), _ptr(ws['eprod']),
                                     chunk, stream_ptr()), 'iamx_ba_lsmr_iterate')

    # Single rank: the 3 x chunk launches of a chunk are captured once into a HIP graph (every
    # pointer they take lives in prob.lsmr_ws; dreg is copied into it) and replayed per chunk:
    # one graph launch instead of 192 kernel launches from the host.
    graph = None
    if not multi and prob.use_graph:
        key = (chunk, prob.O, dreg_dev.data_ptr())
        if ws.get('graph_key') != key:
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), _lib.hold_stream(fresh=True):     # (the capture stream)
                enqueue_launches()
            ws['graph'], ws['graph_key'] = g, key
        graph = ws['graph']

    def enqueue_chunk():
        if graph is not None:
            graph.replay()
        elif not multi:
            enqueue_launches()
        else:
            # two small all-reduces per iteration: 2 scalars, then the raw camera part of
            # J^T ut' + one scalar (7 C + 1 doubles = 157 KB at configs[3]); the point part of
            # every n-vector stays on the rank that owns the points
            xr, tbuf = ws['xr'], ws['tbuf']
            ncam = prob.C * 7
            for it in range(chunk):
                par = it & 1
                for phase in (0, 1, 2):
                    check(L.iamx_ba_lsmr_phase(*mcommon, _ptr(xr), _ptr(tbuf), _ptr(ws['eprod']),
                                               phase, par, stream_ptr()), 'iamx_ba_lsmr_phase')
                    if phase == 0:
                        _dist.allreduce_sum_(xr[:2])
                    elif phase == 1:
                        _dist.allreduce_sum_(tbuf[:ncam + 1])
            prob.fused_phase_iterations += chunk

    # The state block is copied out behind every chunk into its own pinned slot; the NEXT chunk
    # is enqueued before the host waits for that copy, so the device never idles while the host
    # enqueues (a chunk behind a latched stop is made of no-ops).
    ns = st.size
    if prob._state_pin is None:
        prob._state_pin = (torch.empty(ns, dtype=F64).pin_memory(), torch.empty(ns, dtype=F64).pin_memory(),
                           torch.cuda.Event(), torch.cuda.Event())
    slots, events = prob._state_pin[:2], prob._state_pin[2:]

    def snapshot(k):
        slots[k].copy_(ws['state'][:ns], non_blocking=True)
        events[k].record()

    enqueue_chunk()
    snapshot(0)
    cur = 0
    for _ in range(int(maxiter) // chunk + 3):
        enqueue_chunk()
        snapshot(cur ^ 1)
        events[cur].synchronize()
        st = slots[cur].numpy().copy()
        if st[_R['ISTOP']] != 0:
            break
        cur ^= 1
    else:
        raise _lib.IamxError('fused LSMR did not latch a stop condition')
    if multi:
        # the point entries of x live on their owners (the others never left 0): complete x
        _dist.allreduce_sum_(x[prob.C * 7:n])
    ph.__exit__()
    if st[_R['ISTOP']] == 8:
        raise _lib.IamxError('fused LSMR broke down (NaN in the recurrence)')
    return (result(x), int(st[_R['ISTOP']]), int(st[_R['ITN']]), float(st[_R['NORMR']]),
            float(st[_R['NORMAR']]))


def schur_solve(prob, d_dev, dreg_dev, eta=None, maxiter=None, chunk=4, to_host=True, qtol=None):
    """The Gauss-Newton step of  min || [J diag(d); diag(dreg)] p - [r; 0] ||  through the normal
    equations (csrc/ba_schur.hip): points eliminated exactly, the reduced camera system solved by
    block-Jacobi preconditioned conjugate gradients until the preconditioned residual has
    dropped by the factor `eta` or the decrease of the quadratic model has levelled off
    (iteration i lowers it by less than qtol / i of the total so far: the truncated-Newton test
    of Nash & Sofer) -- SciPy's TRF only uses the step to span a 2-D subspace together with the
    gradient and minimises the exact model in it (trf.py:315-327), so what the step has to
    deliver is model decrease, not a small residual in the weakly determined (gauge-like)
    directions --, the point part back-substituted.  J and r are what residual_jac() left on
    the device.  Several ranks (observations sharded by point): one all-reduce of C x 35 doubles
    per solve and one of C x 7 doubles per CG iteration; the recurrence is replicated.
    Returns (step, istop, iterations, sqrt(r.z), 0.0) like lsmr_device()."""
    n, C, P, O = prob.n, prob.C, prob.P, prob.O
    dev = prob.dev
    L = lib()
    multi = prob.world > 1
    if qtol is None:
        # the solver's own settings come as a pair; an explicit eta alone means "to this residual"
        qtol = prob.schur_qtol if eta is None else 0.0
    eta = prob.schur_eta if eta is None else float(eta)
    maxiter = int(prob.schur_max_iter if maxiter is None else maxiter)
    ws = prob.schur_ws
    wc = prob.with_calib
    nq = C * 7 + (8 if wc else 0)        # the reduced system: cameras (+ the calibration block)
    Jk = _ptr(prob.Jk) if wc else None
    if ws is None:
        z = lambda k: torch.zeros(max(int(k), 1), dtype=F64, device=dev)
        ns = int(L.iamx_ba_schur_state_size())
        ws = prob.schur_ws = dict(Y=z(P * 6), yg=z(P * 3), zp=z(P * 3), sraw=z(C * 35 + 44), minv=z(C * 28 + 36),
                                  t=z(2 * O), qraw=z(C * 7 + 8), part=z(2 * C + 2), x=z(C * 7 + 8), r=z(C * 7 + 8),
                                  z=z(C * 7 + 8), p=z(C * 7 + 8), y=z(C * 7 + 8), state=z(ns), step=z(n),
                                  ck=z(C * 44) if wc else None,
                                  pin=(torch.empty(ns, dtype=F64).pin_memory(),
                                       torch.empty(ns, dtype=F64).pin_memory()),
                                  ev=(torch.cuda.Event(), torch.cuda.Event()))
    ck = _ptr(ws['ck']) if wc else None
    ph = _Phase(prob, 'schur:prepare')
    ph.__enter__()
    a = prob.accumulate()
    nc = C * 7
    gp = _lib.c_void_p(a['g'].data_ptr() + 8 * nc)
    check(L.iamx_ba_schur_prepare(_ptr(prob.Jc), _ptr(prob.Jp), Jk, _ptr(prob.r), _ptr(prob.cam_ptr),
                                  _ptr(prob.pt_idx), O, C, P, _ptr(a['V']), gp, _ptr(d_dev),
                                  _ptr(dreg_dev), _ptr(ws['Y']), _ptr(ws['yg']), _ptr(ws['zp']),
                                  _ptr(ws['sraw']), ck, stream_ptr()), 'iamx_ba_schur_prepare')
    if multi:
        _dist.allreduce_sum_(ws['sraw'][:C * 35 + (44 if wc else 0)])
    check(L.iamx_ba_schur_factor(_ptr(ws['sraw']), _ptr(d_dev), _ptr(dreg_dev), C, P, 1 if wc else 0,
                                 eta, qtol, maxiter,
                                 _ptr(ws['minv']), _ptr(ws['x']), _ptr(ws['r']), _ptr(ws['z']),
                                 _ptr(ws['p']), _ptr(ws['y']), _ptr(ws['state']), stream_ptr()),
          'iamx_ba_schur_factor')
    ph.__exit__()
    ph = _Phase(prob, 'schur:iterate')
    ph.__enter__()
    it_args = (_ptr(prob.Jc), _ptr(prob.Jp), Jk, _ptr(prob.cam_idx), _ptr(prob.pt_idx), _ptr(prob.cam_ptr),
               _ptr(prob.pt_ptr), _ptr(prob.pt_obs), O, C, P, _ptr(d_dev), _ptr(dreg_dev),
               _ptr(ws['Y']), _ptr(ws['minv']), _ptr(ws['t']), _ptr(ws['zp']), _ptr(ws['qraw']),
               _ptr(ws['part']), ck, _ptr(ws['x']), _ptr(ws['r']), _ptr(ws['z']), _ptr(ws['p']), _ptr(ws['y']),
               _ptr(ws['state']))

    enqueued = [0]          # iterations enqueued since the factorisation (selects the state buffer)

    # Chunk sizes: the first one is as long as the previous solve took (the counts drift slowly
    # from one outer iteration to the next; at most 6), the following ones short -- whatever is
    # enqueued behind the latched stop is launched as no-ops, 5 per iteration (and, on several
    # ranks, all-reduced).
    plan = [max(2, min(int(prob.schur_last_itn or chunk), 6, maxiter)), 2, 2, 2, 2]

    def enqueue_chunk():
        k = enqueued[0]
        chunk = plan.pop(0) if plan else 4
        enqueued[0] += chunk
        if not multi:
            check(L.iamx_ba_schur_iterate(*it_args, k, chunk, -1, stream_ptr()), 'iamx_ba_schur_iterate')
            return
        for i in range(chunk):
            check(L.iamx_ba_schur_iterate(*it_args, k + i, 1, 0, stream_ptr()), 'iamx_ba_schur_iterate')
            _dist.allreduce_sum_(ws['qraw'][:nq])
            check(L.iamx_ba_schur_iterate(*it_args, k + i, 1, 1, stream_ptr()), 'iamx_ba_schur_iterate')
        prob.fused_phase_iterations += chunk

    slots, events = ws['pin'], ws['ev']
    ns = slots[0].numel()

    half = ns // 2
    where = [0, 0]          # which state buffer is current behind the chunk a slot snapshots

    def snapshot(k):
        slots[k].copy_(ws['state'][:ns], non_blocking=True)
        events[k].record()
        where[k] = enqueued[0] & 1

    # the NEXT chunk is enqueued before the host waits for the state behind this one (a chunk
    # behind a latched stop is made of no-ops)
    enqueue_chunk()
    snapshot(0)
    cur = 0
    for _ in range(maxiter // 2 + 8):
        enqueue_chunk()
        snapshot(cur ^ 1)
        events[cur].synchronize()
        st = slots[cur].numpy()[where[cur] * half:(where[cur] + 1) * half].copy()
        if st[3] != 0:
            break
        cur ^= 1
    else:
        raise _lib.IamxError('Schur CG did not latch a stop condition')
    ph.__exit__()
    ph = _Phase(prob, 'schur:finish')
    ph.__enter__()
    check(L.iamx_ba_schur_finish(_ptr(prob.Jc), _ptr(prob.Jp), Jk, _ptr(prob.cam_idx), _ptr(prob.pt_ptr),
                                 _ptr(prob.pt_obs), O, C, P, prob.pt_lo, prob.pt_hi, _ptr(d_dev),
                                 _ptr(ws['Y']), _ptr(ws['yg']), _ptr(ws['x']), _ptr(ws['y']),
                                 _ptr(ws['t']), _ptr(ws['step']), stream_ptr()),
          'iamx_ba_schur_finish')
    if multi and not prob.local_points:
        _dist.allreduce_sum_(ws['step'][nc:nc + 3 * P])      # (the calibration entries are replicated)
    ph.__exit__()
    itn = int(st[2])
    prob.schur_last_itn = itn
    prob.inner_iterations.append(itn)
    prob.inner_stops.append(int(st[3]))
    step = ws['step']
    result = prob.download_n(step) if to_host else step[:n].clone()
    return result, int(st[3]), itn, float(np.sqrt(max(st[0], 0.0))), 0.0


def lsmr(prob, d_dev, dreg_dev, **opts):
    """The Gauss-Newton subproblem of one outer iteration: the Schur-complement solve
    (prob.inner == 'schur', default), else LSMR on the whole system as SciPy does it -- fused
    host-free iterations when the problem allows it, else the stepwise form (the only LSMR form
    with calibration columns)."""
    if prob.inner == 'schur' and prob.C and (prob.O or prob.world > 1):
        # (with calibration columns: the bordered form, csrc/ba_schur.hip)
        return schur_solve(prob, d_dev, dreg_dev, to_host=opts.get('to_host', True))
    if not prob.with_calib and not prob.force_stepwise_lsmr and (prob.O or prob.world > 1):
        r = lsmr_device_fused(prob, d_dev, dreg_dev, **opts)
    else:
        r = lsmr_device(prob, d_dev, dreg_dev, **opts)
    prob.inner_iterations.append(r[2])
    return r


# --------------------------------------------------------------------------------------
# reflective step selection (scipy/optimize/_lsq/trf.py select_step) on Gram matrices
# --------------------------------------------------------------------------------------
def _select_step(prob, x, d, diag_h, g_h, p, p_h, Delta, lb, ub, theta, d_dev=None):
    from scipy.optimize._lsq.common import (in_bounds, intersect_trust_region,
                                            minimize_quadratic_1d, step_size_to_bound)

    def quad(G, i, s):        # 0.5 s^T (J_h^T J_h + D) s + g_h^T s
        return 0.5 * (G[i, i] + np.dot(s * diag_h, s)) + np.dot(g_h, s)

    if in_bounds(x + p, lb, ub):
        G = prob.gram(d, [p_h], d_dev)
        return p, p_h, -quad(G, 0, p_h)

    p_stride, hits = step_size_to_bound(x, p, lb, ub)
    r_h = np.copy(p_h)
    r_h[hits.astype(bool)] *= -1
    r = d * r_h
    p = p * p_stride
    p_h = p_h * p_stride
    x_on_bound = x + p
    _, to_tr = intersect_trust_region(p_h, r_h, Delta)
    to_bound, _ = step_size_to_bound(x_on_bound, r, lb, ub)
    r_stride = min(to_bound, to_tr)
    if r_stride > 0:
        r_stride_l = (1 - theta) * p_stride / r_stride
        r_stride_u = theta * to_bound if r_stride == to_bound else to_tr
    else:
        r_stride_l, r_stride_u = 0, -1

    ag_h = -g_h
    G = prob.gram(d, [p_h, r_h, ag_h], d_dev)   # all three model directions in one go
    if r_stride_l <= r_stride_u:
        # 1-d quadratic along r_h from s0 = p_h (scipy build_quadratic_1d with s0)
        a = 0.5 * (G[1, 1] + np.dot(r_h * diag_h, r_h))
        b = np.dot(g_h, r_h) + G[0, 1] + np.dot(p_h * diag_h, r_h)
        c0 = 0.5 * (G[0, 0] + np.dot(p_h * diag_h, p_h)) + np.dot(g_h, p_h)
        r_stride, r_value = minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c=c0)
        r_h = r_h * r_stride + p_h
        r = r_h * d
    else:
        r_value = np.inf

    p = p * theta
    p_h_t = p_h * theta
    p_value = 0.5 * theta * theta * (G[0, 0] + np.dot(p_h * diag_h, p_h)) + theta * np.dot(g_h, p_h)

    ag = d * ag_h
    to_tr = Delta / norm(ag_h)
    to_bound, _ = step_size_to_bound(x, ag, lb, ub)
    ag_stride = theta * to_bound if to_bound < to_tr else to_tr
    a = 0.5 * (G[2, 2] + np.dot(ag_h * diag_h, ag_h))
    b = np.dot(g_h, ag_h)
    ag_stride, ag_value = minimize_quadratic_1d(a, b, 0, ag_stride)
    ag_h = ag_h * ag_stride
    ag = ag * ag_stride

    if p_value < r_value and p_value < ag_value:
        return p, p_h_t, -p_value
    elif r_value < p_value and r_value < ag_value:
        return r, r_h, -r_value
    return ag, ag_h, -ag_value


def trf_device(prob, x0, lb, ub, **kw):
    """scipy/optimize/_lsq/trf.py trf_bounds with tr_solver='lsmr', x_scale='jac',
    loss='linear', on a DeviceBA problem.

    The host-side O(n) vector algebra runs with the BLAS thread pool limited to one thread: a
    multi-threaded np.dot/norm leaves ~100 OpenBLAS workers spinning for a while, and the
    device queue then stalls 60-90 ms in the next LSMR solve (measured, tools/diag_stall2.py;
    profiles/r1_ba_notes.txt).  The vectors are memory-bound; one thread loses nothing."""
    fn = _trf_host if prob.host_logic else _trf_device
    with _lib.hold_stream():              # (one stream for the whole solve; see _lib.hold_stream)
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:               # pragma: no cover
            return fn(prob, x0, lb, ub, **kw)
        with threadpool_limits(limits=1, user_api='blas'):
            return fn(prob, x0, lb, ub, **kw)


def _trf_host(prob, x0, lb, ub, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0,
              callback=None, lsmr_opts=None):
    """the O(n) vector logic in numpy with SciPy's own helpers (DeviceBA.host_logic = True):
    the cross-check of _trf_device, and ~40 % slower at config 4 (every n-vector crosses PCIe)"""
    from scipy.optimize import OptimizeResult
    from scipy.optimize._lsq.common import (CL_scaling_vector, check_termination,
                                            find_active_constraints, make_strictly_feasible,
                                            minimize_quadratic_1d, print_header_nonlinear,
                                            print_iteration_nonlinear, solve_trust_region_2d,
                                            update_tr_radius)
    from scipy.linalg import qr
    lsmr_opts = dict(lsmr_opts or {})
    n = prob.n
    x = make_strictly_feasible(np.asarray(x0, np.float64).copy(), lb, ub)
    prob.set_x(x)
    prob.residual_jac()
    nfev = njev = 1
    cost = prob.cost_of_r(prob.r)
    g = prob.grad()
    scale_inv = prob.colnorm()
    scale_inv[scale_inv == 0] = 1
    scale = 1 / scale_inv

    v, dv = CL_scaling_vector(x, g, lb, ub)
    v[dv != 0] *= scale_inv[dv != 0]
    Delta = norm(x * scale_inv / v ** 0.5)
    if Delta == 0:
        Delta = 1.0
    g_norm = norm(g * v, ord=np.inf)
    if max_nfev is None:
        max_nfev = n * 100
    termination_status = None
    iteration = 0
    step_norm = actual_reduction = None
    lsmr_iters = 0
    r_new = torch.empty_like(prob.r)
    if verbose == 2 and prob.rank == 0:
        print_header_nonlinear()

    while True:
        v, dv = CL_scaling_vector(x, g, lb, ub)
        g_norm = norm(g * v, ord=np.inf)
        if g_norm < gtol:
            termination_status = 1
        if verbose == 2 and prob.rank == 0:
            print_iteration_nonlinear(iteration, nfev, cost, actual_reduction, step_norm, g_norm)
        if termination_status is not None or nfev == max_nfev:
            break

        with _Phase(prob, 'scaling+reg'):
            v[dv != 0] *= scale_inv[dv != 0]
            d = v ** 0.5 * scale
            diag_h = g * dv * scale
            g_h = d * g

            d_dev = prob.upload_n(d)
            # regularisation term (trf.py: build_quadratic_1d along -g_h)
            G = prob.gram(d, [g_h], d_dev)
            a = 0.5 * (G[0, 0] + np.dot(g_h * diag_h, g_h))
            b = -np.dot(g_h, g_h)
            to_tr = Delta / norm(g_h)
            ag_value = minimize_quadratic_1d(a, b, 0, to_tr)[1]
            reg_term = -ag_value / Delta ** 2

        with _Phase(prob, 'lsmr'):
            dreg_dev = prob.upload_n((diag_h + reg_term) ** 0.5)
            gn_h, _istop, itn, _nr, _nar = lsmr(prob, d_dev, dreg_dev, **lsmr_opts)
            lsmr_iters += itn
        with _Phase(prob, 'subspace'):
            S = np.vstack((g_h, gn_h)).T
            S, _ = qr(S, mode='economic')
            GS = prob.gram(d, [S[:, 0].copy(), S[:, 1].copy()], d_dev)
            B_S = GS + np.dot(S.T * diag_h, S)
            g_S = S.T.dot(g_h)

        theta = max(0.995, 1 - g_norm)
        actual_reduction = -1
        while actual_reduction <= 0 and nfev < max_nfev:
            with _Phase(prob, 'select_step'):
                p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
                p_h = S.dot(p_S)
                p = d * p_h
                step, step_h, predicted_reduction = _select_step(prob, x, d, diag_h, g_h, p, p_h,
                                                                 Delta, lb, ub, theta, d_dev)
            with _Phase(prob, 'fun'):
                x_new = make_strictly_feasible(x + step, lb, ub, rstep=0)
                prob.set_x(x_new)
                prob.residual(out=r_new)
                nfev += 1
                step_h_norm = norm(step_h)
                cost_new = prob.cost_of_r(r_new)
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual_reduction = cost - cost_new
            Delta_new, ratio = update_tr_radius(Delta, actual_reduction, predicted_reduction,
                                                step_h_norm, step_h_norm > 0.95 * Delta)
            step_norm = norm(step)
            termination_status = check_termination(actual_reduction, cost, step_norm, norm(x),
                                                   ratio, ftol, xtol)
            if termination_status is not None:
                break
            Delta = Delta_new

        if actual_reduction > 0:
            with _Phase(prob, 'jac+grad'):
                x = x_new
                cost = cost_new                   # (the device already holds x_new: the accepted
                prob.residual_jac()               #  trial was the last one evaluated)
                njev += 1
                g = prob.grad()
                cn = prob.colnorm()
                scale_inv = np.maximum(scale_inv, cn)    # compute_jac_scale(J, scale_inv_old)
                scale = 1 / scale_inv
            if callback is not None:
                callback(x, cost)
        else:
            prob.set_x(x)
            prob.residual()
            step_norm = 0
            actual_reduction = 0
        iteration += 1

    if termination_status is None:
        termination_status = 0
    active_mask = find_active_constraints(x, lb, ub, rtol=xtol)
    return OptimizeResult(x=x, cost=cost, grad=g, optimality=g_norm, active_mask=active_mask,
                          nfev=nfev, njev=njev, status=termination_status,
                          lsmr_iterations=lsmr_iters, iterations=iteration)


# --------------------------------------------------------------------------------------
# the same outer iteration with every n-vector resident on the device (internal order)
# --------------------------------------------------------------------------------------
class VecOps(object):
    """The n-vector kernels of the TRF outer loop (csrc/trf_vec.hip) on float64 device tensors:
    torch holds the memory, libiamx does the arithmetic.  Scalars come back through one small
    read per call (`dots` returns up to 8 inner products at once)."""

    def __init__(self, dev):
        self.dev = dev
        self.scratch = torch.empty(int(lib().iamx_vec_scratch_doubles()), dtype=F64, device=dev)
        # scalar results: SLOTS doubles read by the host in ONE transfer (begin / q_* / fetch)
        self.out = torch.empty(self.SLOTS, dtype=F64, device=dev)
        self._q = 0
        # several ranks with rank-local point parts (_trf_device): (rank, first point entry,
        # number of point entries).  An n-vector then holds the camera part (replicated), this
        # rank's points, and ZEROS for the points of the other ranks; a sum over the entries is
        # rank 0's whole vector + the point part of the others, all-reduced as a scalar.
        self.part = None

    SLOTS = 32

    def _reduce(self, k, op, at=0):
        """k scalars of self.out over the ranks (identity on one rank)"""
        if self.part is not None:
            _dist.allreduce_(self.out[at:at + k], op)

    # ---- queued scalars: every q_* launch writes its results into the next free slots of
    # self.out and returns the index of the first; fetch() is the one host read for all of them
    # (each read drains the queue: the TRF loop pays for round trips, not for arithmetic)
    def begin(self):
        self._q = 0

    def _take(self, k):
        at = self._q
        if at + k > self.SLOTS:
            raise _lib.IamxError('VecOps: more than %d queued scalars' % self.SLOTS)
        self._q = at + k
        return at

    def _slot_ptr(self, at):
        return _lib.c_void_p(self.out.data_ptr() + 8 * at)

    def fetch(self):
        vals = self.out[:self._q].tolist()
        self._q = 0
        return vals

    def new(self, like):
        return torch.empty(like.numel(), dtype=F64, device=self.dev)

    def lincomb(self, a, x, b=0.0, y=None, c=0.0, z=None, out=None):
        out = self.new(x) if out is None else out
        check(lib().iamx_vec_lincomb(x.numel(), float(a), _ptr(x), float(b), _ptr(y), float(c),
                                     _ptr(z), _ptr(out), stream_ptr()), 'iamx_vec_lincomb')
        return out

    def mul(self, x, y=None, s=1.0, out=None):
        out = self.new(x) if out is None else out
        check(lib().iamx_vec_mul(x.numel(), float(s), _ptr(x), _ptr(y), _ptr(out), stream_ptr()),
              'iamx_vec_mul')
        return out

    def sqrt_shift(self, x, shift):
        out = self.new(x)
        check(lib().iamx_vec_sqrt_shift(x.numel(), _ptr(x), float(shift), _ptr(out), stream_ptr()),
              'iamx_vec_sqrt_shift')
        return out

    def dots(self, *terms, n=None, n_vectors=True):
        """terms: (a, b) or (a, w, b) -> [sum a.*b (.*w)] as python floats, one host read.
        n_vectors=False: the operands are not n-vectors (observation space: the caller reduces)"""
        self.begin()
        self.q_dots(*terms, n=n, n_vectors=n_vectors)
        return self.fetch()

    def q_zero(self, k):
        at = self._take(k)
        self.out[at:at + k].zero_()
        return at

    def q_dots(self, *terms, n=None, n_vectors=True):
        """dots() without the host read: the slot of the first of the len(terms) results"""
        import ctypes
        k = len(terms)
        at = self._take(k)
        A = (ctypes.c_void_p * k)(*[t[0].data_ptr() for t in terms])
        B = (ctypes.c_void_p * k)(*[t[-1].data_ptr() for t in terms])
        W = (ctypes.c_void_p * k)(*[(t[1].data_ptr() if len(t) == 3 and t[1] is not None else None)
                                    for t in terms])
        n = terms[0][0].numel() if n is None else n
        if self.part is not None and n_vectors and self.part[0] > 0:
            # this rank's share of a sum over n-vector entries: the point part only
            off, n = 8 * self.part[1], self.part[2]
            A = (ctypes.c_void_p * k)(*[a + off for a in A])
            B = (ctypes.c_void_p * k)(*[b + off for b in B])
            W = (ctypes.c_void_p * k)(*[(w + off if w else None) for w in W])
        check(lib().iamx_vec_dots(n, k, A, B, W, self._slot_ptr(at), _ptr(self.scratch), stream_ptr()),
              'iamx_vec_dots')
        if n_vectors:
            self._reduce(k, 'sum', at)
        return at

    def absmax(self, x, y=None):
        self.begin()
        self.q_absmax(x, y)
        return float(self.fetch()[0])

    def q_absmax(self, x, y=None):
        at = self._take(1)
        check(lib().iamx_vec_absmax_prod(x.numel(), _ptr(x), _ptr(y), self._slot_ptr(at),
                                         _ptr(self.scratch), stream_ptr()), 'iamx_vec_absmax_prod')
        self._reduce(1, 'max', at)
        return at

    # ---- scipy/optimize/_lsq/common.py on device vectors
    def cl_scaling(self, x, g, lb, ub):
        """CL_scaling_vector"""
        v, dv = self.new(x), self.new(x)
        check(lib().iamx_trf_cl_scaling(x.numel(), _ptr(x), _ptr(g), _ptr(lb), _ptr(ub), _ptr(v),
                                        _ptr(dv), stream_ptr()), 'iamx_trf_cl_scaling')
        return v, dv

    def trf_scale(self, v, dv, g, scale_inv, want_v=False):
        """trf.py: v[dv != 0] *= scale_inv; d = sqrt(v) * scale; diag_h = g dv scale; g_h = d g"""
        d, diag_h, g_h = self.new(v), self.new(v), self.new(v)
        v_out = self.new(v) if want_v else None
        check(lib().iamx_trf_scale(v.numel(), _ptr(v), _ptr(dv), _ptr(g), _ptr(scale_inv),
                                   _ptr(v_out), _ptr(d), _ptr(diag_h), _ptr(g_h), stream_ptr()),
              'iamx_trf_scale')
        return (d, diag_h, g_h, v_out) if want_v else (d, diag_h, g_h)

    def jac_scale(self, colsq, scale_inv, first):
        """compute_jac_scale from the column sums of J.^2 (in place on scale_inv)"""
        check(lib().iamx_trf_jac_scale(scale_inv.numel(), _ptr(colsq), _ptr(scale_inv),
                                       1 if first else 0, stream_ptr()), 'iamx_trf_jac_scale')
        return scale_inv

    def step_size_to_bound(self, x, s, lb, ub, want_hits=False):
        """step_size_to_bound: (min step, hits in {-1, 0, 1} when asked for)"""
        self.begin()
        self.q_step_to_bound(x, s, lb, ub)
        step = float(self.fetch()[0])
        if not want_hits:
            return step, None
        hits = self.new(x)
        check(lib().iamx_trf_reflect(x.numel(), _ptr(x), _ptr(s), _ptr(lb), _ptr(ub), step, None,
                                     None, _ptr(hits), stream_ptr()), 'iamx_trf_reflect')
        return step, hits

    def q_step_to_bound(self, x, s, lb, ub):
        at = self._take(1)
        check(lib().iamx_trf_step_to_bound(x.numel(), _ptr(x), _ptr(s), _ptr(lb), _ptr(ub),
                                           self._slot_ptr(at), _ptr(self.scratch), stream_ptr()),
              'iamx_trf_step_to_bound')
        self._reduce(1, 'min', at)
        return at

    def reflect(self, x, s, lb, ub, min_step, p_h):
        """trf.py select_step: p_h with the components that hit a bound first negated"""
        r_h = self.new(x)
        check(lib().iamx_trf_reflect(x.numel(), _ptr(x), _ptr(s), _ptr(lb), _ptr(ub), float(min_step),
                                     _ptr(p_h), _ptr(r_h), None, stream_ptr()), 'iamx_trf_reflect')
        return r_h

    def in_bounds(self, x, lb, ub, p=None):
        """in_bounds(x (+ p), lb, ub)"""
        self.begin()
        self.q_count_outside(x, lb, ub, p)
        return self.fetch()[0] == 0.0

    def q_count_outside(self, x, lb, ub, p=None):
        """number of components of x (+ p) outside [lb, ub] (0 <=> in_bounds)"""
        at = self._take(1)
        check(lib().iamx_trf_count_outside(x.numel(), _ptr(x), _ptr(p), _ptr(lb), _ptr(ub),
                                           self._slot_ptr(at), _ptr(self.scratch), stream_ptr()),
              'iamx_trf_count_outside')
        self._reduce(1, 'sum', at)
        return at

    def strictly_feasible(self, x, lb, ub, step=None):
        """make_strictly_feasible(x (+ step), lb, ub, rstep=0)"""
        out = self.new(x)
        check(lib().iamx_trf_strictly_feasible(x.numel(), _ptr(x), _ptr(step), _ptr(lb), _ptr(ub),
                                               _ptr(out), stream_ptr()), 'iamx_trf_strictly_feasible')
        return out

    def feasible_start(self, x, lb, ub, rstep):
        """make_strictly_feasible(x, lb, ub, rstep) for rstep > 0"""
        out = self.new(x)
        check(lib().iamx_trf_feasible_start(x.numel(), _ptr(x), _ptr(lb), _ptr(ub), float(rstep),
                                            _ptr(out), stream_ptr()), 'iamx_trf_feasible_start')
        return out

    def scaled_start(self, x, scale_inv, v, dv):
        """x * scale_inv / sqrt(v') with v' = v * scale_inv where dv != 0 (trf.py:243-246)"""
        out = self.new(x)
        check(lib().iamx_trf_scaled_start(x.numel(), _ptr(x), _ptr(scale_inv), _ptr(v), _ptr(dv),
                                          _ptr(out), stream_ptr()), 'iamx_trf_scaled_start')
        return out

    def active_constraints(self, x, lb, ub, rtol):
        """find_active_constraints (rtol > 0) as -1 / 0 / +1"""
        out = self.new(x)
        check(lib().iamx_trf_active(x.numel(), _ptr(x), _ptr(lb), _ptr(ub), float(rtol), _ptr(out),
                                    stream_ptr()), 'iamx_trf_active')
        return out


_COMPANION4 = np.zeros((4, 4))
_COMPANION4[1, 0] = _COMPANION4[2, 1] = _COMPANION4[3, 2] = 1.0


def solve_tr_2d(B, g, Delta):
    """scipy/optimize/_lsq/common.py solve_trust_region_2d -- min 0.5 p^T B p + g^T p over
    |p| <= Delta for a symmetric 2 x 2 B -- with the same two cases (the Newton step when B is
    positive definite and the step fits, else the best real root of the same quartic in
    t = tan(phi / 2), found as the eigenvalues of its companion matrix like np.roots does), in
    scalar arithmetic: SciPy's version spends 0.3 ms per call in array plumbing, which the
    device-resident loop would wait for with an idle GPU.  Returns (p, newton_step)."""
    a, b, c = float(B[0][0]), float(B[0][1]), float(B[1][1])
    g0, g1 = float(g[0]), float(g[1])
    if a > 0.0:
        l10 = b / a
        s = c - l10 * b
        if s > 0.0:                                   # B = L diag(a, s) L^T
            y1 = -g1 + l10 * g0
            p1 = y1 / s
            p0 = -g0 / a - l10 * p1
            if p0 * p0 + p1 * p1 <= Delta * Delta:
                return np.array([p0, p1]), True
    D2 = Delta * Delta
    A, Bq, C = a * D2, b * D2, c * D2
    d, f = g0 * Delta, g1 * Delta
    co = (-Bq + d, 2 * (A - C + f), 6 * Bq, 2 * (-A + C + f), -Bq - d)
    if co[0] == 0.0 or co[4] == 0.0 or not all(np.isfinite(co)):
        # (np.roots strips zero end coefficients; leave the rare case to it)
        from scipy.optimize._lsq.common import solve_trust_region_2d
        return solve_trust_region_2d(np.array([[a, b], [b, c]]), np.array([g0, g1]), Delta)
    M = _COMPANION4.copy()
    M[0, 0], M[0, 1], M[0, 2], M[0, 3] = -co[1] / co[0], -co[2] / co[0], -co[3] / co[0], -co[4] / co[0]
    best = None
    for t in np.linalg.eigvals(M):
        if t.imag != 0.0:
            continue
        t = t.real
        q = 1.0 + t * t
        p0, p1 = Delta * 2.0 * t / q, Delta * (1.0 - t * t) / q
        val = 0.5 * (p0 * (a * p0 + b * p1) + p1 * (b * p0 + c * p1)) + g0 * p0 + g1 * p1
        if best is None or val < best[0]:
            best = (val, p0, p1)
    if best is None:
        from scipy.optimize._lsq.common import solve_trust_region_2d
        return solve_trust_region_2d(np.array([[a, b], [b, c]]), np.array([g0, g1]), Delta)
    return np.array([best[1], best[2]]), False


def _select_step_dev(prob, V, x, d, diag_h, g_h, p, p_h, Delta, lb, ub, theta, Ggg, gdg, gg):
    """trf.py select_step on device vectors; the J products go through Gram matrices.  Two host
    reads at most: the first brings the in_bounds test together with the model value of a step
    that stays inside (the common case) and the distance to the bounds; the second everything the
    reflected and the anti-gradient candidates need.  (Ggg, gdg, gg: |J d g_h|^2, g_h.diag_h.g_h
    and g_h.g_h, known since the regularisation term -- the anti-gradient direction is -g_h.)"""
    from scipy.optimize._lsq.common import minimize_quadratic_1d

    y_p = prob.jd(d, p_h)
    V.begin()
    i_out = V.q_count_outside(x, lb, ub, p)
    i_sb = V.q_step_to_bound(x, p, lb, ub)
    i_G = prob.q_pairs((y_p, y_p))
    i_d = V.q_dots((p_h, diag_h, p_h), (g_h, p_h), (p_h, p_h))
    S = V.fetch()
    Gpp, ph_d_ph, g_ph, ph_ph = S[i_G], S[i_d], S[i_d + 1], S[i_d + 2]
    if S[i_out] == 0.0:
        return p, p_h, -(0.5 * (Gpp + ph_d_ph) + g_ph)

    p_stride = S[i_sb]
    r_h = V.reflect(x, p, lb, ub, p_stride, p_h)
    r = V.mul(d, r_h)
    p = V.mul(p, s=p_stride)
    p_h = V.mul(p_h, s=p_stride)
    x_on_bound = V.lincomb(1.0, x, 1.0, p)
    ag = V.mul(d, g_h, s=-1.0)                      # d .* ag_h with ag_h = -g_h
    y_r = prob.jd(d, r_h)
    V.begin()
    i_1 = V.q_dots((r_h, r_h), (p_h, r_h), (r_h, diag_h, r_h), (g_h, r_h), (p_h, diag_h, r_h))
    i_2 = prob.q_pairs((y_r, y_r), (y_p, y_r))
    i_3 = V.q_step_to_bound(x_on_bound, r, lb, ub)
    i_4 = V.q_step_to_bound(x, ag, lb, ub)
    S = V.fetch()
    # the products with the scaled p_h (and with y_p, computed before the scaling)
    G00, ph_d_ph, g_ph = p_stride * p_stride * Gpp, p_stride * p_stride * ph_d_ph, p_stride * g_ph
    G11, G01 = S[i_2], p_stride * S[i_2 + 1]
    rh_d_rh, g_rh, ph_d_rh = S[i_1 + 2], S[i_1 + 3], S[i_1 + 4]
    # intersect_trust_region(p_h, r_h, Delta): positive root of |p_h + t r_h| = Delta
    a, b, c = S[i_1], S[i_1 + 1], p_stride * p_stride * ph_ph - Delta * Delta
    if a == 0:
        raise ValueError("`s` is zero.")
    if c > 0:
        raise ValueError("`x` is not within the trust region.")
    disc = np.sqrt(b * b - a * c)
    q = -(b + np.copysign(disc, b))
    t1, t2 = q / a, c / q
    to_tr = max(t1, t2)
    to_bound = S[i_3]
    r_stride = min(to_bound, to_tr)
    if r_stride > 0:
        r_stride_l = (1 - theta) * p_stride / r_stride
        r_stride_u = theta * to_bound if r_stride == to_bound else to_tr
    else:
        r_stride_l, r_stride_u = 0, -1

    if r_stride_l <= r_stride_u:
        qa = 0.5 * (G11 + rh_d_rh)
        qb = g_rh + G01 + ph_d_rh
        c0 = 0.5 * (G00 + ph_d_ph) + g_ph
        r_stride, r_value = minimize_quadratic_1d(qa, qb, r_stride_l, r_stride_u, c=c0)
    else:
        r_value = np.inf

    p_value = 0.5 * theta * theta * (G00 + ph_d_ph) + theta * g_ph

    to_tr = Delta / np.sqrt(gg)
    to_bound = S[i_4]
    ag_stride = theta * to_bound if to_bound < to_tr else to_tr
    ag_stride, ag_value = minimize_quadratic_1d(0.5 * (Ggg + gdg), -gg, 0, ag_stride)

    # (only the chosen candidate is formed)
    if p_value < r_value and p_value < ag_value:
        return V.mul(p, s=theta), V.mul(p_h, s=theta), -p_value
    elif r_value < p_value and r_value < ag_value:
        r_h = V.lincomb(r_stride, r_h, 1.0, p_h)
        return V.mul(r_h, d), r_h, -r_value
    return V.mul(ag, s=ag_stride), V.mul(g_h, s=-ag_stride), -ag_value


def _trf_device(prob, x0, lb, ub, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0,
                callback=None, lsmr_opts=None):
    """scipy/optimize/_lsq/trf.py trf_bounds, n-vectors on the device in the internal order:
    x, g, the Coleman-Li vectors, the scaled Gauss-Newton step and the candidate steps never
    cross PCIe and are only touched by libiamx kernels (VecOps); per outer iteration the host
    sees a few dozen scalars."""
    from scipy.optimize import OptimizeResult
    from scipy.optimize._lsq.common import (check_termination,
                                            minimize_quadratic_1d, print_header_nonlinear,
                                            print_iteration_nonlinear, update_tr_radius)
    lsmr_opts = dict(lsmr_opts or {})
    lsmr_opts['to_host'] = False
    n = prob.n
    V = prob.vec_ops()
    x_raw, lb_d, ub_d = (t[:n] for t in prob.upload_n_many([np.asarray(x0, np.float64), lb, ub]))
    x = V.feasible_start(x_raw, lb_d, ub_d, 1e-10)        # make_strictly_feasible(x0, lb, ub)
    # Several ranks (observations sharded by point): the point part of x, g, the step and every
    # other n-vector lives on the rank that owns the points -- zeros elsewhere --, the TRF scalars
    # are sums / maxima of per-rank shares (VecOps.part).  Inside the loop only the camera blocks
    # and scalars cross ranks; the point parts are put together once, at the end.
    nc, np3 = prob.C * 7, prob.P * 3
    prob.local_points = prob.world > 1 and not prob.with_calib and prob.inner == 'schur'
    own = None
    if prob.local_points:
        own = torch.zeros(n, dtype=F64, device=prob.dev)
        own[:nc] = 1.0
        own[nc + 3 * prob.pt_lo:nc + 3 * prob.pt_hi] = 1.0
        x = V.mul(x, own)
        V.part = (prob.rank, nc, np3)
    else:
        V.part = None
    prob.set_x_dev(x)
    prob.residual_jac()
    nfev = njev = 1
    V.begin()
    prob.q_cost(prob.r)
    cost = 0.5 * V.fetch()[0]
    g = prob.grad_dev()
    scale_inv = V.jac_scale(prob.colsq_dev(), torch.empty(n, dtype=F64, device=prob.dev), first=True)

    v, dv = V.cl_scaling(x, g, lb_d, ub_d)
    # Delta = norm(x0 * scale_inv / v**0.5) with v[dv != 0] *= scale_inv
    if prob.local_points:
        # (entries of other ranks' points: x is 0 there; keep the divisor away from 0 as well)
        v0 = V.lincomb(1.0, V.mul(v, own), 1.0, 1.0 - own)
    else:
        v0 = v
    t0 = V.scaled_start(x, scale_inv, v0, dv)
    Delta = float(np.sqrt(V.dots((t0, t0))[0]))
    del t0, v0
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = n * 100
    termination_status = None
    iteration = 0
    step_norm = actual_reduction = None
    lsmr_iters = 0
    r_new = torch.empty_like(prob.r)
    if verbose == 2 and prob.rank == 0:
        print_header_nonlinear()

    # Host reads per outer iteration (each one drains the queue): the gradient norm with the
    # regularisation scalars, the Schur solve's stop tests, one projection and one batch for the
    # 2-D subspace, one for the step, one for the trial point.
    while True:
        with _Phase(prob, 'scaling+reg'):
            v, dv = V.cl_scaling(x, g, lb_d, ub_d)
            d, diag_h, g_h = V.trf_scale(v, dv, g, scale_inv)
            y_g = prob.jd(d, g_h)
            V.begin()
            i_n = V.q_absmax(g, v)
            i_G = prob.q_gram([y_g])
            i_d = V.q_dots((g_h, diag_h, g_h), (g_h, g_h))
            S = V.fetch()
            g_norm, Ggg, gdg, gg = S[i_n], S[i_G], S[i_d], S[i_d + 1]
        if g_norm < gtol:
            termination_status = 1
        if verbose == 2 and prob.rank == 0:
            print_iteration_nonlinear(iteration, nfev, cost, actual_reduction, step_norm, g_norm)
        if termination_status is not None or nfev == max_nfev:
            break

        # regularisation term (trf.py: build_quadratic_1d along -g_h)
        a = 0.5 * (Ggg + gdg)
        to_tr = Delta / np.sqrt(gg)
        ag_value = minimize_quadratic_1d(a, -gg, 0, to_tr)[1]
        reg_term = -ag_value / Delta ** 2

        with _Phase(prob, 'lsmr'):
            dreg = V.sqrt_shift(diag_h, reg_term)
            gn_h, _istop, itn, _nr, _nar = lsmr(prob, d, dreg, **lsmr_opts)
            gn_h = gn_h[:n]
            lsmr_iters += itn
        with _Phase(prob, 'subspace'):
            # orthonormal basis {s0, s1} of span{g_h, gn_h} (SciPy: economic QR; the step S p_S
            # does not depend on which orthonormal basis is used): s0 = g_h / |g_h|, and
            # Gram-Schmidt with one re-orthogonalisation for s1.  The first projection is carried
            # out on the vectors (w = gn_h - (s0.gn_h) s0 cancels when the two are nearly
            # parallel); the second one, c = s0.w, is of round-off size and is applied to the
            # scalars: with w' = w - c s0, every product with w' is a combination of the products
            # with w and s0 that one batch returns.  s0 and s1 are never formed.
            sg = np.sqrt(gg)
            w = V.lincomb(1.0, gn_h, -V.dots((g_h, gn_h))[0] / gg, g_h)
            y_w = prob.jd(d, w)
            V.begin()
            i_G = prob.q_pairs((y_g, y_w), (y_w, y_w))
            i_d = V.q_dots((g_h, w), (w, w), (g_h, diag_h, w), (w, diag_h, w))
            S = V.fetch()
            G00, G0w, Gww = Ggg / gg, S[i_G] / sg, S[i_G + 1]
            c, ww = S[i_d] / sg, S[i_d + 1]
            e00, e0w, eww = gdg / gg, S[i_d + 2] / sg, S[i_d + 3]
            wn = float(np.sqrt(max(ww - c * c, 0.0)))
            if wn > 0:
                G01, G11 = (G0w - c * G00) / wn, (Gww - 2 * c * G0w + c * c * G00) / wn ** 2
                e01, e11 = (e0w - c * e00) / wn, (eww - 2 * c * e0w + c * c * e00) / wn ** 2
            else:
                G01 = G11 = e01 = e11 = 0.0
            B_S = np.array([[G00 + e00, G01 + e01], [G01 + e01, G11 + e11]])
            g_S = np.array([sg, 0.0])                   # s0.g_h = |g_h|, s1.g_h = 0

        theta = max(0.995, 1 - g_norm)
        actual_reduction = -1
        while actual_reduction <= 0 and nfev < max_nfev:
            with _Phase(prob, 'select_step'):
                p_S, _ = solve_tr_2d(B_S, g_S, Delta)
                # p_h = p_S[0] s0 + p_S[1] s1 in terms of g_h and w
                cw = float(p_S[1]) / wn if wn > 0 else 0.0
                p_h = V.lincomb((float(p_S[0]) - cw * c) / sg, g_h, cw, w)
                p = V.mul(d, p_h)
                step, step_h, predicted_reduction = _select_step_dev(
                    prob, V, x, d, diag_h, g_h, p, p_h, Delta, lb_d, ub_d, theta, Ggg, gdg, gg)
            with _Phase(prob, 'fun'):
                x_new = V.strictly_feasible(x, lb_d, ub_d, step)
                prob.set_x_dev(x_new)
                prob.residual(out=r_new)
                nfev += 1
                V.begin()
                i_s = V.q_dots((step_h, step_h), (step, step), (x, x))
                i_c = prob.q_cost(r_new)
                S = V.fetch()
                step_h_norm = float(np.sqrt(S[i_s]))
                cost_new = 0.5 * S[i_c]
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual_reduction = cost - cost_new
            Delta_new, ratio = update_tr_radius(Delta, actual_reduction, predicted_reduction,
                                                step_h_norm, step_h_norm > 0.95 * Delta)
            step_norm, x_norm = float(np.sqrt(S[i_s + 1])), float(np.sqrt(S[i_s + 2]))
            termination_status = check_termination(actual_reduction, cost, step_norm, x_norm,
                                                   ratio, ftol, xtol)
            if termination_status is not None:
                break
            Delta = Delta_new

        if actual_reduction > 0:
            with _Phase(prob, 'jac+grad'):
                x = x_new
                cost = cost_new                   # (the device already holds x_new: the accepted
                prob.residual_jac()               #  trial was the last one evaluated)
                njev += 1
                g = prob.grad_dev()
                V.jac_scale(prob.colsq_dev(), scale_inv, first=False)       # compute_jac_scale
            if callback is not None:
                callback(prob.download_n(x), cost)
        else:
            prob.set_x_dev(x)
            prob.residual()
            step_norm = 0
            actual_reduction = 0
        iteration += 1

    if termination_status is None:
        termination_status = 0
    active = V.active_constraints(x, lb_d, ub_d, xtol)
    if prob.local_points:
        # the point parts, put together once (zeros outside a rank's own block)
        active = V.mul(active, own)
        for vec in (x, g, active):
            _dist.allreduce_sum_(vec[nc:nc + np3])
        V.part = None
        prob.local_points = False
    x_out, g_out, active_out = prob.download_n_many([x, g, active])
    return OptimizeResult(x=x_out, cost=cost, grad=g_out, optimality=g_norm,
                          active_mask=active_out.astype(int), nfev=nfev, njev=njev,
                          status=termination_status, lsmr_iterations=lsmr_iters,
                          iterations=iteration)



def gather_residual(prob, n_obs_total):
    """full camera-major residual vector in the reference's observation order (all ranks): every
    rank's slice travels once (all-gather of the residuals and of their positions), not a
    zero-padded vector of all observations through an all-reduce."""
    r = prob.download(prob.r, prob.m)
    full = np.zeros(2 * n_obs_total)
    sel = prob.local_obs
    if prob.world == 1:
        full[2 * sel] = r[0::2]
        full[2 * sel + 1] = r[1::2]
        return full
    pos = torch.from_numpy(np.ascontiguousarray(sel, np.int64)).to(prob.dev)
    parts_r = _dist.allgather_padded(prob.r, prob.m)
    parts_p = _dist.allgather_padded(pos, pos.numel())
    for rr, pp in zip(parts_r, parts_p):
        rr, pp = rr.cpu().numpy(), pp.cpu().numpy()
        full[2 * pp] = rr[0::2]
        full[2 * pp + 1] = rr[1::2]
    return full


def solve(opt, x0, bounds, ftol=1e-4, verbose=0, max_nfev=None, inner=None):
    """Entry point used by Optimizer.run() when opt.solver is 'device' (inner='schur') or
    'device-lsmr' (inner='lsmr')."""
    rank, world = _dist.world()
    n = x0.size
    if isinstance(bounds, (list, tuple)) and np.ndim(bounds[0]) > 0:
        lb, ub = np.asarray(bounds[0], np.float64), np.asarray(bounds[1], np.float64)
    else:
        lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
    wc = opt.optimize_calib == 'global'
    K, dc = opt.K, np.asarray(opt.distCoeffs, np.float64)
    fixed = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], *dc])
    uv = np.concatenate([np.asarray(a, np.float64).reshape(-1, 2)
                         for a in opt.by_camera_points_2d if len(a)])
    prob = DeviceBA(opt.n_cameras, opt.n_points, opt.camera_indices, opt.point_indices, uv, wc,
                    fixed_calib=fixed, rank=rank, world=world)
    if inner is not None:
        prob.inner = inner
    res = trf_device(prob, x0, lb, ub, ftol=ftol, verbose=verbose, max_nfev=max_nfev)
    res.inner_solver = prob.inner
    prob.set_x(res.x)
    prob.residual()
    res.fun = gather_residual(prob, opt.camera_indices.size)
    opt._feedback(res.fun, prob.calib)
    res.message = 'device TRF status %d' % res.status
    res.success = res.status > 0
    return res
