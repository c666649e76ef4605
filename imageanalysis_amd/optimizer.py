"""MI355X-native stand-in for the reference's scripts/lib/optimizer.py (sparse bundle
adjustment).  Same class, methods, argument meaning and return values:

    Optimizer(root)                                                optimizer.py:66-95
      .setup(proj, groups, group_index, matches_list, optimized=False, cam_calib=False)  :283-405
      .fun(params, n_cameras, n_points, by_camera_point_indices, by_camera_points_2d)     :174-279
      .bundle_adjustment_sparsity(n_cameras, n_points, camera_indices, point_indices)     :142-169
      .run() -> (camera_params, points_3d, camera_map_fwd, feat_map_rev, fx, fy, cu, cv, dist) :410-541
      .update_camera_poses(proj)                                                          :543-575
      .refit(proj, matches, groups, group_index)                                          :583-683
so scripts/process.py:380-401, scripts/4a-optimize.py and scripts/4b-mre-by-image.py run
unchanged.

What moves to the GPU: every residual evaluation (the reference loops over cameras calling
cv2.projectPoints) is one launch of csrc/ba_kernels.hip over all observations, and the
Jacobian -- which the reference lets SciPy build by ~20 finite-difference sweeps of fun()
-- is the analytic 2x(7+3[+8]) blocks of the same kernel.  The trust-region driver is
SciPy's own TRF (`solver='scipy'`, identical step logic to the reference) fed with that
Jacobian, or the device-resident restatement in ba_solver.py (`solver='device'`).
"""
import time
from math import pi

import numpy as np

from . import _deps
from .hostlib import transforms as tf

d2r = pi / 180.0
r2d = 180.0 / pi


def _log(*a):
    _deps.logger().log(*a)


def _qlog(*a):
    _deps.logger().qlog(*a)


def get_recenter_affine(src_list, dst_list):
    """similarity transform (scale + rotation + translation) current -> original camera
    positions (optimizer.py:27-45)."""
    _log('get_recenter_affine():')
    src = np.ones((4, len(src_list)))
    dst = np.ones((4, len(dst_list)))
    src[:3] = np.asarray(src_list, np.float64).reshape(-1, 3).T
    dst[:3] = np.asarray(dst_list, np.float64).reshape(-1, 3).T
    A = tf.superimposition_matrix(src, dst, scale=True)
    _log("A:\n", A)
    return A


def transform_points(A, pts_list):
    """optimizer.py:48-61: apply the 4x4 affine to a list of 3-vectors -> list of lists."""
    p = np.asarray(pts_list, np.float64).reshape(-1, 3)
    src = np.ones((4, len(p)))
    src[:3] = p.T
    dst = A.dot(src)
    return [[float(dst[0][i]), float(dst[1][i]), float(dst[2][i])] for i in range(len(p))]


def _rotation_vector(R):
    """rotation matrix -> axis * angle (the vector cv2.Rodrigues(R) returns)."""
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(w) * 0.5
    c = max(-1.0, min(1.0, (np.trace(R) - 1.0) * 0.5))
    theta = np.arccos(c)
    if s >= 1e-5:
        return w * (0.5 / s * theta)
    if c > 0:
        return np.zeros(3)
    # angle ~ pi: axis from the diagonal, signs from the off-diagonal terms
    x, y, z = np.sqrt(np.maximum((np.diag(R) + 1.0) * 0.5, 0.0))
    y = -y if R[0, 1] < 0 else y
    z = -z if R[0, 2] < 0 else z
    if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
        z = -z
    v = np.array([x, y, z])
    return v * (theta / np.linalg.norm(v))


class Optimizer():
    def __init__(self, root):
        self.root = root
        self.camera_map_fwd = {}
        self.camera_map_rev = {}
        self.feat_map_fwd = {}
        self.feat_map_rev = {}
        self.last_mre = None
        self.graph = None
        self.optimize_calib = 'none'
        self.ftol = 1e-4
        self.min_chain_len = 3
        self.with_bounds = True
        self.cam_method = 'ned_quat'
        self.ncp = 7
        self.cam2body = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=float)
        self.body2cam = np.linalg.inv(self.cam2body)
        # 'device' (default): the TRF restatement of ba_solver.py with J resident on the GPU and
        # the Gauss-Newton subproblems solved through the Schur complement (iamx_ba_accumulate +
        # iamx_ba_schur_*); 'device-lsmr': the same outer iteration with SciPy's subproblem
        # solver, LSMR on the whole system (the reference's formulation, ~10x more inner
        # iterations); 'scipy': SciPy's own TRF driven by the device residual / analytic Jacobian
        self.solver = 'device'
        self._dev = None

    # ---------------------------------------------------------------------------------
    # optimizer.py:120-126
    # ---------------------------------------------------------------------------------
    def nedquat2rvectvec(self, ned, quat):
        body2ned = tf.quaternion_matrix(np.array(quat))[:3, :3]
        R = self.body2cam.dot(body2ned.T)
        rvec = _rotation_vector(R)
        tvec = -(R @ np.asarray(ned, np.float64).reshape(3, 1))
        return rvec.reshape(3, 1), tvec

    # ---------------------------------------------------------------------------------
    # optimizer.py:142-169 (same matrix, assembled in one shot)
    # ---------------------------------------------------------------------------------
    def bundle_adjustment_sparsity(self, n_cameras, n_points, camera_indices, point_indices):
        from scipy.sparse import coo_matrix
        m = camera_indices.size * 2
        n = n_cameras * self.ncp + n_points * 3
        if self.optimize_calib == 'global':
            n += 8
        _log('sparsity matrix is %d x %d' % (m, n))
        rows, cols = self._pattern(n_cameras, n_points, camera_indices, point_indices)
        A = coo_matrix((np.ones(rows.size, dtype=int), (rows, cols)), shape=(m, n)).tolil()
        _log('A-matrix non-zero elements:', A.nnz)
        return A

    def _pattern(self, n_cameras, n_points, camera_indices, point_indices):
        O = camera_indices.size
        per = self.ncp + 3 + (8 if self.optimize_calib == 'global' else 0)
        cols = np.empty((O, per), np.int64)
        cols[:, :self.ncp] = camera_indices[:, None] * self.ncp + np.arange(self.ncp)
        cols[:, self.ncp:self.ncp + 3] = n_cameras * self.ncp + point_indices[:, None] * 3 \
            + np.arange(3)
        if per > self.ncp + 3:
            cols[:, self.ncp + 3:] = n_cameras * self.ncp + n_points * 3 + np.arange(8)
        cols = np.repeat(cols, 2, axis=0)                       # rows 2i and 2i+1
        rows = np.repeat(np.arange(2 * O), per)
        return rows, cols.ravel()

    # ---------------------------------------------------------------------------------
    # device problem
    # ---------------------------------------------------------------------------------
    def _device(self, by_camera_point_indices, by_camera_points_2d):
        """Flatten the per-camera lists into the camera-major arrays the kernels take and
        keep them on the device (rebuilt only when other lists are passed in)."""
        import torch
        from . import kernels
        d = self._dev
        if d is not None and d['src'][0] is by_camera_point_indices \
                and d['src'][1] is by_camera_points_2d:
            return d
        dev = kernels.require_gpu()
        counts = [len(a) for a in by_camera_point_indices]
        cam_idx = np.repeat(np.arange(len(counts), dtype=np.int32), counts)
        pt_idx = np.concatenate([np.asarray(a, np.int64) for a in by_camera_point_indices
                                 if len(a)]).astype(np.int32)
        uv = np.concatenate([np.asarray(a, np.float64).reshape(-1, 2)
                             for a in by_camera_points_2d if len(a)])
        d = dict(src=(by_camera_point_indices, by_camera_points_2d),
                 n_obs=int(cam_idx.size),
                 cam_idx_host=cam_idx, pt_idx_host=pt_idx,
                 cam_idx=torch.from_numpy(cam_idx).to(dev),
                 pt_idx=torch.from_numpy(pt_idx).to(dev),
                 uv=torch.from_numpy(np.ascontiguousarray(uv)).to(dev))
        self._dev = d
        return d

    def _unpack(self, params, n_cameras, n_points):
        """params -> (cams C*7, pts P*3, calib[9]) host arrays (optimizer.py:177-193)."""
        ncp = self.ncp
        cams = params[:n_cameras * ncp]
        pts = params[n_cameras * ncp:n_cameras * ncp + n_points * 3]
        if self.optimize_calib == 'global':
            cal = params[n_cameras * ncp + n_points * 3:]
            calib = np.array([cal[0], cal[0], cal[1], cal[2], cal[3], cal[4], cal[5], cal[6],
                              cal[7]], np.float64)
        else:
            K, dc = self.K, np.asarray(self.distCoeffs, np.float64)
            calib = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], dc[0], dc[1], dc[2], dc[3],
                              dc[4]], np.float64)
        return cams, pts, calib

    def _upload(self, params, n_cameras, n_points):
        import torch
        from . import kernels
        dev = kernels.require_gpu()
        cams, pts, calib = self._unpack(np.asarray(params, np.float64), n_cameras, n_points)
        return (torch.from_numpy(np.ascontiguousarray(cams)).to(dev),
                torch.from_numpy(np.ascontiguousarray(pts)).to(dev),
                torch.from_numpy(calib).to(dev))

    # ---------------------------------------------------------------------------------
    # optimizer.py:174-279  residuals, camera-major, (du, dv) interleaved, observed-projected
    # ---------------------------------------------------------------------------------
    def fun(self, params, n_cameras, n_points, by_camera_point_indices, by_camera_points_2d):
        from . import kernels
        d = self._device(by_camera_point_indices, by_camera_points_2d)
        cams, pts, calib = self._upload(params, n_cameras, n_points)
        r = kernels.ba_residual(cams, pts, d['cam_idx'], d['pt_idx'], d['uv'], calib)
        error = r.cpu().numpy()
        self._feedback(error, calib)
        return error

    def _feedback(self, error, calib):
        """operator feedback when the MRE improves by > 0.1 % (optimizer.py:231-257)."""
        mre = np.mean(np.abs(error))
        if self.last_mre is None or 1.0 - mre / self.last_mre > 0.001:
            self.last_mre = mre
            _log('mre: %.3f std: %.3f max: %.2f' % (mre, np.std(error), np.amax(np.abs(error))))
            if self.optimize_calib == 'global':
                c = [float(v) for v in np.asarray(calib.cpu() if hasattr(calib, 'cpu') else calib)]
                _log("K:\n", np.array([[c[0], 0, c[2]], [0, c[1], c[3]], [0, 0, 1]]))
                _log("distCoeffs: %.3f %.3f %.3f %.3f %.3f" % tuple(c[4:9]))

    def jac(self, params, n_cameras, n_points, by_camera_point_indices, by_camera_points_2d):
        """Analytic Jacobian of fun() as a scipy CSR matrix with the sparsity pattern of
        bundle_adjustment_sparsity() (the reference gets it by finite differences)."""
        from scipy.sparse import csr_matrix
        from . import kernels
        d = self._device(by_camera_point_indices, by_camera_points_2d)
        cams, pts, calib = self._upload(params, n_cameras, n_points)
        wc = self.optimize_calib == 'global'
        r, Jc, Jp, Jk = kernels.ba_residual_jac(cams, pts, d['cam_idx'], d['pt_idx'], d['uv'],
                                                calib, with_calib=wc)
        O = d['n_obs']
        blocks = [Jc.cpu().numpy().reshape(2 * O, 7), Jp.cpu().numpy().reshape(2 * O, 3)]
        if wc:
            blocks.append(Jk.cpu().numpy().reshape(2 * O, 8))
        data = np.concatenate(blocks, axis=1)
        if 'csr' not in d or d['csr'][0] != (n_cameras, n_points, wc):
            _rows, cols = self._pattern(n_cameras, n_points, d['cam_idx_host'].astype(np.int64),
                                        d['pt_idx_host'].astype(np.int64))
            per = data.shape[1]
            d['csr'] = ((n_cameras, n_points, wc), cols.astype(np.int32),
                        np.arange(0, 2 * O * per + 1, per, dtype=np.int32))
        n = n_cameras * self.ncp + n_points * 3 + (8 if wc else 0)
        return csr_matrix((data.ravel(), d['csr'][1], d['csr'][2]), shape=(2 * O, n))

    # ---------------------------------------------------------------------------------
    # optimizer.py:283-405
    # ---------------------------------------------------------------------------------
    def setup(self, proj, groups, group_index, matches_list, optimized=False, cam_calib=False):
        _log('Setting up optimizer data structures...')
        self.optimize_calib = 'global' if cam_calib else 'none'
        cam = _deps.camera()

        placed_images = set()
        for name in groups[group_index]:
            placed_images.add(proj.findIndexByName(name))
        _log('Number of placed images:', len(placed_images))

        # camera index remapping, in the iteration order of the set like the reference
        self.camera_map_fwd = {}
        self.camera_map_rev = {}
        for i, index in enumerate(placed_images):
            self.camera_map_fwd[i] = index
            self.camera_map_rev[index] = i
        self.feat_map_fwd = {}
        self.feat_map_rev = {}

        self.K = cam.get_K(optimized)
        self.distCoeffs = np.array(cam.get_dist_coeffs(optimized))

        self.n_cameras = len(placed_images)
        self.camera_params = np.empty(self.n_cameras * self.ncp)
        for cam_idx, global_index in enumerate(placed_images):
            image = proj.image_list[global_index]
            ned, ypr, quat = image.get_camera_pose(optimized)
            self.camera_params[cam_idx * self.ncp:cam_idx * self.ncp + self.ncp] = \
                np.append(ned, quat)

        from .match_cleanup import Chains
        if isinstance(matches_list, Chains) and matches_list.untouched():
            return self._setup_from_arrays(matches_list, placed_images, group_index)
        # one pass over the matches (the reference makes three with the same test)
        pts, obs_cam, obs_feat, obs_uv = [], [], [], []
        cam_rev = self.camera_map_rev
        feat_used = 0
        for i, match in enumerate(matches_list):
            if match[1] != group_index:
                continue
            obs = [m for m in match[2:] if m[0] in placed_images]
            if len(obs) < self.min_chain_len:
                continue
            self.feat_map_fwd[i] = feat_used
            self.feat_map_rev[feat_used] = i
            pts.append(match[0])
            for m in obs:
                obs_cam.append(cam_rev[m[0]])
                obs_feat.append(feat_used)
                obs_uv.append(m[1])
            feat_used += 1
        self.n_points = feat_used
        pts = np.asarray(pts, np.float64).reshape(-1, 3)
        for k in np.nonzero(np.isnan(pts).any(axis=1))[0]:       # optimizer.py:352-353
            print(self.feat_map_rev[int(k)], pts[k])
        self.points_3d = pts.reshape(-1)[:self.n_points * 3].copy() if self.n_points \
            else np.empty(0)
        n_observations = len(obs_cam)

        # camera-major, order of appearance inside a camera (stable sort)
        obs_cam = np.asarray(obs_cam, np.int64)
        obs_feat = np.asarray(obs_feat, np.int64)
        obs_uv = np.asarray(obs_uv, np.float64).reshape(-1, 2)
        order = np.argsort(obs_cam, kind='stable')
        counts = np.bincount(obs_cam, minlength=self.n_cameras) if n_observations else \
            np.zeros(self.n_cameras, np.int64)
        splits = np.cumsum(counts)[:-1]
        self.by_camera_point_indices = [np.array(a) for a in np.split(obs_feat[order], splits)]
        self.by_camera_points_2d = [a.reshape(len(a), 1, 2)
                                    for a in np.split(obs_uv[order], splits)]
        self.camera_indices = obs_cam[order].astype(int)
        self.point_indices = obs_feat[order].astype(int)
        self._dev = None
        _log("num observations:", n_observations)

    def _setup_from_arrays(self, chains, placed_images, group_index):
        """the match pass of setup() on link_matches()'s arrays (match_cleanup.Chains): the same
        selections in the same order, without a python object per chain member"""
        ptr, img, uv = chains.ptr, chains.img, chains.uv
        n = len(ptr) - 1
        n_img = int(img.max()) + 1 if len(img) else 0
        placed = np.zeros(max(n_img, max(placed_images, default=-1) + 1, 1), bool)
        placed[list(placed_images)] = True
        cam_of = np.full(len(placed), -1, np.int64)
        for i, index in self.camera_map_fwd.items():
            cam_of[index] = i
        chain_of = np.repeat(np.arange(n), np.diff(ptr))
        ok_obs = placed[img] & (chains.group[chain_of] == group_index)
        per_chain = np.bincount(chain_of[ok_obs], minlength=n)
        use = np.nonzero(per_chain >= self.min_chain_len)[0]          # ascending = the loop's order
        feat_of = np.full(n, -1, np.int64)
        feat_of[use] = np.arange(len(use))
        self.feat_map_fwd = dict(zip(use.tolist(), range(len(use))))
        self.feat_map_rev = dict(zip(range(len(use)), use.tolist()))
        self.n_points = len(use)
        pts = np.where(chains.has_ned[use, None], chains.ned[use], np.nan)   # (None -> nan, as np.asarray does)
        for k in np.nonzero(np.isnan(pts).any(axis=1))[0]:       # optimizer.py:352-353
            print(self.feat_map_rev[int(k)], pts[k])
        self.points_3d = pts.reshape(-1).copy() if self.n_points else np.empty(0)
        sel = np.nonzero(ok_obs & (feat_of[chain_of] >= 0))[0]         # chain order, member order
        obs_cam, obs_feat, obs_uv = cam_of[img[sel]], feat_of[chain_of[sel]], uv[sel]
        n_observations = len(sel)
        order = np.argsort(obs_cam, kind='stable')
        counts = np.bincount(obs_cam, minlength=self.n_cameras) if n_observations else \
            np.zeros(self.n_cameras, np.int64)
        splits = np.cumsum(counts)[:-1]
        self.by_camera_point_indices = [np.array(a) for a in np.split(obs_feat[order], splits)]
        self.by_camera_points_2d = [a.reshape(len(a), 1, 2)
                                    for a in np.split(obs_uv[order], splits)]
        self.camera_indices = obs_cam[order].astype(int)
        self.point_indices = obs_feat[order].astype(int)
        self._dev = None
        _log("num observations:", n_observations)

    # ---------------------------------------------------------------------------------
    # optimizer.py:410-541
    # ---------------------------------------------------------------------------------
    def _x0(self):
        if self.optimize_calib == 'global':
            return np.hstack((self.camera_params.ravel(), self.points_3d.ravel(),
                              self.K[0, 0], self.K[0, 2], self.K[1, 2], self.distCoeffs))
        return np.hstack((self.camera_params.ravel(), self.points_3d.ravel()))

    def _bounds(self):
        """optimizer.py:425-478: n,e +-3 m, d +-9 m, everything else free; with 'global'
        calibration f,cu,cv +-20 %, p1,p2 +-0.2."""
        if not self.with_bounds:
            return (-np.inf, np.inf)
        n = self.n_cameras * self.ncp + self.n_points * 3
        lower = np.full(n, -np.inf)
        upper = np.full(n, np.inf)
        cp = np.asarray(self.camera_params, np.float64).reshape(self.n_cameras, self.ncp)
        lo = lower[:self.n_cameras * self.ncp].reshape(self.n_cameras, self.ncp)
        up = upper[:self.n_cameras * self.ncp].reshape(self.n_cameras, self.ncp)
        d = 3
        lo[:, 0:2] = cp[:, 0:2] - d
        up[:, 0:2] = cp[:, 0:2] + d
        lo[:, 2] = cp[:, 2] - 3 * d
        up[:, 2] = cp[:, 2] + 3 * d
        lower, upper = lower.tolist(), upper.tolist()
        if self.optimize_calib == 'global':
            tol = 0.2
            cu, cv = self.K[0, 2], self.K[1, 2]
            lower += [self.K[0, 0] * (1 - tol), cu * (1 - tol), cv * (1 - tol),
                      -np.inf, -np.inf, -tol, -tol, -np.inf]
            upper += [self.K[0, 0] * (1 + tol), cu * (1 + tol), cv * (1 + tol),
                      np.inf, np.inf, tol, tol, np.inf]
        return [lower, upper]

    def run(self):
        x0 = self._x0()
        args = (self.n_cameras, self.n_points, self.by_camera_point_indices,
                self.by_camera_points_2d)
        f0 = self.fun(x0, *args)
        mre_start = np.mean(np.abs(f0))
        bounds = self._bounds()

        t0 = time.time()
        if self.solver in ('device', 'device-lsmr'):
            from . import ba_solver
            res = ba_solver.solve(self, x0, bounds, ftol=self.ftol, verbose=2,
                                  inner='schur' if self.solver == 'device' else 'lsmr')
        else:
            from scipy.optimize import least_squares
            res = least_squares(self.fun, x0, jac=self.jac, verbose=2, method='trf',
                                loss='linear', ftol=self.ftol, x_scale='jac', bounds=bounds,
                                args=args)
        t1 = time.time()
        _log("Optimization took %.1f seconds" % (t1 - t0))
        _log("res:", res)

        ncp = self.ncp
        self.camera_params = res.x[:self.n_cameras * ncp].reshape((self.n_cameras, ncp))
        self.points_3d = res.x[self.n_cameras * ncp:self.n_cameras * ncp
                               + self.n_points * 3].reshape((self.n_points, 3))
        if self.optimize_calib == 'global':
            camera_calib = res.x[self.n_cameras * ncp + self.n_points * 3:]
            fx = fy = camera_calib[0]
            cu, cv = camera_calib[1], camera_calib[2]
            distCoeffs_opt = camera_calib[3:]
        else:
            fx, fy = self.K[0, 0], self.K[1, 1]
            cu, cv = self.K[0, 2], self.K[1, 2]
            distCoeffs_opt = self.distCoeffs

        mre_final = np.mean(np.abs(res.fun))
        _log("Starting mean reprojection error: %.2f" % mre_start)
        _log("Final mean reprojection error: %.2f" % mre_final)
        _log("Iterations:", res.njev)
        if getattr(res, 'inner_solver', None):
            # which linear solver produced the Gauss-Newton steps (the reference's least_squares
            # uses LSMR on the full system; the default here is the Schur complement form, see
            # INTEGRATION.md "Solver")
            _log("Inner solver:", res.inner_solver)
        _log("Elapsed time = %.1f sec" % (t1 - t0))
        if self.optimize_calib == 'global':
            _log("Final camera calib:\n", camera_calib)
        self.result = res
        return (self.camera_params, self.points_3d, self.camera_map_fwd, self.feat_map_rev,
                fx, fy, cu, cv, distCoeffs_opt)

    # ---------------------------------------------------------------------------------
    # optimizer.py:543-575
    # ---------------------------------------------------------------------------------
    def update_camera_poses(self, proj):
        _log('Updated the optimized camera poses.')
        for image in proj.image_list:
            image.node.getChild('camera_pose_opt', True).setBool('valid', False)
        for i, cam in enumerate(self.camera_params):
            image = proj.image_list[self.camera_map_fwd[i]]
            ned_orig, ypr_orig, quat_orig = image.get_camera_pose()
            ned = cam[0:3]
            quat = cam[3:7]
            (yaw_rad, pitch_rad, roll_rad) = tf.euler_from_quaternion(quat, "rzyx")
            _log(image.name, ned_orig, '->', ned, 'dist:',
                 np.linalg.norm(np.array(ned_orig) - np.array(ned)))
            image.set_camera_pose(ned, yaw_rad * r2d, pitch_rad * r2d, roll_rad * r2d, opt=True)
            image.placed = True
        proj.save_images_info()

    # ---------------------------------------------------------------------------------
    # optimizer.py:583-683
    # ---------------------------------------------------------------------------------
    def refit(self, proj, matches, groups, group_index):
        from .match_cleanup import Chains
        # (list(matches): a shallow copy whose members are the caller's chains; the array-backed
        #  form is edited through its arrays below)
        matches_opt = matches if (isinstance(matches, Chains) and matches.untouched()) else list(matches)
        group = groups[group_index]
        _log('refitting group size:', len(group))
        src_list, dst_list = [], []
        for name in group:
            image = proj.findImageByName(name)
            ned, ypr, quat = image.get_camera_pose(opt=True)
            src_list.append(ned)
            ned, ypr, quat = image.get_camera_pose()
            dst_list.append(ned)
        A = get_recenter_affine(src_list, dst_list)

        scale, shear, angles, trans, persp = tf.decompose_matrix(A)
        _log('  scale:', scale)
        _log('  shear:', shear)
        _log('  angles:', angles)
        _log('  translate:', trans)
        _log('  perspective:', persp)
        R = tf.euler_matrix(*angles)
        _log("R:\n{}".format(R))

        in_group = set(group)
        camera_list = []
        for image in proj.image_list:
            ned, ypr, quat = image.get_camera_pose(opt=(image.name in in_group))
            camera_list.append(ned)
        new_cams = transform_points(A, camera_list)

        for i, image in enumerate(proj.image_list):
            if image.name not in in_group:
                continue
            ned, [y, p, r], quat = image.get_camera_pose(opt=True)
            image.set_camera_pose(new_cams[i], y, p, r, opt=True)
        proj.save_images_info()

        dist_report = []
        for i, image in enumerate(proj.image_list):
            if image.name not in in_group:
                continue
            ned_orig, ypr_orig, quat_orig = image.get_camera_pose()
            ned, ypr, quat = image.get_camera_pose(opt=True)
            Rbody2ned = tf.quaternion_matrix(np.array(quat))[:3, :3]   # image.get_body2ned(opt)
            newRbody2ned = R[:3, :3].dot(Rbody2ned)
            (yaw, pitch, roll) = tf.euler_from_matrix(newRbody2ned, 'rzyx')
            image.set_camera_pose(new_cams[i], yaw * r2d, pitch * r2d, roll * r2d, opt=True)
            dist = np.linalg.norm(np.array(ned_orig) - np.array(new_cams[i]))
            _qlog("image:", image.name)
            _qlog("  orig pos:", ned_orig)
            _qlog("  fit pos:", new_cams[i])
            _qlog("  dist moved:", dist)
            dist_report.append((dist, image.name))
        proj.save_images_info()

        dist_report = sorted(dist_report, key=lambda fields: fields[0], reverse=False)
        _log("Image movement sorted lowest to highest:")
        for report in dist_report:
            _log(report[1], "dist:", report[0])

        new_feats = transform_points(A, self.points_3d)
        name_in_group = [image.name in in_group for image in proj.image_list]
        if isinstance(matches_opt, Chains) and matches_opt.untouched():
            # the same rule on the arrays: a chain with a member in the group takes its new point
            ch = matches_opt
            rows = np.array([self.feat_map_rev[i] for i in range(len(new_feats))], np.int64)
            in_grp = np.asarray(name_in_group, bool)[ch.img]
            csum = np.concatenate([[0], np.cumsum(in_grp)])
            hit = (csum[ch.ptr[rows + 1]] - csum[ch.ptr[rows]]) > 0
            ch.ned[rows[hit]] = np.asarray(new_feats, np.float64).reshape(-1, 3)[hit]
            ch.has_ned[rows[hit]] = True
            return
        for i, feat in enumerate(new_feats):
            match = matches_opt[self.feat_map_rev[i]]
            if any(name_in_group[m[0]] for m in match[2:]):
                match[0] = feat
