"""ORACLE / TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the
reference's bundle-adjustment residual and problem assembly.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Pinned against tests/golden/ba_*.npz (outputs of the reference's own
scripts/lib/optimizer.py run here through oracle/gen_golden.py; the
cv2.projectPoints inside it was the closed form the reference itself states in
scripts/lib/project.py:300-329, cv2 being absent -- "cv2 native unpinned").
Citations are relative to /root/reference/.
"""
import numpy as np

# scripts/lib/optimizer.py:92-95: cam2body and its inverse
CAM2BODY = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
BODY2CAM = np.linalg.inv(CAM2BODY)
_EPS = np.finfo(float).eps * 4.0       # scripts/lib/archive/transformations.py (module _EPS)


def quaternion_matrix3(q):
    """scripts/lib/archive/transformations.py:1395-1420 (w,x,y,z; normalises q)."""
    q = np.array(q, dtype=np.float64)
    n = float(q @ q)
    if n < _EPS:
        return np.identity(3)
    q = q * np.sqrt(2.0 / n)
    o = np.outer(q, q)
    return np.array([
        [1.0 - o[2, 2] - o[3, 3], o[1, 2] - o[3, 0], o[1, 3] + o[2, 0]],
        [o[1, 2] + o[3, 0], 1.0 - o[1, 1] - o[3, 3], o[2, 3] - o[1, 0]],
        [o[1, 3] - o[2, 0], o[2, 3] + o[1, 0], 1.0 - o[1, 1] - o[2, 2]]])


def camera_rt(cam7):
    """scripts/lib/optimizer.py:120-126 nedquat2rvectvec, without the Rodrigues
    round trip: R = body2cam . body2ned^T, t = -R . ned."""
    body2ned = quaternion_matrix3(cam7[3:7])
    R = BODY2CAM @ body2ned.T
    return R, -R @ np.asarray(cam7[:3], np.float64)


def project(Xc, fx, fy, cu, cv, dist):
    """Pinhole + Brown (k1,k2,p1,p2,k3): scripts/lib/project.py:300-329,
    dist order scripts/lib/camera.py:94."""
    k1, k2, p1, p2, k3 = [float(v) for v in dist]
    x = Xc[:, 0] / Xc[:, 2]
    y = Xc[:, 1] / Xc[:, 2]
    r2 = x * x + y * y
    rad = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    return np.stack([fx * xd + cu, fy * yd + cv], 1)


def residuals(params, n_cameras, n_points, camera_indices, point_indices, points_2d,
              K, dist, calib_global=False):
    """scripts/lib/optimizer.py:174-229 Optimizer.fun: camera-major, per observation
    (du, dv) interleaved, observed - projected.  `camera_indices`/`point_indices`
    are the camera-major arrays setup() builds (:397-404)."""
    params = np.asarray(params, np.float64)
    cams = params[:n_cameras * 7].reshape(n_cameras, 7)
    pts = params[n_cameras * 7:n_cameras * 7 + n_points * 3].reshape(n_points, 3)
    if calib_global:                                   # :181-189
        cal = params[n_cameras * 7 + n_points * 3:]
        fx = fy = cal[0]
        cu, cv = cal[1], cal[2]
        dist = cal[3:8]
    else:
        fx, fy, cu, cv = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    out = np.empty((len(camera_indices), 2))
    for c in np.unique(camera_indices):
        sel = np.nonzero(camera_indices == c)[0]
        R, t = camera_rt(cams[c])
        Xc = pts[point_indices[sel]] @ R.T + t
        out[sel] = points_2d[sel] - project(Xc, fx, fy, cu, cv, dist)
    return out.ravel()


def setup(names, groups, group_index, matches, min_chain_len=3):
    """scripts/lib/optimizer.py:296-404 Optimizer.setup index structures.

    `names` = image basenames in proj.image_list order.  Returns a dict with
    camera_map_fwd (list), feat_map_rev (list), camera_indices, point_indices,
    points_2d (camera-major), and the match indices of the used points.
    Camera order = iteration order of the python set of placed indices (:296-307).
    """
    placed = set()
    for name in groups[group_index]:
        placed.add(names.index(name))
    cam_fwd = list(placed)                             # set iteration order, like :303
    cam_rev = {g: i for i, g in enumerate(cam_fwd)}
    feat_rev = []
    by_cam_pt = [[] for _ in cam_fwd]
    by_cam_uv = [[] for _ in cam_fwd]
    for i, m in enumerate(matches):
        if m[1] != group_index:
            continue
        obs = [o for o in m[2:] if o[0] in placed]
        if len(obs) < min_chain_len:
            continue
        f = len(feat_rev)
        feat_rev.append(i)
        for o in obs:
            by_cam_pt[cam_rev[o[0]]].append(f)
            by_cam_uv[cam_rev[o[0]]].append(o[1])
    cam_idx = np.concatenate([np.full(len(p), c, np.int64) for c, p in enumerate(by_cam_pt)])
    pt_idx = np.concatenate([np.asarray(p, np.int64) for p in by_cam_pt])
    uv = np.concatenate([np.asarray(u, np.float64).reshape(-1, 2) for u in by_cam_uv])
    return dict(camera_map_fwd=cam_fwd, feat_map_rev=feat_rev, camera_indices=cam_idx,
                point_indices=pt_idx, points_2d=uv,
                by_camera_counts=np.array([len(p) for p in by_cam_pt]))
