"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-in for the subset of OpenCV (``cv2``, absent from this container and
from /root/reference) that /root/reference/scripts/lib/{matcher,optimizer,image}.py
touch on the hot path.  It exists so the reference's OWN Python can be imported
here (oracle/gen_golden.py) to produce golden vectors.  What it restates:

* ``FlannBasedMatcher.knnMatch`` -> the exact thing FLANN approximates
  (cv2.BFMatcher(NORM_L2) semantics): for every query row the k
  nearest train rows by L2, ``distance = float32(sqrt(float32(sum((a-b)^2))))``
  (exact for the integer-valued SIFT descriptors: sum <= 128*255^2 < 2^24),
  ties broken by lowest trainIdx.  PINNING: cv2's tie order is unpinned.
* ``projectPoints`` / ``Rodrigues`` -> pinhole + Brown(k1,k2,p1,p2,k3), the
  closed form the reference itself states in scripts/lib/project.py:300-329.
* ``xfeatures2d.matchGMS`` -> delegates to the reference's own pure-Python
  port scripts/lib/archive/gms_matcher.py (imported from /root/reference at
  run time, never copied) with the live call's thresholdFactor.

* ``triangulatePoints`` -> the published linear (DLT) two-view triangulation OpenCV
  implements: per point the 4x4 system [x*P3-P1; y*P3-P2] of both views, solution = right
  singular vector of the smallest singular value (homogeneous, not normalised).

* ``estimateAffinePartial2D`` -> a DETERMINISTIC robust 4-DOF (rotation, uniform scale,
  translation) fit, NOT OpenCV's RANSAC (whose sampling cannot be reproduced without OpenCV):
  closed-form least-squares similarity on all correspondences, then nine re-fits on the
  correspondences whose residual is at most 200, 50, 10, 3, 3, 3, 3, 3, 3 px under the
  current model (a re-fit needs >= 2 of them, else the loop stops).  This pins what the
  reference does WITH the matrix (scripts/lib/smart.py:66-115,138-192,251-283: decomposition,
  course arithmetic, property-tree weighting); the fit itself is this documented stand-in,
  implemented a second time, independently, by iamx_similarity_pairs.

Anything else raises AttributeError on purpose.
"""
import math
import sys
import types

import numpy as np

NORM_L2 = 4
NORM_HAMMING = 6
RANSAC = 8
LMEDS = 4
IMREAD_ANYCOLOR = 4
IMREAD_ANYDEPTH = 2
IMREAD_IGNORE_ORIENTATION = 128


class KeyPoint(object):
    def __init__(self, x=0.0, y=0.0, size=0.0, angle=-1.0, response=0.0,
                 octave=0, class_id=-1):
        # cv2.KeyPoint stores float32 members
        self.pt = (float(np.float32(x)), float(np.float32(y)))
        self.size = float(np.float32(size))
        self.angle = float(np.float32(angle))
        self.response = float(np.float32(response))
        self.octave = int(octave)
        self.class_id = int(class_id)


class DMatch(object):
    def __init__(self, queryIdx=-1, trainIdx=-1, distance=float('inf')):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = -1
        self.distance = float(np.float32(distance))


def _d2_matrix(des1, des2):
    a = np.asarray(des1, dtype=np.float64)
    b = np.asarray(des2, dtype=np.float64)
    na = (a * a).sum(axis=1)
    nb = (b * b).sum(axis=1)
    d2 = na[:, None] + nb[None, :] - 2.0 * (a @ b.T)   # exact: all integers < 2^53
    return d2


class FlannBasedMatcher(object):
    """Exact brute-force stand-in (see module docstring)."""

    def __init__(self, index_params=None, search_params=None):
        self.index_params = index_params
        self.search_params = search_params

    def knnMatch(self, des1, des2, k=2):
        d2 = _d2_matrix(des1, des2)
        out = []
        # stable argsort -> ties resolved by lowest trainIdx
        order = np.argsort(d2, axis=1, kind='stable')[:, :k]
        for q in range(d2.shape[0]):
            row = []
            for t in order[q]:
                dist = np.sqrt(np.float32(d2[q, t]))      # float32 sqrt, like cv2
                row.append(DMatch(q, int(t), float(dist)))
            out.append(row)
        return out


BFMatcher = FlannBasedMatcher


# ---------------------------------------------------------------------------
# GMS: delegate to the reference's archived pure-Python port
# ---------------------------------------------------------------------------
class _Size(object):
    def __init__(self, wh):
        self.width = wh[0]
        self.height = wh[1]


def _matchGMS(size1, size2, kp1, kp2, matches, withRotation=False,
              withScale=False, thresholdFactor=6.0):
    import importlib.util
    import io
    import contextlib
    path = '/root/reference/scripts/lib/archive/gms_matcher.py'
    spec = importlib.util.spec_from_file_location('_ref_gms_matcher', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.THRESHOLD_FACTOR = thresholdFactor      # live call passes 5.0 (matcher.py:285)
    uv1 = [kp.pt for kp in kp1]
    uv2 = [kp.pt for kp in kp2]
    with contextlib.redirect_stdout(io.StringIO()):
        gms = mod.GmsMatcher(uv1, _Size(size1), uv2, _Size(size2), matches)
        mask, _n = gms.GetInlierMask(withScale, withRotation)
    return [m for m, keep in zip(matches, mask) if keep]


xfeatures2d = types.SimpleNamespace(matchGMS=_matchGMS)


# ---------------------------------------------------------------------------
# Rodrigues / projectPoints
# ---------------------------------------------------------------------------
def Rodrigues(src):
    src = np.asarray(src, dtype=np.float64)
    if src.size == 3:
        r = src.reshape(3)
        theta = math.sqrt(float(r @ r))
        if theta < 1e-300:
            return np.identity(3), None
        k = r / theta
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = (math.cos(theta) * np.identity(3) + (1 - math.cos(theta)) * np.outer(k, k)
             + math.sin(theta) * Kx)
        return R, None
    R = src.reshape(3, 3)
    # rotation matrix -> rotation vector (robust near 0 and pi)
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = math.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = max(-1.0, min(1.0, (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5))
    theta = math.acos(c)
    if s < 1e-5:
        if c > 0:
            rvec = np.zeros(3)
        else:
            t = (R[0, 0] + 1) * 0.5
            x = math.sqrt(max(t, 0.0))
            t = (R[1, 1] + 1) * 0.5
            y = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
            t = (R[2, 2] + 1) * 0.5
            z = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
            if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
                z = -z
            n = math.sqrt(x * x + y * y + z * z)
            rvec = np.array([x, y, z]) * (theta / n)
    else:
        vth = 0.5 / s * theta
        rvec = np.array([rx, ry, rz]) * vth
    return rvec.reshape(3, 1), None


def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs):
    X = np.asarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    R, _ = Rodrigues(np.asarray(rvec, dtype=np.float64).reshape(3))
    t = np.asarray(tvec, dtype=np.float64).reshape(3)
    K = np.asarray(cameraMatrix, dtype=np.float64)
    d = np.zeros(5)
    if distCoeffs is not None:
        dc = np.asarray(distCoeffs, dtype=np.float64).ravel()
        d[:min(5, dc.size)] = dc[:5]
    k1, k2, p1, p2, k3 = d
    Xc = X @ R.T + t
    x = Xc[:, 0] / Xc[:, 2]
    y = Xc[:, 1] / Xc[:, 2]
    r2 = x * x + y * y
    radial = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * radial + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * radial + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    u = K[0, 0] * xd + K[0, 2]
    v = K[1, 1] * yd + K[1, 2]
    return np.stack([u, v], axis=1).reshape(-1, 1, 2), None


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2):
    """4xN homogeneous points (see module docstring)."""
    P = [np.asarray(projMatr1, np.float64), np.asarray(projMatr2, np.float64)]
    x = [np.asarray(projPoints1, np.float64), np.asarray(projPoints2, np.float64)]
    n = x[0].shape[1]
    out = np.zeros((4, n))
    for i in range(n):
        A = np.zeros((4, 4))
        for j in range(2):
            A[2 * j] = x[j][0, i] * P[j][2] - P[j][0]
            A[2 * j + 1] = x[j][1, i] * P[j][2] - P[j][1]
        out[:, i] = np.linalg.svd(A)[2][3]
    return out


SIMILARITY_THRESHOLDS = (200.0, 50.0, 10.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0)


def _fit_similarity(P, Q, w):
    n = w.sum()
    if n < 2:
        return None
    cp = (P * w[:, None]).sum(0) / n
    cq = (Q * w[:, None]).sum(0) / n
    Pc, Qc = P - cp, Q - cq
    den = (w * (Pc * Pc).sum(1)).sum()
    if den == 0:
        return None
    a = (w * (Pc * Qc).sum(1)).sum() / den
    b = (w * (Pc[:, 0] * Qc[:, 1] - Pc[:, 1] * Qc[:, 0])).sum() / den
    A = np.array([[a, -b], [b, a]])
    t = cq - A.dot(cp)
    return np.hstack([A, t.reshape(2, 1)])


def estimateAffinePartial2D(from_pts, to_pts, *args, **kwargs):
    """(2x3 float64 matrix or None, inlier mask [N,1] uint8) -- see the module docstring."""
    P = np.asarray(from_pts, np.float64).reshape(-1, 2)
    Q = np.asarray(to_pts, np.float64).reshape(-1, 2)
    w = np.ones(len(P))
    M = _fit_similarity(P, Q, w)
    if M is None:
        return None, None
    for thr in SIMILARITY_THRESHOLDS:
        res = np.sqrt((((P.dot(M[:, :2].T) + M[:, 2]) - Q) ** 2).sum(1))
        w2 = (res <= thr).astype(np.float64)
        new = _fit_similarity(P, Q, w2)
        if new is None:
            break
        M, w = new, w2
    return M, w.astype(np.uint8).reshape(-1, 1)
