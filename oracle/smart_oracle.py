"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatements of the two per-pair computations of the reference's scripts/lib/smart.py that
the product runs on the GPU, for the CPU tests of find_matches' pair loop
(tests/test_find_matches_loop.py) and as a second opinion beside the device kernels:

* triangulate_down(): scripts/lib/smart.py:26-63 triangulate_features() -- the published linear
  (DLT) two-view triangulation cv2.triangulatePoints implements: per match the 4x4 system
  [x P3 - P1; y P3 - P2] of both views, solution = right singular vector of the smallest singular
  value, then / w; returns the NED "down" row.
* fit_similarity(): scripts/lib/smart.py:66-89 find_affine() -- the reference asks
  cv2.estimateAffinePartial2D (RANSAC, not reproducible without OpenCV: PARITY UNPINNED for the
  fit itself); this is the documented deterministic stand-in used on BOTH sides of every
  comparison in this repository (oracle/shims/cv2.py for the golden generator, iamx_similarity_pairs
  on the device): least-squares 4-DOF similarity on all matches, then nine re-fits on the matches
  within 200, 50, 10, 3, 3, 3, 3, 3, 3 px of the current model (a re-fit needs >= 2 of them).

Pinned through tests/golden/smart_grid.pkl and find_matches_strip.pkl: outputs of the reference's
own lib/smart.py + lib/matcher.py run with the shims (oracle/gen_golden.py G8, G9).
"""
import numpy as np

SIMILARITY_THRESHOLDS = (200.0, 50.0, 10.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0)


def triangulate_down(P1, P2, K, uv1, uv2):
    """P1, P2: 3x4 [R | t]; K 3x3; uv1, uv2 [n, 2] pixels -> float64 [n] NED down"""
    IK = np.linalg.inv(np.asarray(K, float))
    P = [np.asarray(P1, float).reshape(3, 4), np.asarray(P2, float).reshape(3, 4)]
    n = len(uv1)
    out = np.zeros(n)
    for i in range(n):
        A = np.zeros((4, 4))
        for j, uv in enumerate((uv1, uv2)):
            x = IK[0, 0] * uv[i][0] + IK[0, 1] * uv[i][1] + IK[0, 2]
            y = IK[1, 0] * uv[i][0] + IK[1, 1] * uv[i][1] + IK[1, 2]
            A[2 * j] = x * P[j][2] - P[j][0]
            A[2 * j + 1] = y * P[j][2] - P[j][1]
        X = np.linalg.svd(A)[2][3]
        out[i] = X[2] / X[3]
    return out


def _fit(P, Q, w):
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
    return np.hstack([A, (cq - A.dot(cp)).reshape(2, 1)])


def fit_similarity(from_pts, to_pts):
    """2x3 matrix mapping from_pts onto to_pts, or None"""
    P = np.asarray(from_pts, np.float64).reshape(-1, 2)
    Q = np.asarray(to_pts, np.float64).reshape(-1, 2)
    M = _fit(P, Q, np.ones(len(P)))
    if M is None:
        return None
    for thr in SIMILARITY_THRESHOLDS:
        res = np.sqrt((((P.dot(M[:, :2].T) + M[:, 2]) - Q) ** 2).sum(1))
        new = _fit(P, Q, (res <= thr).astype(np.float64))
        if new is None:
            break
        M = new
    return M
