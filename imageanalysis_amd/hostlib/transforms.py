"""The handful of rigid-transform helpers the BA path needs, in numpy.

The reference imports them from the third-party `transformations` package (C. Gohlke; an
archived copy sits at scripts/lib/archive/transformations.py).  They are restated here --
same conventions: quaternions w,x,y,z; euler axes strings; 4x4 homogeneous matrices -- so the
optimizer module does not depend on that package.  Pinned by tests/golden/ba_*_refit.pkl
(outputs of the reference's own update_camera_poses()/refit()).

Citations: scripts/lib/archive/transformations.py
  quaternion_matrix :1395-1420, quaternion_from_euler :1276-1330, euler_from_matrix :1115-1170,
  euler_matrix :1051-1112, decompose_matrix :730-815, affine_matrix_from_points :889-995,
  superimposition_matrix :998-1046, quaternion_multiply :1499-1512.
"""
import math

import numpy as np

_EPS = np.finfo(float).eps * 4.0
_NEXT_AXIS = [1, 2, 0, 1]
# axes string -> (firstaxis, parity, repetition, frame); only the ones the path uses
_AXES = {'sxyz': (0, 0, 0, 0), 'rzyx': (0, 0, 0, 1)}


def quaternion_matrix(quaternion):
    """4x4 rotation matrix of a (w,x,y,z) quaternion; the quaternion is normalised first."""
    q = np.array(quaternion, dtype=np.float64)
    n = float(np.dot(q, q))
    if n < _EPS:
        return np.identity(4)
    q = q * math.sqrt(2.0 / n)
    o = np.outer(q, q)
    return np.array([
        [1.0 - o[2, 2] - o[3, 3], o[1, 2] - o[3, 0], o[1, 3] + o[2, 0], 0.0],
        [o[1, 2] + o[3, 0], 1.0 - o[1, 1] - o[3, 3], o[2, 3] - o[1, 0], 0.0],
        [o[1, 3] - o[2, 0], o[2, 3] + o[1, 0], 1.0 - o[1, 1] - o[2, 2], 0.0],
        [0.0, 0.0, 0.0, 1.0]])


def quaternion_from_euler(ai, aj, ak, axes='sxyz'):
    firstaxis, parity, repetition, frame = _AXES[axes]
    i = firstaxis + 1
    j = _NEXT_AXIS[i + parity - 1] + 1
    k = _NEXT_AXIS[i - parity] + 1
    if frame:
        ai, ak = ak, ai
    if parity:
        aj = -aj
    ai, aj, ak = ai / 2.0, aj / 2.0, ak / 2.0
    ci, si = math.cos(ai), math.sin(ai)
    cj, sj = math.cos(aj), math.sin(aj)
    ck, sk = math.cos(ak), math.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    q = np.empty(4)
    q[0] = cj * cc + sj * ss
    q[i] = cj * sc - sj * cs
    q[j] = cj * ss + sj * cc
    q[k] = cj * cs - sj * sc
    if parity:
        q[j] *= -1.0
    return q


def quaternion_multiply(quaternion1, quaternion0):
    """Hamilton product q1 * q0 of (w,x,y,z) quaternions, term order as the reference's."""
    w0, x0, y0, z0 = quaternion0
    w1, x1, y1, z1 = quaternion1
    return np.array([-x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
                     x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
                     -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                     x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0], dtype=np.float64)


def euler_from_matrix(matrix, axes='sxyz'):
    firstaxis, parity, repetition, frame = _AXES[axes]
    i = firstaxis
    j = _NEXT_AXIS[i + parity]
    k = _NEXT_AXIS[i - parity + 1]
    M = np.array(matrix, dtype=np.float64)[:3, :3]
    cy = math.sqrt(M[i, i] * M[i, i] + M[j, i] * M[j, i])
    if cy > _EPS:
        ax = math.atan2(M[k, j], M[k, k])
        ay = math.atan2(-M[k, i], cy)
        az = math.atan2(M[j, i], M[i, i])
    else:
        ax = math.atan2(-M[j, k], M[j, j])
        ay = math.atan2(-M[k, i], cy)
        az = 0.0
    if parity:
        ax, ay, az = -ax, -ay, -az
    if frame:
        ax, az = az, ax
    return ax, ay, az


def euler_from_quaternion(quaternion, axes='sxyz'):
    return euler_from_matrix(quaternion_matrix(quaternion), axes)


def euler_matrix(ai, aj, ak, axes='sxyz'):
    firstaxis, parity, repetition, frame = _AXES[axes]
    i = firstaxis
    j = _NEXT_AXIS[i + parity]
    k = _NEXT_AXIS[i - parity + 1]
    if frame:
        ai, ak = ak, ai
    if parity:
        ai, aj, ak = -ai, -aj, -ak
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    M = np.identity(4)
    M[i, i] = cj * ck
    M[i, j] = sj * sc - cs
    M[i, k] = sj * cc + ss
    M[j, i] = cj * sk
    M[j, j] = sj * ss + cc
    M[j, k] = sj * cs - sc
    M[k, i] = -sj
    M[k, j] = cj * si
    M[k, k] = cj * ci
    return M


def decompose_matrix(matrix):
    """-> (scale[3], shear[3], angles[3] ('sxyz'), translate[3], perspective[4])."""
    M = np.array(matrix, dtype=np.float64).T
    if abs(M[3, 3]) < _EPS:
        raise ValueError("M[3, 3] is zero")
    M /= M[3, 3]
    P = M.copy()
    P[:, 3] = 0.0, 0.0, 0.0, 1.0
    if not np.linalg.det(P):
        raise ValueError("matrix is singular")
    scale = np.zeros((3,))
    shear = [0.0, 0.0, 0.0]
    angles = [0.0, 0.0, 0.0]
    if any(abs(M[:3, 3]) > _EPS):
        perspective = np.dot(M[:, 3], np.linalg.inv(P.T))
        M[:, 3] = 0.0, 0.0, 0.0, 1.0
    else:
        perspective = np.array([0.0, 0.0, 0.0, 1.0])
    translate = M[3, :3].copy()
    M[3, :3] = 0.0
    row = M[:3, :3].copy()
    scale[0] = math.sqrt(np.dot(row[0], row[0]))
    row[0] /= scale[0]
    shear[0] = np.dot(row[0], row[1])
    row[1] -= row[0] * shear[0]
    scale[1] = math.sqrt(np.dot(row[1], row[1]))
    row[1] /= scale[1]
    shear[0] /= scale[1]
    shear[1] = np.dot(row[0], row[2])
    row[2] -= row[0] * shear[1]
    shear[2] = np.dot(row[1], row[2])
    row[2] -= row[1] * shear[2]
    scale[2] = math.sqrt(np.dot(row[2], row[2]))
    row[2] /= scale[2]
    shear[1:] = [s / scale[2] for s in shear[1:]]
    if np.dot(row[0], np.cross(row[1], row[2])) < 0:
        np.negative(scale, scale)
        np.negative(row, row)
    angles[1] = math.asin(-row[0, 2])
    if math.cos(angles[1]):
        angles[0] = math.atan2(row[1, 2], row[2, 2])
        angles[2] = math.atan2(row[0, 1], row[0, 0])
    else:
        angles[0] = math.atan2(-row[2, 1], row[1, 1])
        angles[2] = 0.0
    return scale, shear, angles, translate, perspective


def superimposition_matrix(v0, v1, scale=False):
    """Similarity (scale=True) / rigid transform mapping point set v0 onto v1, least squares
    via SVD.  v0, v1: [>=3, N] arrays (homogeneous 4th row is ignored)."""
    v0 = np.array(v0, dtype=np.float64)[:3].copy()
    v1 = np.array(v1, dtype=np.float64)[:3].copy()
    ndims = 3
    t0 = -np.mean(v0, axis=1)
    M0 = np.identity(ndims + 1)
    M0[:ndims, ndims] = t0
    v0 += t0.reshape(ndims, 1)
    t1 = -np.mean(v1, axis=1)
    M1 = np.identity(ndims + 1)
    M1[:ndims, ndims] = t1
    v1 += t1.reshape(ndims, 1)
    u, s, vh = np.linalg.svd(np.dot(v1, v0.T))
    R = np.dot(u, vh)
    if np.linalg.det(R) < 0.0:
        R -= np.outer(u[:, ndims - 1], vh[ndims - 1, :] * 2.0)
        s[-1] *= -1.0
    M = np.identity(ndims + 1)
    M[:ndims, :ndims] = R
    if scale:
        v0 *= v0
        v1 *= v1
        M[:ndims, :ndims] *= math.sqrt(np.sum(v1) / np.sum(v0))
    M = np.dot(np.linalg.inv(M1), np.dot(M, M0))
    M /= M[ndims, ndims]
    return M
