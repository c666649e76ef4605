"""MI355X-path replacement of the reference's scripts/lib/smart.py -- the whole module surface
(`from imageanalysis_amd.smart import *` is the shim file): the per-pair
ground-surface and yaw-error estimates find_matches() keeps while it runs
(lib/matcher.py:987-1005) and its discard policy depends on, and what process.py asks of the
module around it.

    triangulate_features(i1, i2)        smart.py:26-63    two-view DLT of the pair's matches
    estimate_surface_elevation(i1, i2)  smart.py:117-130  -mean / std of the "down" coordinate
    update_surface_estimate(i1, i2)     smart.py:196-250  /smart/<image>/tri_surface_pairs/...
    get_surface_estimate, load, save    smart.py:283-340

    find_affine / decompose_affine      smart.py:66-115   similarity between a pair's keypoints
    estimate_yaw_error(i1, i2)          smart.py:138-192  yaw error from that matrix + poses
    update_yaw_error_estimate(i1, i2)   smart.py:251-283  /smart/<image>/yaw_pairs/... + average
    get_yaw_error_estimate(i1)          smart.py:285-290
    update_srtm_elevations(proj)        smart.py:319-324  process.py:220 (delegates to lib.srtm)
    set_yaw_error_estimates(proj)       smart.py:326-331  process.py:240

The triangulation and the similarity fit run on the GPU (csrc/triangulate.hip:
iamx_triangulate_pairs, iamx_similarity_pairs; find_matches does a whole batch of pairs in one
launch each and hands the results to record_surface_estimate / record_yaw_error_estimate).
The reference obtains the matrix from cv2.estimateAffinePartial2D (RANSAC); here it is a
deterministic robust fit (least squares, then re-fits on the matches within 200, 50, 10, 3, ...
px), so yaw estimates agree with the reference's to the extent the two fits do (both see
GMS-filtered, cross-checked matches).  Inside the reference environment this module works on the
reference's own /smart tree (props.getNode) and smart.json goes through props_json, so the rest
of process.py shares its state; lib.srtm (tile download + interpolation, out of scope here) is
used where the reference uses it."""
import json
import os

import numpy as np

from . import _deps

smart_node = _deps.getNode("/smart", True)
CAM2BODY = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=float)      # lib/image.py:50-52


def projection_matrix(image):
    """[R | t] with R = body2cam . ned2body, t = -R . ned  (lib/image.py:542-553 get_proj,
    without the detour through a Rodrigues vector)."""
    ned, _ypr, _quat = image.get_camera_pose()
    cam2body = image.get_cam2body() if hasattr(image, 'get_cam2body') else CAM2BODY
    R = np.linalg.inv(cam2body).dot(np.asarray(image.get_body2ned()).T)
    return np.hstack([R, -R.dot(np.asarray(ned, float).reshape(3, 1))])


def triangulate_down(i1, i2, pairs):
    """NED "down" of the DLT-triangulated matches `pairs` ([[kp1, kp2], ...]) of one image pair."""
    import torch
    from . import kernels
    from .kernels import _ptr, check, lib, stream_ptr
    from .matcher import _kp_xy
    cam = _deps.camera()
    dev = kernels.require_gpu()
    pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
    n = len(pairs)
    if n == 0:
        return np.zeros(0)
    xy1, xy2 = _kp_xy(i1), _kp_xy(i2)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    PROJ = np.stack([projection_matrix(i1).ravel(), projection_matrix(i2).ravel()])
    IK = np.linalg.inv(cam.get_K())
    out = torch.empty(n, dtype=torch.float64, device=dev)
    args = (t(np.array([[0, 1]]), torch.int32), t(PROJ, torch.float64), t(IK.ravel(), torch.float64),
            t(np.array([0, len(xy1)]), torch.int64), t(np.concatenate([xy1, xy2]), torch.float32),
            t(np.array([n]), torch.int32), t(pairs, torch.int32))
    check(lib().iamx_triangulate_pairs(*[_ptr(a) for a in args], 1, n, _ptr(out), stream_ptr()),
          'iamx_triangulate_pairs')
    return out.cpu().numpy()


def triangulate_features(i1, i2):
    """smart.py:26-63: the [4, N] homogeneous NED points of the pair's matches after
    `points /= points[3]` (rows north, east, down, 1), None where the reference returns None.
    (iamx_triangulate_pairs_xyz; find_matches itself uses the batched down-only form.)"""
    import torch
    from . import kernels
    from .kernels import _ptr, check, lib, stream_ptr
    from .matcher import _kp_xy
    if i1 == i2 or i2.name not in i1.match_list or len(i1.match_list[i2.name]) == 0:
        return None
    if not i1.kp_list or not len(i1.kp_list):
        i1.load_features()
    if not i2.kp_list or not len(i2.kp_list):
        i2.load_features()
    dev = kernels.require_gpu()
    pairs = np.asarray(i1.match_list[i2.name], np.int32).reshape(-1, 2)
    n = len(pairs)
    xy1, xy2 = _kp_xy(i1), _kp_xy(i2)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    PROJ = np.stack([projection_matrix(i1).ravel(), projection_matrix(i2).ravel()])
    IK = np.linalg.inv(_deps.camera().get_K())
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    args = (t(np.array([[0, 1]]), torch.int32), t(PROJ, torch.float64), t(IK.ravel(), torch.float64),
            t(np.array([0, len(xy1)]), torch.int64), t(np.concatenate([xy1, xy2]), torch.float32),
            t(np.array([n]), torch.int32), t(pairs, torch.int32))
    check(lib().iamx_triangulate_pairs_xyz(*[_ptr(a) for a in args], 1, n, _ptr(out), stream_ptr()),
          'iamx_triangulate_pairs_xyz')
    points = np.ones((4, n))
    points[:3] = out.cpu().numpy().T
    return points


# ---- the yaw-error FEEDBACK of the pair loop (lib/matcher.py:987-993) ---------------------------
# After every pair the reference rewrites both images' camera poses from the running yaw-error
# estimate (update_yaw_error_estimate -> Image.set_aircraft_yaw_error_estimate,
# lib/image.py:434-457), and the NEXT pair either image takes part in triangulates with those
# poses (triangulate_features -> get_proj, lib/image.py:542-553).  The estimate itself only
# depends on the similarity fits (pose independent), so the whole chain is a prefix computation
# over the schedule: PoseFeedback replays it in schedule order from the rounds' yaw values and
# hands every pair with matches the two projection matrices the reference would have used.
NATIVE_FEEDBACK = True      # False: PoseFeedback replays in python (tests compare the two)


def pose_matrices(yaw_deg, pitch_deg, roll_deg, ned, body2cam_q, body2cam_m, half_trig=None):
    """[n, 12] row-major [R | t] of cameras whose AIRCRAFT attitude is (yaw, pitch, roll) degrees
    ('rzyx') and whose mount offset is the quaternion body2cam_q: the camera pose
    Image.set_aircraft_yaw_error_estimate() stores (lib/image.py:441-457: ned2cam = ned2body *
    body2cam) turned into get_proj()'s matrix (lib/image.py:542-553: R = body2cam . ned2body,
    t = -R . ned), for n poses at once.  Term by term the arithmetic of transformations.py
    (quaternion_from_euler 'rzyx', quaternion_multiply, quaternion_matrix); the half-angle sines
    and cosines come from math like there."""
    import math
    d2r = math.pi / 180.0
    n = len(yaw_deg)
    trig = lambda a: (np.array(list(map(math.cos, a))), np.array(list(map(math.sin, a))))
    # quaternion_from_euler(yaw, pitch, roll, 'rzyx'): frame 1 swaps the first and last angle
    if half_trig is not None:      # (cos, sin of roll / 2 and pitch / 2 per pose, formed once per image)
        ci, si, cj, sj = half_trig
    else:
        ci, si = trig((np.asarray(roll_deg, float) * d2r / 2.0).tolist())
        cj, sj = trig((np.asarray(pitch_deg, float) * d2r / 2.0).tolist())
    ck, sk = trig((np.asarray(yaw_deg, float) * d2r / 2.0).tolist())
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    w1, x1, y1, z1 = cj * cc + sj * ss, cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc
    w0, x0, y0, z0 = [float(v) for v in body2cam_q]
    q = np.stack([-x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
                  x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
                  -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                  x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0], 1)
    # quaternion_matrix: q *= sqrt(2 / |q|^2); outer products
    nn = (q * q).sum(1)
    q = q * np.sqrt(2.0 / nn)[:, None]
    o = q[:, :, None] * q[:, None, :]
    body2ned = np.empty((n, 3, 3))
    body2ned[:, 0, 0] = 1.0 - o[:, 2, 2] - o[:, 3, 3]
    body2ned[:, 0, 1] = o[:, 1, 2] - o[:, 3, 0]
    body2ned[:, 0, 2] = o[:, 1, 3] + o[:, 2, 0]
    body2ned[:, 1, 0] = o[:, 1, 2] + o[:, 3, 0]
    body2ned[:, 1, 1] = 1.0 - o[:, 1, 1] - o[:, 3, 3]
    body2ned[:, 1, 2] = o[:, 2, 3] - o[:, 1, 0]
    body2ned[:, 2, 0] = o[:, 1, 3] - o[:, 2, 0]
    body2ned[:, 2, 1] = o[:, 2, 3] + o[:, 1, 0]
    body2ned[:, 2, 2] = 1.0 - o[:, 1, 1] - o[:, 2, 2]
    R = np.einsum('ij,nkj->nik', np.asarray(body2cam_m, float), body2ned)       # body2cam . ned2body
    t = -np.einsum('nij,nj->ni', R, np.asarray(ned, float).reshape(n, 3))
    return np.concatenate([R, t[:, :, None]], 2).reshape(n, 12)


class PoseFeedback(object):
    """The camera poses the reference's pair loop would hold when it reaches each pair
    (lib/matcher.py:918-1005), replayed from the rounds of find_matches.

    State per image: the yaw-error estimate its LAST pair so far left (the weighted average over
    /smart/<image>/yaw_pairs after a pair with a similarity fit, 0 after a pair without matches
    or without a fit: update_yaw_error_estimate, lib/smart.py:251-283), or "untouched" -- then the
    image's stored camera pose stands.  feed() takes one round (all ranks' pairs, schedule order)
    and returns, for its pairs with matches, the estimate of both images BEFORE the pair."""

    def __init__(self, image_list, native=None):
        self.image_list = image_list
        n = len(image_list)
        self.touched = np.zeros(n, bool)
        self.value = [0] * n                 # python numbers: what the reference passes on
        self._entries = {}                   # image index -> [sorted partner names, {name: (err, w, dist)}]
        self._base = None
        self._base_of = {}
        native = NATIVE_FEEDBACK if native is None else native
        self._native = self._native_state(image_list) if native and n else None

    def __del__(self):
        h, self._native = getattr(self, '_native', None), None
        free = getattr(self, '_free', None)
        if h is not None and free is not None:
            free(h)

    def _native_state(self, image_list):
        """the replay in libiamx (iamx_yaw_feedback_*: host C++, ~0.1 us per pair instead of ~10 us
        of python per pair with matches), seeded with the entries the tree already holds; None when
        an entry names a partner outside the project (the python replay below handles any name)"""
        import ctypes
        from . import _lib
        names = [im.name for im in image_list]
        index_of = {n_: k for k, n_ in enumerate(names)}
        if len(index_of) != len(names):
            return None
        order = sorted(range(len(names)), key=names.__getitem__)
        rank = np.empty(len(names), np.int32)
        rank[order] = np.arange(len(names), dtype=np.int32)
        seeds = []
        for x, name in enumerate(names):
            inode = smart_node.getChild(name, False) if smart_node.hasChild(name) else None
            if inode is None or not inode.hasChild("yaw_pairs"):
                continue
            acc = _yaw_pairs_of(name, inode.getChild("yaw_pairs", True))
            if not acc:
                continue
            if any(k not in index_of for k in acc):
                return None
            seeds.append((x, np.array([index_of[k] for k in acc], np.int32),
                          np.array([v[0] for v in acc.values()], np.float64),
                          np.array([v[1] for v in acc.values()], np.float64),
                          np.array([v[2] for v in acc.values()], np.float64)))
        L = _lib.lib()
        _p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        h = L.iamx_yaw_feedback_new(len(names), _p(rank))
        if not h:
            raise MemoryError("iamx_yaw_feedback_new")
        h = ctypes.c_void_p(h)
        self._free = L.iamx_yaw_feedback_free         # (kept: modules may be gone when __del__ runs)
        for x, partner, err, w, dist in seeds:
            _lib.check(L.iamx_yaw_feedback_seed(h, x, len(partner), _p(partner), _p(err), _p(w), _p(dist)),
                       'iamx_yaw_feedback_seed')
        return h

    # -- the running average of one image, lib/smart.py:265-283 ------------------------------
    def _state(self, x):
        st = self._entries.get(x)
        if st is None:
            name = self.image_list[x].name
            inode = smart_node.getChild(name, True)
            acc = dict(_yaw_pairs_of(name, inode.getChild("yaw_pairs", True)))
            st = self._entries[x] = [sorted(acc), acc]
        return st

    def _record(self, x, other, values):
        """the pair entry of image x for partner `other` and the average over x's entries"""
        import bisect
        keys, acc = self._state(x)
        if other not in acc:
            bisect.insort(keys, other)
        acc[other] = (float("%.1f" % values[0]), int(float("%.1f" % values[3])), float("%.1f" % values[1]))
        total, count = 0, 0
        for k in keys:                                   # the tree's child order (sorted)
            err, w, dist_m = acc[k]
            if dist_m >= 0.5 and abs(err) <= 30:
                total += err * w
                count += w
        return total / count if count > 0 else 0

    def feed(self, seq, pi, pj, quiet, hit_rows, yv_f, yv_r, ok):
        """one round: seq / pi / pj / quiet over ALL its pairs (seq ascending), hit_rows = the
        rows with matches, yv_f / yv_r [h][4] their (yaw_error, dist, course, weight) per
        direction, ok [h][2] whether that direction has a similarity fit.
        -> (e1 [h], e2 [h], fresh1 [h], fresh2 [h]): the estimate of pair.i1 / pair.i2 when the
        pair is reached, and whether that image is still untouched (stored pose)."""
        import bisect
        h = len(hit_rows)
        if self._native is not None:
            return self._feed_native(pi, pj, quiet, hit_rows, yv_f, yv_r, ok)
        e1, e2 = [0] * h, [0] * h
        f1, f2 = np.zeros(h, bool), np.zeros(h, bool)
        names = None
        if h:
            if self._base is None or self._base[0] is None:
                self._base = ([im.name for im in self.image_list],) + tuple(self._base[1:] if self._base else ())
            names = self._base[0]
            hi, hj = pi[hit_rows], pj[hit_rows]
            is_hit_img = np.zeros(len(self.image_list), bool)
            is_hit_img[hi] = True
            is_hit_img[hj] = True
            # quiet pairs of this round that touch an image with matches in this round, per image
            qrows = np.nonzero(quiet)[0]
            qsets = {}
            if len(qrows):
                qi, qj, qs = pi[qrows], pj[qrows], seq[qrows]
                img = np.concatenate([qi[is_hit_img[qi]], qj[is_hit_img[qj]]])
                sq = np.concatenate([qs[is_hit_img[qi]], qs[is_hit_img[qj]]])
                if len(img):
                    order = np.lexsort((sq, img))
                    img, sq = img[order], sq[order]
                    cut = np.nonzero(np.diff(img))[0] + 1
                    for a, b in zip(np.concatenate([[0], cut]).tolist(), np.concatenate([cut, [len(img)]]).tolist()):
                        qsets[int(img[a])] = sq[a:b].tolist()
            last_event = {}                                  # image -> seq of its last pair with matches here
            touched, value = self.touched, self.value
            hs = seq[hit_rows].tolist()
            for t, (x, y, s) in enumerate(zip(hi.tolist(), hj.tolist(), hs)):
                for side, im_ in ((0, x), (1, y)):
                    qs_ = qsets.get(im_)
                    if qs_:
                        k = bisect.bisect_left(qs_, s)
                        if k and qs_[k - 1] > last_event.get(im_, -1):
                            touched[im_] = True              # a quiet pair in between: back to 0
                            value[im_] = 0
                    if side == 0:
                        e1[t], f1[t] = value[x], not touched[x]
                    else:
                        e2[t], f2[t] = value[y], not touched[y]
                # the pair's own updates: image 1, then image 2 (lib/matcher.py:990-993)
                value[x] = self._record(x, names[y], yv_f[t]) if ok[t][0] else 0
                value[y] = self._record(y, names[x], yv_r[t]) if ok[t][1] else 0
                touched[x] = touched[y] = True
                last_event[x] = last_event[y] = s
        # every image whose LAST pair of the round is a quiet one ends the round at 0
        n_img = len(self.image_list)
        newest = np.full(n_img, -1, np.int64)
        newest_j = np.full(n_img, -1, np.int64)
        newest[pi] = seq                                     # (seq ascends: the last assignment stays)
        newest_j[pj] = seq
        np.maximum(newest, newest_j, out=newest)
        seen = np.nonzero(newest >= 0)[0]
        if len(seen):
            row_of = np.searchsorted(seq, newest[seen])
            for x in seen[quiet[row_of]].tolist():
                self.value[x] = 0
            self.touched[seen] = True
        return e1, e2, f1, f2

    def _feed_native(self, pi, pj, quiet, hit_rows, yv_f, yv_r, ok):
        import ctypes
        from . import _lib
        h = len(hit_rows)
        c = lambda a, dt: np.ascontiguousarray(a, dt)
        _p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        pi_, pj_, q_ = c(pi, np.int32), c(pj, np.int32), c(quiet, np.uint8)
        hr, f_, r_, ok_ = c(hit_rows, np.int64), c(yv_f, np.float64), c(yv_r, np.float64), c(ok, np.uint8)
        e1, e2 = np.zeros(h), np.zeros(h)
        f1, f2 = np.zeros(h, np.uint8), np.zeros(h, np.uint8)
        rc = _lib.lib().iamx_yaw_feedback_feed(self._native, len(pi_), _p(pi_), _p(pj_), _p(q_), h, _p(hr),
                                               _p(f_), _p(r_), _p(ok_), _p(e1), _p(e2), _p(f1), _p(f2))
        if rc and b'infinity' in (_lib.lib().iamx_last_error() or b''):
            raise OverflowError("cannot convert float infinity to integer")     # python's int(inf)
        _lib.check(rc, 'iamx_yaw_feedback_feed')
        return e1.tolist(), e2.tolist(), f1.astype(bool), f2.astype(bool)

    def _sync_native(self):
        if self._native is not None:
            import ctypes
            from . import _lib
            n = len(self.image_list)
            value, touched = np.zeros(n), np.zeros(n, np.uint8)
            _lib.check(_lib.lib().iamx_yaw_feedback_state(self._native, value.ctypes.data_as(ctypes.c_void_p),
                                                          touched.ctypes.data_as(ctypes.c_void_p)),
                       'iamx_yaw_feedback_state')
            self.value, self.touched = value.tolist(), touched.astype(bool)

    def _pose_base(self, x):
        """(aircraft yaw, cos / sin of roll / 2 and pitch / 2, camera ned): what
        set_aircraft_yaw_error_estimate() and get_proj() read; none of them changes inside one
        find_matches call"""
        import math
        hit = self._base_of.get(x)
        if hit is None:
            im = self.image_list[x]
            _lla, ypr, _q = im.get_aircraft_pose()
            d2r = math.pi / 180.0
            hr, hp = ypr[2] * d2r / 2.0, ypr[1] * d2r / 2.0
            hit = self._base_of[x] = (ypr[0], math.cos(hr), math.sin(hr), math.cos(hp), math.sin(hp),
                                      tuple(im.get_camera_pose()[0]))
        return hit

    def projections(self, images, estimates):
        """[n, 12] projection matrices of `images` (indices) under the yaw-error `estimates`"""
        cam = _deps.camera()
        il = self.image_list
        base = [self._pose_base(x) for x in images]
        yaw = [b[0] + e for b, e in zip(base, estimates)]
        if self._base is None or len(self._base) < 3:
            first = il[images[0]]
            body2cam_m = first.get_body2cam() if hasattr(first, 'get_body2cam') else np.linalg.inv(CAM2BODY)
            self._base = (self._base[0] if self._base else None, np.asarray(cam.get_body2cam(), float),
                          np.asarray(body2cam_m, float))
        cols = np.array([b[1:5] for b in base], np.float64).reshape(-1, 4)
        return pose_matrices(yaw, None, None, np.array([b[5] for b in base], float).reshape(-1, 3),
                             self._base[1], self._base[2],
                             half_trig=(cols[:, 0], cols[:, 1], cols[:, 2], cols[:, 3]))

    def settle(self):
        """the call is over: every touched image gets its last estimate, the way the reference's
        last set_aircraft_yaw_error_estimate() of that image left it"""
        self._sync_native()
        for x in np.nonzero(self.touched)[0].tolist():
            self.image_list[x].set_aircraft_yaw_error_estimate(self.value[x])


_frozen = None          # id(image) -> (ned array, aircraft yaw) while find_matches runs


_deferred = None        # set of image names whose weighted averages are due (begin_batch() ...)


def begin_batch():
    """find_matches' batched form: record_surface_estimate / record_yaw_error_estimate write
    the pair entries as always, but the weighted averages over an image's pairs (a pass over all
    of them, sorted -- O(k) per new pair, O(k^2) per image on an all-pairs schedule) are formed
    by flush_aggregates(), once, instead of after every pair.  The averages a reader finds in
    the tree after the flush are the reference's: they only depend on the entries."""
    global _deferred
    _deferred = set()


_pending = []           # record_round() entries not yet written to the tree


def record_round(pairs):
    """find_matches' round form of record_surface_estimate + record_yaw_error_estimate (both
    directions): pairs = [(i1, i2, avg, std, dist_m, yaw_values_fwd, yaw_values_rev)] with
    yaw_values = (yaw_error, dist, relative course, weight) or None.  Nothing is written yet:
    flush_aggregates() puts the entries into the property tree in one pass (a node per pair
    and direction, built directly instead of through ~30 tree calls per pair) and forms the
    weighted averages.  Needs begin_batch()."""
    _pending.extend(pairs)


def materialize_pending():
    _materialize_pending()


def _materialize_pending():
    """the tree entries of everything record_round() was given, in order (same values as
    record_surface_estimate / _record_yaw write, pair by pair)"""
    global _pending
    if not _pending:
        return
    pend, _pending = _pending, []
    from .hostlib import props_compat
    fast = type(smart_node) is props_compat.PropertyNode
    Node = props_compat.PropertyNode
    tri_of, yaw_of = {}, {}

    def nodes(name):
        hit = tri_of.get(name)
        if hit is None:
            inode = smart_node.getChild(name, True)
            tri = inode.getChild("tri_surface_pairs", True)
            yaw = inode.getChild("yaw_pairs", True)
            hit = tri_of[name] = (tri, _pairs_of(name, tri))
            yaw_of[name] = (yaw, _yaw_pairs_of(name, yaw))
        return hit

    if fast:
        # our own tree: a pair's node is made in one go -- a fresh node whose dictionary is set
        # at once when the pair has no entry yet (the normal case inside one find_matches call)
        new_node = Node.__new__
        add = _deferred.add
        for i1, i2, avg, std, dist_m, yv_f, yv_r in pend:
            n1, n2 = i1.name, i2.name
            hit1 = tri_of.get(n1) or nodes(n1)
            hit2 = tri_of.get(n2) or nodes(n2)
            if avg is not None:
                surface_m, stddev = float("%.1f" % avg), float("%.1f" % std)
                weight, dist_i = int(dist_m * dist_m), int(dist_m)
                entry = (surface_m, weight, stddev)
                for (tri, acc), other in ((hit1, n2), (hit2, n1)):
                    td = tri.__dict__
                    pn = td.get(other)
                    if type(pn) is not Node:
                        pn = td[other] = new_node(Node)
                    d = pn.__dict__
                    d["surface_m"], d["weight"], d["stddev"], d["dist_m"] = surface_m, weight, stddev, dist_i
                    acc[other] = entry
                add(n1)
                add(n2)
            for me, other, yv in ((n1, n2, yv_f), (n2, n1, yv_r)):
                if yv is None:
                    continue
                yaw, yacc = yaw_of[me]
                ye, yd = float("%.1f" % yv[0]), float("%.1f" % yv[1])
                yc, yw = float("%.1f" % yv[2]), float("%.1f" % yv[3])
                yd_ = yaw.__dict__
                pn = yd_.get(other)
                if type(pn) is not Node:
                    pn = yd_[other] = new_node(Node)
                d = pn.__dict__
                d["yaw_error"], d["dist_m"], d["relative_crs"], d["weight"] = ye, yd, yc, yw
                yacc[other] = (ye, int(yw), yd)
                add(me)
        return
    for i1, i2, avg, std, dist_m, yv_f, yv_r in pend:
        n1, n2 = i1.name, i2.name
        (tri1, acc1), (tri2, acc2) = nodes(n1), nodes(n2)
        if avg is not None:
            surface_m, stddev = float("%.1f" % avg), float("%.1f" % std)
            weight, dist_i = int(dist_m * dist_m), int(dist_m)
            for tri, acc, other in ((tri1, acc1, n2), (tri2, acc2, n1)):
                pn = tri.getChild(other, True)
                pn.setFloat("surface_m", surface_m)
                pn.setInt("weight", weight)
                pn.setFloat("stddev", stddev)
                pn.setInt("dist_m", dist_i)
                acc[other] = (surface_m, weight, stddev)
            _deferred.add(n1)
            _deferred.add(n2)
        for me, other, yv in ((n1, n2, yv_f), (n2, n1, yv_r)):
            if yv is None:
                continue
            yaw, yacc = yaw_of[me]
            ye, yd = float("%.1f" % yv[0]), float("%.1f" % yv[1])
            yc, yw = float("%.1f" % yv[2]), float("%.1f" % yv[3])
            pn = yaw.getChild(other, True)
            pn.setFloat("yaw_error", ye)
            pn.setFloat("dist_m", yd)
            pn.setFloat("relative_crs", yc)
            pn.setFloat("weight", yw)
            yacc[other] = (ye, int(yw), yd)
            _deferred.add(me)


def flush_aggregates():
    """-> {image name: weighted yaw error over its pairs} for the images recorded since
    begin_batch() / the last flush; writes the pending pair entries, tri_surface_m and yaw_error
    of those images"""
    global _deferred
    out = {}
    if _deferred is not None:
        _materialize_pending()
    if not _deferred:
        return out
    for name in sorted(_deferred):
        node = smart_node.getChild(name, True)
        if name in _pair_cache:
            _surface_average(node, _pair_cache[name][1])
        if name in _yaw_cache:
            out[name] = _yaw_average(node, _yaw_cache[name][1])
    _deferred = set()
    return out


def frozen_ned(image_list, with_yaw=False):
    """[n, 3] camera positions of image_list (and the [n] aircraft yaw angles), cached while
    the poses are frozen"""
    global _frozen_ned
    if not (_frozen is not None and _frozen_ned is not None and _frozen_ned[0] is image_list
            and len(_frozen_ned[1]) == len(image_list)):
        pairs = [_ned_yaw(im) for im in image_list]
        _frozen_ned = (image_list, np.array([p[0] for p in pairs], np.float64).reshape(-1, 3),
                       np.array([p[1] for p in pairs], np.float64))
    return (_frozen_ned[1], _frozen_ned[2]) if with_yaw else _frozen_ned[1]


def yaw_errors_from_affines(ned1, air_yaw1, ned2, aff):
    """yaw_error_from_affine() for n pairs at once: ned1, ned2 [n,3], air_yaw1 [n], aff [n,6]
    (row major 2x3) -> (yaw_error, dist, relative course, weight), each [n]"""
    r2d = 180.0 / np.pi
    tx, ty = aff[:, 2], aff[:, 5]
    with np.errstate(divide='ignore', invalid='ignore'):
        weight = np.where(np.abs(ty) > 0, np.abs(ty / tx), np.abs(tx))
    diff = ned2 - ned1
    dist = np.sqrt((diff * diff).sum(1))
    with np.errstate(divide='ignore', invalid='ignore'):
        direction = diff / dist[:, None]
    crs_gps = 90 - np.arctan2(direction[:, 0], direction[:, 1]) * r2d
    crs_gps = np.where(crs_gps < 0, crs_gps + 360, crs_gps)
    crs_gps = np.where(crs_gps > 360, crs_gps - 360, crs_gps)
    w, h = _deps.camera().get_image_params()
    cx, cy = int(w * 0.5), int(h * 0.5)
    fcx, fcy = float(np.float32(cx)), float(np.float32(cy))
    newx = aff[:, 0] * fcx + aff[:, 1] * fcy + aff[:, 2]
    newy = aff[:, 3] * fcx + aff[:, 4] * fcy + aff[:, 5]
    crs_aff = 90 - np.arctan2(cy - newy, newx - cx) * r2d
    yaw_error = crs_gps - (air_yaw1 + crs_aff)
    yaw_error = np.where(yaw_error < -180, yaw_error + 360, yaw_error)
    yaw_error = np.where(yaw_error > 180, yaw_error - 360, yaw_error)
    return yaw_error, dist, crs_aff, weight


def record_yaw_values(i1, i2, values):
    """record_yaw_error_estimate() with the pair's (yaw_error, dist, relative course, weight)
    already computed (find_matches does a whole round at once); values None: no fit"""
    if values is None:
        return 0
    return _record_yaw(i1, i2, *values)


_frozen_ned = None


def freeze_poses(on):
    """find_matches brackets its pair loop with this: camera POSITIONS and aircraft yaw angles do
    not change inside one call (the yaw-error feedback rewrites the camera attitude only,
    lib/image.py:434-457), and reading one back from the property tree costs more than a pair's
    share of the kernels"""
    global _frozen, _frozen_ned, _deferred
    _frozen = {} if on else None
    _frozen_ned = None
    if not on:
        flush_aggregates()
        _deferred = None


def _ned_yaw(im):
    if _frozen is None:
        return np.array(im.get_camera_pose()[0]), im.get_aircraft_pose()[1][0]
    hit = _frozen.get(id(im))
    if hit is None:
        hit = _frozen[id(im)] = (np.array(im.get_camera_pose()[0]), im.get_aircraft_pose()[1][0])
    return hit


def _pair_distance(i1, i2):
    return np.linalg.norm(_ned_yaw(i2)[0] - _ned_yaw(i1)[0])


def estimate_surface_elevation(i1, i2):
    dist_m = _pair_distance(i1, i2)
    if i1 == i2 or i2.name not in i1.match_list or len(i1.match_list[i2.name]) == 0:
        return None, None, dist_m
    z = triangulate_down(i1, i2, i1.match_list[i2.name])
    return -np.average(z), np.std(z), dist_m


_pair_cache = {}        # image name -> (tri node, {other name: (surface_m, weight, stddev)})


def _pairs_of(name, tri_node):
    """the image's tri_surface_pairs children as plain tuples (built from the tree once, then
    kept in step by record_surface_estimate; load() drops it)"""
    entry = _pair_cache.get(name)
    acc = entry[1] if entry is not None and entry[0] is tri_node else None
    if acc is None:                      # first touch, or the tree was reset / reloaded under us
        acc = {}
        for child in tri_node.getChildren():
            pn = tri_node.getChild(child)
            if pn is not None:
                acc[child] = (pn.getFloat("surface_m"), pn.getInt("weight"), pn.getFloat("stddev"))
        _pair_cache[name] = (tri_node, acc)
    return acc


def record_surface_estimate(i1, i2, avg, std, dist_m):
    """the property-tree bookkeeping of update_surface_estimate (smart.py:203-250)"""
    if avg is None:
        return None, None
    i1_node = smart_node.getChild(i1.name, True)
    i2_node = smart_node.getChild(i2.name, True)
    tri1_node = i1_node.getChild("tri_surface_pairs", True)
    tri2_node = i2_node.getChild("tri_surface_pairs", True)
    weight = dist_m * dist_m
    cutoff_std = 25             # more than this suggests a bad set of matches
    for node, tri, me, other in ((i1_node, tri1_node, i1, i2), (i2_node, tri2_node, i2, i1)):
        acc = _pairs_of(me.name, tri)
        pair_node = tri.getChild(other.name, True)
        pair_node.setFloat("surface_m", float("%.1f" % avg))
        pair_node.setInt("weight", weight)
        pair_node.setFloat("stddev", float("%.1f" % std))
        pair_node.setInt("dist_m", dist_m)
        acc[other.name] = (pair_node.getFloat("surface_m"), pair_node.getInt("weight"),
                           pair_node.getFloat("stddev"))
    for node, me in ((i1_node, i1), (i2_node, i2)):
        if _deferred is not None:
            _deferred.add(me.name)
        else:
            _surface_average(node, _pair_cache[me.name][1])
    return avg, std


def _surface_average(node, acc):
    cutoff_std = 25
    total, count = 0, 0
    for child in sorted(acc):                           # the tree's child order
        surface_m, w, stddev = acc[child]
        if stddev < cutoff_std:
            total += surface_m * w
            count += w
    if count > 0:
        node.setFloat("tri_surface_m", float("%.1f" % (total / count)))


def update_surface_estimate(i1, i2):
    avg, std, dist_m = estimate_surface_elevation(i1, i2)
    return record_surface_estimate(i1, i2, avg, std, dist_m)


# ---- yaw error (smart.py:66-115, 138-192, 251-283) --------------------------------------------
def find_affine(i1, i2):
    """2x3 similarity from i2's pixels to i1's over the pair's matches (None without matches)"""
    import torch
    from . import kernels
    from .kernels import _ptr, check, lib, stream_ptr
    from .matcher import _kp_xy
    if i1 == i2 or i2.name not in i1.match_list or len(i1.match_list[i2.name]) == 0:
        return None
    dev = kernels.require_gpu()
    pairs = np.asarray(i1.match_list[i2.name], np.int32).reshape(-1, 2)
    xy1, xy2 = _kp_xy(i1), _kp_xy(i2)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    aff = torch.empty((1, 2, 6), dtype=torch.float64, device=dev)
    ok = torch.empty((1, 2), dtype=torch.int32, device=dev)
    args = (t(np.array([[0, 1]]), torch.int32), t(np.array([0, len(xy1)]), torch.int64),
            t(np.concatenate([xy1, xy2]), torch.float32), t(np.array([len(pairs)]), torch.int32),
            t(pairs, torch.int32))
    check(lib().iamx_similarity_pairs(*[_ptr(a) for a in args], 1, len(pairs), _ptr(aff), _ptr(ok),
                                      stream_ptr()), 'iamx_similarity_pairs')
    if not int(ok[0, 0].item()):
        return None
    return aff[0, 0].cpu().numpy().reshape(2, 3)


def decompose_affine(affine):
    """(rotation deg, tx, ty, sx, sy) of a 2x3 matrix"""
    a, b, tx = affine[0]
    c, d, ty = affine[1]
    sx = np.sqrt(a * a + b * b) * (-1.0 if a < 0.0 else 1.0)
    sy = np.sqrt(c * c + d * d) * (-1.0 if d < 0.0 else 1.0)
    angle_deg = np.arctan2(-b, a) * 180.0 / np.pi
    if angle_deg < -180.0:
        angle_deg += 360.0
    if angle_deg > 180.0:
        angle_deg -= 360.0
    return angle_deg, tx, ty, sx, sy


def yaw_error_from_affine(i1, i2, affine):
    """estimate_yaw_error with the matrix given: (yaw_error, dist_m, relative course, weight)"""
    if affine is None:
        return None, None, None, None
    r2d = 180.0 / np.pi
    _rot, tx, ty, _sx, _sy = decompose_affine(affine)
    weight = abs(ty / tx) if abs(ty) > 0 else abs(tx)
    (ned1, air_yaw1), (ned2, _y2) = _ned_yaw(i1), _ned_yaw(i2)
    diff = ned2 - ned1
    dist = np.linalg.norm(diff)
    direction = diff / dist
    crs_gps = 90 - np.arctan2(direction[0], direction[1]) * r2d
    if crs_gps < 0:
        crs_gps += 360
    if crs_gps > 360:
        crs_gps -= 360
    # centre pixel of i2 in i1's pixel coordinates
    w, h = _deps.camera().get_image_params()
    cx, cy = int(w * 0.5), int(h * 0.5)
    newc = np.asarray(affine).dot(np.float32([cx, cy, 1.0]))[:2]
    cdiff = [newc[0] - cx, cy - newc[1]]
    crs_aff = 90 - np.arctan2(cdiff[1], cdiff[0]) * r2d
    yaw_error = crs_gps - (air_yaw1 + crs_aff)
    if yaw_error < -180:
        yaw_error += 360
    if yaw_error > 180:
        yaw_error -= 360
    return yaw_error, dist, crs_aff, weight


def estimate_yaw_error(i1, i2):
    return yaw_error_from_affine(i1, i2, find_affine(i1, i2))


_yaw_cache = {}         # image name -> (yaw_pairs node, {other name: (yaw_error, weight, dist_m)})


def _yaw_pairs_of(name, yaw_node):
    """the image's yaw_pairs children as plain tuples (like _pairs_of: read from the tree once,
    then kept in step; an image with k partners would otherwise re-read k nodes per new pair)"""
    entry = _yaw_cache.get(name)
    acc = entry[1] if entry is not None and entry[0] is yaw_node else None
    if acc is None:
        acc = {}
        for child in yaw_node.getChildren():
            pn = yaw_node.getChild(child)
            if pn is not None:
                acc[child] = (pn.getFloat("yaw_error"), pn.getInt("weight"), pn.getFloat("dist_m"))
        _yaw_cache[name] = (yaw_node, acc)
    return acc


def record_yaw_error_estimate(i1, i2, affine):
    """the property-tree bookkeeping of update_yaw_error_estimate (smart.py:251-283): the pair's
    entry under /smart/<i1>/yaw_pairs and the weighted average over the image's pairs (pairs
    closer than 0.5 m or with more than 30 deg of error do not count; weights are truncated to
    integers exactly like the reference's getInt)"""
    if affine is None:
        return 0
    yaw_error, dist, crs_affine, weight = yaw_error_from_affine(i1, i2, affine)
    if yaw_error is None:
        return 0
    return _record_yaw(i1, i2, yaw_error, dist, crs_affine, weight)


def _record_yaw(i1, i2, yaw_error, dist, crs_affine, weight):
    i1_node = smart_node.getChild(i1.name, True)
    yaw_node = i1_node.getChild("yaw_pairs", True)
    acc = _yaw_pairs_of(i1.name, yaw_node)
    pair_node = yaw_node.getChild(i2.name, True)
    pair_node.setFloat("yaw_error", "%.1f" % yaw_error)
    pair_node.setFloat("dist_m", "%.1f" % dist)
    pair_node.setFloat("relative_crs", "%.1f" % crs_affine)
    pair_node.setFloat("weight", "%.1f" % weight)
    acc[i2.name] = (pair_node.getFloat("yaw_error"), pair_node.getInt("weight"),
                    pair_node.getFloat("dist_m"))
    if _deferred is not None:
        _deferred.add(i1.name)
        return None                                     # (the caller asks flush_aggregates())
    return _yaw_average(i1_node, acc)


def current_yaw_average(name):
    """the weighted yaw error over the pairs recorded for `name` so far (what
    record_yaw_error_estimate returns when it is not deferred)"""
    entry = _yaw_cache.get(name)
    return _yaw_average(None, entry[1]) if entry is not None else 0


def _yaw_average(i1_node, acc):
    total, count = 0, 0
    for child in sorted(acc):                           # the tree's child order
        err, w, dist_m = acc[child]
        if dist_m >= 0.5 and abs(err) <= 30:
            total += err * w
            count += w
    if count > 0:
        if i1_node is not None:
            i1_node.setFloat("yaw_error", float("%.1f" % (total / count)))
        return total / count
    return 0


def update_yaw_error_estimate(i1, i2):
    return record_yaw_error_estimate(i1, i2, find_affine(i1, i2))


def get_yaw_error_estimate(i1):
    i1_node = smart_node.getChild(i1.name, True)
    return i1_node.getFloat("yaw_error") if i1_node.hasChild("yaw_error") else 0.0


def get_surface_estimate(i1, i2):
    i1_node = smart_node.getChild(i1.name, True)
    i2_node = smart_node.getChild(i2.name, True)
    vals = [n.getFloat("tri_surface_m") for n in (i1_node, i2_node) if n.hasChild("tri_surface_m")]
    if vals:
        return sum(vals) / len(vals)
    return (i1_node.getFloat("srtm_surface_m") + i2_node.getFloat("srtm_surface_m")) * 0.5


# ---- what process.py asks of the module around find_matches (smart.py:319-331) ----------------
def update_srtm_elevations(proj, ned_interp=None):
    """smart.py:319-324 (process.py:220): /smart/<image>/srtm_surface_m = the SRTM ground under
    every camera, to 0.1 m.  The lookup is lib.srtm's (tile download + interpolation, out of
    scope here), which process.py:218 initialises itself; `ned_interp` replaces it for callers
    without the reference tree."""
    if ned_interp is None:
        srtm = _deps.srtm()
        if srtm is None:
            raise RuntimeError("update_srtm_elevations: lib.srtm is not importable (it is the "
                               "reference's SRTM tile reader); pass ned_interp=")
        ned_interp = srtm.ned_interp
    for image in proj.image_list:
        ned, _ypr, _quat = image.get_camera_pose()
        surface = np.asarray(ned_interp([ned[0], ned[1]]), float).ravel()[0]
        image_node = smart_node.getChild(image.name, True)
        image_node.setFloat("srtm_surface_m", float("%.1f" % surface))


def set_yaw_error_estimates(proj):
    """smart.py:326-331 (process.py:240), as written there: the value is read from the image's
    `yaw_pairs` node -- the per-partner table, where a key "yaw_error" only exists for an image of
    that name -- not from /smart/<image>/yaw_error where update_yaw_error_estimate() puts the
    average; so the reference applies 0.0 to every image, and so does this."""
    for image in proj.image_list:
        image_node = smart_node.getChild(image.name, True)
        yaw_node = image_node.getChild("yaw_pairs", True)
        yaw_error_deg = yaw_node.getFloat("yaw_error")
        image.set_aircraft_yaw_error_estimate(yaw_error_deg)


# ---- smart.json (props_json inside the reference environment, plain json here) ----------------
def _to_dict(node):
    # (only without the props package: `node` is hostlib.props_compat.PropertyNode, whose
    #  attributes are the children / enumerated lists / scalars themselves)
    out = {}
    for k, v in node.__dict__.items():
        if isinstance(v, list):
            if v:
                out[k] = [float(x) for x in v]
            else:
                out[k] = v
        elif hasattr(v, 'getChild'):
            out[k] = _to_dict(v)
        else:
            out[k] = v
    return out


def _has_enum_lists(node):
    """enumerated lists need _to_dict's float conversion; the smart tree has none in practice"""
    for img in node.__dict__.values():
        if isinstance(img, list):
            return True
        if hasattr(img, '__dict__'):
            for v in img.__dict__.values():
                if isinstance(v, list):
                    return True
    return False


def _from_dict(node, d):
    for k, v in d.items():
        if isinstance(v, dict):
            _from_dict(node.getChild(k, True), v)
        elif isinstance(v, bool):
            node.setBool(k, v)
        elif isinstance(v, int):
            node.setInt(k, v)
        elif isinstance(v, float):
            node.setFloat(k, v)
        elif isinstance(v, list):
            node.setLen(k, len(v))
            for i, x in enumerate(v):
                node.setFloatEnum(k, i, x)
        else:
            node.setString(k, v)


def load(analysis_dir):
    _pair_cache.clear()
    _yaw_cache.clear()
    if analysis_dir is None:
        return
    path = os.path.join(analysis_dir, "smart.json")
    if _deps.HAVE_PROPS:
        import props_json
        props_json.load(path, smart_node)
    elif os.path.exists(path):
        with open(path) as f:
            _from_dict(smart_node, json.load(f))


def save(analysis_dir, polite=False):
    """polite: called on a helper thread beside a loop that launches device work (find_matches
    writes the file ahead of time): give the interpreter lock up after every image's record -- a
    thread that holds it for 0.3 ms at a time and never sleeps makes the launching thread wait up
    to the 5 ms switch interval at each of its own dozens of lock hand-overs per round."""
    if analysis_dir is None:
        return
    path = os.path.join(analysis_dir, "smart.json")
    if _deps.HAVE_PROPS:
        import props_json
        props_json.save(path, smart_node)
    else:
        if len(smart_node.__dict__) > 200 and not _has_enum_lists(smart_node):
            # (one C-encoder pass with a callback per node instead of a python copy of the tree)
            import time

            def records(top):
                for k in sorted(top):
                    yield '%s: %s' % (json.dumps(k), json.dumps(top[k], sort_keys=True,
                                                                default=lambda o: o.__dict__))
                    if polite:
                        time.sleep(0.0002)
            with open(path, 'w') as f:
                f.write('{\n')
                f.write(',\n'.join(records(smart_node.__dict__)))
                f.write('\n}\n')
            return
        tree = _to_dict(smart_node)
        with open(path, 'w') as f:
            if len(tree) <= 200:
                json.dump(tree, f, indent=4, sort_keys=True)
            else:
                # python's json only has a C encoder for the compact form; a survey of
                # thousands of images carries millions of per-pair leaves (one line per image)
                f.write('{\n')
                f.write(',\n'.join('%s: %s' % (json.dumps(k), json.dumps(tree[k], sort_keys=True))
                                   for k in sorted(tree)))
                f.write('\n}\n')
