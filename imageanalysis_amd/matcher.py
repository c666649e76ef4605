"""MI355X-native stand-in for the reference's scripts/lib/matcher.py (live part, :1-1197).

Same module-level entry points, arguments and on-disk results:
    configure()                                           matcher.py:43-80
    find_matches(proj, K, strategy, transform, sort, review)   :852-1031
    saveMatches(image_list, check_if_dirty=False)         :1033-1040
    raw_matches / basic_pair_matches / bidirectional_pair_matches   :203-347
    filter_duplicates / filter_cross_check                :157-200
so scripts/process.py:290-292 runs unchanged.

What differs is where the arithmetic happens: descriptors live in HBM as int8, the k=2
nearest-neighbour search is exact brute force on the i8 MFMA kernel (csrc/match_knn2.hip)
instead of FLANN's randomised KD-trees, pairs are processed in batches
(both directions of many pairs per launch), and the quality-metric threshold runs on the
device.  The python below keeps the reference's ordering rules (stable metric sort, clip to
2000, GMS, first-come de-duplication, cross check) so the match lists are identical to the
reference run on an exact matcher.

Only the 'traditional' strategy (the default of process.py / 3a-matching.py) and SIFT
descriptors (integer valued 0..255) are on this path.
"""
import contextlib
import ctypes
import gc
import os
import time
from math import sqrt

import numpy as np

from . import _deps, _lib
from .matchpairs import MatchPairs, empty_huge
from .keypoints import KeyPointList
from ._deps import getNode
from .gms import gms_inlier_mask

# host threads of the bulk routines between a round's packed results and its match lists
# (libiamx: iamx_pairs_fwd_rev, iamx_segment_mean_std); the GPU box grants a process 16 cores
_HOST_THREADS = 6

detector_node = getNode('/config/detector', True)
matcher_node = getNode('/config/matcher', True)

detect_scale = 0.40
the_matcher = None
max_distance = None
min_pairs = 25

MYMAX = 2000            # matcher.py:265
SAVE_INTERVAL = 300     # seconds between the periodic saves of find_matches (matcher.py:923)
PAIRS_PER_BATCH = 16384 # unordered pairs per device batch (per-batch host costs are ~3 ms: 4096 -> 16384
                        # took 0.5 s off the 2812-image all-pairs survey, profiles/r4_fm_config2.txt)
BATCH_BYTES = 24 << 30  # ... as far as one batch's device workspace stays below this (three are pooled:
                        # 72 GB of the 288 GB; 512 instead of 128 pairs per batch at 50 k keypoints)
PREWARM_ROUNDS = 32     # a call with at least this many rounds fills its buffer pools on a helper thread first
EARLY_SMART_ROUNDS = 8  # rounds in a row without a match before smart.json is written ahead of time
early_smart_stats = {'written': 0, 'current_at_end': 0}     # (tests / diagnosis)
PACK_CAP = 4 << 20      # matches a batch's packed download holds (more: that batch's slots are copied)
# Which sweep a round of find_matches takes.  The symmetric sweep (one MFMA pass per image pair)
# leaves the rows whose bounds pass the metric test to an exact stage that runs at half the sweep's
# rate; on overlapping frames of real imagery 15-30 % of the rows are such candidates.  Round 5
# built the route the round-4 review proposed -- a round whose last measured candidate share exceeds
# DENSE_SHARE takes the one-direction bound form (two sweeps per pair, survivors finished exactly),
# every DENSE_PROBE-th routed round measures again -- and MEASURED it (profiles/r5_*): break-even on
# 512 rendered frames (3.36 s symmetric, 3.44 s routed, f = 0.17), SLOWER on the workloads with many
# true survivors: the dense-overlap workload of bench.py (f = 0.28) runs 0.63x as fast in the
# one-direction form (its per-survivor finish costs more than the exact stage it replaces), the
# dense 400-image survey 76 k instead of 80 k pairs/s through find_matches (same code otherwise;
# configs[2], whose dense rounds are few, is unaffected) -- and the parity-partitioned layout it
# needs is another 140 B per descriptor row.  So the default is
# 'never'; 'auto' / 'always' remain for A/B runs (IAMX_DENSE_ROUTE).  Results are identical in all
# modes (tests/test_mirror_gpu.py).
DENSE_ROUTE = os.environ.get('IAMX_DENSE_ROUTE', 'never')
DENSE_SHARE = 0.30
DENSE_PROBE = 4
_route = {'share': None, 'since_probe': 0, 'rounds': [0, 0]}     # rounds: [symmetric, one-direction]


def _route_reset():
    _route.update(share=None, since_probe=0, rounds=[0, 0])


def _route_next():
    """True: the next round takes the one-direction form"""
    if DENSE_ROUTE == 'always':
        return True
    if DENSE_ROUTE != 'auto' or _route['share'] is None or _route['share'] < DENSE_SHARE:
        return False
    _route['since_probe'] += 1
    if _route['since_probe'] >= DENSE_PROBE:
        _route['since_probe'] = 0
        return False
    return True


def _workspace_bytes_per_pair(rows):
    """device workspace one unordered image pair of `rows`-descriptor images needs in a batch:
    the per-row partial bounds of the symmetric sweep (workgroups of the register-resident image
    x padded rows of the streamed one x 8 B) dominate -- 0.13 MB at 4096 rows, 20 MB at 50 k --,
    then ~48 B per query row and direction (distances, candidates, survivors), ~41 B for the
    narrow exact stage (class mask, result slots, items, partial results, group byte) and the
    per-pair result slots of the filters"""
    rows = max(int(rows), 1)
    wg_rows = 1024 if rows >= 4096 else (512 if rows >= 2048 else 256)
    cap = (rows + 127) // 128 * 128
    return ((rows + wg_rows - 1) // wg_rows) * cap * 8 + 2 * rows * 90 + 96 * 1024


def _batch_bytes():
    """BATCH_BYTES, unless device memory is really short: three workspaces are pooled, so the
    figure shrinks to a quarter of what this process could still obtain -- the device's free
    memory PLUS the blocks torch's caching allocator holds but does not use (a second
    find_matches call in the same process, or a large resident arena, must not shrink the
    batches: the partition of the schedule into rounds would depend on allocator state) --,
    never below 256 MB"""
    try:
        import torch
        free, _total = torch.cuda.mem_get_info()
        free += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
    except Exception:                     # noqa: BLE001  (no device yet: the 288 GB figure)
        return BATCH_BYTES
    return int(max(256 << 20, min(BATCH_BYTES, free // 4)))


def _pairs_per_batch(rows):
    """largest power of two <= PAIRS_PER_BATCH (at least 16) whose batch fits BATCH_BYTES"""
    if PAIRS_PER_BATCH < 16:
        return PAIRS_PER_BATCH
    n = max(16, min(PAIRS_PER_BATCH, _batch_bytes() // _workspace_bytes_per_pair(rows)))
    p = 16
    while p * 2 <= n:
        p *= 2
    return p


def _log(*a):
    _deps.logger().log(*a)


def _qlog(*a):
    _deps.logger().qlog(*a)


# --------------------------------------------------------------------------------------
# device side
# --------------------------------------------------------------------------------------
class DeviceMatcher(object):
    """Holds the survey's descriptors in HBM (one growing arena) and runs batched k=2 NN."""

    def __init__(self):
        self._slots = {}          # image name -> (slot, n_rows)
        self._counts = []
        self._store = None
        self._pending = []
        self._kp = {}             # slot -> (xy float32 [n,2], key2 int32 [n,2]) host copies
        self._kp_dev = None       # (n_slots, kp_off, xy, key2) device arena of the post filter
        self._proj = {}           # slot -> (pose, [R|t] row major, epoch) for the surface triangulation
        self._pose_epoch = None   # set by find_matches: camera poses do not change inside one call
        self._adopted = {}        # image name -> n_rows: features that arrived from another rank

    # cv2-style single pair call (returns numpy (idx[nq,2], dist[nq,2] float32))
    def knnMatch(self, des1, des2, k=2):
        from . import kernels
        if k != 2:
            raise ValueError("the device matcher is a k=2 search")
        idx, d2 = kernels.knn2(np.asarray(des1), np.asarray(des2))
        return idx.cpu().numpy(), np.sqrt(d2.cpu().numpy().astype(np.float32))

    def slot_of(self, image, pre=None):
        """Device slot of an image's descriptors; uploaded once per image (a host-side cache
        flush + reload of the same image re-uses the rows already in HBM).  pre: (xy, key2) of the
        image's keypoints when the caller computed them already (find_matches' registration does,
        on a few threads)."""
        ent = self._slots.get(image.name)
        if ent is not None and image.name in self._adopted and \
                (image.des_list is None or not len(image.des_list)):
            return ent[0]                       # detected on another rank (adopt())
        n = int(image.des_list.shape[0])
        if ent is not None and ent[1] == n:
            return ent[0]
        self._adopted.pop(image.name, None)     # (re-)detected here after all
        slot = len(self._counts)
        self._slots[image.name] = (slot, n)
        self._counts.append(n)
        self._pending.append((slot, image.des_list))
        if pre is None:
            xy = _kp_xy(image)
            pre = (xy, kp_key2(xy))
        self._kp[slot] = pre
        return slot

    def slot_known(self, image):
        """the image's slot without looking at its features when it is registered already (the
        host copy of an image's features may have been flushed since): the surface stage of the
        booking rank asks for images other ranks matched"""
        ent = self._slots.get(image.name)
        if ent is not None:
            return ent[0]
        _ensure_features(image)
        return self.slot_of(image)

    def adopt(self, name, des, xy):
        """Register an image whose features were detected by ANOTHER rank (dist.exchange_features):
        des uint8 [n,128] (device or host), xy float32 [n,2] host.  The image object itself keeps
        no kp_list / des_list on this rank."""
        n = int(des.shape[0])
        slot = len(self._counts)
        self._slots[name] = (slot, n)
        self._adopted[name] = n
        self._counts.append(n)
        self._pending.append((slot, des))
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self._kp[slot] = (xy, kp_key2(xy))
        return slot

    def rows_of(self, image):
        """descriptor rows of an image: adopted count, else len(des_list)"""
        n = self._adopted.get(image.name)
        if n is not None and (image.des_list is None or not len(image.des_list)):
            return n
        return int(image.des_list.shape[0]) if image.des_list is not None and \
            len(getattr(image.des_list, 'shape', ())) else 0

    def keypoints(self):
        """device arena of kp.pt and their "%.2f" keys for every slot (rebuilt when images
        were added): (kp_off int64 [n_slots], xy float32 [total,2], key2 int32 [total,2])"""
        import torch
        from . import _lib
        n = len(self._counts)
        cur = self._kp_dev
        if cur is None or cur[0] != n:
            # only the NEW slots go up (one concatenation, one copy each), into arenas that grow
            # geometrically: find_matches meets a few new images in every round of its first
            # stretch, and re-uploading every keypoint of every image each time was O(images^2)
            dev = _lib.require_gpu()
            n0 = cur[0] if cur is not None else 0
            rows0 = cur[4] if cur is not None else 0
            cnt = [len(self._kp[s][0]) for s in range(n0, n)]
            off = rows0 + np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            rows = int(off[-1])
            if cur is None or cur[1].shape[0] < n or cur[2].shape[0] < rows:
                cap_n = max(n, 2 * (cur[1].shape[0] if cur is not None else 0), 64)
                cap_r = max(rows, 2 * (cur[2].shape[0] if cur is not None else 0), 1 << 16)
                d_off = torch.zeros(cap_n, dtype=torch.int64, device=dev)
                d_xy = torch.empty((cap_r, 2), dtype=torch.float32, device=dev)
                d_k2 = torch.empty((cap_r, 2), dtype=torch.int32, device=dev)
                if cur is not None:
                    d_off[:n0].copy_(cur[1][:n0])
                    d_xy[:rows0].copy_(cur[2][:rows0])
                    d_k2[:rows0].copy_(cur[3][:rows0])
            else:
                d_off, d_xy, d_k2 = cur[1], cur[2], cur[3]
            if n > n0:
                xy = np.concatenate([self._kp[s][0] for s in range(n0, n)] + [np.zeros((0, 2), np.float32)])
                k2 = np.concatenate([self._kp[s][1] for s in range(n0, n)] + [np.zeros((0, 2), np.int32)])
                d_off[n0:n].copy_(torch.from_numpy(off[:-1].copy()))
                d_xy[rows0:rows].copy_(torch.from_numpy(np.ascontiguousarray(xy, np.float32)))
                d_k2[rows0:rows].copy_(torch.from_numpy(np.ascontiguousarray(k2, np.int32)))
            self._kp_dev = (n, d_off, d_xy, d_k2, rows)
        return self._kp_dev[1:4]

    def want_train_layout(self):
        """From now on the arena carries the parity-partitioned copy the one-direction sweep reads
        (+50 % of the arena: 140 B per descriptor row).  -> False when that copy would take more
        than a quarter of the free device memory -- the round then stays on the symmetric sweep,
        which needs nothing extra (a 10 000-frame survey's arena is 104 GB without it)."""
        if getattr(self, '_train_layout', False):
            return True
        try:
            import torch
            free, _total = torch.cuda.mem_get_info()
        except Exception:                 # noqa: BLE001
            return False
        need = (sum(self._counts) + 128 * len(self._counts)) * 140
        if need > free // 4:
            return False
        self._train_layout = True
        return True

    def store(self):
        """(Re)build the arena when new images arrived; old rows are copied on the device."""
        from . import kernels
        pend = self._pending
        want_train = getattr(self, '_train_layout', False)
        _tm = [time.perf_counter()] if _round_trace is not None else None
        if self._store is not None and len(self._store.counts) < len(self._counts) and \
                self._store.has_train_layout == bool(want_train) and \
                self._store.counts == self._counts[:len(self._store.counts)] and \
                self._store.try_extend(self._counts[len(self._store.counts):]):
            pass                                   # the new images fit behind the old ones
        elif self._store is None or len(self._store.counts) != len(self._counts):
            # (no parity-partitioned copy: find_matches' batches hold both directions of every
            #  pair -- a third less arena, 104 instead of 155 GB for 10 000 frames of 37 k keypoints)
            # Capacity: find_matches says how many images the survey has (expect_images); the
            # rows so far give the mean.  Without that, half again what is needed now -- a call that
            # meets undetected images registers a few hundred per round, and re-allocating and
            # copying the arena every time was 10 of 42 s at 4186 frames.
            need = sum(self._counts) + 256 * len(self._counts)
            n_exp = max(int(getattr(self, 'expect_images', 0)), len(self._counts))
            reserve = int(need * (n_exp / float(max(len(self._counts), 1))) * 1.03) if n_exp > len(self._counts) else 0
            if self._store is not None and len(self._store.counts):
                reserve = max(reserve, int(1.5 * need))
            try:
                import torch
                free, _t = torch.cuda.mem_get_info()
                free += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
                if reserve * 290 > free // 2:        # (never more than half of what is left)
                    reserve = 0
            except Exception:                         # noqa: BLE001
                reserve = 0
            new = kernels.DescriptorStore(self._counts, train_layout=want_train,
                                          reserve_rows=reserve, reserve_images=n_exp if reserve else 0)
            if self._store is not None and len(self._store.counts):
                old = self._store
                n_old, n_old2, k = int(old.offsets[-1]), int(old.offsets2[-1]), len(old.counts)
                new.desc[:n_old].copy_(old.desc[:n_old])
                new.norm_q[:n_old].copy_(old.norm_q[:n_old])
                new.norm_t[:n_old].copy_(old.norm_t[:n_old])
                # ... and the train-side (parity partitioned) form the fast kernel reads
                if new.has_train_layout and not old.has_train_layout:
                    old.ensure_train_layout()
                if old.has_train_layout and new.has_train_layout:
                    new.desc2[:n_old2].copy_(old.desc2[:n_old2])
                    new.norm2[:n_old2].copy_(old.norm2[:n_old2])
                    new.cinit[:n_old2].copy_(old.cinit[:n_old2])
                    new.perm[:n_old2].copy_(old.perm[:n_old2])
                    new.meta[:k].copy_(old.meta[:k])
                # ... and the sorted form of the symmetric sweep
                n_old3 = int(old.offsets3[-1])
                for name in ('desc3', 'sn2', 'sct', 'sperm', 'sinv'):
                    getattr(new, name)[:n_old3].copy_(getattr(old, name)[:n_old3])
            self._store = new
        if _tm is not None:
            _tm.append(time.perf_counter())
        # a run of consecutive new slots whose descriptors are host arrays goes up in ONE step
        # (DescriptorStore.set_images); anything else image by image
        keep = []
        bulk = []
        if len(pend) >= 8:
            k = 0
            while k < len(pend) and pend[k][0] == pend[0][0] + k and isinstance(pend[k][1], np.ndarray) \
                    and pend[k][1].dtype in (np.float32, np.uint8) and pend[k][1].ndim == 2 \
                    and pend[k][1].shape[1] == 128:
                k += 1
            if k >= 8:
                bulk, pend = pend[:k], pend[k:]
                keep.append(self._store.set_images(bulk[0][0], [d for _s, d in bulk]))
        keep += [self._store.set_image(slot, des if hasattr(des, 'data_ptr') else
                                       np.ascontiguousarray(des), sync=False)
                 for slot, des in pend]
        if _tm is not None:
            _tm.append(time.perf_counter())
        if keep:
            import torch
            torch.cuda.current_stream().synchronize()        # one sync for the whole batch
        if _tm is not None:
            _round_trace.append(('pre', 'store(): arena %.3f enqueue %.3f sync %.3f (%d images)'
                                 % (_tm[1] - _tm[0], _tm[2] - _tm[1], time.perf_counter() - _tm[2], len(pend) + len(bulk)),
                                 time.perf_counter()))
        del keep
        self._pending = []
        if want_train and not self._store.has_train_layout:
            self._store.ensure_train_layout()
        return self._store


def _kp_xy(image):
    """[N,2] float32 array of kp.pt, cached on the image while kp_list is the same object.
    The cache holds a WEAK reference to the list (a plain-list kp_list, which cannot be weakly
    referenced, is identified by id + length): find_matches' periodic flush sets kp_list to
    None and the keypoint objects of ~50 k-keypoint images must go with it, as in the reference
    (scripts/lib/matcher.py:1008-1026) -- a strong reference here kept every list ever matched."""
    import weakref
    kl = image.kp_list
    if isinstance(kl, KeyPointList):
        return kl.xy()                                 # array backed: no objects, no cache needed
    cache = getattr(image, '_iamx_xy', None)
    if cache is not None:
        tag, xy = cache
        if tag is None:
            return xy                                  # array handed over without a list (bench)
        if isinstance(tag, weakref.ref):
            if tag() is kl and kl is not None:
                return xy
        elif kl is not None and tag == (id(kl), len(kl)):
            return xy
    xy = np.array([kp.pt for kp in kl], np.float32).reshape(-1, 2)
    try:
        tag = weakref.ref(kl)
    except TypeError:
        tag = (id(kl), len(kl))
    image._iamx_xy = (tag, xy)
    return xy


def kp_key2(xy):
    """[N,2] int32: round-half-even(100 * kp.pt) computed exactly -- two keypoints have the same
    "%.2f-%.2f" % kp.pt string (matcher.py:166-167) iff their key pairs are equal.  kp.pt holds
    float32 values; x * 2**40 is an exact integer for every float32 in [2**-16, 2**14) and
    smaller values print as 0.00 either way."""
    if isinstance(xy, np.ndarray) and xy.dtype == np.float32 and xy.ndim == 2 and xy.shape[1] == 2 \
            and xy.flags.c_contiguous and len(xy) >= 256:
        # (one pass in libiamx: the numpy form below is six temporaries of the array's size)
        try:
            key = np.empty(xy.shape, np.int32)
            rc = _lib.lib().iamx_kp_key2(xy.ctypes.data_as(ctypes.c_void_p), len(xy),
                                         key.ctypes.data_as(ctypes.c_void_p))
        except OSError:
            rc = None
        if rc == 0:
            return key
        if rc is not None:
            raise ValueError("keypoint coordinates outside [0, 16384)")
    x = np.asarray(xy, np.float64).reshape(-1, 2)
    if x.size and (x.min() < 0 or x.max() >= 16384.0):
        raise ValueError("keypoint coordinates outside [0, 16384)")
    m = (x * float(1 << 40)).astype(np.int64) * 100
    q, rem = m >> 40, m & ((1 << 40) - 1)
    half = 1 << 39
    q = q + ((rem > half) | ((rem == half) & ((q & 1) == 1)))
    return q.astype(np.int32)


# --------------------------------------------------------------------------------------
# configuration -- matcher.py:43-80
# --------------------------------------------------------------------------------------
def configure():
    global detect_scale, the_matcher, max_distance, min_pairs
    detect_scale = detector_node.getFloat('scale')
    detector_str = detector_node.getString('detector')
    if detector_str == 'SIFT':
        max_distance = 270.0
    elif detector_str in ('SURF', 'ORB', 'Star'):
        _log("Detector", detector_str, "is not on the MI355X matching path (SIFT only:",
             "integer valued 128-D descriptors)")
        quit()
    else:
        _log("Detector not specified or not known:", detector_str)
        quit()
    the_matcher = DeviceMatcher()
    min_pairs = matcher_node.getFloat('min_pairs')
    try:                                          # (page-locked upload staging ready before find_matches)
        from . import kernels
        kernels.prewarm_upload_stage()
    except Exception:                             # noqa: BLE001  (no device / no library: find_matches says so)
        pass


# --------------------------------------------------------------------------------------
# filters -- matcher.py:157-200
# --------------------------------------------------------------------------------------
def _dedupe(xy1, xy2, idx_pairs):
    """first-come-wins on the "%.2f-%.2f" keys of either end point (matcher.py:157-182)."""
    if len(idx_pairs) == 0:
        return [], 0
    p = np.asarray(idx_pairs, np.int64).reshape(-1, 2)
    k1 = ["%.2f-%.2f" % (x, y) for x, y in xy1[p[:, 0]].tolist()]
    k2 = ["%.2f-%.2f" % (x, y) for x, y in xy2[p[:, 1]].tolist()]
    used1, used2, out = set(), set(), []
    for pair, a, b in zip(idx_pairs, k1, k2):
        if a in used1 or b in used2:
            continue
        used1.add(a)
        used2.add(b)
        out.append(pair)
    return out, len(idx_pairs) - len(out)


def filter_duplicates(i1, i2, idx_pairs):
    result, count = _dedupe(_kp_xy(i1), _kp_xy(i2), idx_pairs)
    if count > 0:
        _qlog("  removed %d/%d duplicate features" % (count, len(idx_pairs)))
    return result


def filter_cross_check(idx_pairs1, idx_pairs2):
    """keep p in fwd iff [p1,p0] in rev; rev becomes the mirror of the kept fwd (:187-200)."""
    rev = set((int(a), int(b)) for a, b in idx_pairs2)
    new1 = [pair for pair in idx_pairs1 if (int(pair[1]), int(pair[0])) in rev]
    new2 = [[pair[1], pair[0]] for pair in new1]
    if len(idx_pairs1) != len(new1) or len(idx_pairs2) != len(new2):
        _qlog("  cross check: (%d, %d) => (%d, %d)" % (len(idx_pairs1), len(idx_pairs2),
                                                       len(new1), len(new2)))
    return new1, new2


# --------------------------------------------------------------------------------------
# single pair entry points -- matcher.py:203-347
# --------------------------------------------------------------------------------------
def raw_matches(i1, i2, k=2):
    """Returns (idx[nq,2] int32, dist[nq,2] float32) -- the content of cv2's list of DMatch
    pairs -- or [] under the reference's guards (:205-210)."""
    if i1.des_list is None or i2.des_list is None:
        return []
    if len(i1.des_list.shape) == 0 or i1.des_list.shape[0] <= 1:
        return []
    if len(i2.des_list.shape) == 0 or i2.des_list.shape[0] <= 1:
        return []
    if the_matcher is None:
        configure()
    matches = the_matcher.knnMatch(i1.des_list, i2.des_list, k=k)
    _qlog("  raw matches:", len(matches[0]))
    return matches


def _threshold_sort_clip(q_rows, t_rows, metric):
    """matcher.py:258-269: stable sort by metric (survivors only), clip to the best 2000."""
    order = np.argsort(metric, kind='stable')[:MYMAX]
    return np.stack([q_rows[order], t_rows[order]], axis=1)


def _post_filter(i1, i2, thresh_pairs, xy=None):
    """matcher.py:271-300 from the thresholded list on: min_pairs gate, GMS, de-dup, gate."""
    if len(thresh_pairs) < min_pairs:
        return []
    size = _camera_size()
    xy1, xy2 = xy if xy is not None else (_kp_xy(i1), _kp_xy(i2))
    mask = gms_inlier_mask(xy1, xy2, size, size, thresh_pairs, with_rotation=True,
                           with_scale=False, threshold_factor=5.0)
    idx_pairs = [[int(a), int(b)] for a, b in thresh_pairs[mask]]
    idx_pairs, count = _dedupe(xy1, xy2, idx_pairs)
    if count > 0:
        _qlog("  removed %d/%d duplicate features" % (count, count + len(idx_pairs)))
    _qlog("  initial matches =", len(idx_pairs))
    if len(idx_pairs) < min_pairs:
        return []
    return idx_pairs


def basic_pair_matches(i1, i2):
    matches = raw_matches(i1, i2)
    match_ratio = matcher_node.getFloat('match_ratio')
    if len(matches) == 0:
        raise ZeroDivisionError("float division by zero")       # :232 sum / len(matches)
    idx, dist = matches
    d0 = dist[:, 0].astype(np.float64)
    d1 = dist[:, 1].astype(np.float64)
    good = d0 <= d1 * match_ratio                                # :225-235 statistics
    _qlog("  avg dist:", d0.sum() / len(d0))
    if good.any():
        _qlog("  avg good dist:", d0[good].sum() / good.sum(), "(%d)" % good.sum())
    _qlog("  max good dist:", d0[good].max() if good.any() else 0)
    if np.any(d1 == 0.0):
        raise ZeroDivisionError("float division by zero")       # :255
    metric = d0 * (d0 / d1)
    keep = np.nonzero(metric < max_distance * match_ratio)[0]
    thresh_pairs = _threshold_sort_clip(keep.astype(np.int32), idx[keep, 0], metric[keep])
    _qlog("  quality matches:", len(keep))
    if len(keep) > MYMAX:
        _qlog("  clipping to:", MYMAX)
    return _post_filter(i1, i2, thresh_pairs)


def bidirectional_pair_matches(i1, i2, review=False):
    if i1 == i2:
        _log("We shouldn't see this, but i1 == i2", i1.name, i2.name)
        return [], []
    idx_pairs1 = basic_pair_matches(i1, i2)
    if len(idx_pairs1) >= min_pairs:
        idx_pairs2 = basic_pair_matches(i2, i1)
    else:
        idx_pairs2 = []
    return filter_cross_check(idx_pairs1, idx_pairs2)


# --------------------------------------------------------------------------------------
# pair schedule -- matcher.py:852-916
# --------------------------------------------------------------------------------------
_WORK_MATRIX_MAX = 4096      # images up to which the schedule is built from the n x n distance matrix (~0.4 GB of temporaries)


def _work_arrays(proj, sort):
    """the pair schedule as arrays (dist float64, i int32, j int32), i < j"""
    image_list = proj.image_list
    ned = np.array([im.get_camera_pose()[0] for im in image_list], np.float64).reshape(-1, 3)
    intervals = np.linalg.norm(ned[1:] - ned[:-1], axis=1)
    median = float(np.median(intervals))
    average = float(np.average(intervals))
    _log("Median pair interval: %.1f m" % median)
    _log("Average pair interval: %.1f m" % average)
    _log("Max adjacent interval: %.1f m" % np.max(intervals))
    if median < average:
        median = average
    median_int = int(round(median))
    if median_int == 0:
        median_int = 1
    min_dist = matcher_node.getFloat("min_dist") if matcher_node.hasChild("min_dist") else 0
    max_dist = matcher_node.getFloat("max_dist") if matcher_node.hasChild("max_dist") \
        else median_int * 4
    # 'neighbours' = the reference at HEAD (distance window disabled by `if False and`,
    # :896-903); 'distance' = that window; 'all-pairs' = BASELINE.json's schedule.
    schedule = matcher_node.getString("schedule") if matcher_node.hasChild("schedule") \
        else "neighbours"
    _log('Generating work list for range:', min_dist, '-', max_dist)
    n = len(image_list)
    interval = median_int * 1.3
    if n <= _WORK_MATRIX_MAX:
        # the distance MATRIX, then the selected upper-triangle entries in row-major order (what
        # triu_indices + two gathers of [pairs, 3] coordinates deliver, 3x faster; the sums are
        # formed in np.linalg.norm's order, so the distances are bit-identical)
        acc = None
        for d in range(3):
            c = ned[:, d]
            diff = c[None, :] - c[:, None]
            sq = diff * diff
            acc = sq if acc is None else acc + sq
        D = np.sqrt(acc)
        upper = np.triu(np.ones((n, n), bool), 1)
        if schedule == "all-pairs":
            mask = upper
        elif schedule == "distance":
            mask = upper & (((D >= min_dist) & (D <= max_dist)) | ~np.triu(np.ones((n, n), bool), 5))
        else:
            mask = upper & ~np.triu(np.ones((n, n), bool), 5)
        dist = D[mask]
        idx = np.arange(n, dtype=np.int32)
        ii = np.broadcast_to(idx[:, None], (n, n))[mask]
        jj = np.broadcast_to(idx, (n, n))[mask]
    else:
        ii, jj = np.triu_indices(n, k=1)
        dist = np.linalg.norm(ned[jj] - ned[ii], axis=1)
        if schedule == "all-pairs":
            sel = np.ones(len(ii), bool)
        elif schedule == "distance":
            sel = ((dist >= min_dist) & (dist <= max_dist)) | (np.abs(ii - jj) <= 4)
        else:
            sel = np.abs(ii - jj) <= 4
        ii, jj, dist = ii[sel], jj[sel], dist[sel]
    # python's round() is round-half-even, like np.rint
    steps = np.rint(dist / interval)
    ddist = steps * interval
    if sort:
        # stable, like the reference's sorted(); the key is a small integer (numpy radix-sorts
        # 16-bit keys: 4x faster than the merge sort of the float64 distances, same permutation)
        key = steps.astype(np.int16) if len(steps) and 0 <= steps.min() and steps.max() < 32767 else ddist
        order = np.argsort(key, kind='stable')
        ddist, ii, jj = ddist[order], ii[order], jj[order]
    return ddist.astype(np.float64), ii.astype(np.int32), jj.astype(np.int32)


def _work_list(proj, sort):
    """[[dist, i, j], ...] -- the reference's work_list (matcher.py:886-916)"""
    d, i, j = _work_arrays(proj, sort)
    return [[float(a), int(b), int(c)] for a, b, c in zip(d.tolist(), i.tolist(), j.tolist())]


# --------------------------------------------------------------------------------------
# the batched pair loop -- matcher.py:918-1031
# --------------------------------------------------------------------------------------
FREEZE_MIN_OBJECTS = 2_000_000     # tracked objects a find_matches call must leave before gc.freeze()


def _tracked_objects():
    # (with the collector disabled the young generation's counter just accumulates: container
    #  allocations minus deallocations since the last collection, O(1) to read)
    return gc.get_count()[0]


@contextlib.contextmanager
def _no_gc():
    """find_matches keeps tens of millions of 2-element lists alive (the match_list contract);
    every generation-2 pass of the cyclic collector would walk all of them again"""
    was = gc.isenabled()
    gc.disable()
    before = _tracked_objects() if was else 0
    try:
        yield
    finally:
        if was:
            # The lists made inside are acyclic and stay: out of the collector's generations, or
            # the first allocation after enable() pays a full pass over all of them (~0.3 s).
            # freeze() also pins whatever garbage CYCLES exist at that moment -- exception
            # tracebacks, objects holding device or page-locked buffers -- for the life of the
            # process, so only a call that left a survey's worth of objects freezes (a small
            # call's walk is cheap anyway).  (Collecting the young generations first, as was tried
            # in round 5, IS the walk over everything the call made -- the collector was off, all of
            # it sits in generation 0: 0.3 s at the end of the 2812-image survey,
            # profiles/r5_fm_config2_final.txt against r4_fm_config2_run.txt.)
            if hasattr(gc, 'freeze') and _tracked_objects() - before > FREEZE_MIN_OBJECTS:
                gc.freeze()
            gc.enable()


def _ensure_features(image):
    if image.kp_list is None or image.des_list is None or not len(image.kp_list) \
            or not len(image.des_list):
        image.detect_features(detect_scale)


def _rows_of(image):
    dm = the_matcher
    if isinstance(dm, DeviceMatcher):
        return dm.rows_of(image)
    return int(image.des_list.shape[0]) if image.des_list is not None and \
        len(getattr(image.des_list, 'shape', ())) else 0


def _have_features(image):
    """True when the device matcher can match this image without touching the detector: its
    features are on the host, or they arrived from the rank that detected it"""
    dm = the_matcher
    if isinstance(dm, DeviceMatcher) and image.name in dm._adopted:
        return True
    return not (image.kp_list is None or image.des_list is None or not len(image.kp_list)
                or not len(image.des_list))


def detect_features_sharded(proj, images=None):
    """Several ranks: every image is detected (or loaded from its cache) by exactly ONE rank
    (dist.owner_of_images: contiguous blocks), then the uint8 descriptors and the keypoint
    positions are exchanged (dist.exchange_features: one large transfer per owner and buffer,
    RCCL over xGMI between GPUs) and registered with the device matcher -- SURVEY.md 8e "SIFT
    detect".  Without it every rank of an all-pairs schedule runs the detector on every image.
    `images`: indices into proj.image_list (default: all).  Returns the keypoint counts."""
    import torch
    from . import dist as _dist
    rank, ws = _dist.world()
    if the_matcher is None:
        configure()
    idx = list(range(len(proj.image_list))) if images is None else [int(i) for i in images]
    owner_local = _dist.owner_of_images(len(idx), ws)
    own = {}
    failure = None
    try:
        for k, i in enumerate(idx):
            if owner_local[k] == rank:
                im = proj.image_list[i]
                _ensure_features(im)
                des = np.asarray(im.des_list)
                own[k] = (np.clip(np.rint(des), 0, 255).astype(np.uint8) if des.dtype != np.uint8 else des,
                          _kp_xy(im))
    except (Exception, SystemExit) as exc:
        if ws == 1:
            raise
        failure = exc
    if ws == 1:
        return np.array([len(own[k][0]) for k in range(len(idx))], np.int64)
    # a detector failure on one rank (quit() on an image-size mismatch, an unreadable file) is
    # re-raised on every rank before the exchange: the others would wait in it forever
    _dist.raise_on_any_rank(failure)
    dev = None
    if torch.distributed.get_backend() == 'nccl':
        from . import _lib
        dev = _lib.require_gpu()
    counts, desc, xy = _dist.exchange_features(owner_local, own, rank, ws, device=dev)
    off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    xy_host = xy.cpu().numpy()
    dm = the_matcher
    for k, i in enumerate(idx):
        if owner_local[k] != rank and isinstance(dm, DeviceMatcher):
            im = proj.image_list[i]
            if im.name not in dm._slots:
                dm.adopt(im.name, desc[int(off[k]):int(off[k + 1])], xy_host[int(off[k]):int(off[k + 1])])
    return counts


_host_sets = {}          # (n, clip, surface) -> free page-locked result buffer sets
_arena = None
_ws_pool = []            # free kernels.PairWorkspace objects (a round's outputs; reused)
_post_pool = {}          # (n, clip, surface) -> free device result sets of the per-pair filters


def _upload_arena():
    global _arena
    from . import kernels
    if _arena is None:
        _arena = kernels.UploadArena()
    return _arena


def _workspace(rows, pairs):
    """a PairWorkspace with room for `rows` query rows / `pairs` ordered pairs from the pool (a
    round's workspace goes back when its results have been read)"""
    from . import kernels
    for k, w in enumerate(_ws_pool):
        if w.max_rows >= rows and w.max_pairs >= pairs:
            return _ws_pool.pop(k)
    if len(_ws_pool) >= 3:                       # (a survey's rounds are all the same size)
        _ws_pool.pop(0)
    return kernels.PairWorkspace(int(rows * 1.05) + 1024, pairs)


def _post_set(n, clip, dev, surface):
    """surface: False, True (triangulated heights + similarity fits) or 'fit' (the fits only)"""
    import torch
    free = _post_pool.setdefault((n, clip, surface), [])
    if free:
        return free.pop()
    post = dict(key=(n, clip, surface),
                cnt=torch.empty(n, dtype=torch.int32, device=dev),
                pairs=torch.empty((n, clip, 2), dtype=torch.int32, device=dev),
                scratch=torch.empty((n, 2, clip, 2), dtype=torch.int32, device=dev),
                stat=torch.empty((n, 4), dtype=torch.int32, device=dev),
                status=torch.empty(n, dtype=torch.int32, device=dev))
    if surface is True:
        post['z'] = torch.empty((n, clip), dtype=torch.float64, device=dev)
    if surface:
        post['aff'] = torch.empty((n, 2, 6), dtype=torch.float64, device=dev)
        post['aff_ok'] = torch.empty((n, 2), dtype=torch.int32, device=dev)
    return post


def _tensor_bytes(obj):
    import torch
    vals = obj.values() if isinstance(obj, dict) else vars(obj).values()
    seen, total = set(), 0
    for t in vals:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            key = t.untyped_storage().data_ptr()
            if key not in seen:
                seen.add(key)
                total += t.untyped_storage().nbytes()
    return total


def device_memory_report():
    """bytes of HBM this module holds right now: the descriptor arena (what grows with the
    survey), the keypoint arena, and the pooled per-round buffers (what does not)"""
    dm = the_matcher if isinstance(the_matcher, DeviceMatcher) else None
    arena = _tensor_bytes(dm._store) if dm is not None and dm._store is not None else 0
    kp = sum(t.untyped_storage().nbytes() for t in dm._kp_dev[1:4]) if dm is not None and dm._kp_dev else 0
    pooled = sum(_tensor_bytes(w) for w in _ws_pool)
    post = sum(_tensor_bytes(p_) for lst in _post_pool.values() for p_ in lst)
    return dict(descriptor_arena_bytes=int(arena), keypoint_arena_bytes=int(kp),
                pooled_workspace_bytes=int(pooled), pooled_result_set_bytes=int(post),
                images=len(dm._counts) if dm is not None else 0,
                descriptor_rows=int(sum(dm._counts)) if dm is not None else 0)


def device_memory_model(n_images, rows_per_image, train_layout=False):
    """HBM the matching stage needs for a survey of n_images x rows_per_image descriptors, bytes:
      arena      per row: 128 B int8 + 3 x 4 B (norms, key scratch) in the original order, 128 B +
                 4 x 4 B in the sorted order (+ 128 B + 3 x 4 B parity partitioned with
                 train_layout; find_matches does without), rows padded to 128 per image;
                 keypoints 16 B per row (kp.pt float32 x 2, "%.2f" keys int32 x 2)
      per round  a workspace of <= BATCH_BYTES (the per-row partial bounds of the symmetric sweep
                 dominate), TWO of them pooled + a third while a round is in flight, and the
                 per-pair result slots of the filters (clip x 56 B per pair)
    Only the arena grows with the survey; the rest is bounded by BATCH_BYTES whatever N is."""
    rows = (int(rows_per_image) + 127) // 128 * 128
    per_row = (128 + 12) + (128 + 16) + ((128 + 12) if train_layout else 0)
    arena = n_images * rows * per_row + n_images * int(rows_per_image) * 16
    ppb = _pairs_per_batch(1.25 * rows_per_image)
    ws = ppb * _workspace_bytes_per_pair(1.25 * rows_per_image)
    clip = 2000
    return dict(arena_bytes=int(arena), workspace_bytes_each=int(ws), pairs_per_batch=int(ppb),
                result_set_bytes_each=int(ppb * clip * (8 + 16 + 8) + ppb * 200),
                peak_bytes=int(arena + 3 * ws + 2 * ppb * clip * 32))


def _recycle(h):
    """a finished round's device buffers back to their pools"""
    if h.get('ws') is not None:
        _ws_pool.append(h['ws'])
        h['ws'] = None
    post = h.get('post')
    if post is not None and 'key' in post:
        _post_pool[post['key']].append(post)
        h['post'] = None


def _host_set(n, clip, surface):
    """page-locked landing buffers for one batch's results (reused: allocating pinned memory
    costs milliseconds)"""
    import torch
    free = _host_sets.setdefault((n, clip, surface), [])
    if free:
        return free.pop()
    pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
    hs = dict(key=(n, clip, surface), zero_div=pin(2, torch.int32),
              count=pin(2 * n, torch.int32), cand=pin(2 * n, torch.int32))
    if clip:
        # the matches of the pairs that have some, packed back to back by the device
        # (iamx_match_pack_results writes these page-locked buffers directly)
        # (grows with the batch.  The closest pairs of a survey carry ~1700 matches each and a
        #  distance-sorted schedule puts 16 384 of them into ONE round: with room for 1024 per pair
        #  the first three rounds of the 2812-image survey overflowed into the slot-by-slot path,
        #  0.45 s each -- tools/find_matches_rate.py --trace)
        cap = min(n * clip, max(PACK_CAP, n * 2048))
        hs.update(cnt=pin(n, torch.int32), status=pin(n, torch.int32), cap=cap,
                  off=pin(n + 1, torch.int64), pk_pairs=pin((cap, 2), torch.int32))
        if surface:
            hs['pk_z'] = pin(cap, torch.float64)
            hs['aff'] = pin((n, 2, 6), torch.float64)
            hs['aff_ok'] = pin((n, 2), torch.int32)
    return hs


def _prewarm_pools(n, rows, surface, dev, stream, sets=3):
    """The first rounds of a survey each met an empty pool: a workspace of gigabytes, the device
    result set of the filters, the page-locked landing buffers (hundreds of megabytes whose pages
    are mapped on first touch) -- 0.4-0.5 s per round and set, with the device idle
    (tools/find_matches_rate.py --trace).  This allocates what `sets` rounds of n pairs with
    `rows`-descriptor images rotate through and touches the host buffers, on a helper thread beside
    the bookkeeping between the schedule and the first launch; everything goes through the pools'
    own constructors, so a round that asks for another size simply allocates as before."""
    import torch
    clip = int(_lib.lib().iamx_match_postfilter_clip())
    with torch.cuda.device(dev), torch.cuda.stream(stream):
        host = [_host_set(n, clip, surface) for _ in range(sets)]
        for hs in host:
            for t in hs.values():
                if isinstance(t, torch.Tensor) and t.numel() * t.element_size() >= (1 << 20):
                    _lib.check(_lib.lib().iamx_touch_pages(ctypes.c_void_p(t.data_ptr()),
                                                           t.numel() * t.element_size(), _HOST_THREADS),
                               'iamx_touch_pages')
        for hs in host:
            _host_sets[hs['key']].append(hs)
        post = [_post_set(n, clip, torch.device('cuda', dev), surface) for _ in range(2)]
        for p_ in post:
            _post_pool[p_['key']].append(p_)
        work = [_workspace(2 * n * int(rows), 2 * n) for _ in range(2)]
        # the bound tables of the symmetric sweep (kernels.sym_tables: a column table of the padded
        # row count per pair, the per-row partial bounds of every workgroup of the other image)
        cap3 = (int(rows) + 127) // 128 * 128
        wg_rows = 1024 if rows >= 4096 else (512 if rows >= 2048 else 256)
        for w in work:
            w.ensure_sym(n * cap3, n * ((int(rows) + wg_rows - 1) // wg_rows) * cap3)
        _ws_pool.extend(work)


def _launch_batch(batch, match_ratio, device_filters=True, surface=False, one_direction=False):
    """batch: list of (i1, i2) image objects.  ENQUEUES, on the current stream, the device k=2
    NN + metric threshold for both directions of every pair, the per-pair filters (sort/clip,
    GMS, de-dup, gates, cross check: iamx_match_postfilter) unless `device_filters` is False,
    with `surface` the DLT triangulation of every pair's matches (one launch for the batch), and
    the downloads of the results into page-locked buffers.  Nothing is waited for: the returned
    handle goes to _finish_batch(), and find_matches launches the next batch before it finishes
    this one so that the GPU works while python builds the match lists."""
    import torch
    from . import kernels
    from .kernels import _ptr, check, lib, stream_ptr
    dm = the_matcher
    by_id = {}                                   # one slot_of() per image, not per pair
    if isinstance(batch, _PairView):
        # a round of find_matches: index arrays, no tuple per pair
        slot_of = np.zeros(len(batch.image_list), np.int32)
        for u in batch.uniq.tolist():
            im = batch.image_list[u]
            slot_of[u] = dm.slot_of(im)
            by_id[id(im)] = (int(slot_of[u]), im)
        slots = np.stack([slot_of[batch.pi], slot_of[batch.pj]], 1)
    else:
        for pair in batch:
            for im in pair:
                if id(im) not in by_id:
                    by_id[id(im)] = (dm.slot_of(im), im)
        slots = [(by_id[id(a)][0], by_id[id(b)][0]) for a, b in batch]
    if dm._pending or dm._kp_dev is None or dm._kp_dev[0] != len(dm._counts):
        # new images: the descriptor / keypoint arenas are about to be rebuilt, and the previous
        # round's side-stream kernels may still be reading the old ones
        torch.cuda.current_stream().wait_stream(_side_stream())
    if one_direction and not dm.want_train_layout():      # (the parity-partitioned copy, built on first use)
        one_direction = False
    store = dm.store()
    arena = _upload_arena()
    arena.begin()
    d_proj = d_ik = None
    if surface is True and device_filters:
        # (every small table of the batch goes up with ONE asynchronous copy: a pageable upload
        #  blocks the host until the previous batch's kernels have run)
        from . import smart as _smart
        PROJ = np.zeros((len(dm._counts), 12))
        epoch = dm._pose_epoch
        for s, im in by_id.values():
            hit = dm._proj.get(s)
            if hit is None or epoch is None or hit[2] is not epoch:
                # (reading a pose back from the property tree costs more than a pair's share of
                #  the kernels: once per image and find_matches call, else whenever it changed)
                pose = im.get_camera_pose()
                if hit is None or hit[0] != pose:
                    hit = (pose, _smart.projection_matrix(im).ravel(), epoch)
                else:
                    hit = (hit[0], hit[1], epoch)
                dm._proj[s] = hit
            PROJ[s] = hit[1]
        IK = np.linalg.inv(np.asarray(_deps.camera().get_K(), float))
        d_proj = arena.put(PROJ)
        d_ik = arena.put(np.ascontiguousarray(IK.ravel()))
    sl = np.asarray(slots, np.int32).reshape(-1, 2)
    ordered = np.concatenate([sl, sl[:, ::-1]])
    pb = kernels.PairBatch(store, ordered, sym=False if one_direction else None, arena=arena)
    arena.commit()
    _route['rounds'][0 if pb.sym else 1] += 1
    ws = _workspace(pb.rows, pb.n_pairs)
    # (the kernels only ever RAISE this flag: a pooled workspace that reported a zero distance --
    #  the ZeroDivisionError of matcher.py:255 -- would fail every later batch that draws it)
    ws.flags.zero_()
    if device_filters:
        kp_off, xy, key2 = dm.keypoints()        # (may upload: before the kernels, like PROJ)
    thresh = max_distance * match_ratio
    # The sweep fills the machine for ~95 % of a round; everything behind it -- candidate test,
    # exact stage, per-pair filters, triangulation, packing, the downloads -- is small, latency
    # bound work that runs on a SECOND stream beside the next round's sweep (each round has its
    # own workspace / result set from the pools, handed back only after the host has read them).
    main = torch.cuda.current_stream()
    side = _side_stream()
    if pb.rows and pb.n_pairs:
        pb.run_knn2_fast(ws)
    swept = torch.cuda.Event()
    swept.record(main)
    side.wait_event(swept)
    with torch.cuda.stream(side):
        h = _launch_batch_tail(batch, pb, ws, thresh, device_filters, surface, d_proj, d_ik,
                               kp_off if device_filters else None, xy if device_filters else None,
                               key2 if device_filters else None)
    # (the side stream reads these tables; the caching allocator knows them by the stream they were
    #  made on, so a table that outgrew the upload arena's slot -- a fresh main-stream tensor --
    #  must stay referenced until the host has seen the batch's `done` event)
    h['tables'] = (d_proj, d_ik, kp_off if device_filters else None,
                   xy if device_filters else None, key2 if device_filters else None)
    return h


_side = {}


def _side_stream():
    """the second stream of _launch_batch, one per device"""
    import torch
    dev = torch.cuda.current_device()
    st = _side.get(dev)
    if st is None:
        st = _side[dev] = torch.cuda.Stream(device=dev)
    return st


def _launch_batch_tail(batch, pb, ws, thresh, device_filters, surface, d_proj, d_ik, kp_off, xy, key2):
    """the part of _launch_batch() behind the sweep (enqueued on the side stream)"""
    import torch
    from .kernels import _ptr, check, lib, stream_ptr
    if pb.rows and pb.n_pairs:
        pb.run_filter_fast(ws, thresh)
    n = len(batch)
    post = None
    clip = 0
    if device_filters:
        cam_w, cam_h = _camera_size()
        L = lib()
        clip = int(L.iamx_match_postfilter_clip())
        dev = xy.device
        post = _post_set(n, clip, dev, surface)
        check(L.iamx_match_postfilter(_ptr(ws.surv_off), _ptr(ws.surv_cnt), _ptr(ws.surv_q),
                                      _ptr(ws.surv_t), _ptr(ws.surv_metric), _ptr(pb.d_pairs),
                                      _ptr(kp_off), _ptr(xy), _ptr(key2), n, float(cam_w),
                                      float(cam_h), float(min_pairs), 5.0, _ptr(post['cnt']),
                                      _ptr(post['pairs']), _ptr(post['scratch']), _ptr(post['stat']),
                                      _ptr(post['status']), stream_ptr()), 'iamx_match_postfilter')
        if surface:
            post['tri_cnt'] = post['cnt']           # (0 for the pairs left to the host filters)
        if surface is True:
            check(L.iamx_triangulate_pairs(_ptr(pb.d_pairs), _ptr(d_proj), _ptr(d_ik), _ptr(kp_off),
                                           _ptr(xy), _ptr(post['tri_cnt']), _ptr(post['pairs']), n,
                                           clip, _ptr(post['z']), stream_ptr()),
                  'iamx_triangulate_pairs')
        if surface:
            # the similarity between the two images' keypoints, both ways (yaw-error estimate)
            check(L.iamx_similarity_pairs(_ptr(pb.d_pairs), _ptr(kp_off), _ptr(xy),
                                          _ptr(post['tri_cnt']), _ptr(post['pairs']), n, clip,
                                          _ptr(post['aff']), _ptr(post['aff_ok']), stream_ptr()),
                  'iamx_similarity_pairs')
    hs = _host_set(n, clip, surface if post is not None else False)
    hs['zero_div'].copy_(ws.flags, non_blocking=True)
    hs['count'].copy_(ws.surv_cnt[:2 * n], non_blocking=True)
    if pb.sym and pb.rows and pb.n_pairs:
        hs['cand'].copy_(ws.seg_count[:2 * n], non_blocking=True)     # candidate rows per ordered pair
    if post is not None:
        hs['cnt'].copy_(post['cnt'], non_blocking=True)
        hs['status'].copy_(post['status'], non_blocking=True)
        has_z = 'z' in post
        check(lib().iamx_match_pack_results(_ptr(post['cnt']), _ptr(post['status']), _ptr(post['pairs']),
                                            _ptr(post['z']) if has_z else None, n, clip, hs['cap'],
                                            _ptr(hs['off']), _ptr(hs['pk_pairs']),
                                            _ptr(hs['pk_z']) if has_z else None, stream_ptr()),
              'iamx_match_pack_results')
        if 'aff' in post:
            hs['aff'].copy_(post['aff'], non_blocking=True)
            hs['aff_ok'].copy_(post['aff_ok'], non_blocking=True)
    done_ev = torch.cuda.Event()
    done_ev.record()
    return dict(batch=batch, n=n, ws=ws, pb=pb, post=post, host=hs, done=done_ev, surface=surface,
                sym=bool(pb.sym and pb.rows and pb.n_pairs), rows=int(pb.rows))


def _finish_batch(h):
    """Waits for the batch's results and builds the python lists of the callers' contract.
    Returns per pair (match_fwd, match_rev, n_fwd_quality, n_rev_quality); when the batch was
    launched with `surface` a fifth entry: the pair's ground-surface statistics
    (avg, std, dist_m) of smart.estimate_surface_elevation(), or None where the pair took the
    host filter path (the caller then asks smart per pair)."""
    h['done'].synchronize()
    hs, n, ws, post = h['host'], h['n'], h['ws'], h['post']
    try:
        if int(hs['zero_div'][0]):
            raise ZeroDivisionError("float division by zero")       # matcher.py:255
        if int(hs['zero_div'][1]):
            # (never seen: the bound forms' exact stage covers every candidate row by
            #  construction; a .match file written past this would be silently wrong)
            raise RuntimeError("libiamx: %d query rows left unresolved by the exact stage"
                               % int(hs['zero_div'][1]))
        if h.get('sym'):
            _route['share'] = float(hs['cand'].numpy()[:2 * n].sum(dtype=np.int64)) / max(h['rows'], 1)
        count = hs['count'].numpy().astype(np.int64)
        first = sq = st = sm = status = cnt = lists = None
        z_rows = {}
        if post is not None:
            cnt, status = hs['cnt'].numpy(), hs['status'].numpy()
            lists, zs = _unpacked(hs, post, n)
            if zs is not None:
                aff, aff_ok = hs['aff'].numpy(), hs['aff_ok'].numpy()
                z_rows = {int(k): (zs(k),
                                   aff[k, 0].reshape(2, 3).copy() if aff_ok[k, 0] else None,
                                   aff[k, 1].reshape(2, 3).copy() if aff_ok[k, 1] else None)
                          for k in np.nonzero((status == 0) & (cnt > 0))[0]}
        if post is None or status.any():
            # survivor arrays: only the host filter path reads them (blocking copies)
            first, count, sq, st, sm = ws.survivors(h['pb'].n_pairs)
        out = []
        with _no_gc():
            _collect_batch(out, h['batch'], n, count, first, sq, st, sm, post is not None, status,
                           cnt, lists, z_rows, h['surface'])
    finally:
        _host_sets[hs['key']].append(hs)
        _recycle(h)
    return out


def _unpacked(hs, post, n):
    """accessors of a finished batch's packed download: (pairs_of(k) -> int32 [cnt, 2],
    z_of(k) -> float64 [cnt] or None).  A batch with more matches than the packed buffers hold
    has its slots copied the plain way."""
    cnt, off = hs['cnt'].numpy(), hs['off'].numpy()
    if int(off[n]) <= hs['cap']:
        pk, pz = hs['pk_pairs'].numpy(), hs['pk_z'].numpy() if 'pk_z' in hs and 'z' in post else None
        return ((lambda k: pk[off[k]:off[k] + cnt[k]]),
                (lambda k: pz[off[k]:off[k] + cnt[k]]) if pz is not None else None)
    full = post['pairs'].cpu().numpy()
    fz = post['z'].cpu().numpy() if 'z' in post else None
    return ((lambda k: full[k, :cnt[k]]), (lambda k: fz[k, :cnt[k]]) if fz is not None else None)


class _PairView(object):
    """the (image, image) pairs of one round of find_matches as two index arrays into
    proj.image_list (a sequence of pairs for the code that wants one)"""

    def __init__(self, image_list, pi, pj):
        self.image_list, self.pi, self.pj = image_list, pi, pj
        self.uniq = np.unique(np.concatenate([pi, pj]))

    def __len__(self):
        return len(self.pi)

    def __getitem__(self, k):
        return self.image_list[int(self.pi[k])], self.image_list[int(self.pj[k])]

    def __iter__(self):
        il = self.image_list
        return ((il[a], il[b]) for a, b in zip(self.pi.tolist(), self.pj.tolist()))


class _RoundResult(object):
    """what a round delivers, as arrays over its n pairs: n_fwd / n_rev (quality matches per
    direction), cc (cross checked matches), quiet (nothing left after the filters); for the h
    pairs WITH matches hit_rows (ascending) and their match rows fwd_all[lo[t]:hi[t]] (rev_all:
    the same rows with the columns swapped); with `fit` the pair distances and, per direction,
    the yaw values of the similarity fit (yaw_error, dist, relative course, weight) / whether a
    fit exists; mean / std of the triangulated "down" are filled in by the surface stage.
    `src`: where the surface stage finds the match rows without another copy (the page-locked
    buffer the device packed them into, or the device buffer a gather landed in), `release`
    hands the round's landing buffers back to their pool."""
    __slots__ = ('n', 'n_fwd', 'n_rev', 'cc', 'quiet', 'hit_rows', 'lo', 'hi', 'fwd_all', 'rev_all',
                 'fit', 'dist', 'same', 'yv_f', 'yv_r', 'aff_ok', 'mean', 'std', 'src', 'release')

    def __init__(self, n=0):
        self.n = n
        self.hit_rows = np.zeros(0, np.int64)
        self.lo = self.hi = np.zeros(0, np.int64)
        self.fwd_all = self.rev_all = np.zeros((0, 2), np.int32)
        self.fit = False
        self.dist = self.same = self.yv_f = self.yv_r = self.aff_ok = self.mean = self.std = None
        self.src = None
        self.release = None

    @property
    def hits(self):
        return _LazyHits(self)

    def done(self):
        if self.release is not None:
            self.release()
            self.release = None
        self.src = None


class _LazyHits(object):
    """the pairs WITH matches of a round as [(k, fwd, rev, surf)] -- k the pair's row, fwd / rev
    array-backed match lists, surf = (avg, std, dist_m, None, None, yaw values fwd, rev) or None
    -- made on access from the round's arrays (len(), indexing, slicing, iteration)"""
    __slots__ = ('R',)

    def __init__(self, R):
        self.R = R

    def __len__(self):
        return len(self.R.hit_rows)

    def _one(self, t):
        R = self.R
        a, b = int(R.lo[t]), int(R.hi[t])
        surf = None
        if R.fit:
            if R.same[t] or R.mean is None:
                surf = _NO_SURFACE
            else:
                surf = (-float(R.mean[t]), float(R.std[t]), float(R.dist[t]), None, None,
                        tuple(R.yv_f[t].tolist()) if R.aff_ok[t, 0] else None,
                        tuple(R.yv_r[t].tolist()) if R.aff_ok[t, 1] else None)
        return (int(R.hit_rows[t]), MatchPairs.of_array(R.fwd_all[a:b]), MatchPairs.of_array(R.rev_all[a:b]), surf)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._one(t) for t in range(*k.indices(len(self)))]
        return self._one(k if k >= 0 else k + len(self))

    def __iter__(self):
        return (self._one(t) for t in range(len(self)))


def _similarity_of_lists(view, rows, lists):
    """similarity fits of a few pairs whose match lists were made on the host (pairs the device
    filters handed back): [(aff [2, 6], ok [2])] through iamx_similarity_pairs"""
    import torch
    from .kernels import _ptr, check, lib, stream_ptr
    dm = the_matcher
    kp_off, xy, _k2 = dm.keypoints()
    dev = xy.device
    out = []
    for k, fwd in zip(rows, lists):
        i1, i2 = view[k]
        pairs = np.ascontiguousarray(np.asarray(fwd, np.int32).reshape(-1, 2))
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
        aff = torch.empty((1, 2, 6), dtype=torch.float64, device=dev)
        ok = torch.empty((1, 2), dtype=torch.int32, device=dev)
        d_img = t(np.array([[dm.slot_of(i1), dm.slot_of(i2)]]), torch.int32)
        d_cnt, d_pairs = t(np.array([len(pairs)]), torch.int32), t(pairs, torch.int32)
        check(lib().iamx_similarity_pairs(_ptr(d_img), _ptr(kp_off), _ptr(xy), _ptr(d_cnt), _ptr(d_pairs),
                                          1, max(len(pairs), 1), _ptr(aff), _ptr(ok), stream_ptr()),
              'iamx_similarity_pairs')
        out.append((aff[0].cpu().numpy(), ok[0].cpu().numpy()))
    return out


def _finish_batch_arrays(h):
    """_finish_batch() for find_matches: the same wait and the same results, but only the pairs
    that HAVE matches become python objects (and only when they are booked) -- on an all-pairs
    schedule 95-99 % of the pairs end with nothing and are just a mask here.  h: the handle of
    _launch_batch(view, ...).  -> _RoundResult"""
    if isinstance(h, list):
        # (a stand-in of _launch_batch / _finish_batch that already returns per-pair tuples:
        #  the CPU tests of the multi-rank logic)
        return _round_from_tuples(h)
    from . import smart as _smart
    _t0 = time.perf_counter()
    h['done'].synchronize()
    _t1 = time.perf_counter()
    hs, n, ws, post = h['host'], h['n'], h['ws'], h['post']
    R = _RoundResult(n)
    keep_host = False
    try:
        if int(hs['zero_div'][0]):
            raise ZeroDivisionError("float division by zero")       # matcher.py:255
        if int(hs['zero_div'][1]):
            # (never seen: the bound forms' exact stage covers every candidate row by
            #  construction; a .match file written past this would be silently wrong)
            raise RuntimeError("libiamx: %d query rows left unresolved by the exact stage"
                               % int(hs['zero_div'][1]))
        if h.get('sym'):
            _route['share'] = float(hs['cand'].numpy()[:2 * n].sum(dtype=np.int64)) / max(h['rows'], 1)
        count = hs['count'].numpy()
        R.n_fwd, R.n_rev = count[:n].astype(np.int64), count[n:2 * n].astype(np.int64)
        if post is None:
            raise ValueError("find_matches runs the device filters")
        cnt, status = hs['cnt'].numpy(), hs['status'].numpy()
        R.cc = np.where(status == 0, cnt, 0).astype(np.int64)
        R.quiet = (status == 0) & (cnt == 0)
        dev_rows = np.nonzero((status == 0) & (cnt > 0))[0]
        host_rows = np.nonzero(status != 0)[0]
        R.fit = bool(h['surface']) and 'aff' in hs
        view = h['batch']
        off_h = hs['off'].numpy()
        packed = int(off_h[n]) <= hs['cap']
        _hp = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
        c = cnt[dev_rows].astype(np.int64)
        if len(dev_rows) and packed:
            # the round's pair rows, forward and reversed, as TWO arrays the match lists are views
            # of (the download is a pinned buffer the next round reuses: copied once, one threaded
            # pass writes the copy and its column-swapped twin)
            src = hs['pk_pairs'].numpy()[:int(off_h[n])]
            R.fwd_all, R.rev_all = empty_huge(src.shape, np.int32), empty_huge(src.shape, np.int32)
            _lib.check(_lib.lib().iamx_pairs_fwd_rev(_hp(src), len(src), _hp(R.fwd_all), _hp(R.rev_all),
                                                     _HOST_THREADS), 'iamx_pairs_fwd_rev')
            lo = off_h[dev_rows].astype(np.int64)
            if R.fit:
                # (the surface stage reads the rows where the device packed them and writes the
                #  triangulated heights beside them: the landing buffers stay out of the pool
                #  until R.done())
                R.src = dict(kind='pinned', pairs=hs['pk_pairs'], z=hs['pk_z'], total=int(off_h[n]))
                keep_host = True
        elif len(dev_rows):
            # a batch whose matches did not fit the packed download: the slots, copied plainly
            full = post['pairs'].cpu().numpy()
            R.fwd_all = np.concatenate([full[k, :cnt[k]] for k in dev_rows.tolist()])
            R.rev_all = np.ascontiguousarray(R.fwd_all[:, ::-1])
            lo = np.concatenate([[0], np.cumsum(c)[:-1]]).astype(np.int64)
        else:
            lo = np.zeros(0, np.int64)
        R.hit_rows, R.lo, R.hi = dev_rows.astype(np.int64), lo, lo + c
        aff = hs['aff'].numpy()[dev_rows] if R.fit else None
        aff_ok = hs['aff_ok'].numpy()[dev_rows] if R.fit else None
        if len(host_rows):
            # pairs the device filters handed back (more candidates than their buffers hold):
            # the host filters, per pair, as before; their rows go behind the packed ones
            first, count_s, sq, st, sm = ws.survivors(h['pb'].n_pairs)
            extra_rows, extra = [], []
            for k in host_rows.tolist():
                i1, i2 = view[k]
                _ensure_features(i1)
                _ensure_features(i2)
                xy1, xy2 = _kp_xy(i1), _kp_xy(i2)
                a, b = first[k], first[k] + count_s[k]
                fwd = _post_filter(i1, i2, _threshold_sort_clip(sq[a:b], st[a:b], sm[a:b]), (xy1, xy2))
                rev = []
                if len(fwd) >= min_pairs:
                    a, b = first[n + k], first[n + k] + count_s[n + k]
                    rev = _post_filter(i2, i1, _threshold_sort_clip(sq[a:b], st[a:b], sm[a:b]),
                                       (xy2, xy1))
                fwd, rev = filter_cross_check(fwd, rev)
                R.cc[k] = len(fwd)
                if len(fwd) == 0 and len(rev) == 0:
                    R.quiet[k] = True
                else:
                    extra_rows.append(k)
                    extra.append(np.asarray(fwd, np.int32).reshape(-1, 2))
            if extra_rows:
                base = len(R.fwd_all)
                R.fwd_all = np.concatenate([R.fwd_all] + extra)
                R.rev_all = np.ascontiguousarray(R.fwd_all[:, ::-1])
                ec = np.array([len(e) for e in extra], np.int64)
                elo = base + np.concatenate([[0], np.cumsum(ec)[:-1]]).astype(np.int64)
                rows_all = np.concatenate([R.hit_rows, np.array(extra_rows, np.int64)])
                lo_all, hi_all = np.concatenate([R.lo, elo]), np.concatenate([R.hi, elo + ec])
                if R.fit:
                    fits = _similarity_of_lists(view, extra_rows, extra)
                    aff = np.concatenate([aff, np.stack([f[0] for f in fits])])
                    aff_ok = np.concatenate([aff_ok, np.stack([f[1] for f in fits])])
                order = np.argsort(rows_all, kind='stable')
                R.hit_rows, R.lo, R.hi = rows_all[order], lo_all[order], hi_all[order]
                if R.fit:
                    aff, aff_ok = aff[order], aff_ok[order]
                if R.src is not None:
                    # (the device-packed rows no longer cover every pair with matches: the surface
                    #  stage uploads the round's rows instead)
                    R.src = None
        if R.fit and len(R.hit_rows):
            ned, air_yaw = _smart.frozen_ned(view.image_list, with_yaw=True)
            a_idx, b_idx = view.pi[R.hit_rows], view.pj[R.hit_rows]
            R.dist = np.linalg.norm(ned[b_idx] - ned[a_idx], axis=1)
            R.same = a_idx == b_idx
            # the yaw error of both directions of every pair from the similarity fits
            # (smart.yaw_error_from_affine, vectorised over the round)
            R.yv_f = np.stack(_smart.yaw_errors_from_affines(ned[a_idx], air_yaw[a_idx], ned[b_idx],
                                                             aff[:, 0]), 1)
            R.yv_r = np.stack(_smart.yaw_errors_from_affines(ned[b_idx], air_yaw[b_idx], ned[a_idx],
                                                             aff[:, 1]), 1)
            R.aff_ok = aff_ok.astype(bool)
        elif R.fit:
            R.dist, R.same = np.zeros(0), np.zeros(0, bool)
            R.yv_f = R.yv_r = np.zeros((0, 4))
            R.aff_ok = np.zeros((0, 2), bool)
    finally:
        if keep_host and R.src is not None:
            R.release = lambda hs_=hs: _host_sets[hs_['key']].append(hs_)
        else:
            R.src = None
            _host_sets[hs['key']].append(hs)
        _recycle(h)
    if _round_trace is not None:        # (tools/find_matches_rate.py --trace)
        _round_trace.append(('finish', _t1 - _t0, time.perf_counter() - _t1, len(R.hit_rows)))
    return R


_round_trace = None


def _round_from_tuples(results):
    """per-pair tuples (fwd, rev, n_fwd, n_rev[, fit]) -> _RoundResult; fit = (dist_m, yaw values
    fwd or None, yaw values rev or None) where a stand-in of the device step supplies them"""
    n = len(results)
    R = _RoundResult(n)
    R.n_fwd = np.array([r[2] for r in results], np.int64).reshape(n)
    R.n_rev = np.array([r[3] for r in results], np.int64).reshape(n)
    R.cc = np.array([len(r[0]) for r in results], np.int64).reshape(n)
    R.quiet = np.array([len(r[0]) == 0 and len(r[1]) == 0 for r in results], bool).reshape(n)
    rows = np.nonzero(~R.quiet)[0]
    lists = [np.asarray(results[k][0], np.int32).reshape(-1, 2) for k in rows.tolist()]
    c = np.array([len(a) for a in lists], np.int64)
    R.hit_rows = rows.astype(np.int64)
    R.lo = np.concatenate([[0], np.cumsum(c)[:-1]]).astype(np.int64) if len(c) else np.zeros(0, np.int64)
    R.hi = R.lo + c
    R.fwd_all = np.concatenate(lists + [np.zeros((0, 2), np.int32)])
    R.rev_all = np.ascontiguousarray(R.fwd_all[:, ::-1])
    R.fit = bool(results) and all(len(r) > 4 and r[4] is not None for r in results)
    if R.fit:
        fits = [results[k][4] for k in rows.tolist()]
        zero = (0.0, 0.0, 0.0, 0.0)
        R.dist = np.array([f[0] for f in fits], np.float64).reshape(len(fits))
        R.same = np.zeros(len(fits), bool)
        R.yv_f = np.array([f[1] if f[1] is not None else zero for f in fits], np.float64).reshape(len(fits), 4)
        R.yv_r = np.array([f[2] if f[2] is not None else zero for f in fits], np.float64).reshape(len(fits), 4)
        R.aff_ok = np.array([[f[1] is not None, f[2] is not None] for f in fits], bool).reshape(len(fits), 2)
    return R


def _match_batch(batch, match_ratio, device_filters=True, surface=False):
    """one batch, start to end (see _launch_batch / _finish_batch)"""
    return _finish_batch(_launch_batch(batch, match_ratio, device_filters, surface))


class _Empty(list):
    """the shared empty match list of a pair without matches; its mutators raise (one object
    stands for every quiet pair of a batch).  find_matches stores a fresh [] per pair and image,
    like the reference."""
    __slots__ = ()

    def _immutable(self, *a, **k):
        raise TypeError("the shared empty match list is immutable: copy it (list(x)) first")

    append = extend = insert = __iadd__ = __imul__ = __setitem__ = _immutable


_NO_MATCHES = _Empty()
_NO_SURFACE = (None, None, 0.0, None, None)      # nothing to record (smart.py:200-201)


def _collect_batch(out, batch, n, count, first, sq, st, sm, have_post, status, cnt, lists, z_rows,
                   surface):
    """host side of a batch: the per-pair python lists the callers' contract asks for"""
    from . import smart as _smart
    count = count.tolist()
    quiet = None
    if have_post:
        # pairs the device filters finished with nothing left (most pairs of an all-pairs
        # schedule): one shared result, no per-pair work beyond the counts of the log
        quiet = ((status == 0) & (cnt == 0)).tolist()
    for k in range(n):
        n_fwd, n_rev = count[k], count[n + k]
        if quiet is not None and quiet[k]:
            out.append((_NO_MATCHES, _NO_MATCHES, n_fwd, n_rev, _NO_SURFACE) if surface
                       else (_NO_MATCHES, _NO_MATCHES, n_fwd, n_rev))
            continue
        if have_post and status[k] == 0:
            # array-backed lists (matchpairs.py): `lists` is a page-locked buffer that the next
            # batch reuses, the pair's rows are copied out of it once
            c = cnt[k]
            if c:
                both = np.empty((2, c, 2), np.int32)
                both[0] = lists(k)
                both[1] = both[0, :, ::-1]
                fwd, rev = MatchPairs(both[0]), MatchPairs(both[1])
            else:
                fwd, rev = [], []
        else:
            i1, i2 = batch[k]
            _ensure_features(i1)         # the keypoints may have been flushed since the launch
            _ensure_features(i2)
            xy1, xy2 = _kp_xy(i1), _kp_xy(i2)
            a, b = first[k], first[k] + count[k]
            fwd = _post_filter(i1, i2, _threshold_sort_clip(sq[a:b], st[a:b], sm[a:b]), (xy1, xy2))
            rev = []
            if len(fwd) >= min_pairs:
                a, b = first[n + k], first[n + k] + count[n + k]
                rev = _post_filter(i2, i1, _threshold_sort_clip(sq[a:b], st[a:b], sm[a:b]),
                                   (xy2, xy1))
            fwd, rev = filter_cross_check(fwd, rev)
        if not surface:
            out.append((fwd, rev, n_fwd, n_rev))
            continue
        surf = None
        if have_post and status[k] == 0:
            i1, i2 = batch[k]
            if i1 == i2 or k not in z_rows:
                surf = _NO_SURFACE
            else:
                zk, aff_fwd, aff_rev = z_rows[k]
                surf = (-np.average(zk), np.std(zk), _smart._pair_distance(i1, i2), aff_fwd, aff_rev)
        out.append((fwd, rev, n_fwd, n_rev, surf))


def _camera_size():
    cam = _deps.camera()
    w, h = cam.get_image_params()
    if not w or not h:
        _log("Zero image sizes will crash matchGMS():", w, h)
        _log("Recommend removing all meta/*.feat files and")
        _log("rerun the matching step.")
        _log("... or do some coding to add this information to the")
        _log("ImageAnalysis/meta/<image_name>.json files")
        quit()
    return w, h


def _launch_lines(lines, match_ratio, surface=False, raw=None):
    """lines: [(dist, i, j, i1, i2)]: enqueue the batch (see _launch_batch); raw: the descriptor
    row counts of every line's two images when the caller already has them"""
    batch = [(l[3], l[4]) for l in lines]
    if raw is None:
        raw = [(_rows_of(l[3]), _rows_of(l[4])) for l in lines]
    handle = _launch_batch(batch, match_ratio, surface=True) if surface \
        else _launch_batch(batch, match_ratio)
    return lines, raw, handle


def _finish_lines(launched):
    """Returns [(i, j, match_fwd, match_rev, surf)]; surf is the pair's (avg, std, dist_m)
    surface statistics when the launch asked for them, else None."""
    lines, raw, handle = launched
    results = _finish_batch(handle)
    out, records = [], []
    for (dist, i, j, i1, i2), (raw1, raw2), res in zip(lines, raw, results):
        match_fwd, match_rev, n_fwd, n_rev = res[:4]
        # the reference's seven qlog() lines per pair (matcher.py:311-343), one log record per
        # batch: at millions of pairs the per-line bookkeeping costs more than the GPU work
        records.append("Matching %s vs %s\n  separation (approx) = %.0f (m)\n  raw matches: %d\n"
                       "  quality matches: %d\n  raw matches: %d\n  quality matches: %d\n"
                       "  cross checked matches: %d"
                       % (i1.name, i2.name, dist, raw1, n_fwd, raw2, n_rev, len(match_fwd)))
        out.append((i, j, match_fwd, match_rev, res[4] if len(res) > 4 else None))
    if records:
        _qlog("\n".join(records))
    return out


def _process_batch(lines, match_ratio, surface=False):
    return _finish_lines(_launch_lines(lines, match_ratio, surface))


def find_matches(proj, K, strategy="smart", transform="homography", sort=False, review=False):
    """transform / review are accepted and unused, as on the reference's live path.

    With torch.distributed initialised (one process per GPU) the pairs that still need
    matching are dealt to the ranks in contiguous blocks (dist.shard_pairs); after every round
    the per-pair match lists are gathered on rank 0, which keeps the survey's bookkeeping
    (match lists, surface estimates) and writes the files -- the other ranks only record the
    pairs they matched themselves."""
    with _no_gc():
        try:
            _find_matches(proj, K, strategy, transform, sort, review)
        finally:
            if isinstance(the_matcher, DeviceMatcher):
                the_matcher._pose_epoch = None
            if _deps.smart() is not None:
                _deps.smart().freeze_poses(False)


def _find_matches(proj, K, strategy, transform, sort, review):
    if strategy != "traditional":
        _log("Match strategy", strategy, "is not on the MI355X path; only 'traditional'",
             "(bidirectional k=2 NN + metric + GMS + cross check) is.")
        quit()
    if the_matcher is None:
        configure()
    if isinstance(the_matcher, DeviceMatcher):
        the_matcher._pose_epoch = object()
    _route_reset()
    smart = _deps.smart()
    if smart is not None:
        # (camera positions and aircraft yaw angles do not change inside one call -- what the
        #  feedback changes is the camera ATTITUDE --, and reading one back from the property
        #  tree costs more than a pair's share of the kernels: smart.frozen_ned)
        smart.freeze_poses(True)
    run = _MatchRun(proj, sort)
    run.schedule()
    run.prepare()
    run.rounds()
    run.finish()
    print('Pair-wise matches successfully saved.')


# the wire layout of one rank's share of a round (dist.pack_arrays order)
_PART_FIELDS = ('seq', 'pi', 'pj', 'dist', 'raw1', 'raw2', 'n_fwd', 'n_rev', 'cc', 'quiet',
                'hit_rows', 'lo', 'hi', 'fwd_all', 'fit', 'hit_dist', 'same', 'yv_f', 'yv_r', 'aff_ok')
_PART_FWD = _PART_FIELDS.index('fwd_all')


class _Part(object):
    """one rank's share of one round on the booking rank: the pairs (seq = position in the
    schedule, image indices, log figures) and their _RoundResult"""
    __slots__ = ('seq', 'pi', 'pj', 'dist', 'raw1', 'raw2', 'R')

    def __init__(self, seq, pi, pj, dist, raw1, raw2, R):
        self.seq, self.pi, self.pj, self.dist, self.raw1, self.raw2, self.R = seq, pi, pj, dist, raw1, raw2, R

    def to_wire(self):
        from . import dist as _dist
        R = self.R
        h = len(R.hit_rows)
        z1, z4, z2 = np.zeros(h), np.zeros((h, 4)), np.zeros((h, 2), bool)
        fit = R.fit and R.dist is not None
        # (only the rows of pairs with matches, back to back in hit order)
        if h and not (R.lo[0] == 0 and np.array_equal(R.lo[1:], R.hi[:-1]) and R.hi[-1] == len(R.fwd_all)):
            fwd = np.concatenate([R.fwd_all[a:b] for a, b in zip(R.lo.tolist(), R.hi.tolist())])
            c = R.hi - R.lo
            lo = np.concatenate([[0], np.cumsum(c)[:-1]]).astype(np.int64)
            hi = lo + c
        else:
            fwd, lo, hi = R.fwd_all, R.lo, R.hi
        return _dist.pack_arrays([
            np.asarray(self.seq, np.int64), np.asarray(self.pi, np.int32), np.asarray(self.pj, np.int32),
            np.asarray(self.dist, np.float64), np.asarray(self.raw1, np.int64), np.asarray(self.raw2, np.int64),
            np.asarray(R.n_fwd, np.int64), np.asarray(R.n_rev, np.int64), np.asarray(R.cc, np.int64),
            np.asarray(R.quiet, bool), np.asarray(R.hit_rows, np.int64), np.asarray(lo, np.int64),
            np.asarray(hi, np.int64), np.ascontiguousarray(fwd, np.int32).reshape(-1, 2),
            np.array([bool(R.fit)]), np.asarray(R.dist if fit else z1, np.float64),
            np.asarray(R.same if fit else np.zeros(h, bool), bool),
            np.asarray(R.yv_f if fit else z4, np.float64).reshape(h, 4),
            np.asarray(R.yv_r if fit else z4, np.float64).reshape(h, 4),
            np.asarray(R.aff_ok if fit else z2, bool).reshape(h, 2)])

    @classmethod
    def from_wire(cls, host, dev):
        """host: the uint8 buffer of to_wire(); dev: the same bytes on the device (RCCL gather)
        or None"""
        from . import dist as _dist
        arrs, offs = _dist.unpack_arrays(host)
        f = dict(zip(_PART_FIELDS, arrs))
        R = _RoundResult(len(f['seq']))
        R.n_fwd, R.n_rev, R.cc, R.quiet = f['n_fwd'], f['n_rev'], f['cc'], f['quiet']
        R.hit_rows, R.lo, R.hi = f['hit_rows'], f['lo'], f['hi']
        R.fwd_all = f['fwd_all']
        R.rev_all = np.ascontiguousarray(R.fwd_all[:, ::-1])
        R.fit = bool(f['fit'][0])
        if R.fit:
            R.dist, R.same, R.yv_f, R.yv_r, R.aff_ok = f['hit_dist'], f['same'], f['yv_f'], f['yv_r'], f['aff_ok']
            if dev is not None and len(R.fwd_all):
                import torch
                o = offs[_PART_FWD]
                R.src = dict(kind='device', total=len(R.fwd_all),
                             pairs=dev[o:o + R.fwd_all.nbytes].view(torch.int32).view(-1, 2))
        return cls(f['seq'], f['pi'], f['pj'], f['dist'], f['raw1'], f['raw2'], R)


_surface = {}


def _surface_stream():
    """the stream of the surface stage's triangulation, one per device: beside the next round's
    sweep (main stream) and its tail (side stream, queued behind that sweep)"""
    import torch
    dev = torch.cuda.current_device()
    st = _surface.get(dev)
    if st is None:
        st = _surface[dev] = torch.cuda.Stream(device=dev)
    return st


_z_pool = []             # page-locked float64 landing buffers of the surface stage


def _z_buffer(total):
    import torch
    for k, t in enumerate(_z_pool):
        if t.numel() >= total:
            return _z_pool.pop(k)
    return torch.empty(max(int(total * 1.25), 1 << 16), dtype=torch.float64, pin_memory=True)


_surface_up = None       # kernels.UploadArena of the surface stage's tables (its own stream)


def _surface_device(image_list, jobs, waiter=None):
    """The device half of the surface stage: jobs = [dict(pi, pj [h] image indices, proj
    [h, 2, 12], m_off [h + 1], pairs = host int32 [T, 2], src = _RoundResult.src or None)] ->
    per job the float64 [T] NED "down" of every match (iamx_triangulate_packed: every pair with
    its own two projection matrices).  Match rows are read where they already are -- the
    page-locked buffer the device packed them into, the device buffer of the round's gather --
    and uploaded only when they are in neither; heights land in page-locked memory directly.
    waiter(event): what the host does until the event has happened (default: wait)."""
    global _surface_up
    import torch
    from . import kernels
    from .kernels import _ptr, check, lib
    dm = the_matcher
    dev = kernels.require_gpu()
    # image index -> slot of the keypoint arena, once per image of the round
    slot_of = {}
    slots = []
    for job in jobs:
        both = np.stack([job['pi'], job['pj']], 1)
        for x in np.unique(both).tolist():
            if x not in slot_of:
                slot_of[x] = dm.slot_known(image_list[x])
        table = np.zeros(int(both.max()) + 1 if both.size else 1, np.int32)
        for x, s_ in slot_of.items():
            if x < len(table):
                table[x] = s_
        slots.append(table[both].astype(np.int32))
    kp_off, xy, _k2 = dm.keypoints()
    IK = np.ascontiguousarray(np.linalg.inv(np.asarray(_deps.camera().get_K(), float)).ravel())
    st = _surface_stream()
    outs, keep = [], []
    with torch.cuda.stream(st):
        if _surface_up is None:
            _surface_up = kernels.UploadArena(nbytes=16 << 20, slots=3)
        _surface_up.begin()
        up = _surface_up.put
        d_ik = up(IK)
        launches = []
        for job, sl in zip(jobs, slots):
            total = int(job['m_off'][-1])
            src = job.get('src')
            if src is not None and src['kind'] == 'pinned':
                d_pairs, z = src['pairs'], src['z']
                own_z = False
            else:
                d_pairs = src['pairs'] if src is not None else \
                    torch.from_numpy(np.ascontiguousarray(job['pairs'])).to(dev, non_blocking=True)
                z, own_z = _z_buffer(total), True
            launches.append(((up(sl), up(job['proj']), up(job['m_off'])), d_pairs, z, len(sl), total))
            outs.append((z, total, own_z))
        _surface_up.commit()
        for tabs, d_pairs, z, h, total in launches:
            keep.append((tabs, d_pairs))
            check(lib().iamx_triangulate_packed(_ptr(tabs[0]), _ptr(tabs[1]), _ptr(d_ik), _ptr(kp_off),
                                                _ptr(xy), _ptr(tabs[2]), _ptr(d_pairs), h, total,
                                                _ptr(z), ctypes.c_void_p(st.cuda_stream)),
                  'iamx_triangulate_packed')
        ev = torch.cuda.Event()
        ev.record(st)
    if waiter is not None:
        waiter(ev)
    ev.synchronize()
    res = []
    for z, total, own_z in outs:
        zz = z.numpy()[:total]
        if own_z:
            zz = zz.copy()
            _z_pool.append(z)
        res.append(zz)
    del keep
    return res


def _segment_mean_std(z, starts, counts):
    """per segment np.mean / np.std of z[start : start + count] (smart.py:117-130), summed in
    numpy's order (iamx_segment_mean_std)"""
    mean, std = np.empty(len(counts)), np.empty(len(counts))
    if len(counts):
        _hp = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
        z = np.ascontiguousarray(z, np.float64)
        starts, counts = np.ascontiguousarray(starts, np.int64), np.ascontiguousarray(counts, np.int64)
        _lib.check(_lib.lib().iamx_segment_mean_std(_hp(z), _hp(starts), _hp(counts), len(counts), len(z),
                                                    _hp(mean), _hp(std), _HOST_THREADS),
                   'iamx_segment_mean_std')
    return mean, std


class _MatchRun(object):
    """One find_matches() call -- scripts/lib/matcher.py:852-1031 -- as its stages:
        schedule()   the work list (:856-916), the skip rule (:946-951), rank 0's list on every rank
        prepare()    sharded detection, pairs per round, pools, prefetch, the images already in memory
        rounds()     per round: launch (this rank's pairs of the round, dist.round_slice) ->
                     finish (arrays) -> exchange (one byte tensor per rank to rank 0) ->
                     surface stage (rank 0: pose feedback in schedule order + triangulation) ->
                     bookkeeping (queued, worked off while the host would wait for the device)
        finish()     the images' final poses, .match files, smart.json
    """
    BOOK_CHUNK = 64     # pairs with matches booked per step (~2 ms): see drain()

    def __init__(self, proj, sort):
        from . import dist as _dist
        self.proj, self.sort = proj, sort
        self.image_list = proj.image_list
        self.names = [im.name for im in self.image_list]
        self.rank, self.ws = _dist.world()
        self.smart = _deps.smart()
        self.surface = self.smart is not None
        self.t_start = time.time()
        self.device = isinstance(the_matcher, DeviceMatcher)
        self.failure = None
        self.prefetcher = None
        self.feedback = None
        self.seen = np.zeros(len(self.image_list), bool)      # images with a pair processed in this call
        self.n_done = 0
        self._stored_proj = {}
        self.first_pos = None

    # ---- the schedule ------------------------------------------------------------------------
    def _register_early(self):
        """One rank: the images whose features are in memory go to the device (descriptor arena,
        keypoint arena: ~0.3 ms of host time each) on a helper thread WHILE the schedule is built
        and sorted -- both are seconds on a survey of thousands of images, and the first round of
        a distance-sorted schedule needs nearly every image.  -> (thread, failures) or None"""
        image_list = self.image_list
        if not (self.ws == 1 and self.device and len(image_list) > 64):
            return None
        import threading
        import torch as _torch
        _dev = _torch.cuda.current_device()
        _stream = _torch.cuda.current_stream()
        ready = [im for im in image_list
                 if im.des_list is not None and im.kp_list is not None
                 and len(getattr(im.des_list, 'shape', ())) == 2 and len(im.des_list) > 1]
        failed = []

        def _register():
            try:
                with _torch.cuda.device(_dev), _torch.cuda.stream(_stream):
                    _t = [time.perf_counter()]
                    # (kp.pt arrays and their "%.2f" keys: 0.2 ms per image, outside the interpreter
                    #  lock -- eight threads; the slots are handed out in list order afterwards)
                    from concurrent.futures import ThreadPoolExecutor

                    def _keys(im):
                        xy = _kp_xy(im)
                        return xy, kp_key2(xy)
                    with ThreadPoolExecutor(max_workers=8, thread_name_prefix='iamx-keys') as ex:
                        pre = list(ex.map(_keys, ready))
                    for im, pk in zip(ready, pre):
                        the_matcher.slot_of(im, pre=pk)
                    _t.append(time.perf_counter())
                    the_matcher.store()
                    _t.append(time.perf_counter())
                    the_matcher.keypoints()
                    _t.append(time.perf_counter())
                    if _round_trace is not None:
                        _round_trace.append(('pre', 'registered: slots %.3f store %.3f keypoints %.3f'
                                             % (_t[1] - _t[0], _t[2] - _t[1], _t[3] - _t[2]), _t[3]))
            except BaseException as exc:          # noqa: BLE001  (re-raised on the calling thread)
                failed.append(exc)
        if len(ready) <= 1:
            return None
        th = threading.Thread(target=_register, name='iamx-register')
        th.start()
        return th, failed

    def schedule(self):
        from . import dist as _dist
        from .matchpairs import MatchDict, QuietLedger
        image_list, names = self.image_list, self.names
        if isinstance(the_matcher, DeviceMatcher):
            the_matcher.expect_images = len(image_list)      # (capacity of the descriptor arena)
        early = self._register_early()
        if _round_trace is not None:
            _round_trace.append(('pre', 'helper started', time.perf_counter()))
        try:
            wd, wi, wj = _work_arrays(self.proj, self.sort)
            if _round_trace is not None:
                _round_trace.append(('pre', 'schedule built', time.perf_counter()))
        finally:
            if early is not None:
                early[0].join()
        if _round_trace is not None:
            _round_trace.append(('pre', 'registration joined', time.perf_counter()))
        if early is not None and early[1]:
            raise early[1][0]
        self.match_ratio = matcher_node.getFloat('match_ratio')
        # ---- skip rule (:946-951), evaluated up front: a pair's state is only changed by itself
        if any(im.match_list for im in image_list):
            index_of = {n_: k for k, n_ in enumerate(names)}
            n_img = len(names)
            tried = np.zeros((n_img, n_img), bool)
            found = np.zeros((n_img, n_img), bool)
            for k, im in enumerate(image_list):
                for other, lst in im.match_list.items():
                    o = index_of.get(other)
                    if o is not None:
                        tried[k, o] = True
                        found[k, o] = len(lst) > 0
            both = tried[wi, wj] & tried[wj, wi]
            retry = both & ~found[wi, wj]
            skip = both & found[wi, wj]
            for k in np.nonzero(retry)[0][:50].tolist():
                _log("Retrying: ", names[wi[k]], "vs", names[wj[k]], "(no matches found previously)")
            for k in np.nonzero(skip)[0][:50].tolist():
                _log("Skipping: ", names[wi[k]], "vs", names[wj[k]], "already done.")
            if retry.sum() > 50 or skip.sum() > 50:
                _log("... %d pairs retried (no matches found previously), %d skipped (already done)"
                     % (int(retry.sum()), int(skip.sum())))
            keep = ~skip
            wd, wi, wj = wd[keep], wi[keep], wj[keep]
        if self.ws > 1:
            # Results are gathered on rank 0 only, so after an earlier find_matches call in this
            # process the other ranks' match lists are incomplete and their skip rule would keep a
            # different list: rank 0's list is THE list (every collective below assumes the ranks
            # agree on it -- sharded detection, the batch size, the rounds, the deal)
            wd, wi, wj = _dist.broadcast_object((wd, wi, wj) if self.rank == 0 else None, src=0)
        self.wd, self.wi, self.wj = wd, wi, wj
        self.n_pending = len(wi)
        # every image's match_list becomes a MatchDict tied to this call's ledger of quiet pairs
        self.ledger = QuietLedger(names, capacity=self.n_pending)
        for k, im in enumerate(image_list):
            if not isinstance(im.match_list, MatchDict):
                im.match_list = MatchDict(im.match_list)
            im.match_list.attach(self.ledger, k)
        if _round_trace is not None:
            _round_trace.append(('pre', 'ledger attached', time.perf_counter()))

    # ---- before the first launch -------------------------------------------------------------
    def _mine(self, rnd):
        from . import dist as _dist
        return _dist.round_slice(self.n_pending, rnd, self.ppb, self.rank, self.ws)

    def prepare(self):
        from . import dist as _dist
        from . import image as _image
        image_list, wi, wj = self.image_list, self.wi, self.wj
        rank, ws, n_pending = self.rank, self.ws, self.n_pending
        self.save_time = time.time()
        self.save_interval = SAVE_INTERVAL     # seconds
        _log("Processing worklist matches:")
        if ws > 1 and self.device and n_pending:
            # every image of the work list is detected by ONE rank; descriptors and keypoint
            # positions are exchanged once (a collective: the work list is the same on every rank)
            detect_features_sharded(self.proj, np.unique(np.concatenate([wi, wj])).tolist())
        # pairs per device batch: bounded by the workspace a batch needs at this survey's keypoint
        # counts (the images of a survey carry similar numbers: the largest count known so far, or
        # that of the first image of this rank's share, stands for all); the ranks agree on the
        # smallest value so that they run the same number of rounds (the gather is a collective)
        ppb = PAIRS_PER_BATCH
        known = []
        try:
            if n_pending > rank and self.device:
                known = [_rows_of(im) for im in image_list if _have_features(im)]
                if not known:
                    first = image_list[int(wi[rank])]
                    _ensure_features(first)
                    known = [_rows_of(first)]
                ppb = _pairs_per_batch(1.25 * max(known))
        except (Exception, SystemExit) as exc:
            if ws == 1:
                raise
            self.failure = exc           # re-raised on every rank by the first exchange
        if ws > 1:
            ppb = min(_dist.allgather_objects(ppb))
        self.ppb = ppb
        self.n_rounds = (n_pending + ws * ppb - 1) // (ws * ppb) if n_pending else 0
        # a survey of many full rounds: the pools its rounds rotate through are filled (and the
        # page-locked buffers touched) on a helper thread while the bookkeeping below runs
        self.warm = None
        if ws == 1 and self.device and self.failure is None and known and self.n_rounds >= PREWARM_ROUNDS:
            import threading
            import torch as _torch

            def _warm(n_=ppb, rows_=max(known), dev_=_torch.cuda.current_device(),
                      stream_=_torch.cuda.current_stream(), surface_='fit' if self.surface else False):
                try:
                    _prewarm_pools(n_, rows_, surface_, dev_, stream_)
                except Exception:             # noqa: BLE001  (best effort: a round allocates what it misses)
                    # (out of memory half way: give back what the pools hold so that the first real
                    #  round starts from a clean allocator instead of failing where this did)
                    _ws_pool.clear()
                    _post_pool.clear()
                    _torch.cuda.empty_cache()
            self.warm = threading.Thread(target=_warm, name='iamx-prewarm')
            self.warm.start()
        # ---- images this rank will have to detect / load, in the order the rounds reach them:
        # their JPEG decode or cache load runs ahead on worker threads (image.prefetch)
        if ws == 1:
            mine_imgs = np.stack([wi, wj], 1).ravel()
        else:
            sel = np.concatenate([self._mine(r) for r in range(self.n_rounds)] + [np.zeros(0, np.int64)])
            mine_imgs = np.stack([wi[sel], wj[sel]], 1).ravel()
        # (first occurrence of every image without sorting the 2 x pairs entries: assigning positions
        #  in reverse order leaves the smallest one; np.unique took 0.2 s on the 2812-image survey)
        first_pos = np.full(len(image_list), -1, np.int64)
        first_pos[mine_imgs[::-1]] = np.arange(len(mine_imgs) - 1, -1, -1, dtype=np.int64)
        mine_uniq = np.nonzero(first_pos >= 0)[0]
        first_at = first_pos[mine_uniq]
        self.first_pos = first_pos
        need = []
        for k in mine_imgs[np.sort(first_at)].tolist():
            im = image_list[k]
            if not _have_features(im) and \
                    getattr(type(im), 'detect_features', None) is _image.detect_features:
                need.append(im)
        self.prefetcher = _image.prefetch(need, scale=detect_scale) if need else None
        # The images whose features are in memory already go to the device in ONE step (descriptor
        # arena, keypoint arena): the first rounds of a distance-sorted schedule meet ~35 new images
        # each, and growing the arenas image by image rebuilt and re-synchronised them every round
        # (the device idled through the first ~70 rounds of a 2812-image survey: r3_fm_timeline)
        if self.device and self.failure is None:
            ready = [image_list[k] for k in mine_uniq.tolist()
                     if image_list[k].des_list is not None and image_list[k].kp_list is not None
                     and len(getattr(image_list[k].des_list, 'shape', ())) == 2 and len(image_list[k].des_list) > 1]
            if len(ready) > 1:
                for im in ready:
                    the_matcher.slot_of(im)
                the_matcher.store()
                the_matcher.keypoints()
        self.rows = np.zeros(len(image_list), np.int64)           # descriptor rows of the images seen so far
        self.rows_known = np.zeros(len(image_list), bool)         # (reset by the periodic cache flush)
        if self.surface and self.rank == 0:
            self.smart.begin_batch()
            self.feedback = self.smart.PoseFeedback(image_list)
        self._init_booking()

    # ---- one round on this rank ----------------------------------------------------------------
    def launch_round(self, rnd):
        sl = self._mine(rnd)
        if not len(sl):
            return None
        image_list, rows, rows_known = self.image_list, self.rows, self.rows_known
        view = _PairView(image_list, self.wi[sl], self.wj[sl])
        # per IMAGE of the round, not per pair: time stamp of the descriptor cache, detection
        # if the features are not there, the row count the log quotes
        now = time.time()
        # (in the order the prefetcher was given them -- first use in the schedule --: an image asked
        #  for ahead of its turn is not in flight yet and would be decoded and detected INLINE, one
        #  at a time, while the workers run ahead on images nobody waits for: 33 s instead of 10 s
        #  for 1024 frames met undetected, the way scripts/process.py:290 calls find_matches)
        uniq = view.uniq
        if self.first_pos is not None:
            uniq = uniq[np.argsort(self.first_pos[uniq], kind='stable')]
        for k in uniq.tolist():
            im = image_list[k]
            im.desc_timestamp = now
            if rows_known[k]:
                continue
            if not _have_features(im):
                _ensure_features(im)
            rows[k] = _rows_of(im)
            rows_known[k] = True
            if rows[k] <= 1:
                # raw_matches() returns [] and basic_pair_matches divides by len([]) (:232)
                raise ZeroDivisionError("float division by zero")
        handle = _launch_batch(view, self.match_ratio, surface='fit' if self.surface else False,
                               one_direction=_route_next())
        return view, sl, handle

    def finish_round(self, launched):
        """-> _Part: the round's pairs of this rank and their results as arrays"""
        view, sl, handle = launched
        R = _finish_batch_arrays(handle)
        return _Part(sl, view.pi, view.pj, self.wd[sl], self.rows[view.pi], self.rows[view.pj], R)

    def exchange(self, part):
        """every rank's part of the round on rank 0 (one byte tensor per rank, dist.gather_arrays);
        a failure on one rank is re-raised on every rank.  -> [_Part] on rank 0, [] elsewhere"""
        from . import dist as _dist
        if self.ws == 1:
            if self.failure is not None:
                raise self.failure
            return [part] if part is not None else []
        dev = None
        if self.device:
            import torch
            if torch.distributed.get_backend() == 'nccl':
                dev = _lib.require_gpu()
        buf = part.to_wire() if part is not None and self.failure is None else np.zeros(0, np.uint8)
        got = _dist.gather_arrays(buf, self.failure, device=dev)
        if part is not None:
            self.n_done_own = getattr(self, 'n_done_own', 0) + len(part.seq)
        if got is None:
            if part is not None:
                part.R.done()
            return []
        parts = []
        for r, (host, d) in enumerate(got):
            if r == self.rank:
                if part is not None:
                    parts.append(part)
            elif len(host):
                parts.append(_Part.from_wire(host, d))
        return parts

    # ---- the surface stage (booking rank) ------------------------------------------------------
    def _projection(self, x):
        hit = self._stored_proj.get(x)
        if hit is None:
            hit = self._stored_proj[x] = self.smart.projection_matrix(self.image_list[x]).ravel()
        return hit

    def surface_stage(self, parts):
        """lib/matcher.py:987-993 for the whole round: the pose feedback replayed in schedule order
        (smart.PoseFeedback), then every pair with matches triangulated with the two camera poses
        the reference's loop would hold when it reaches the pair; fills mean / std of the parts"""
        if self.feedback is None or not parts:
            for p_ in parts:
                p_.R.done()
            return
        fit_parts = [p_ for p_ in parts if p_.R.fit]
        if len(fit_parts) != len(parts):
            raise RuntimeError("find_matches: a rank delivered a round without similarity fits")
        # the round in schedule order (the ranks' shares interleave: seq = base + r + W t)
        if len(parts) == 1:
            p0 = parts[0]
            seq, pi, pj, quiet = p0.seq, p0.pi, p0.pj, p0.R.quiet
            hit_pos = p0.R.hit_rows
            part_of = np.zeros(len(hit_pos), np.int64)
            hit_t = np.arange(len(hit_pos), dtype=np.int64)
            yv_f, yv_r, ok = p0.R.yv_f, p0.R.yv_r, p0.R.aff_ok
        else:
            seq = np.concatenate([p_.seq for p_ in parts])
            order = np.argsort(seq, kind='stable')
            inv = np.empty(len(order), np.int64)
            inv[order] = np.arange(len(order))
            seq = seq[order]
            pi = np.concatenate([p_.pi for p_ in parts])[order]
            pj = np.concatenate([p_.pj for p_ in parts])[order]
            quiet = np.concatenate([p_.R.quiet for p_ in parts])[order]
            starts = np.concatenate([[0], np.cumsum([len(p_.seq) for p_ in parts])])
            pos = np.concatenate([inv[starts[k] + p_.R.hit_rows] for k, p_ in enumerate(parts)]
                                 + [np.zeros(0, np.int64)]).astype(np.int64)
            part_id = np.concatenate([np.full(len(p_.R.hit_rows), k, np.int64) for k, p_ in enumerate(parts)]
                                     + [np.zeros(0, np.int64)])
            t_in = np.concatenate([np.arange(len(p_.R.hit_rows), dtype=np.int64) for p_ in parts]
                                  + [np.zeros(0, np.int64)])
            o2 = np.argsort(pos, kind='stable')
            hit_pos, part_of, hit_t = pos[o2], part_id[o2], t_in[o2]
            cat = lambda name, shape: np.concatenate(
                [np.asarray(getattr(p_.R, name)).reshape((-1,) + shape) for p_ in parts])[o2]
            yv_f, yv_r, ok = cat('yv_f', (4,)), cat('yv_r', (4,)), cat('aff_ok', (2,))
        e1, e2, f1, f2 = self.feedback.feed(seq, pi, pj, quiet, hit_pos, yv_f, yv_r, ok)
        h = len(hit_pos)
        if h:
            hi_, hj_ = pi[hit_pos], pj[hit_pos]
            proj = np.empty((h, 2, 12))
            for side, (imgs, est, fresh) in enumerate(((hi_, e1, f1), (hj_, e2, f2))):
                moved = np.nonzero(~fresh)[0]
                if len(moved):
                    proj[moved, side] = self.feedback.projections(imgs[moved].tolist(),
                                                                  [est[t] for t in moved.tolist()])
                for t in np.nonzero(fresh)[0].tolist():
                    proj[t, side] = self._projection(int(imgs[t]))
            jobs, where = [], []
            for k, p_ in enumerate(parts):
                R = p_.R
                mine = np.nonzero(part_of == k)[0]            # ascending hit_t: the part's hit order
                if not len(mine):
                    continue
                lo, hi2 = R.lo, R.hi
                contiguous = bool(np.array_equal(lo[1:], hi2[:-1]))
                src = R.src if contiguous else None
                if contiguous:
                    base = int(lo[0]) if src is None else 0
                    m_off = np.concatenate([lo, hi2[-1:]]).astype(np.int64) - base
                    if src is not None and src['kind'] == 'pinned':
                        m_off[-1] = int(hi2[-1])              # (the packed buffer: offsets as they are)
                    pairs = R.fwd_all[int(lo[0]):int(hi2[-1])] if src is None else None
                else:
                    c = hi2 - lo
                    m_off = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
                    pairs = np.concatenate([R.fwd_all[a:b] for a, b in zip(lo.tolist(), hi2.tolist())])
                jobs.append(dict(pi=hi_[mine], pj=hj_[mine], proj=proj[mine], m_off=m_off,
                                 pairs=pairs, src=src))
                where.append((k, m_off))
            zs = _surface_device(self.image_list, jobs, waiter=self.drain_until)
            for (k, m_off), z in zip(where, zs):
                R = parts[k].R
                R.mean, R.std = _segment_mean_std(z, m_off[:-1], np.diff(m_off))
        for p_ in parts:
            if p_.R.mean is None and p_.R.fit:
                p_.R.mean = p_.R.std = np.zeros(len(p_.R.hit_rows))
            p_.R.done()

    # ---- bookkeeping (booking rank) ------------------------------------------------------------
    def _init_booking(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        self.backlog = []
        # the .match bytes of the rounds' pair lists (what saveMatches will write) are made on ONE
        # background thread, in libiamx without the interpreter lock (matchpairs.prepickle)
        self.pickler = ThreadPoolExecutor(max_workers=1, thread_name_prefix='iamx-pickle')
        self.pickling = []
        # Finished rounds whose pairs with matches are not booked yet (generators of book_steps()).
        # The schedule is sorted by distance: the pairs WITH matches -- all the per-pair host work,
        # ~30 us each -- sit in the first rounds, where the device would idle while they are booked,
        # and hundreds of rounds without a match follow, where the host would idle.  So a finished
        # round is only queued; its pairs are booked a chunk at a time in the moments the host would
        # otherwise WAIT for the device (drain(until=...)), or at the latest before a save.
        self.to_book = deque()
        # smart.json ahead of time.  Its content only changes when a pair WITH matches is booked; on a
        # distance-sorted schedule those all sit in the first rounds, and the host then idles through
        # hundreds of quiet ones.  Once everything seen is booked and several rounds in a row brought
        # nothing, the file is written on a helper thread beside the waits; the end of the call writes
        # it again only if a pair with matches was booked after that (booking first waits for a
        # writer that is still reading the tree).
        self.hits_booked = 0
        self.rounds_without_hits = 0
        self.early_smart = {'thread': None, 'hits': -1, 'error': None}

    def _early_smart_join(self):
        t = self.early_smart['thread']
        if t is not None and t.is_alive():
            t.join()

    def maybe_save_smart_early(self):
        smart = self.smart
        if self.rank != 0 or smart is None or self.early_smart['thread'] is not None \
                or self.rounds_without_hits < EARLY_SMART_ROUNDS:
            return
        # (quiet rounds: the host has time -- a few booking steps if any are left, ~2 ms each)
        for _ in range(4):
            if self.backlog or self.to_book:
                self.drain_step()
        if self.backlog or self.to_book or self.hits_booked == 0:
            return
        import threading
        smart.flush_aggregates()
        self.early_smart['hits'] = self.hits_booked
        early_smart, proj = self.early_smart, self.proj

        def _write():
            try:
                smart.save(proj.analysis_dir, polite=True)
            except BaseException as exc:          # noqa: BLE001  (the end of the call writes it again)
                early_smart['error'] = exc
        early_smart['thread'] = threading.Thread(target=_write, name='iamx-smart-save-early')
        early_smart['thread'].start()
        early_smart_stats['written'] += 1

    def drain_step(self):
        self._early_smart_join()
        if self.backlog:
            _kind, payload = self.backlog.pop(0)
            self.smart.record_round(payload)
            self.smart.materialize_pending()
        elif self.to_book:
            try:
                next(self.to_book[0])
            except StopIteration:
                self.to_book.popleft()

    def drain_until(self, event):
        self.drain(event)

    def drain(self, until=None):
        """work the queues off -- all of it, or while the event `until` has not happened yet
        (one ~2 ms step at a time); all of it includes the pickles in flight"""
        while (self.backlog or self.to_book) and (until is None or not until.query()):
            self.drain_step()
        if until is None:
            while self.pickling:
                self.pickling.pop(0).result()

    def book_steps(self, part):
        """the bookkeeping of one rank's round, as a generator: the once-per-round part first, then
        BOOK_CHUNK pairs with matches per step.  Nothing here depends on the order in which rounds
        or chunks are booked: match_list entries and the ledger carry the pair's seq and the newest
        seq wins; what DOES depend on the order -- the poses -- was settled by the surface stage."""
        names = self.names
        seq, pi, pj, quiet = part.seq, part.pi, part.pj, part.R.quiet
        raw1, raw2, n_fwd, n_rev = part.raw1, part.raw2, part.R.n_fwd, part.R.n_rev
        # ---- the log: the reference's seven qlog() lines per pair (matcher.py:311-343) for the
        # pairs with matches, one summary line for the round's pairs without (at millions of
        # pairs the per-pair lines cost more than the GPU work they describe)
        nq = int(quiet.sum())
        if nq:
            _qlog("%d pairs without matches (%s vs %s ... %s vs %s): raw matches %d..%d, "
                  "quality matches %d + %d in total"
                  % (nq, names[pi[0]], names[pj[0]], names[pi[-1]], names[pj[-1]],
                     int(min(raw1.min(), raw2.min())), int(max(raw1.max(), raw2.max())),
                     int(n_fwd[quiet].sum()), int(n_rev[quiet].sum())))
            # ---- pairs without matches: the ledger (two dictionary entries per pair, later)
            self.ledger.add(pi[quiet], pj[quiet], seq[quiet])
        self.seen[pi] = True
        self.seen[pj] = True
        hits = part.R.hits
        for c0 in range(0, len(hits), self.BOOK_CHUNK):
            yield
            self._book_hits(hits[c0:c0 + self.BOOK_CHUNK], part)

    def _book_hits(self, hits, part):
        from .matchpairs import prepickle
        names, image_list = self.names, self.image_list
        pi, pj, seq, dist = part.pi, part.pj, part.seq, part.dist
        raw1, raw2, n_fwd, n_rev, cc = part.raw1, part.raw2, part.R.n_fwd, part.R.n_rev, part.R.cc
        self._early_smart_join()
        self.hits_booked += len(hits)
        # (python numbers for the chunk's pairs once: numpy scalars cost ~1 us each to format)
        ks = np.fromiter((h[0] for h in hits), np.int64, len(hits))
        pik, pjk, sqk = pi[ks].tolist(), pj[ks].tolist(), seq[ks].tolist()
        records = ["Matching %s vs %s\n  separation (approx) = %.0f (m)\n  raw matches: %d\n"
                   "  quality matches: %d\n  raw matches: %d\n  quality matches: %d\n"
                   "  cross checked matches: %d" % (names[a_], names[b_], d_, r1_, f_, r2_, r_, c_)
                   for a_, b_, d_, r1_, f_, r2_, r_, c_ in zip(
                       pik, pjk, dist[ks].tolist(), raw1[ks].tolist(), n_fwd[ks].tolist(),
                       raw2[ks].tolist(), n_rev[ks].tolist(), cc[ks].tolist())]
        round_records = [] if self.surface else None
        for t_, (k, match_fwd, match_rev, surf) in enumerate(hits):
            i, j, sq = pik[t_], pjk[t_], sqk[t_]
            i1, i2 = image_list[i], image_list[j]
            if isinstance(match_fwd, np.ndarray):
                match_fwd, match_rev = MatchPairs(match_fwd), MatchPairs(match_rev)
            i1.match_list.set_in_order(i2.name, match_fwd, sq)
            i2.match_list.set_in_order(i1.name, match_rev, sq)
            i1.matches_clean = i2.matches_clean = False
            # ---- surface / yaw bookkeeping and the discard policy (:987-1005)
            avg = std = None
            if round_records is not None and surf is not None and len(surf) > 5:
                # the round's entries go to the property tree in one pass (smart.record_round);
                # the averages over an image's pairs are formed when the call ends
                avg, std = surf[0], surf[1]
                round_records.append((i1, i2, avg, std, surf[2], surf[5], surf[6]))
                if avg and std:
                    # (the pair's line of the log, in the chunk's record: one log call per chunk)
                    records.append("  %s %s surface est: %.1f std: %.1f" % (i1.name, i2.name, avg, std))
            if std and std >= 50 and len(match_fwd) < 100:
                _log("Std dev of surface triangulation blew up, matches are probably bad so "
                     "discarding them!", i1.name, i2.name, "avg:", avg, "std:", std,
                     "count:", len(match_fwd))
                i1.match_list.set_in_order(i2.name, [], sq)
                i2.match_list.set_in_order(i1.name, [], sq)
        if records:
            _qlog("\n".join(records))
        # the tree entries of these pairs (smart.record_round): a backlog item of their own
        if round_records:
            self.backlog.append(('smart', round_records))
        # ... and the .match bytes of their lists (what saveMatches will write): background thread
        lists = [dict.get(image_list[int(x)].match_list, image_list[int(y)].name)
                 for k_, _f, _r, _s in hits for x, y in ((pi[k_], pj[k_]), (pj[k_], pi[k_]))]
        self.pickling.append(self.pickler.submit(prepickle, lists))
        while len(self.pickling) > 256 and self.pickling[0].done():
            self.pickling.pop(0).result()

    # ---- the loop ------------------------------------------------------------------------------
    def rounds(self):
        """software pipeline: the GPU works on round r+1 while python turns round r into lists.
        An exception on one rank (ZeroDivisionError of a <= 1-descriptor image, quit() on an
        image-size mismatch) -- in the batch-size estimate, in the launch of round 0 or inside a
        round -- travels with the results and is re-raised on EVERY rank: the others would
        otherwise wait in the collective forever"""
        ws = self.ws
        in_flight = None
        if _round_trace is not None:
            _round_trace.append(('pre', 'ready to launch', time.perf_counter()))
        if self.warm is not None:
            self.warm.join()
        if _round_trace is not None:
            _round_trace.append(('pre', 'pools warm', time.perf_counter()))
        if self.n_rounds and self.failure is None:
            try:
                in_flight = self.launch_round(0)
            except (Exception, SystemExit) as exc:
                if ws == 1:
                    raise
                self.failure = exc
        for rnd in range(self.n_rounds):
            part = None
            try:
                if self.failure is None:
                    coming = self.launch_round(rnd + 1) if rnd + 1 < self.n_rounds else None
                    if in_flight is not None and isinstance(in_flight[2], dict):
                        self.drain(in_flight[2]['done'])     # instead of waiting for the round's results
                    part = self.finish_round(in_flight) if in_flight is not None else None
                    in_flight = coming
            except (Exception, SystemExit) as exc:
                if ws == 1:
                    raise
                self.failure, part = exc, None
            parts = self.exchange(part)
            if self.rank == 0:
                self.surface_stage(parts)
                for p_ in parts:
                    steps = self.book_steps(p_)
                    next(steps, None)             # the once-per-round part now, the pairs with matches later
                    nh = len(p_.R.hit_rows)
                    if nh:
                        self.to_book.append(steps)
                    self.n_done += len(p_.seq)
                    self.rounds_without_hits = 0 if nh else self.rounds_without_hits + 1
                self.maybe_save_smart_early()
            elif part is not None:
                self.n_done += len(part.seq)
            t_elapsed = time.time() - self.t_start
            # (ranks != 0 only see their own pairs: their progress is that of their own share)
            total = self.n_pending if self.rank == 0 else max(self.n_pending // ws, 1)
            percent = self.n_done / float(max(total, 1))
            t_remain = (t_elapsed / percent - t_elapsed) if percent > 0 else 0.0
            _qlog("%.1f%% done: %.1f (min) remaining" % (percent * 100.0, t_remain / 60.0))
            # ---- periodic save + host descriptor cache flush (:1008-1026)
            if time.time() >= self.save_time + self.save_interval:
                self.periodic_save()

    def periodic_save(self):
        proj = self.proj
        if self.rank == 0:
            self.drain()
            self._early_smart_join()
            if self.smart is not None:
                self.smart.flush_aggregates()
            for k in np.nonzero(self.seen)[0].tolist():
                self.image_list[k].matches_clean = False
            saveMatches(proj.image_list, check_if_dirty=True)
            if self.smart is not None:
                self.smart.save(proj.analysis_dir)
        self.save_time = time.time()
        time_list = [[i3.desc_timestamp, i3] for i3 in proj.image_list if i3.des_list is not None]
        time_list = sorted(time_list, key=lambda fields: fields[0], reverse=True)
        cache_size = 20 + 5 * (int(sqrt(len(proj.image_list))) + 1)
        flush_list = time_list[cache_size:]
        _qlog("flushing keypoint/descriptor cache - size: %d (over by: %d)"
              % (cache_size, len(flush_list)))
        for line in flush_list:
            _qlog('  clearing descriptors for:', line[1].name)
            line[1].kp_list = None
            line[1].des_list = None
            line[1].uv_list = None
        self.rows_known[:] = False

    def finish(self):
        proj, smart = self.proj, self.smart
        if self.prefetcher is not None:
            self.prefetcher.close()
        self.drain()
        self._early_smart_join()
        if self.rank == 0:
            if smart is not None:
                smart.flush_aggregates()
            if self.feedback is not None:
                # every image's pose as its LAST pair of the call left it (lib/matcher.py:990-993)
                self.feedback.settle()
            # (quiet pairs dirty both images' match lists, like the reference's assignments)
            for k in np.nonzero(self.seen)[0].tolist():
                self.image_list[k].matches_clean = False
            # smart.json is written beside the .match files (file writes release the interpreter),
            # unless the copy written ahead of time is still current
            saver = None
            es = self.early_smart
            smart_current = es['thread'] is not None and es['error'] is None and es['hits'] == self.hits_booked
            early_smart_stats['current_at_end'] += bool(smart_current)
            if smart is not None and not smart_current:
                import threading
                saver_error = []

                def _save_smart():
                    try:
                        smart.save(proj.analysis_dir)
                    except BaseException as exc:       # re-raised below, where the reference's call sits
                        saver_error.append(exc)
                saver = threading.Thread(target=_save_smart, name='iamx-smart-save')
                saver.start()
            try:
                saveMatches(proj.image_list)
            finally:
                if saver is not None:
                    saver.join()
            if saver is not None and saver_error:
                raise saver_error[0]
        self.pickler.shutdown(wait=True)


def saveMatches(image_list, check_if_dirty=False):
    # (writer threads and a concurrent smart.save were measured on the 2812-image survey: the
    #  interpreter lock makes four writers take twice as long as one)
    _log('saving matches and image meta data ...')
    for image in image_list:
        if check_if_dirty:
            if not image.matches_clean:
                image.save_matches()
        else:
            image.save_matches()
