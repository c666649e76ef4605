"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the node; "gloo" in the CPU tests).  The path shards by *image* (detection), by *pair*
(matching) and by *point* (BA); the data-path exchanges are

  * features: every image is detected by ONE rank (`owner_of_images`), then the uint8
    descriptors and float32 keypoint positions of all images are exchanged once
    (`exchange_features`: keypoint counts as objects, then ONE `all_gather_into_tensor` per buffer
    through `gather_store_shards` -- the ranks' blocks padded to the largest); bench.py
    all-gathers its equal-sized packed stores in place (1.47 GB per layout for the 2812-image
    survey);
  * match lists: the pairs of a round are dealt round-robin (`round_slice`: rank r takes pairs
    r, r + W, ... of the round's stretch of the schedule, so every rank sees the same mix of near
    and far pairs); the variable-length per-pair results go to rank 0 ONLY as ONE flat byte
    tensor per rank (`pack_arrays` / `gather_arrays`: a fixed layout of arrays -- counts, packed
    match rows, similarity fits --, gathered with `torch.distributed.gather`; on RCCL the bytes
    travel GPU to GPU and rank 0's surface stage reads the match rows where they land), which
    keeps the survey's bookkeeping and writes the files; a failure on one rank is re-raised on
    every rank first (`raise_on_any_rank`), and rank 0's list of pending pairs is what every
    rank shards (`broadcast_object`);
  * BA: observations AND the point part of every n-vector are sharded by point
    (`point_range`, `shard_observations_by_point`); per inner iteration the ranks all-reduce the
    camera-side part only (7 C + 1 doubles), per outer iteration the camera blocks of the
    normal equations (`allreduce_sum_`), see ba_solver.py.

xGMI is point to point (7 links x ~153 GB/s per GPU): the store exchange is done as a few
large collectives (one all-gather per buffer), never per image.
"""
import numpy as np
import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced split of range(n_items): sizes differ by at most one."""
    lo = (n_items * rank) // world_size
    hi = (n_items * (rank + 1)) // world_size
    return lo, hi


def owner_of_images(n_images, world_size):
    """image i is detected / packed by rank owner[i] (contiguous blocks: IO locality)."""
    owner = np.empty(n_images, np.int32)
    for r in range(world_size):
        lo, hi = shard_bounds(n_images, r, world_size)
        owner[lo:hi] = r
    return owner


def shard_pairs(pairs, rank, world_size):
    """Deal a train-major ordered list of unordered pairs to ranks in contiguous blocks so
    that (a) both directions of a pair stay on one rank (the cross check needs both) and
    (b) consecutive pairs on a rank share their train image (L2 reuse in knn2_pairs_kernel).
    Returns the slice owned by `rank`; the union over ranks is exactly `pairs`, in order."""
    lo, hi = shard_bounds(len(pairs), rank, world_size)
    return pairs[lo:hi]


def gather_store_shards(buffers, row_offsets, owner, rank, world_size, group=None):
    """In-place gather of a sharded packed store.

    buffers: list of (tensor, row_width) -- e.g. (desc.view(-1), 128), (norm_q, 1), (norm_t, 1),
    each covering ALL images; row_offsets: int64 [n_images+1] first packed row of each image;
    owner: int [n_images], contiguous blocks (owner_of_images).  On entry rank r has filled the
    rows of its own images; on exit every rank has every row.  ONE all-gather per buffer
    (all_gather_into_tensor: RCCL's ring over xGMI, every link busy at once): the ranks' blocks
    differ in size, so they travel padded to the largest one and are copied into place.
    """
    import torch.distributed as dist
    if world_size == 1:
        return
    owner = np.asarray(owner)
    spans = []
    for r in range(world_size):
        mine = np.nonzero(owner == r)[0]
        spans.append((int(row_offsets[mine[0]]), int(row_offsets[mine[-1] + 1])) if len(mine) else (0, 0))
    longest = max(hi - lo for lo, hi in spans)
    if longest == 0:
        return
    for buf, width in buffers:
        send = torch.zeros(longest * width, dtype=buf.dtype, device=buf.device)
        lo, hi = spans[rank]
        send[:(hi - lo) * width].copy_(buf[lo * width:hi * width])
        recv = torch.empty(world_size * longest * width, dtype=buf.dtype, device=buf.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        for r, (lo, hi) in enumerate(spans):
            if hi > lo and r != rank:
                buf[lo * width:hi * width].copy_(recv[r * longest * width:r * longest * width + (hi - lo) * width])
        del send, recv


def exchange_features(owner, own_images, rank, world_size, device=None, group=None):
    """Sharded detection -> every rank holds every image's descriptors and keypoint positions.

    owner: int [n_images] (owner_of_images: contiguous blocks); own_images: {image index:
    (des uint8 [n,128], xy float32 [n,2])} for the images this rank detected (or loaded from
    their cache).  Exchanges the keypoint counts (python objects, bytes), then ONE flat uint8
    descriptor buffer and ONE flat float32 coordinate buffer covering all images, filled by
    their owners block by block and all-gathered (gather_store_shards: one
    all_gather_into_tensor per buffer -- the RCCL descriptor all-gather of the north star).
    Returns (counts int64 [n_images], desc uint8 [sum n, 128], xy float32 [sum n, 2]) as torch
    tensors on `device` (CPU for gloo)."""
    n_images = len(owner)
    mine = {int(i): int(len(d)) for i, (d, _xy) in own_images.items()}
    parts = allgather_objects(mine, group=group)
    counts = np.zeros(n_images, np.int64)
    for r, part in enumerate(parts):
        for i, n in part.items():
            if owner[i] != r:
                raise ValueError("image %d was detected by rank %d but belongs to rank %d"
                                 % (i, r, owner[i]))
            counts[i] = n
    off = np.zeros(n_images + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    dev = device if device is not None else torch.device('cpu')
    desc = torch.zeros((max(int(off[-1]), 1), 128), dtype=torch.uint8, device=dev)
    xy = torch.zeros((max(int(off[-1]), 1), 2), dtype=torch.float32, device=dev)
    for i, (d, p) in own_images.items():
        lo, hi = int(off[i]), int(off[i + 1])
        if hi > lo:
            desc[lo:hi] = torch.from_numpy(np.ascontiguousarray(d, np.uint8)).to(dev)
            xy[lo:hi] = torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(dev)
    gather_store_shards([(desc.view(-1), 128), (xy.view(-1), 2)], off, np.asarray(owner), rank,
                        world_size, group=group)
    return counts, desc[:int(off[-1])], xy[:int(off[-1])]


def allgather_objects(obj, group=None):
    """Every rank contributes one picklable object; returns the list ordered by rank."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return [obj]
    out = [None] * ws
    dist.all_gather_object(out, obj, group=group)
    return out


def broadcast_object(obj, src=0, group=None):
    """rank `src`'s picklable object on every rank"""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return obj
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def _raise_note(r, n):
    if n[0] == 'ZeroDivisionError':
        raise ZeroDivisionError(n[1])
    if n[0] == 'SystemExit':
        raise SystemExit("rank %d quit: %s" % (r, n[1]))
    raise RuntimeError("rank %d failed: %s: %s" % (r, n[0], n[1]))


def raise_on_any_rank(failure, group=None):
    """A collective: all-gathers a one-line note of `failure` (an exception raised on this rank,
    or None) and re-raises the first one on EVERY rank -- the rank that failed raises its own
    exception -- so that no rank is left waiting in the collective that would follow."""
    rank, ws = world()
    if ws == 1:
        if failure is not None:
            raise failure
        return
    note = None if failure is None else (type(failure).__name__, str(failure))
    for r, n in enumerate(allgather_objects(note, group=group)):
        if n is not None:
            if r == rank and failure is not None:
                raise failure
            _raise_note(r, n)


def gather_results(results, failure=None, dst=0, group=None):
    """find_matches' per-round exchange.  The per-pair match lists go to rank `dst` ONLY (it
    keeps the survey's bookkeeping and writes the files; match consolidation and everything after
    it are rank-0 steps): returns the parts of all ranks, ordered by rank, on `dst` and just
    this rank's own part elsewhere.  `failure`: an exception raised on this rank while it
    produced its part -- a one-line note of it is all-gathered first and the failure is re-raised
    on EVERY rank, so that no rank is left waiting in a collective."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        if failure is not None:
            raise failure
        return [results]
    raise_on_any_rank(failure, group=group)
    parts = [None] * ws if rank == dst else None
    dist.gather_object(results, parts, dst=dst, group=group)
    return parts if rank == dst else [results]


def round_slice(n_items, rnd, per_rank, rank, world_size):
    """find_matches' deal: round `rnd` covers items [rnd * W * per_rank, (rnd + 1) * W * per_rank)
    of the schedule and rank r takes every W-th of them starting at r.  A distance-sorted schedule
    puts the overlapping pairs (all the exact-stage, filter and match-list work) first: contiguous
    blocks gave all of them to rank 0.  -> the item indices of `rank` (ascending int64)."""
    base = rnd * world_size * per_rank
    end = min(base + world_size * per_rank, n_items)
    return np.arange(base + rank, end, world_size, dtype=np.int64)


_WIRE_DTYPES = [np.dtype(t) for t in ('uint8', 'int32', 'int64', 'float64', 'bool', 'float32')]


def pack_arrays(arrays):
    """list of numpy arrays (<= 2-D, dtypes of _WIRE_DTYPES) -> ONE uint8 array: an int64 header
    (count, then per array: dtype code, rows, columns or -1, byte offset) and the arrays' bytes,
    each 16-byte aligned.  unpack_arrays() gives views of the same bytes back."""
    k = len(arrays)
    head = np.zeros(1 + 4 * k, np.int64)
    head[0] = k
    off = (head.nbytes + 15) & ~15
    arrs = []
    for t, a in enumerate(arrays):
        a = np.ascontiguousarray(a)
        code = _WIRE_DTYPES.index(a.dtype)
        if a.ndim == 0 or a.ndim > 2:
            raise ValueError("pack_arrays: 1-D or 2-D arrays only")
        head[1 + 4 * t:5 + 4 * t] = (code, a.shape[0], a.shape[1] if a.ndim == 2 else -1, off)
        arrs.append((off, a))
        off = (off + a.nbytes + 15) & ~15
    buf = np.zeros(max(off, 16), np.uint8)
    buf[:head.nbytes] = head.view(np.uint8)
    for o, a in arrs:
        if a.nbytes:
            buf[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    return buf


def unpack_arrays(buf):
    """-> (list of array views into buf, list of their byte offsets)"""
    buf = np.asarray(buf, np.uint8)
    k = int(buf[:8].view(np.int64)[0])
    head = buf[:8 * (1 + 4 * k)].view(np.int64)
    out, offs = [], []
    for t in range(k):
        code, d0, d1, off = (int(v) for v in head[1 + 4 * t:5 + 4 * t])
        dt = _WIRE_DTYPES[code]
        n = d0 * (d1 if d1 >= 0 else 1)
        a = buf[off:off + n * dt.itemsize].view(dt)
        out.append(a.reshape(d0, d1) if d1 >= 0 else a)
        offs.append(off)
    return out, offs


def gather_arrays(buf, failure=None, dst=0, group=None, device=None):
    """find_matches' per-round exchange: every rank's pack_arrays() buffer -> rank `dst`, as ONE
    tensor gather (sizes and failure notes travel in the small object all-gather before it; a
    failure on any rank is re-raised on EVERY rank, like gather_results).  `device`: where the
    bytes travel (the GPU for RCCL, None = host for gloo).
    -> on dst: [(host uint8 array, device uint8 tensor or None)] per rank; elsewhere None."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        if failure is not None:
            raise failure
        return [(buf, None)]
    note = None if failure is None else (type(failure).__name__, str(failure))
    infos = allgather_objects((note, 0 if buf is None else int(len(buf))), group=group)
    for r, (n, _size) in enumerate(infos):
        if n is not None:
            if r == rank and failure is not None:
                raise failure
            _raise_note(r, n)
    sizes = [sz for _n, sz in infos]
    longest = max(max(sizes), 16)
    dev = device if device is not None else torch.device('cpu')
    mine = torch.zeros(longest, dtype=torch.uint8, device=dev)
    if len(buf):
        mine[:len(buf)].copy_(torch.from_numpy(buf))
    parts = [torch.empty(longest, dtype=torch.uint8, device=dev) for _ in range(ws)] \
        if rank == dst else None
    dist.gather(mine, parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for r, t in enumerate(parts):
        host = buf if r == rank else t[:sizes[r]].cpu().numpy()
        out.append((host, t[:sizes[r]] if device is not None else None))
    return out


def allreduce_sum_(tensor, group=None):
    import torch.distributed as dist
    rank, ws = world()
    if ws > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def allreduce_(tensor, op, group=None):
    """op: 'sum' | 'max' | 'min' (the scalars of the BA outer iteration)"""
    import torch.distributed as dist
    rank, ws = world()
    if ws > 1:
        dist.all_reduce(tensor, op={'sum': dist.ReduceOp.SUM, 'max': dist.ReduceOp.MAX,
                                    'min': dist.ReduceOp.MIN}[op], group=group)
    return tensor


def allgather_padded(tensor, count, group=None):
    """every rank's first `count` (its own) entries of a 1-D tensor -> (list of per-rank tensors).
    Slices of different length travel padded to the longest one."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return [tensor[:count]]
    counts = allgather_objects(int(count))
    longest = max(counts)
    buf = torch.zeros(longest, dtype=tensor.dtype, device=tensor.device)
    buf[:count].copy_(tensor[:count])
    parts = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(parts, buf, group=group)
    return [p_[:c] for p_, c in zip(parts, counts)]


def point_owner(point_indices, n_points, world_size):
    """rank that owns each point (contiguous blocks of point ids, balanced by observation
    count): point p belongs to rank floor(world * (observations of points < p) / total)."""
    point_indices = np.asarray(point_indices)
    per_point = np.bincount(point_indices, minlength=n_points)
    csum = np.cumsum(per_point)
    total = int(csum[-1]) if len(csum) else 0
    before = csum - per_point
    return np.minimum((before * world_size) // max(total, 1), world_size - 1)


def point_range(point_indices, n_points, rank, world_size):
    """[lo, hi): the contiguous block of point ids owned by `rank` (the blocks of all ranks
    partition range(n_points))."""
    owner = point_owner(point_indices, n_points, world_size)
    lo = int(np.searchsorted(owner, rank, side='left'))
    hi = int(np.searchsorted(owner, rank, side='right'))
    return lo, hi


def shard_observations_by_point(point_indices, n_points, rank, world_size):
    """BA: all observations of a point live on one rank (point blocks and point elimination
    stay rank-local; only camera-side sums are reduced).  Points are dealt in contiguous
    blocks balanced by observation count.  Returns the indices (into the camera-major
    observation list) owned by `rank`, in ascending order (camera-major order is preserved)."""
    point_indices = np.asarray(point_indices)
    owner = point_owner(point_indices, n_points, world_size)
    return np.nonzero(owner[point_indices] == rank)[0]
