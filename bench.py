#!/usr/bin/env python3
"""bench.py -- image-pairs matched / s on MI355X (BASELINE.json metric), with roofline and a
CPU baseline on the same line.

Workload, every N: the survey BASELINE.json's metric is quoted on (configs[2]) -- 2812 synthetic
images x 4096 keypoints x 128-D SIFT-like descriptors, brute-force L2 2-NN in BOTH directions for
all 3 952 266 image pairs, plus the reference's quality-metric filter and survivor compaction on
the device (metric M1).  A "step" = one pass over all pairs of the rank's share
(strong scaling: the total work is fixed; it fits one GPU, 7 s per step).  N>1: each rank owns
1/N of the images' descriptors ("detected there"), packs them, they are all-gathered over RCCL
inside the step (the path's one exchange step), every rank matches its share of the pair schedule,
and the survivors of every launch travel to rank 0 as fixed-layout tensors (the product's
per-round gather, dist.gather_arrays) -- no other data-path collective.  configs[1] (500 images,
one MI355X) follows as an extra key at N = 1.  The BA section runs configs[3] point-sharded.

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

KPTS = 4096
DIM = 128
FLOP_PER_PAIR = 2.0 * KPTS * KPTS * DIM          # one distance matrix serves both directions
I8_DENSE_PEAK_TFLOPS = 5000.0                    # the task's dense i8 / fp8 nameplate (2 x the 2.5 PF bf16 dense figure); the guide has no i8 spec line,
                                                 # only the micro-benchmark below -- `frac` is against this, `frac_of_guide_ubench_3944` against that
I8_UBENCH_TOPS = 3944.0                          # the guide's measured i8 MFMA micro-benchmark rate
KNN2SYM_TRAFFIC_FILE = 'r6_knn2sym_traffic.json' # tools/update_traffic_json.py (PMC passes)
CONFIG1_IMAGES = 500                             # configs[1]: C(500, 2) = 124 750 pairs
CONFIG2_IMAGES = 2812                            # configs[2]: 3 952 266 pairs
E2E_FRAMES = 128                                 # configs[4] slice of the default run (rendered 20 MP frames)
MATCH_RATIO = 0.75
MAX_DISTANCE = 270.0


def synth_descriptors(n_img, first, count, device, seed=1234):
    """SIFT-like u8 descriptors for images [first, first+count): gamma(0.6)
    -> L2 normalise -> clip 0.2 -> renormalise -> x512 -> saturate; image j copies 30 % of
    image j-1's rows with integer noise U[-6,6].  Data generation only (torch), not the path."""
    out = torch.empty((count, KPTS, DIM), dtype=torch.uint8, device=device)
    alpha = torch.full((KPTS, DIM), 0.6, device=device)

    def base(j):
        g = torch.Generator(device=device)
        g.manual_seed(seed + j)
        x = torch._standard_gamma(alpha, generator=g)
        x = x / x.norm(dim=1, keepdim=True)
        x = x.clamp(max=0.2)
        x = x / x.norm(dim=1, keepdim=True)
        return (x * 512.0).round().clamp(0, 255), g

    prev = None                                # base (pre-copy) descriptors of image j-1
    for j in range(max(first - 1, 0), first + count):
        cur, g = base(j)
        base_j = cur.clone()
        if prev is not None:
            k = int(0.3 * KPTS)
            src = torch.randperm(KPTS, generator=g, device=device)[:k]
            dst = torch.randperm(KPTS, generator=g, device=device)[:k]
            noise = torch.randint(-6, 7, (k, DIM), generator=g, device=device)
            cur[dst] = (prev[src] + noise).clamp(0, 255)
        prev = base_j
        if j >= first:
            out[j - first] = cur.to(torch.uint8)
    return out


def pair_schedule(n_img, rank, world, sub_batch):
    """All unordered pairs, dealt to ranks in contiguous blocks of the train-major order.  Returns
    the rank's launches: ordered-pair arrays [fwd ..., rev ...] of `sub_batch` / 2 image pairs
    each (both directions of a pair in the same launch), and the rank's unordered pair count."""
    ii, jj = np.triu_indices(n_img, k=1)
    order = np.lexsort((ii, jj))                              # train-major: L2 reuse of the train
    ii, jj = ii[order], jj[order]
    n = len(ii)
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    und = np.stack([ii[lo:hi], jj[lo:hi]], 1).astype(np.int32)
    half = max(sub_batch // 2, 1)
    launches = [np.concatenate([und[s:s + half], und[s:s + half, ::-1]]) for s in range(0, len(und), half)]
    return launches, hi - lo


def match_section(args, rank, world, dev, dist, one_gpu, n_img, steps, warmup, verify_pairs):
    """The matching half of the metric on an `n_img`-image survey: `warmup` untimed steps, then
    exactly `steps` timed ones between barriers (max over ranks).  Returns the figures of the
    JSON line (seconds, pairs, survivor counts, the sweep's roofline, the oracle self-check)."""
    from imageanalysis_amd import kernels

    # ---- survey: configs[1] on one GPU, configs[2] (strong scaling) on several.  Image slots
    #      are dealt to the ranks in equal blocks (the last block may hold spare slots that no
    #      pair refers to), so the all-gather is one equal-sized collective per buffer
    per = (n_img + world - 1) // world
    n_slots = per * world
    first, mine = rank * per, per

    # ---- this rank's images (as if it had detected them there), packed locally
    raw = synth_descriptors(n_slots, first, mine, dev)
    store = kernels.DescriptorStore([KPTS] * n_slots)
    rows_per = int(store.offsets[1] - store.offsets[0])
    rows_per3 = int(store.offsets3[1] - store.offsets3[0])
    src_off = (torch.arange(mine + 1, dtype=torch.int64, device=dev) * KPTS)
    pack_scratch = torch.empty(3 * mine * KPTS, dtype=torch.int32, device=dev)

    def pack_and_gather():
        """pack own images into the stores (original order for the exact re-scan, sorted order
        for the sweep), then RCCL all-gather in place: rank r's shard already sits at offset
        r*shard of the receive buffer."""
        L, _ptr, sp = kernels.lib(), kernels._ptr, kernels.stream_ptr()
        o = int(store.offsets[first])
        kernels.check(L.iamx_desc_pack_u8(_ptr(raw), mine * KPTS, _ptr(store.desc[o:]),
                                          _ptr(store.norm_q[o:]), _ptr(store.norm_t[o:]), sp),
                      'iamx_desc_pack_u8')
        kernels.check(L.iamx_desc3_pack_batch_u8(_ptr(raw), _ptr(src_off), _ptr(store.img_off3[first:]),
                                                 mine, mine * KPTS, KPTS, _ptr(store.desc3),
                                                 _ptr(store.sn2), _ptr(store.sct), _ptr(store.sperm),
                                                 _ptr(store.sinv), _ptr(pack_scratch), sp),
                      'iamx_desc3_pack_batch_u8')
        bufs = [(store.desc, rows_per * DIM), (store.norm_q, rows_per),
                (store.norm_t, rows_per),          # (the exact stage reads the TRAIN image's norms)
                (store.desc3, rows_per3 * DIM), (store.sn2, rows_per3),
                (store.sct, rows_per3), (store.sperm, rows_per3)]
        if args.one_direction:       # parity-partitioned train layout of the one-direction form
            kernels.check(L.iamx_desc2_pack_batch_u8(_ptr(raw), _ptr(src_off), _ptr(store.img_off2[first:]),
                                                     mine, mine * KPTS, KPTS, _ptr(store.desc2),
                                                     _ptr(store.norm2), _ptr(store.cinit),
                                                     _ptr(store.perm), _ptr(store.meta[first]),
                                                     _ptr(pack_scratch), sp), 'iamx_desc2_pack_batch_u8')
            rows_per2 = int(store.offsets2[1] - store.offsets2[0])
            bufs = bufs[:3] + [(store.desc2, rows_per2 * DIM), (store.norm2, rows_per2),
                               (store.cinit, rows_per2), (store.perm, rows_per2), (store.meta, 4)]
        if dist is not None:
            for buf, width in bufs:
                flat = buf.view(-1)
                shard = per * width
                if one_gpu:          # gloo: no in-place all_gather_into_tensor on device memory
                    parts = [torch.empty(shard, dtype=flat.dtype, device=dev) for _ in range(world)]
                    dist.all_gather(parts, flat[rank * shard:(rank + 1) * shard].clone())
                    flat.copy_(torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(flat, flat[rank * shard:(rank + 1) * shard])

    launches, n_pairs_rank = pair_schedule(n_img, rank, world, args.sub_batch)
    batches = [kernels.PairBatch(store, o, sym=not args.one_direction) for o in launches]
    ws = kernels.PairWorkspace(max(b.rows for b in batches), max(b.n_pairs for b in batches))
    thresh = MAX_DISTANCE * MATCH_RATIO
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in batches]
    survivors = torch.zeros(1, dtype=torch.int64, device=dev)
    # The threshold / compaction / finish kernels of a launch (~0.65 ms, latency bound) run on a
    # second stream beside the NEXT launch's sweep (two workspaces, events both ways).
    overlap = not args.no_overlap and len(batches) > 1
    if overlap:
        runner = kernels.OverlappedSweeps(ws.max_rows, ws.max_pairs, first_workspace=ws)
        ws_pair = runner.ws

    candidates = torch.zeros(1, dtype=torch.int64, device=dev)

    # ---- N > 1: what the product does with a round's results (matcher._MatchRun.exchange:
    #      every rank's share of a round -> rank 0 as ONE fixed-layout tensor gather).  Here a
    #      "round" is GATHER_GROUP launches; a rank's share is, per launch, the survivor count of
    #      every ordered pair and the first GATHER_CAP compacted (query row, train row) survivors
    #      (the synthetic survey leaves ~4 k per launch; anything beyond the cap is counted and
    #      reported, never silently dropped).  Inside the timed step, on the side stream.
    GATHER_GROUP, GATHER_CAP = 4, 65536
    gather = None
    if dist is not None:
        nl = torch.tensor([len(batches)], dtype=torch.int64, device=dev)
        dist.all_reduce(nl, op=dist.ReduceOp.MAX)
        n_groups = (int(nl.item()) + GATHER_GROUP - 1) // GATHER_GROUP
        max_pairs = max(b.n_pairs for b in batches)
        words = GATHER_GROUP * (max_pairs + 2 * GATHER_CAP)
        gather = {"groups": n_groups, "words": words, "sent": 0,
                  "stage": [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(2)],
                  "recv": [[torch.empty(words, dtype=torch.int32, device=dev) for _ in range(world)]
                           for _ in range(2)] if rank == 0 else None,
                  "truncated": torch.zeros(1, dtype=torch.int64, device=dev),
                  "received": torch.zeros(1, dtype=torch.int64, device=dev)}

    def send_group(g):
        st = gather["stage"][g & 1]
        if one_gpu:                  # gloo debug mode: host tensors
            parts = [torch.empty(gather["words"], dtype=torch.int32) for _ in range(world)] if rank == 0 else None
            dist.gather(st.cpu(), parts, dst=0)
            if rank == 0:
                gather["received"].add_(sum(int(p_.view(GATHER_GROUP, -1)[:, :max_pairs].sum()) for p_ in parts))
        else:
            dist.gather(st, gather["recv"][g & 1] if rank == 0 else None, dst=0)
            if rank == 0:            # (rank 0 reads what arrived: the counts of every rank's share)
                for p_ in gather["recv"][g & 1]:
                    gather["received"].add_(p_.view(GATHER_GROUP, -1)[:, :max_pairs].sum())
        gather["sent"] = g + 1

    def count_survivors(b, w, k=None):
        survivors.add_(w.surv_cnt[:b.n_pairs].sum())
        candidates.add_(w.seg_count[:b.n_pairs].sum())     # rows that passed the bound test
        if gather is not None and k is not None:
            g, slot = divmod(k, GATHER_GROUP)
            st = gather["stage"][g & 1]
            if slot == 0:
                st.zero_()
            base = slot * (max_pairs + 2 * GATHER_CAP)
            st[base:base + b.n_pairs].copy_(w.surv_cnt[:b.n_pairs])
            st[base + max_pairs:base + max_pairs + GATHER_CAP].copy_(w.surv_q[:GATHER_CAP])
            st[base + max_pairs + GATHER_CAP:base + max_pairs + 2 * GATHER_CAP].copy_(w.surv_t[:GATHER_CAP])
            gather["truncated"].add_((w.surv_cnt[:b.n_pairs].sum() - GATHER_CAP).clamp_(min=0))
            if slot == GATHER_GROUP - 1 or k == len(batches) - 1:
                send_group(g)

    def step(timed_events=False):
        pack_and_gather()
        if gather is not None:
            gather["sent"] = 0
        counter = iter(range(len(batches)))
        if overlap:
            runner.run(batches, thresh, after_filter=lambda b, w: count_survivors(b, w, next(counter)),
                       sweep_events=ev if timed_events else None)
        else:
            for k, (b, (e0, e1)) in enumerate(zip(batches, ev)):
                if timed_events:
                    e0.record()
                b.run_knn2_fast(ws)
                if timed_events:
                    e1.record()
                b.run_filter_fast(ws, thresh)
                count_survivors(b, ws, k)
        if gather is not None:
            # (a rank with fewer launches than the longest share still takes part in every gather)
            for g in range(gather["sent"], gather["groups"]):
                gather["stage"][g & 1].zero_()
                send_group(g)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    survivors.zero_()
    candidates.zero_()
    if gather is not None:
        gather["truncated"].zero_()
        gather["received"].zero_()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(timed_events=True)
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    # (each rank's own clock from the common start barrier to the moment ITS work of the K steps
    #  was done would need a second sync point; what is reported per rank is the barrier-to-barrier
    #  time it measured -- the spread shows clock skew only -- and, below, its busy time)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [round(float(x.item()), 4) for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([n_pairs_rank], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        total_pairs = int(cnt.item())
        dist.all_reduce(survivors)                 # whole-job counts, like `value`
        dist.all_reduce(candidates)
    else:
        total_pairs = n_pairs_rank

    # ---- roofline of the dominant kernel, HIP events of the last step (recorded on the stream
    #      the sweeps are launched on)
    k_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
    per_rank_busy = None
    if dist is not None:
        # every rank's sweep-kernel seconds of the last step (its share of the schedule)
        tb = torch.tensor([sum(k_ms) * 1e-3], dtype=torch.float64, device=dev)
        allb = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allb, tb)
        per_rank_busy = [round(float(x.item()), 4) for x in allb]
    k_pairs = [b.n_pairs / 2.0 for b in batches]           # unordered pairs per launch
    achieved = sum(k_pairs) * FLOP_PER_PAIR / (sum(k_ms) * 1e-3) / 1e12
    traffic, traffic_src, mfma_busy, mfma_busy_src = None, None, None, None
    sweeps = 2 if args.one_direction else 1                # MFMA passes per distance matrix
    tf = os.path.join(REPO, 'profiles', 'r1_knn2v2_traffic.json' if args.one_direction
                      else KNN2SYM_TRAFFIC_FILE)
    if world == 1 and os.path.exists(tf):
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE); PMC cannot be collected from inside
        with open(tf) as fp:
            t = json.load(fp)
        # ... of THIS kernel: a summary measured on another instantiation of the sweep is refused
        kid = kernels.lib().iamx_knn2sym_kernel_id(2).decode() if not args.one_direction else None
        if kid is None or t.get("kernel", "").startswith(kid):
            traffic, traffic_src = t["hbm_bytes_per_launch"], t["source"]
            mfma_busy, mfma_busy_src = t.get("mfma_busy"), t.get("mfma_busy_source")
        else:
            traffic_src = mfma_busy_src = ("refused: %s was measured on %s, the library launches %s"
                                           % (os.path.basename(tf), t.get("kernel", "?").split(' (')[0], kid))
    roofline = {"bound": "mfma", "kernel": "knn2v2_kernel" if args.one_direction else "knn2sym_kernel",
                "achieved": round(achieved, 2), "peak": I8_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / I8_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                # MFMA pipe busy fraction from the committed PMC pass (the north star's
                # "MFMA utilisation" figure)
                "mfma_busy": mfma_busy, "mfma_busy_source": mfma_busy_src,
                "launches": len(batches), "avg_launch_ms": round(sum(k_ms) / len(k_ms), 4),
                "flop_per_launch": sum(k_pairs) / len(k_pairs) * FLOP_PER_PAIR,
                # algorithmic = one distance matrix per unordered pair.  The
                # symmetric sweep executes exactly that on the MFMA pipe (+ the exact re-scan of
                # the candidate rows on the VALU, off this kernel); the one-direction form runs
                # every matrix twice.  Sustained i8 MFMA ceiling with real operand bits ~3200
                # TOP/s (power limited, profiles/r1_ubench_mfma_clock.txt)
                "mfma_passes_per_matrix": sweeps,
                "executed_tflops": round(sweeps * achieved, 1),
                "executed_frac_of_peak": round(sweeps * achieved / I8_DENSE_PEAK_TFLOPS, 4),
                # the guide's measured i8 micro-benchmark ceiling (MI355X_MICROARCH.md: >= 3944 TOPS)
                "frac_of_guide_ubench_3944": round(achieved / I8_UBENCH_TOPS, 4)}

    # ---- self-check outside the timed region: a sample of ordered pairs of the store that was
    #      just timed (neighbours, which overlap, and far pairs), through the same path, against
    #      the oracle (oracle/cpu_ref.c) -- survivor rows, train rows, metrics
    verified = None
    if rank == 0 and verify_pairs > 0:
        verified = verify_sample(kernels, store, raw, first, mine, n_img, thresh,
                                 verify_pairs, not args.one_direction)

    ws_unresolved = sum(int(w.unresolved.item()) for w in (ws_pair if overlap else [ws]))
    cpu_sample = raw[:2].cpu().numpy() if (rank == 0 and mine >= 2) else None
    return {"dt": dt, "total_pairs": total_pairs, "survivors": int(survivors.item()),
            "candidates": int(candidates.item()), "unresolved": int(ws_unresolved),
            "verified": verified, "roofline": roofline, "cpu_sample": cpu_sample,
            "launches": len(batches), "per_rank_seconds": per_rank,
            "per_rank_sweep_seconds": per_rank_busy,
            "gather": None if gather is None else {
                "what": "per %d launches every rank's survivor counts + first %d survivor rows per "
                        "launch -> rank 0, one fixed-layout tensor gather (RCCL), inside the timed step"
                        % (GATHER_GROUP, GATHER_CAP),
                "gathers_per_step": gather["groups"], "bytes_per_rank_and_gather": gather["words"] * 4,
                "survivors_beyond_cap": int(gather["truncated"].item()),
                "survivor_count_seen_by_rank0": int(gather["received"].item())}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--images', type=int, default=0, help='override the survey size')
    ap.add_argument('--sub-batch', type=int, default=8192, help='ordered pairs per launch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--one-direction', action='store_true',
                    help='A/B: the round-1 form, one MFMA sweep per ORDERED pair (iamx_knn2v2_pairs)')
    ap.add_argument('--verify-pairs', type=int, default=64,
                    help='ordered pairs of the timed store checked against the oracle afterwards')
    ap.add_argument('--no-overlap', action='store_true',
                    help='filter kernels on the sweep stream (default: on a second stream)')
    ap.add_argument('--no-survey', action='store_true',
                    help='skip the one-step 2812-image survey sub-record (N = 1 only)')
    ap.add_argument('--no-ba', action='store_true', help='skip the bundle-adjustment section')
    ap.add_argument('--no-sift', action='store_true', help='skip the feature-detection section')
    ap.add_argument('--sift-first', action='store_true',
                    help='run the feature-detection section before the matching section (diagnosis of the '
                         'single-stream detection time: tools/sift_stream_bisect.py)')
    ap.add_argument('--sift-last', action='store_true',
                    help='run the feature-detection section behind the BA section, where rounds 1-4 had it')
    ap.add_argument('--ba-iters', type=int, default=0,
                    help='TRF iterations to time (0: until ftol = 1e-4 stops the solve, the reference\'s call)')
    ap.add_argument('--no-sift-full', action='store_true',
                    help='SIFT section without the scale 1.0 (20 MP) detects (counter passes: every detect the same size)')
    ap.add_argument('--no-e2e', action='store_true',
                    help='skip the configs[4] slice (24 rendered 20 MP frames through the drop-in chain)')
    ap.add_argument('--e2e-full', type=int, default=0, metavar='N',
                    help='BASELINE configs[4] at scale: N rendered 5472 x 3648 frames (N >= 512 asked '
                         'for) through the whole chain on one GPU, neighbour + distance-window '
                         'schedule; rendering and writing N JPEGs takes ~0.1 s per frame, untimed')
    ap.add_argument('--e2e-fused', action='store_true',
                    help='with --e2e-full: no separate detection stage; find_matches detects on demand '
                         'as scripts/process.py calls it (detection overlaps the first rounds of matching)')
    ap.add_argument('--e2e-window', type=int, default=0, metavar='F',
                    help='with --e2e-full: render, detect and delete the JPEGs F frames at a time '
                         '(a survey whose JPEGs do not fit the scratch disk: 10 000 x 20 MP = 80 GB)')
    ap.add_argument('--e2e', type=int, default=0, metavar='N',
                    help='also run the whole chain (detect -> match -> link -> triangulate -> BA, '
                         'BASELINE configs[4] shape) on N rendered images through the drop-in entry '
                         'points and report per-stage seconds')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher, one rank per GPU
        # (the same command line the driver uses for N > 1)
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                  '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
                                  '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d under WORLD_SIZE=%d: the launcher's rank count and "
                         "--gpus disagree" % (args.gpus, world))
    # IAMX_BENCH_DEBUG_ONE_GPU=1: all ranks on device 0 over gloo -- a smoke test of the N > 1
    # code path on a 1-GPU box (RCCL refuses two ranks on one device); never used for numbers
    one_gpu = os.environ.get('IAMX_BENCH_DEBUG_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    n_img = args.images or CONFIG2_IMAGES
    sift_early = None
    if args.sift_first and not args.no_sift:
        sift_early = sift_bench(rank, world, dev, dist, args)
    m = match_section(args, rank, world, dev, dist, one_gpu, n_img, args.steps, args.warmup,
                      args.verify_pairs)
    # The feature-detection section runs HERE, directly behind the headline section.  Behind the
    # survey / dense-overlap / BA sections of the same process its single-stream loop reads 2.8 ms
    # per detection where the same kernels, buffers and loop read 1.66 ms here and 1.70 ms in a
    # fresh process -- while eight detections in flight read 1.39 ms in both places
    # (profiles/r5_bench_sift_{inplace,first}.json, r5_sift_stream_bisect.txt: not the stream
    # count, page-locked memory, allocator state, loaded code objects or ba_bench on its own).
    if sift_early is None and not args.no_sift and not args.sift_last:
        try:
            from threadpoolctl import threadpool_limits as _tl
            _q = _tl(limits=1, user_api='blas')
        except ImportError:                               # pragma: no cover
            import contextlib as _cl
            _q = _cl.nullcontext()
        # (the self-check at the end of the matching section ran the C oracle on every OpenMP
        #  thread; the workers spin for a while after a parallel region and, under the 16-core
        #  quota, throttle the host half of the detections that follow: 72 instead of 350 images/s)
        time.sleep(1.0)
        with _q:
            sift_early = sift_bench(rank, world, dev, dist, args)
    dt, total_pairs, roofline, verified = m["dt"], m["total_pairs"], m["roofline"], m["verified"]
    cpu_sample = m["cpu_sample"]
    # ---- configs[1] (the one-MI355X configuration of BASELINE.json: 500 images, 124 750 pairs) as
    #      an extra key at N = 1: same kernels, same timed-step contents, 0.23 s per step
    survey = None
    if world == 1 and n_img != CONFIG1_IMAGES and not args.no_survey and not args.one_direction:
        torch.cuda.empty_cache()
        s = match_section(args, rank, world, dev, dist, one_gpu, CONFIG1_IMAGES, 5, 1,
                          min(args.verify_pairs, 16))
        survey = {"workload": "configs[1]: %d synthetic images x %d kpts x 128-D, all %d pairs, "
                              "5 timed steps after 1 warm-up step" % (CONFIG1_IMAGES, KPTS, s["total_pairs"]),
                  "value": round(s["total_pairs"] * 5 / s["dt"], 1), "unit": "pairs/s",
                  "seconds_per_step": round(s["dt"] / 5, 4), "pairs_per_step": s["total_pairs"],
                  "launches": s["launches"], "survivors_per_step": s["survivors"] // 5,
                  "candidates_per_step": s["candidates"] // 5, "unresolved": s["unresolved"],
                  "verify": s["verified"],
                  "roofline": {k: s["roofline"][k] for k in ("bound", "kernel", "achieved", "peak",
                                                             "unit", "frac", "avg_launch_ms")}}
        del s
        torch.cuda.empty_cache()
    dense = None
    if rank == 0 and world == 1 and not args.no_survey and not args.one_direction:
        dense = dense_overlap_bench(dev)
        torch.cuda.empty_cache()
    # ---- second half of the metric: sparse bundle adjustment (BASELINE configs[3])
    ba = None
    # The GPU sections below run with the BLAS thread pool limited to one thread: after a
    # multi-threaded numpy call ~100 OpenBLAS workers keep spinning for a while and the device
    # queue of whatever is timed next stalls for tens of milliseconds (profiles/r1_ba_notes.txt;
    # seen here as 13 instead of 4 ms per SIFT frame right after the BA section's numpy setup).
    try:
        from threadpoolctl import threadpool_limits
        quiet_blas = threadpool_limits(limits=1, user_api='blas')
    except ImportError:                                   # pragma: no cover
        import contextlib
        quiet_blas = contextlib.nullcontext()
    sift = cleanup = None
    with quiet_blas:
        if not args.no_ba:
            torch.cuda.empty_cache()
            ba = ba_bench(rank, world, dev, dist, args)
        if not args.no_sift:
            sift = sift_early if sift_early is not None else sift_bench(rank, world, dev, dist, args)
        cleanup = cleanup_bench(args) if rank == 0 else None
    # BASELINE configs[4] as a slice at its own frame size: 24 rendered 5472 x 3648 JPEGs through
    # detect -> match -> link -> triangulate -> BA (the drop-in entry points, host side included)
    e2e = e2e_small = e2e_big = None
    if rank == 0 and world == 1:
        if not args.no_e2e:
            # 128 rendered 20 MP frames (13 s of untimed rendering), the distance-window schedule
            # of a real survey; `--e2e-full N` runs the same at N >= 512
            e2e = e2e_bench(E2E_FRAMES, full_frame=True, schedule='distance')
        if args.e2e_full > 0:
            e2e_big = e2e_bench(args.e2e_full, full_frame=True, schedule='distance',
                                window=args.e2e_window, fused=args.e2e_fused)
        if args.e2e > 0:
            e2e_small = e2e_bench(args.e2e)
    # CPU baselines of the BA and SIFT sections run AFTER every timed GPU section: their OpenMP /
    # OpenBLAS worker threads keep spinning for a while after a parallel region and a GPU
    # section timed right behind them loses 4x (measured: 3.9 -> 20.9 ms per SIFT frame)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if ba is not None:
            ba["cpu_baseline"] = ba_cpu_baseline()
        if sift is not None and _SIFT_SAMPLE is not None:
            sift["cpu_baseline"] = sift_cpu_baseline()

    out = None
    if rank == 0:
        host_post = host_postprocess_rate()              # (times a device kernel: before the
        cpu = None                                       #  OpenMP baseline, see above)
        if not args.no_cpu_baseline and world == 1:      # CPU baselines: rank 0 at N=1 only
            cpu = cpu_baseline(cpu_sample)
        value = total_pairs * args.steps / dt
        out = {
            "metric": "image_pairs_matched_per_sec", "value": round(value, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 (int8 MFMA, int32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s: %d synthetic images x %d kpts x 128-D, all-pairs "
                                   "brute-force L2 2-NN both directions + metric filter + "
                                   "compaction" % ("configs[1]" if n_img == CONFIG1_IMAGES else
                                                   "configs[2]" if n_img == CONFIG2_IMAGES else
                                                   "custom survey", n_img, KPTS),
                       "images": n_img, "kpts": KPTS, "pairs_per_step": total_pairs,
                       "parallelism": "pair-shard x%d%s" % (world, " + RCCL descriptor all-gather"
                                                            if world > 1 else "")},
            "survivors_per_step": m["survivors"] // max(args.steps, 1),
            "candidates_per_step": m["candidates"] // max(args.steps, 1),
            "unresolved": m["unresolved"],
            "verified_pairs": verified["verified_pairs"] if verified else 0, "verify": verified,
            "roofline": roofline, "cpu_baseline": cpu, "config1_500": survey,
            "per_rank_seconds": m.get("per_rank_seconds"),
            "per_rank_sweep_seconds": m.get("per_rank_sweep_seconds"), "gather": m.get("gather"),
            "dense_overlap": dense,
            "host_postprocess": host_post, "ba": ba,
            "sift": sift, "cleanup": cleanup,
        }
        out["e2e"] = e2e
        if e2e_big is not None:
            out["e2e_full"] = e2e_big
        if e2e_small is not None:
            out["e2e_quarter_frames"] = e2e_small
        # LAST key, short: the driver keeps the tail of this line -- every headline figure of the
        # sections above in one place (same numbers, nothing new)
        g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
        out["summary"] = {
            "match_pairs_per_sec": out["value"], "match_frac_i8_peak": g(roofline, "frac"),
            "mfma_busy": g(roofline, "mfma_busy"), "match_traffic_bytes": g(roofline, "traffic"),
            "config1_500_pairs_per_sec": g(survey, "value"), "config1_500_seconds": g(survey, "seconds_per_step"),
            "dense_overlap_pairs_per_sec": g(dense, "pairs_per_sec"),
            "dense_overlap_routed_pairs_per_sec": g(dense, "routed_pairs_per_sec"),
            "ba_seconds_to_ftol": g(ba, "seconds_to_ftol"), "ba_trf_it_per_sec": g(ba, "value"),
            "ba_residual_frac_out_of_cache": g(ba, "residual", "out_of_cache", "frac"),
            "ba_residual_frac_of_stream_same_mix": g(ba, "residual", "out_of_cache", "frac_of_stream_same_mix"),
            "ba_residual_cache_resident_frac_of_hbm_peak": g(ba, "residual", "cache_resident", "frac_of_hbm_peak"),
            "ba_jac_frac_out_of_cache": g(ba, "residual_jac", "out_of_cache", "frac"),
            "ba_jac_cache_resident_frac_of_hbm_peak": g(ba, "residual_jac", "cache_resident", "frac_of_hbm_peak"),
            "ba_schur_iteration_frac": g(ba, "schur_iteration", "frac"),
            "sift_frames_per_sec": g(sift, "value"), "sift_frac_hbm": g(sift, "roofline", "frac"),
            "sift_traffic_bytes": g(sift, "roofline", "traffic"),
            "sift_ms_per_detect_on_stream": g(sift, "ms_per_image_detector_kernels"),
            "sift_frac_hbm_8_in_flight": g(sift, "concurrent_8", "frac"),
            "e2e_stage_seconds": g(e2e, "stage_seconds"),
            "e2e_images": g(e2e, "images"), "e2e_images_per_sec": g(e2e, "images_per_sec_end_to_end"),
            "e2e_total_seconds": g(e2e, "total_seconds"), "e2e_peak_hbm_bytes": g(e2e, "peak_hbm_bytes"),
            "e2e_mre_px": g(e2e, "ba", "mean_abs_residual_px_after"),
            "cpu_baseline_pairs_per_sec": g(cpu, "value"), "cpu_cores": g(cpu, "cores"),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


def verify_sample(kernels, store, raw, first, mine, n_img, thresh, n_check, sym):
    """oracle check of `n_check` ordered pairs among this rank's own images (their raw
    descriptors are at hand): the shipped path on the timed store vs oracle/cpu_ref.c"""
    from oracle import cpu_ref
    hi = min(first + mine, n_img)
    cand = [(first, first + 1), (first + 1, first), (first + 1, first + 2), (first + 2, first + 1),
            (first, hi - 1), (hi - 1, first), (first + 3, hi - 2), (hi - 2, first + 3)]
    if hi < n_img or first > 0:
        # several ranks: also pairs whose other image was packed by ANOTHER rank and arrived
        # through the all-gather (its descriptors are regenerated here from the seed)
        other = hi if hi < n_img else first - 1
        cand = cand[:4] + [(first, other), (other, first)] + cand[4:]
    # ... then more neighbours (overlapping views: most survivors) and far pairs, spread over the store
    span = max(hi - first, 1)
    for k in range(4, 4 + 2 * n_check):
        a = first + (k * 7919) % span
        cand.append((a, a + 1) if k % 2 else (a, first + (a - first + span // 2 + k) % span))
    cand = [(a, b) for a, b in cand if 0 <= a < n_img and 0 <= b < n_img and a != b
            and (first <= a < hi or first <= b < hi)]
    und = []
    for a, b in cand:
        if (a, b) not in und and (b, a) not in und:
            und.append((a, b))
    und = und[:max(1, n_check // 2)]
    ordered = np.array(und + [(b, a) for a, b in und], np.int32)
    pb = kernels.PairBatch(store, ordered, sym=sym)
    w = kernels.PairWorkspace(pb.rows, pb.n_pairs)
    pb.run(w, thresh)
    torch.cuda.synchronize()
    f, c, sq, st, sm = w.survivors(pb.n_pairs)
    host = {}
    for p, (a, b) in enumerate(ordered):
        for i in (a, b):
            if i not in host:
                host[i] = (raw[i - first] if first <= i < hi else
                           synth_descriptors(n_img, int(i), 1, raw.device)[0]).cpu().numpy()
        ridx, rd2 = cpu_ref.knn2_l2_u8(host[a], host[b])
        d = np.sqrt(rd2.astype(np.float32)).astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            metric = d[:, 0] * (d[:, 0] / d[:, 1])
        keep = np.nonzero(metric < thresh)[0]
        lo, hi_ = f[p], f[p] + c[p]
        ok = (np.array_equal(sq[lo:hi_], keep) and np.array_equal(st[lo:hi_], ridx[keep, 0])
              and np.array_equal(sm[lo:hi_], metric[keep]))
        if not ok:
            raise RuntimeError("bench self-check: pair (%d, %d) differs from the oracle" % (a, b))
    return {"verified_pairs": int(len(ordered)), "survivors_checked": int(c.sum()),
            "against": "oracle/cpu_ref.c", "form": "symmetric sweep, form %d" % pb.sym_form
            if pb.sym else "one-direction sweep, %d-row workgroups" % pb.fast_rows}


def dense_overlap_bench(dev, oracle_pairs=4, n_img=12, rows=16384):
    """The regime `value` does not see: image pairs that really overlap.  bench.py's survey copies
    30 % of an image's rows from its predecessor only, so 0.1 % of the rows of an average pair are
    candidates of the sweep's bound test and the exact stage (symexact_*) is a rounding error; on
    rendered 20 MP frames a third to two thirds of a pair's rows are (profiles/r4_e2e128_kernel_
    stats.txt).  Here: 12 images x 16 384 rows, every image 55 % noisy copies of rows of image 0,
    all 66 pairs both ways in one launch; sweep and filter / exact stage timed with events on the
    launch stream; a few ordered pairs checked against oracle/cpu_ref.c afterwards."""
    from imageanalysis_amd import kernels
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    alpha = torch.full((rows, DIM), 0.6, device=dev)

    def fresh(n):
        x = torch._standard_gamma(alpha[:n], generator=g)
        x = x / x.norm(dim=1, keepdim=True)
        x = x.clamp(max=0.2)
        x = x / x.norm(dim=1, keepdim=True)
        return (x * 512.0).round().clamp(0, 255)
    base = fresh(rows)
    imgs = [base.to(torch.uint8)]
    for _ in range(n_img - 1):
        src = torch.randint(0, rows, (rows,), generator=g, device=dev)
        im = (base[src] + torch.randint(-3, 4, (rows, DIM), generator=g, device=dev)).clamp(0, 255)
        new = torch.rand(rows, generator=g, device=dev) < 0.45
        im[new] = fresh(rows)[new]
        imgs.append(im.to(torch.uint8))
    host = [im.cpu().numpy() for im in imgs]
    store = kernels.DescriptorStore.from_arrays(host)
    und = [(a, b) for a in range(n_img) for b in range(a + 1, n_img)]
    ordered = np.array(und + [(b, a) for a, b in und], np.int32)
    pb = kernels.PairBatch(store, ordered, sym=True)
    ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
    thresh = MAX_DISTANCE * MATCH_RATIO
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    best = None
    for _ in range(3):
        ev[0].record()
        pb.run_knn2_fast(ws)
        ev[1].record()
        pb.run_filter_fast(ws, thresh)
        ev[2].record()
        torch.cuda.synchronize()
        t = (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]))
        best = t if best is None or sum(t) < sum(best) else best
    cand = int(ws.seg_count[:pb.n_pairs].sum().item())
    first, count, sq, st, sm = ws.survivors(pb.n_pairs)
    from oracle import cpu_ref
    checked = 0
    for p in list(range(oracle_pairs // 2)) + [len(und) + k for k in range(oracle_pairs // 2)]:
        a, b = ordered[p]
        ridx, rd2 = cpu_ref.knn2_l2_u8(host[a], host[b])
        d = np.sqrt(rd2.astype(np.float32)).astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            metric = d[:, 0] * (d[:, 0] / d[:, 1])
        keep = np.nonzero(metric < thresh)[0]
        lo, hi = first[p], first[p] + count[p]
        if not (np.array_equal(sq[lo:hi], keep) and np.array_equal(st[lo:hi], ridx[keep, 0])
                and np.array_equal(sm[lo:hi], metric[keep])):
            raise RuntimeError("dense-overlap self-check: pair (%d, %d) differs from the oracle" % (a, b))
        checked += 1
    # the same pairs through the one-direction bound form (what find_matches routes a dense round
    # to, matcher.DENSE_ROUTE): two sweeps per pair, survivors finished exactly; same survivors
    pb1 = kernels.PairBatch(store, ordered, sym=False)
    ws1 = kernels.PairWorkspace(pb1.rows, pb1.n_pairs)
    best1 = None
    for _ in range(3):
        ev[0].record()
        pb1.run_knn2_fast(ws1)
        ev[1].record()
        pb1.run_filter_fast(ws1, thresh)
        ev[2].record()
        torch.cuda.synchronize()
        t = (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]))
        best1 = t if best1 is None or sum(t) < sum(best1) else best1
    f1, c1, sq1, st1, sm1 = ws1.survivors(pb1.n_pairs)
    for p_ in range(pb.n_pairs):
        lo, hi, lo1, hi1 = first[p_], first[p_] + count[p_], f1[p_], f1[p_] + c1[p_]
        if not (np.array_equal(sq[lo:hi], sq1[lo1:hi1]) and np.array_equal(st[lo:hi], st1[lo1:hi1])
                and np.array_equal(sm[lo:hi], sm1[lo1:hi1])):
            raise RuntimeError("dense-overlap self-check: the two forms differ on ordered pair %d" % p_)
    if int(ws1.unresolved.item()):
        raise RuntimeError("dense-overlap: unresolved rows in the one-direction form")
    one_dir = {"sweeps_ms": round(best1[0], 3), "filter_and_finish_ms": round(best1[1], 3),
               "pairs_per_sec": round(len(und) / (sum(best1) * 1e-3), 1),
               "pairs_per_sec_in_4096_row_units": round(len(und) * (rows / float(KPTS)) ** 2 / (sum(best1) * 1e-3), 1),
               "speedup_over_symmetric": round(sum(best) / sum(best1), 3),
               "survivors_equal_symmetric_form": True}
    del pb1, ws1
    flop_exact = 2.0 * cand * rows * DIM
    flop_sweep = 2.0 * len(und) * rows * rows * DIM
    return {"workload": "%d images x %d rows, 55 %% of every image's rows are noisy copies of rows of "
                        "image 0; all %d pairs, both directions, one launch" % (n_img, rows, len(und)),
            "candidate_rows": cand, "candidate_share": round(cand / float(pb.rows), 4),
            "survivors": int(count.sum()), "sweep_ms": round(best[0], 3),
            "filter_and_exact_ms": round(best[1], 3),
            "pairs_per_sec": round(len(und) / (sum(best) * 1e-3), 1),
            "pairs_per_sec_in_4096_row_units": round(len(und) * (rows / float(KPTS)) ** 2 / (sum(best) * 1e-3), 1),
            "sweep_tflops": round(flop_sweep / (best[0] * 1e-3) / 1e12, 1),
            "exact_stage_tflops": round(flop_exact / (best[1] * 1e-3) / 1e12, 1),
            "exact_stage": "symnarrow_kernel: a candidate is re-scanned only against the classes of train "
                           "rows the sweep's group minima / block bounds leave open (1/8 of the image or "
                           "one row block per task of 256 (candidate, class) items, tiles through LDS, "
                           "pruned on the candidate test's bound; exact_stage_tflops counts the FULL "
                           "scan's flops it replaces) + candidate test, item bucketing, merge, compaction "
                           "in the same interval; IAMX_EXACT_NARROW=0: symexact_wg_kernel's full scan",
            "one_direction_form": one_dir,
            "routed_pairs_per_sec": round(max(len(und) / (sum(best) * 1e-3), one_dir["pairs_per_sec"]), 1),
            "verified_pairs": checked, "against": "oracle/cpu_ref.c"}


def e2e_bench(n_images, full_frame=False, schedule=None, window=0, fused=False):
    """BASELINE configs[4] shape on one GPU: a rendered survey of n_images JPEGs on disk goes
    through the drop-in entry points exactly as scripts/process.py:236-407 drives the reference's
    modules -- Image.detect_features, matcher.find_matches, match_cleanup.*, groups.compute,
    Optimizer.setup / run / update_camera_poses -- with per-stage wall seconds (host side
    included: JPEG decode, cache files, python lists, .match pickles).  full_frame: the survey
    is rendered at configs[4]'s own frame size, 5472 x 3648 (20 MP), and detected at the
    reference's default scale 0.4 (scripts/lib/matcher.py:38); otherwise at a quarter of the pixels
    and scale 1.0.  schedule: 'all-pairs' (default with full_frame: the slice of a few dozen
    frames), 'distance' (neighbours in the list + every pair inside the reference's distance
    window, scripts/lib/matcher.py:886-903: the shape of a survey of hundreds / thousands of
    frames) or 'neighbours' (the reference at HEAD).
    fused: no separate detection stage -- find_matches() meets undetected images and detects them
    on demand, exactly as scripts/process.py:290 calls it (the reference has no detection stage of
    its own either, lib/matcher.py:961-968): the decode / SIFT of the images later rounds need runs
    on the prefetch workers' streams WHILE the first rounds of the distance-sorted schedule are
    matched; the stage is reported as "detect+match".
    window > 0: the survey does not fit the scratch disk as JPEGs (10 000 frames of 20 MP are
    80 GB): it is rendered `window` frames at a time, every window goes through detect_features
    (timed: the detect stage is the sum over the windows) and its JPEGs are deleted; the cache
    files, which are what the later stages read, stay."""
    import contextlib
    import io
    import shutil
    import tempfile
    from imageanalysis_amd import groups, image as iimg, match_cleanup, matcher, optimizer, synth
    from imageanalysis_amd._deps import getNode
    from imageanalysis_amd.hostlib import camera
    cols = max(2, int(round(math.sqrt(n_images * 2.0))))
    rows = max(2, (n_images + cols - 1) // cols)
    tmp = tempfile.mkdtemp(prefix='iamx_e2e_')
    out = {"images": rows * cols, "grid": [rows, cols]}
    try:
        scale = 0.4 if full_frame else 1.0
        an = os.path.join(tmp, 'ImageAnalysis')
        os.makedirs(os.path.join(an, 'cache'))
        os.makedirs(os.path.join(an, 'meta'))
        getNode('/config/directories', True).setString('project_dir', tmp)
        matcher.detector_node.setString('detector', 'SIFT')
        matcher.detector_node.setFloat('scale', scale)
        matcher.matcher_node.setFloat('match_ratio', 0.75)
        matcher.matcher_node.setInt('min_pairs', 25)
        matcher.matcher_node.setInt('min_chain_len', 0)
        if schedule is None and full_frame:
            schedule = 'all-pairs'
        if schedule is not None:
            matcher.matcher_node.setString('schedule', schedule)
        out["schedule"] = schedule or 'neighbours'
        torch.cuda.reset_peak_memory_stats()
        # (at >= 1024 frames the float32 des_list of the reference's contract is tens of GB of
        #  host memory: image.DES_LIST_U8 keeps the same values as uint8; cache files unchanged)
        des_u8_was = iimg.DES_LIST_U8
        iimg.DES_LIST_U8 = n_images >= 1024
        out["des_list_dtype"] = 'uint8' if iimg.DES_LIST_U8 else 'float32'
        sidecar_was = iimg.USE_DESC_SIDECAR
        if window:
            # (the raw uint8 sidecar of the .desc files is another 4.7 MB per frame: off where
            #  the scratch disk is the limit; the reference's two cache files are written as always)
            iimg.USE_DESC_SIDECAR = False
            out["window_frames"] = int(window)

        class Proj(object):
            analysis_dir = an

            def findIndexByName(self, name):
                return names.index(name) if name in names else None

            def findImageByName(self, name):
                return self.image_list[names.index(name)] if name in names else None

            def save_images_info(self):
                pass

        proj = Proj()
        proj.image_list = []
        quiet = contextlib.redirect_stdout(io.StringIO())
        stages = {}
        profile_stages = os.environ.get('IAMX_E2E_PROFILE', '').split(',')

        def timed(key, fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if key in profile_stages:              # (diagnosis: cProfile of the named stages to stderr)
                import cProfile
                import pstats
                pr = cProfile.Profile()
                with quiet:
                    r = pr.runcall(fn)
                pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(22)
            else:
                with quiet:
                    r = fn()
            torch.cuda.synchronize()
            stages[key] = round(stages.get(key, 0.0) + time.perf_counter() - t, 3)
            return r

        configured = []
        names = []

        def on_frames(names_, truth_, logged_, K_, first, stop):
            """the frames [first, stop) are on disk: their Image objects, and in window mode their
            detection right away (then the JPEGs go)"""
            if not configured:
                W_, H_ = int(2 * K_[0, 2]), int(2 * K_[1, 2])
                node = getNode('/config/camera', True)
                node.__dict__.pop('K_opt', None)
                node.__dict__.pop('dist_coeffs_opt', None)
                camera.set_K(K_[0, 0], K_[1, 1], K_[0, 2], K_[1, 2])
                camera.set_dist_coeffs([0.0] * 5)
                camera.set_image_params(W_, H_)
                camera.set_mount_params(0.0, -90.0, 0.0)
                matcher.configure()
                configured.append(True)
            fresh = []
            for k in range(first, stop):
                names.append(names_[k])
                ned, ypr = logged_[k]
                im = iimg.Image(an, names_[k])
                # (camera pose + the aircraft attitude that leads to it under the nadir mount:
                #  find_matches re-derives the camera pose whenever it updates an image's yaw error)
                im.set_pose_from_camera(ned.tolist(), *ypr.tolist())
                getNode('/smart', True).getChild(names_[k], True).setFloat('tri_surface_m', 0.0)
                proj.image_list.append(im)
                fresh.append(im)
            if window:
                timed("detect", lambda: detect(fresh))
                for im in fresh:
                    with contextlib.suppress(OSError):
                        os.remove(im.image_file)

        def detect(images):
            pf = iimg.prefetch(images, scale=scale)
            for im in images:
                im.detect_features(scale)
            pf.close()
            iimg.cacheio.wait()

        t0 = time.perf_counter()
        if full_frame:
            _n, truth, logged, K = synth.make_rendered_survey(tmp, rows, cols, device='cuda',
                                                              chunk=window or rows * cols,
                                                              on_frames=on_frames, **synth.FULL_FRAME)
        else:
            _n, truth, logged, K = synth.make_rendered_survey(tmp, rows, cols, chunk=window or rows * cols,
                                                              on_frames=on_frames)
        out["render_seconds_untimed"] = round(time.perf_counter() - t0 - stages.get("detect", 0.0), 2)
        W, H = int(2 * K[0, 2]), int(2 * K[1, 2])
        if not window and not fused:
            timed("detect", lambda: detect(proj.image_list))
        out["fused_detect_and_match"] = bool(fused and not window)
        def match():
            trace = None
            if os.environ.get('IAMX_MATCH_TRACE'):       # phases of the call on stderr (diagnosis)
                trace = matcher._round_trace = []
                orig_launch, t_call = matcher._launch_batch, time.perf_counter()

                def launch(*a, **k):
                    t = time.perf_counter()
                    r = orig_launch(*a, **k)
                    trace.append(('launch', time.perf_counter() - t, t - t_call))
                    return r
                matcher._launch_batch = launch
            matcher.find_matches(proj, K, strategy='traditional', transform='homography', sort=True)
            t_fm = time.perf_counter()
            iimg.cacheio.wait()
            if trace is not None:
                matcher._launch_batch, matcher._round_trace = orig_launch, None
                for ent in trace:
                    if ent[0] == 'pre':
                        print('trace pre   %-22s +%.3f s' % (ent[1], ent[2] - t_call), file=sys.stderr)
                    elif ent[0] == 'launch':
                        print('trace launch at +%.3f s took %.3f s' % (ent[2], ent[1]), file=sys.stderr)
                    else:
                        print('trace %s' % (ent,), file=sys.stderr)
                print('trace find_matches returned +%.3f s, cache writes done +%.3f s'
                      % (t_fm - t_call, time.perf_counter() - t_call), file=sys.stderr)
        timed("detect+match" if out["fused_detect_and_match"] else "match", match)
        for im in proj.image_list:                       # (the periodic flush may have dropped some)
            if im.kp_list is None:
                im.load_features()
        out["keypoints_per_image"] = int(np.mean([len(im.kp_list) for im in proj.image_list]))
        out["image_pairs_matched"] = sum(len(im.match_list) for im in proj.image_list) // 2
        out["route_rounds"] = {"mode": matcher.DENSE_ROUTE, "symmetric": matcher._route['rounds'][0],
                               "one_direction": matcher._route['rounds'][1],
                               "last_candidate_share": matcher._route['share']}
        out["hbm_after_match"] = dict(matcher.device_memory_report(),
                                      allocated_bytes=int(torch.cuda.memory_allocated()),
                                      peak_allocated_bytes=int(torch.cuda.max_memory_allocated()))
        out["hbm_model"] = matcher.device_memory_model(len(names), out["keypoints_per_image"],
                                                       train_layout=matcher._route['rounds'][1] > 0)
        out["image_pairs_with_matches"] = sum(len(v) > 0 for im in proj.image_list
                                              for v in im.match_list.values()) // 2

        def consolidate():
            match_cleanup.merge_duplicates(proj)
            match_cleanup.check_for_pair_dups(proj)
            match_cleanup.check_for_1vn_dups(proj)
            direct = match_cleanup.make_match_structure(proj)
            return match_cleanup.link_matches(proj, direct)
        grouped = timed("consolidate", consolidate)
        out["chains"] = len(grouped)

        def triangulate():
            match_cleanup.triangulate_smart(proj, grouped)
            return groups.compute(proj.image_list, grouped)
        group_list = timed("triangulate_group", triangulate)
        opt = optimizer.Optimizer(an)
        timed("ba_setup", lambda: opt.setup(proj, group_list, 0, grouped, optimized=False,
                                            cam_calib=False))
        x0 = opt._x0()
        args_ = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
        with quiet:
            mre0 = float(np.mean(np.abs(opt.fun(x0, *args_))))
        timed("ba_run", opt.run)
        mre1 = float(np.mean(np.abs(opt.result.fun)))
        timed("write_poses", lambda: opt.update_camera_poses(proj))
        # relative geometry against the truth the frames were rendered from (gauge: one scale)
        est = np.array([im.get_camera_pose(opt=True)[0] for im in proj.image_list])
        tru = np.array([t[0] for t in truth])
        db_est = np.linalg.norm(est[:, None, :] - est[None, :, :], axis=2)
        db_tru = np.linalg.norm(tru[:, None, :] - tru[None, :, :], axis=2)
        gauge = float((db_est * db_tru).sum() / (db_tru * db_tru).sum())
        out["groups"] = [len(g) for g in group_list]
        out["peak_hbm_bytes"] = int(torch.cuda.max_memory_allocated())
        import resource
        out["host_peak_rss_bytes"] = int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss) * 1024
        out["baseline_scale"] = round(gauge, 5)
        out["max_baseline_error_m"] = round(float(np.abs(db_est - gauge * db_tru).max()), 4)
        total = sum(stages.values())
        out.update({"stage_seconds": stages, "total_seconds": round(total, 3),
                    "images_per_sec_end_to_end": round(len(names) / total, 2),
                    "ba": {"cameras": int(opt.n_cameras), "points": int(opt.n_points),
                           "observations": int(opt.camera_indices.size),
                           "iterations": int(opt.result.njev),
                           "mean_abs_residual_px_before": round(mre0, 3),
                           "mean_abs_residual_px_after": round(mre1, 3)},
                    "image_size": [W, H], "detect_scale": scale,
                    "note": "drop-in entry points, host side included; rendered %dx%d frames (%s), "
                            "detector scale %.1f"
                            % (W, H, "configs[4]'s own FC6310S frame" if full_frame else
                               "the FC6310S field of view at a quarter of its pixels", scale)})
    finally:
        matcher.matcher_node.__dict__.pop('schedule', None)
        try:
            iimg.DES_LIST_U8 = des_u8_was
            iimg.USE_DESC_SIDECAR = sidecar_was
        except NameError:
            pass
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def host_postprocess_rate():
    """GMS / de-dup / cross-check are host post-processing, timed separately from
    `value`: one pair = both directions with 2000 thresholded matches each (the clip limit of
    scripts/lib/matcher.py:267), 70 % of them on a consistent motion, through
    gms_inlier_mask + _dedupe + filter_cross_check (imageanalysis_amd/matcher.py), one core."""
    from imageanalysis_amd import gms, matcher
    rng = np.random.default_rng(5)
    n, W, H = 4096, 5472.0, 3648.0
    xy1 = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n)], 1).astype(np.float32)
    xy2 = xy1.copy()
    xy2[:, 0] = np.clip(xy1[:, 0] + 300.0, 0, W - 1)
    q = rng.permutation(n)[:2000]
    t = q.copy()
    bad = rng.random(2000) < 0.3
    t[bad] = rng.integers(0, n, bad.sum())
    # survivors of both directions as the matching stage delivers them: query rows ascending
    metric = rng.uniform(1.0, 200.0, 2000)
    of, orv = np.argsort(q, kind='stable'), np.argsort(t, kind='stable')
    sf = (q[of].astype(np.int32), t[of].astype(np.int32), metric[of])
    sr = (t[orv].astype(np.int32), q[orv].astype(np.int32), metric[orv])

    def one():
        out = []
        for a, b, sv in ((xy1, xy2, sf), (xy2, xy1, sr)):
            pr = matcher._threshold_sort_clip(*sv)
            m = gms.gms_inlier_mask(a, b, (W, H), (W, H), pr, with_rotation=True, with_scale=False,
                                    threshold_factor=5.0)
            out.append(matcher._dedupe(a, b, [[int(u), int(v)] for u, v in pr[m]])[0])
        return matcher.filter_cross_check(out[0], out[1])

    one()
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        kept = one()
        reps += 1
    dt = time.perf_counter() - t0
    out = {"host": {"value": round(reps / dt, 1), "unit": "pairs/s", "cores": 1,
                    "sample": "2 x 2000 matches per pair, %d kept after GMS + de-dup + cross-check"
                              % len(kept[0])}}
    # the same pair through iamx_match_postfilter (one workgroup per pair), 2048 pairs per launch
    from imageanalysis_amd import kernels
    from imageanalysis_amd.kernels import _ptr
    dev = torch.device('cuda', torch.cuda.current_device())
    n_pairs = 2048
    L = kernels.lib()
    clip = int(L.iamx_match_postfilter_clip())
    t = lambda a, dt_: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt_)
    sq = np.concatenate([np.tile(sf[0], n_pairs), np.tile(sr[0], n_pairs)])
    st = np.concatenate([np.tile(sf[1], n_pairs), np.tile(sr[1], n_pairs)])
    sm = np.concatenate([np.tile(sf[2], n_pairs), np.tile(sr[2], n_pairs)])
    d = dict(off=t(np.arange(2 * n_pairs + 1) * 2000, torch.int64),
             cnt=t(np.full(2 * n_pairs, 2000), torch.int32), q=t(sq, torch.int32),
             t=t(st, torch.int32), m=t(sm, torch.float64),
             pairs=t(np.array([[0, 1]] * n_pairs + [[1, 0]] * n_pairs), torch.int32),
             kp_off=t(np.array([0, n]), torch.int64), xy=t(np.concatenate([xy1, xy2]), torch.float32),
             key2=t(matcher.kp_key2(np.concatenate([xy1, xy2])), torch.int32),
             out_cnt=torch.empty(n_pairs, dtype=torch.int32, device=dev),
             out_pairs=torch.empty((n_pairs, clip, 2), dtype=torch.int32, device=dev),
             scratch=torch.empty((n_pairs, 2, clip, 2), dtype=torch.int32, device=dev),
             stat=torch.empty((n_pairs, 4), dtype=torch.int32, device=dev),
             status=torch.empty(n_pairs, dtype=torch.int32, device=dev))

    def launch():
        kernels.check(L.iamx_match_postfilter(_ptr(d['off']), _ptr(d['cnt']), _ptr(d['q']), _ptr(d['t']),
                                              _ptr(d['m']), _ptr(d['pairs']), _ptr(d['kp_off']),
                                              _ptr(d['xy']), _ptr(d['key2']), n_pairs, W, H, 25.0, 5.0,
                                              _ptr(d['out_cnt']), _ptr(d['out_pairs']),
                                              _ptr(d['scratch']), _ptr(d['stat']), _ptr(d['status']),
                                              kernels.stream_ptr()), 'iamx_match_postfilter')

    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    got = d['out_pairs'][0, :int(d['out_cnt'][0].item())].cpu().numpy()
    if not np.array_equal(got, np.array(kept[0]).reshape(-1, 2)):
        raise RuntimeError("device post filter disagrees with the host filters")
    out["device"] = {"value": round(n_pairs / (ms * 1e-3), 1), "unit": "pairs/s",
                     "kernel": "postfilter_kernel", "ms_per_launch": round(ms, 3),
                     "pairs_per_launch": n_pairs,
                     "kept_per_pair": int(d['out_cnt'][0].item())}
    return out


def cleanup_bench(args):
    """SURVEY.md 8f ranks 1-2 (between matching and BA): chain linking of the pair-wise matches
    (native host code) and the initial ground-plane triangulation (device), on a synthetic strip
    of 600 images x 4 neighbours x 800 matches; CPU baseline = a literal python transcription
    of the reference's linking loop on a 1/4 sample."""
    from imageanalysis_amd import kernels, match_cleanup
    from imageanalysis_amd.kernels import _ptr
    rng = np.random.default_rng(7)
    n_img, n_kp, per = 600, 4096, 800

    class _KP(object):
        __slots__ = ('pt',)

    class _Img(object):
        pass

    class _Proj(object):
        pass

    proj = _Proj()
    proj.image_list = []
    for i in range(n_img):
        im = _Img()
        im.name = 'S%04d' % i
        xy = np.stack([rng.uniform(0, 5471, n_kp), rng.uniform(0, 3647, n_kp)], 1).astype(np.float32)
        im.kp_list = None
        im._iamx_xy = (None, xy)
        im.match_list = {}
        proj.image_list.append(im)
    for i in range(n_img):
        for j in range(i + 1, min(i + 5, n_img)):
            a, b = rng.choice(n_kp, per, replace=False), rng.choice(n_kp, per, replace=False)
            proj.image_list[i].match_list['S%04d' % j] = np.stack([a, b], 1).tolist()
            proj.image_list[j].match_list['S%04d' % i] = np.stack([b, a], 1).tolist()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        direct = match_cleanup.make_match_structure(proj)
        t1 = time.perf_counter()
        grouped = match_cleanup.link_matches(proj, direct)
        t_link = time.perf_counter() - t1
    out = {"link_matches": {"value": round(len(direct) / t_link, 1), "unit": "pair matches/s",
                            "pair_matches": len(direct), "chains": len(grouped),
                            "seconds": round(t_link, 3),
                            "make_match_structure_seconds": round(t1 - t0, 3)}}
    if not args.no_cpu_baseline:
        sample = direct[:len(direct) // 4]

        def python_rules(matches):               # scripts/lib/match_cleanup.py:246-286
            matches = [list(m) for m in matches]
            while True:
                new, lookup = [], {}
                for match in matches:
                    index = -1
                    for p in match[2:]:
                        key = "%d-%d" % (p[0], p[1])
                        if key in lookup:
                            index = lookup[key]
                            break
                    if index < 0:
                        for p in match[2:]:
                            lookup["%d-%d" % (p[0], p[1])] = len(new)
                        new.append(list(match))
                    else:
                        existing = new[index]
                        for p in match[2:]:
                            if not any(p[0] == e[0] for e in existing[2:]):
                                existing.append(list(p))
                                lookup["%d-%d" % (p[0], p[1])] = index
                if len(new) == len(matches):
                    return new
                matches = new

        t0 = time.perf_counter()
        python_rules(sample)
        dt = time.perf_counter() - t0
        out["link_matches"]["cpu_baseline"] = {
            "value": round(len(sample) / dt, 1), "unit": "pair matches/s", "cores": 1, "kind": "port",
            "sample": "python transcription of the reference loop on %d pair matches in %.1f s"
                      % (len(sample), dt)}
    # triangulation on the device: the features of the linked chains
    dev = torch.device('cuda', torch.cuda.current_device())
    n = len(grouped)
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum([len(m) - 2 for m in grouped], out=ptr[1:])
    obs_img = np.array([p[0] for m in grouped for p in m[2:]], np.int32)
    obs_uv = np.array([p[1] for m in grouped for p in m[2:]], np.float64).reshape(-1, 2)
    M = rng.normal(0, 1e-3, (n_img, 9))
    M[:, 8] = 1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = [t(a) for a in (M, rng.normal(0, 50, (n_img, 3)), np.zeros(n_img), obs_img, obs_uv, ptr)]
    res = torch.empty((n, 3), dtype=torch.float64, device=dev)
    sky = torch.zeros(1, dtype=torch.int32, device=dev)
    L = kernels.lib()

    def launch():
        kernels.check(L.iamx_triangulate_ground(_ptr(d[0]), _ptr(d[1]), _ptr(d[2]), n_img, _ptr(d[3]),
                                                _ptr(d[4]), _ptr(d[5]), n, _ptr(res), _ptr(sky),
                                                kernels.stream_ptr()), 'iamx_triangulate_ground')

    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = len(obs_img) * 20.0 + n * (8.0 + 24.0)           # idx + uv per observation, ptr + out per feature
    out["triangulate"] = {"value": round(n / (ms * 1e-3), 1), "unit": "features/s", "features": n,
                          "observations": int(len(obs_img)), "ms_per_launch": round(ms, 4),
                          "roofline": {"bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1),
                                       "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(by / (ms * 1e-3) / 1e9 / 8000.0, 4)}}
    return out


def sift_bench(rank, world, dev, dist, args):
    """Feature detection on BASELINE's 20 MP frames (5472x3648, detect scale 0.4): CLAHE + resize +
    SIFT on the device, per image, inputs resident in HBM, keypoints/descriptors delivered to
    the host in canonical order.  Images shard over ranks (independent), no collective."""
    from imageanalysis_amd import kernels, synth
    n_local = 4
    imgs = [synth.make_survey_image(seed=100 * rank + i, device=dev) for i in range(2)]
    scale = 0.4

    def detect(img):
        scaled = kernels.equalize_resize(img, scale)
        return scaled, kernels.sift_detect(scaled, cap=400000)

    scaled, (kp, _o, _d) = detect(imgs[0])                  # warm-up (workspace, code objects)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    nkp = 0
    for i in range(n_local):
        nkp += len(detect(imgs[i % 2])[1][0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # device-only time of the detector kernels (no download / sort)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L = kernels.lib()
    h, w = scaled.shape[0], scaled.shape[1]
    need = int(L.iamx_sift_workspace_bytes(h, w))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    kpd = torch.empty((400000, 8), dtype=torch.float32, device=dev)
    dd = torch.empty((400000, 128), dtype=torch.uint8, device=dev)
    nn = torch.zeros(1, dtype=torch.int32, device=dev)
    def enqueue(b=None):
        b = b or (ws, kpd, dd, nn)
        kernels.check(L.iamx_sift_detect(kernels._ptr(scaled), h, w, 3, 0.04, 10.0, 1.6,
                                         kernels._ptr(b[0]), need, kernels._ptr(b[1]), kernels._ptr(b[2]),
                                         400000, kernels._ptr(b[3]), torch.cuda.current_stream().cuda_stream),
                      'iamx_sift_detect')
    # steady state: 3 untimed detections on these buffers, then N_K back to back.  (Rounds 1-4
    # timed FOUR detections behind a synchronize: 2.8 ms each where the same kernels take 1.8 ms
    # once the queue is full and the clocks are up -- tools/sift_stream_time.py.)
    N_K = 20
    for _ in range(3):
        enqueue()
    torch.cuda.synchronize()
    t0h = time.perf_counter()
    e0.record()
    for _ in range(N_K):
        enqueue()
    e1.record()
    t_host = (time.perf_counter() - t0h) / N_K
    torch.cuda.synchronize()
    t_k = e0.elapsed_time(e1) / N_K * 1e-3
    # ... and what image.prefetch runs: 8 detector threads in flight, a buffer set and stream each
    # (whole-job rate, wall clock: the frames' small kernels fill each other's dependency stalls)
    t_conc = None
    if dist is None:
        import threading
        K_DET = 8
        sets = [(torch.empty(need, dtype=torch.uint8, device=dev), torch.empty((400000, 8), dtype=torch.float32, device=dev),
                 torch.empty((400000, 128), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
                for _ in range(K_DET)]
        streams = [torch.cuda.Stream() for _ in range(K_DET)]

        def worker(k, reps):
            with torch.cuda.stream(streams[k]):
                for _ in range(reps):
                    enqueue(sets[k])
        for reps in (2, N_K):
            torch.cuda.synchronize()
            t0c = time.perf_counter()
            th = [threading.Thread(target=worker, args=(k, reps)) for k in range(K_DET)]
            [t.start() for t in th]
            [t.join() for t in th]
            torch.cuda.synchronize()
            t_conc = (time.perf_counter() - t0c) / (K_DET * reps)
        del sets, streams
    if dist is not None:
        t = torch.tensor([dt, t_k], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_k = [float(v) for v in t.tolist()]
    # SURVEY.md 8d also asks for scale = 1.0 (the full 20 MP frame)
    full = None
    try:
        if args.no_sift_full:
            raise RuntimeError("skipped (--no-sift-full)")
        kernels.sift_detect(imgs[0], cap=1500000)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        kp1 = kernels.sift_detect(kernels.equalize_resize(imgs[1], 1.0), cap=1500000)[0]
        torch.cuda.synchronize()
        full = {"ms_per_image": round((time.perf_counter() - t1) * 1e3, 2), "keypoints": len(kp1),
                "detect_image": "5472x3648"}
    except Exception as e:                                  # noqa: BLE001 (e.g. not enough HBM left)
        full = {"error": str(e)[:200]}
    alg = 469.0 * h * w                                     # bytes per image
    sift_tr, sift_tr_src = None, None
    _p = os.path.join(REPO, 'profiles', AUX_TRAFFIC_FILE)
    if os.path.exists(_p):
        with open(_p) as fp:
            _d = json.load(fp)
        sift_tr = _d.get("sift", {}).get("hbm_bytes_per_frame") or None
        sift_tr_src = "profiles/%s (%s)" % (AUX_TRAFFIC_FILE, _d.get("source", ""))
    cpu = None                                              # filled in by main() at the end
    global _SIFT_SAMPLE
    _SIFT_SAMPLE = scaled.cpu().numpy() if rank == 0 else None
    return {"metric": "sift_images_per_sec", "value": round(n_local * world / dt, 2),
            "image": "5472x3648 synthetic, CLAHE + resize 0.4 -> %dx%d detect image" % (w, h),
            "keypoints_per_image": nkp // n_local, "ms_per_image": round(dt / n_local * 1e3, 2),
            "ms_per_image_detector_kernels": round(t_k * 1e3, 3),
            "host_enqueue_ms_per_image": round(t_host * 1e3, 3),
            "roofline": {"bound": "hbm", "kernels": "pyramid + extrema + orientation + descriptor",
                         "achieved": round(alg / t_k / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(alg / t_k / 1e9 / 8000.0, 4), "bytes_per_image": alg,
                         "traffic": sift_tr, "traffic_source": sift_tr_src,
                         "timing": "hipEvents around %d whole detects on the launch stream, steady state "
                                   "(3 untimed before); per-kernel durations: profiles/r6_kernel_stats.txt "
                                   "(rocprofv3 --kernel-trace --stats of the bench command)" % N_K},
            "concurrent_8": None if t_conc is None else {
                "ms_per_image": round(t_conc * 1e3, 3), "achieved": round(alg / t_conc / 1e9, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(alg / t_conc / 1e9 / 8000.0, 4),
                "timing": "wall clock, 8 threads x %d detections, one stream and buffer set per thread" % N_K},
            "scale_1_0": full, "cpu_baseline": cpu, "dtype": "f32 pyramid, f64 histograms",
            "parallelism": "image-shard x%d" % world}


def ba_bench(rank, world, dev, dist, args):
    """BA iterations / s on BASELINE configs[3] (2812 cameras, ~300 k points, ~2 M observations,
    synthetic): one "iteration" = one TRF outer iteration (Jacobian build +
    LSMR Gauss-Newton step + >= 1 residual evaluation).  Observations are sharded by point over
    the ranks; J^T u and the m-dots are all-reduced (RCCL)."""
    from imageanalysis_amd import ba_solver, synth
    p = synth.make_ba_problem()
    C, P, O = len(p['cams0']), len(p['pts0']), len(p['cam_idx'])
    K = p['K']
    calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
    prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib,
                              rank=rank, world=world)
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    lb = np.full(x0.size, -np.inf)
    ub = np.full(x0.size, np.inf)
    cp = p['cams0']
    for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):          # optimizer.py:425-446
        lb[j:C * 7:7] = cp[:, j] - dlt
        ub[j:C * 7:7] = cp[:, j] + dlt
    prob.set_x(x0)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, reps):
        fn()
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    f_res, f_jac = prob.bound_launchers()          # ctypes arguments marshalled once
    t_res = timed(f_res, 100)
    t_jac = timed(f_jac, 50)
    o_local = prob.O
    # The same two kernels on a working set that cannot sit in the 256 MiB Infinity Cache: six
    # copies of the problem (6 x 125 MB for the residual, 6 x 439 MB with the Jacobian blocks) in
    # rotation, so that every launch finds its inputs evicted by the five launches before it.
    # (`timed` above re-launches ONE problem back to back: its 125 MB come out of the cache.)
    t_res_cold = t_jac_cold = copy_bw = None
    stream_bw = {}
    n_rot = 6
    if world == 1:
        rot = [prob]
        for _ in range(n_rot - 1):
            q = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib,
                                   rank=rank, world=world)
            q.set_x(x0)
            rot.append(q)
        launch = [q.bound_launchers() for q in rot]

        def rot_res():
            for fr, _fj in launch:
                fr()

        def rot_jac():
            for _fr, fj in launch:
                fj()
        t_res_cold = timed(rot_res, 20) / n_rot
        t_jac_cold = timed(rot_jac, 10) / n_rot
        del launch, rot
        torch.cuda.empty_cache()
        # yardsticks on the same box, same moment (iamx_hbm_copy16: a grid-stride stream kernel, 16 B
        # per lane and step -- the form the hardware guide measures): six buffers in rotation, 125 MB
        # moved per launch, as a plain copy (62.5 MB in, 62.5 MB out) and in the residual kernel's
        # own mix (three words read per word written: 94 MB in, 31 MB out)
        import ctypes
        from imageanalysis_amd import _lib
        Lc = _lib.lib()
        stp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        stream_bw = {}
        for k in (1, 3):
            n_out = int(125.5e6 / (k + 1)) // 16
            srcs = [torch.empty(2 * n_out * k, dtype=torch.float64, device=dev).normal_() for _ in range(n_rot)]
            dsts = [torch.empty(2 * n_out, dtype=torch.float64, device=dev) for _ in range(n_rot)]

            def rot_copy():
                for a_, b_ in zip(srcs, dsts):
                    Lc.iamx_hbm_copy16(ctypes.c_void_p(a_.data_ptr()), ctypes.c_void_p(b_.data_ptr()),
                                       n_out, k, 1024, stp)
            stream_bw[k] = n_out * 16 * (k + 1) / (timed(rot_copy, 20) / n_rot) / 1e9
            del srcs, dsts
            torch.cuda.empty_cache()
        copy_bw = stream_bw[1]
    # untimed warm-up solve (workspace and allocator blocks of every branch of the step selection,
    # code-object load), like --warmup for matching; the collector has the matching section's heap
    # behind it before the clock starts
    ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4,
                         max_nfev=(args.ba_iters + 1) if args.ba_iters else None, verbose=0)
    sync()
    import gc
    gc.collect()
    del prob.inner_iterations[:]
    t0 = time.perf_counter()
    res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4,
                               max_nfev=(args.ba_iters + 1) if args.ba_iters else None, verbose=0)
    sync()
    dt = time.perf_counter() - t0
    inner_its = list(prob.inner_iterations)
    # the same outer iteration with SciPy's subproblem formulation (LSMR on the whole system,
    # Optimizer.solver = 'device-lsmr'), 3 iterations: what the Schur solver replaced
    lsmr_ref = None
    if world == 1:
        prob.inner = 'lsmr'
        ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=2, verbose=0)
        sync()
        del prob.inner_iterations[:]
        t1 = time.perf_counter()
        res_l = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=4, verbose=0)
        sync()
        dt_l = time.perf_counter() - t1
        lsmr_ref = {"value": round(res_l.iterations / dt_l, 3), "unit": "TRF iterations/s",
                    "iterations": int(res_l.iterations), "lsmr_iterations": int(sum(prob.inner_iterations)),
                    "seconds": round(dt_l, 3),
                    "rms_residual_px": round(float(np.sqrt(2.0 * res_l.cost / (2 * O))), 3)}
        prob.inner = 'schur'
    if dist is not None:
        t = torch.tensor([dt, t_res, t_jac], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_res, t_jac = [float(v) for v in t.tolist()]
    prob.set_x(res.x)
    mre = float(np.sqrt(2.0 * res.cost / (2 * O)))
    HBM = 8000.0
    # the dominant BA kernels: one LSMR iteration = J v and J^T u plus the vector updates.
    # ALGORITHMIC bytes (products with the STORED scaled f64 blocks):
    #   forward + camera adjoint  O*(160 J + 8 idx + 32 ut r/w) + n*32
    #   point adjoint             O*(48 Jp + 16 ut gather + 4 idx) + n*40
    #   update                    n*56
    # The shipped kernels are matrix free (they re-derive the blocks from camera and point), so
    # they MOVE fewer bytes than that: forward O*(4 idx + 32 ut r/w + 24 point-side products out),
    # adjoint O*(4 idx + 24 products gathered) + the point / camera table gathers that stay in the
    # Infinity Cache + n*128 ("executed_bytes_per_iteration", compulsory traffic only).
    lsmr = None
    if world == 1:
        prob.residual_jac()
        cn = prob.colnorm()
        cn[cn == 0] = 1
        d_dev = prob.upload_n(1.0 / cn)
        dreg = prob.upload_n(np.full(prob.n, 1e-3))
        its = 256
        ba_solver.lsmr_device_fused(prob, d_dev, dreg, atol=0, btol=0, conlim=0, maxiter=64)
        sync()
        t1 = time.perf_counter()
        ba_solver.lsmr_device_fused(prob, d_dev, dreg, atol=0, btol=0, conlim=0, maxiter=its)
        sync()
        t_it = (time.perf_counter() - t1) / its
        by = O * (200.0 + 68.0) + prob.n * 128.0
        ex = O * 88.0 + prob.n * 128.0
        lsmr = {"bound": "hbm", "kernels": "lsmr_fwd (+camera adjoint) + lsmr_adj (+sums, stopping tests) + lsmr_update3",
                "achieved": round(by / t_it / 1e9, 1), "peak": HBM, "unit": "GB/s",
                "frac": round(by / t_it / 1e9 / HBM, 4), "us_per_iteration": round(t_it * 1e6, 1),
                "bytes_per_iteration": by, "form": "matrix-free",
                # `frac` prices the ALGORITHMIC bytes of SURVEY 8d (products with a stored J);
                # the matrix-free kernels move ~2.7x fewer, and that working set (ut 31 MB,
                # tables 13 MB, n-vectors) is Infinity-Cache resident -- so frac_executed, not
                # frac, is what the memory system actually delivers
                "executed_bytes_per_iteration": ex,
                "achieved_executed": round(ex / t_it / 1e9, 1),
                "frac_executed": round(ex / t_it / 1e9 / HBM, 4),
                "working_set": "Infinity-Cache resident (< 256 MB)"}
        lsmr["traffic"], lsmr["traffic_source"] = aux_traffic(
            ('lsmr_fwd_kernel', 'lsmr_adj_kernel', 'lsmr_update3_kernel'), 'lsmr_update3_kernel')
        lsmr["timing"] = ("wall clock around %d fused iterations, queue kept full; per-kernel durations: "
                          "profiles/r6_kernel_stats.txt" % its)
    schur_it = None
    if world == 1:
        # one CG iteration of the Schur solver = three passes over the stored Jacobian blocks
        # (ALGORITHMIC bytes: forward Jc 112 + t 16, points Jp 48 + t 16 + slot 4, adjoint
        # Jc 112 + Jp 48 + t 16 + z 24 + idx 4 = 400 B per observation) + the one-workgroup update
        its = 64
        ba_solver.schur_solve(prob, d_dev, dreg, eta=0.0, maxiter=8, to_host=False)
        sync()
        t1 = time.perf_counter()
        _s, istop, itn, _rz, _ = ba_solver.schur_solve(prob, d_dev, dreg, eta=0.0, maxiter=its,
                                                       to_host=False)
        sync()
        t_sol = time.perf_counter() - t1
        t8 = time.perf_counter()
        ba_solver.schur_solve(prob, d_dev, dreg, eta=0.0, maxiter=8, to_host=False)
        sync()
        t_sol8 = time.perf_counter() - t8
        t_cg = (t_sol - t_sol8) / max(itn - 8, 1)             # (prepare / finish cancel)
        by = O * 400.0 + C * 7 * 8 * 12.0
        schur_it = {"bound": "hbm", "kernels": "schur_fwd + schur_pt + schur_adj + schur_update",
                    "achieved": round(by / t_cg / 1e9, 1), "peak": HBM, "unit": "GB/s",
                    "frac": round(by / t_cg / 1e9 / HBM, 4), "us_per_iteration": round(t_cg * 1e6, 1),
                    "bytes_per_iteration": by, "form": "stored blocks",
                    "iterations_timed": int(itn) - 8}
        schur_it["traffic"], schur_it["traffic_source"] = aux_traffic(
            ('schur_fwd_kernel', 'schur_pt_kernel', 'schur_adj_kernel', 'schur_pq_kernel',
             'schur_update1_kernel', 'schur_update2_kernel'), 'schur_fwd_kernel')
        schur_it["timing"] = ("wall clock, difference of a %d- and an 8-iteration solve; per-kernel "
                              "durations: profiles/r6_kernel_stats.txt, profiles/r3_ba_schur_trace.txt"
                              % its)
    cpu = None                                              # filled in by main() at the end
    def cold(t, bytes_per_obs):
        if t is None:
            return None
        bw = bytes_per_obs * o_local / t / 1e9
        mix = stream_bw.get(3) if bytes_per_obs == 64 else None
        return {"achieved": round(bw, 1), "frac": round(bw / HBM, 4),
                "us_per_launch": round(t * 1e6, 2),
                # what a pure stream kernel reaches over the same kind of rotating working set, same
                # box, same moment (iamx_hbm_copy16, 1024 workgroups): as a copy, and in this kernel's
                # own read : write mix -- the ceiling a gather + f64 arithmetic kernel is measured at
                "stream_copy_gbs": None if copy_bw is None else round(copy_bw, 1),
                "frac_of_stream_copy": None if copy_bw is None else round(bw / copy_bw, 4),
                "stream_same_mix_gbs": None if mix is None else round(mix, 1),
                "frac_of_stream_same_mix": None if mix is None else round(bw / mix, 4),
                "working_set": "%d problem copies in rotation (%.0f MB > the 256 MiB Infinity Cache)"
                               % (n_rot, n_rot * bytes_per_obs * o_local / 1e6),
                "timing": "hipEvents around %d rotations" % (20 if bytes_per_obs == 64 else 10)}

    def hbm_first(bytes_per_obs, t_hot, t_cold, kernel, timing):
        """the roofline object of a BA kernel: `achieved` / `frac` are the OUT-OF-CACHE figures (a
        rotating working set that cannot sit in the 256 MiB Infinity Cache) whenever they were
        measured (N = 1); back-to-back launches of one 125 / 439 MB problem are served from that
        cache and are reported as what they are, `cache_resident`"""
        c = cold(t_cold, bytes_per_obs)
        hot = {"achieved": round(bytes_per_obs * o_local * world / t_hot / 1e9, 1),
               "frac_of_hbm_peak": round(bytes_per_obs * o_local / t_hot / 1e9 / HBM, 4),
               "us_per_launch": round(t_hot * 1e6, 2),
               "working_set": "one problem, launched back to back: Infinity-Cache resident -- a "
                              "cache-level figure, NOT an HBM fraction"}
        tr = aux_traffic((kernel,), kernel)
        out = {"bound": "hbm", "peak": HBM, "unit": "GB/s", "bytes_per_obs": bytes_per_obs,
               "achieved": c["achieved"] if c else hot["achieved"],
               "frac": c["frac"] if c else hot["frac_of_hbm_peak"],
               "regime": "out of cache (rotating working set)" if c else "cache resident (N > 1: not rotated)",
               "out_of_cache": c, "cache_resident": hot, "traffic": tr[0], "traffic_source": tr[1],
               "timing": timing}
        return out
    return {"metric": "ba_iterations_per_sec", "value": round(res.iterations / dt, 3),
            # time to the reference's stopping rule (ftol = 1e-4) is the figure that compares with
            # another inner solver: the Schur path takes more, cheaper outer iterations than LSMR
            "seconds_to_ftol": round(dt, 4),
            "iterations": int(res.iterations), "njev": int(res.njev), "nfev": int(res.nfev),
            "status": int(res.status),
            "inner_solver": "Schur complement + block-Jacobi CG (iamx_ba_accumulate, iamx_ba_schur_*), "
                            "forcing term eta = %g" % prob.schur_eta,
            "inner_iterations": int(sum(inner_its)),
            "inner_iterations_per_solve_max": int(max(inner_its)) if inner_its else 0,
            "seconds": round(dt, 3), "lsmr_reference": lsmr_ref,
            "cameras": C, "points": P, "observations": O, "rms_residual_px": round(mre, 3),
            "residual_evals_per_sec": round(1.0 / t_res, 1),
            "residual": hbm_first(64.0, t_res, t_res_cold, 'ba_residual_pipe_kernel',
                                  "hipEvents around 100 launches / 20 rotations (kernel: ba_residual_pipe_kernel, "
                                  "the persistent pipelined walk; IAMX_BA_RESIDUAL=lds selects the "
                                  "one-chain-per-workgroup form, tools/ba_resid_ab.py compares them)"),
            "residual_jac": hbm_first(224.0, t_jac, t_jac_cold, 'ba_residual_jac_kernel',
                                      "hipEvents around 50 launches / 10 rotations (ba_residual_jac_kernel)"),
            "schur_iteration": schur_it, "lsmr_iteration": lsmr, "cpu_baseline": cpu,
            "dtype": "f64", "parallelism": "point-shard x%d" % world}


AUX_TRAFFIC_FILE = 'r6_ba_sift_traffic.json'


def aux_traffic(bases, per):
    """HBM bytes per `per`-kernel launch of the kernels whose base name is in `bases`, summed
    (an iteration = every kernel of it), from the committed PMC passes of this command
    (profiles/r4_ba_sift_traffic.json, tools/aux_traffic_json.py); (None, reason) without it"""
    path = os.path.join(REPO, 'profiles', AUX_TRAFFIC_FILE)
    if not os.path.exists(path):
        return None, None
    with open(path) as fp:
        doc = json.load(fp)
    ks = doc.get("kernels", {})
    n_per = sum(v["dispatches"] for k, v in ks.items() if k.split('<')[0] == per)
    tot = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for k, v in ks.items()
              if k.split('<')[0] in bases)
    if n_per == 0 or tot == 0:
        return None, "profiles/%s holds no %s launches" % (AUX_TRAFFIC_FILE, per)
    return int(tot / n_per), "profiles/%s (%s)" % (AUX_TRAFFIC_FILE, doc.get("source", ""))


_SIFT_SAMPLE = None


def sift_cpu_baseline():
    """The reference's detector is cv2.SIFT_create().detectAndCompute (scripts/lib/image.py:324;
    OpenCV is not installed here): the C / OpenMP restatement of the same published algorithm
    (oracle/sift_ref.c, the parity oracle of the kernels) on the bench's own detect image, all
    host cores, for ~10 s."""
    from oracle import cpu_ref, sift_oracle
    cpu_ref.set_num_threads(cpu_ref.host_cores())      # (a cgroup quota, not the 256 threads the box shows)
    gray = sift_oracle.bgr_to_gray(_SIFT_SAMPLE)
    cpu_ref.sift_detect(gray)                                # warm (thread pool, page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        kps, _des = cpu_ref.sift_detect(gray)
        n += 1
        el = time.perf_counter() - t0
        if el > 10.0 or n >= 200:
            break
    return {"value": round(n / el, 3), "unit": "images/s", "cores": cpu_ref.num_threads(),
            "kind": "port",
            "sample": "%d x the %dx%d detect image (%d keypoints) in %.1f s with oracle/sift_ref.c "
                      "(OpenMP; CLAHE + resize not included)" % (n, gray.shape[1], gray.shape[0],
                                                                 len(kps), el)}


def ba_cpu_baseline():
    """The reference's solver call (scipy least_squares, trf, jac_sparsity, x_scale='jac',
    ftol=1e-4; scripts/lib/optimizer.py:491-501) on the host cores, with the residual from the
    oracle's OpenMP C restatement instead of the per-camera python loop (generous to the CPU),
    on a REDUCED instance of the same synthetic scene (300 cameras, ~195 k observations,
    SURVEY.md 8d) for 3 TRF iterations; the full-size figure is a linear extrapolation in the
    observation count."""
    from scipy.optimize import least_squares
    from scipy.sparse import csr_matrix
    from oracle import cpu_ref
    from imageanalysis_amd import synth
    cpu_ref.set_num_threads(cpu_ref.host_cores())
    p = synth.make_ba_problem(rows=10, cols=30, n_points=32000, n_obs=195000)
    C, P, O = len(p['cams0']), len(p['pts0']), len(p['cam_idx'])
    K = p['K']
    intr = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
    cam, pt = p['cam_idx'].astype(np.int64), p['pt_idx'].astype(np.int64)
    cols = np.concatenate([cam[:, None] * 7 + np.arange(7), C * 7 + pt[:, None] * 3 + np.arange(3)], 1)
    rows = np.repeat(np.arange(2 * O), 10)
    A = csr_matrix((np.ones(20 * O, np.int8), (rows, np.repeat(cols, 2, axis=0).ravel())),
                   shape=(2 * O, C * 7 + P * 3))
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    lb, ub = np.full(x0.size, -np.inf), np.full(x0.size, np.inf)
    for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
        lb[j:C * 7:7] = p['cams0'][:, j] - dlt
        ub[j:C * 7:7] = p['cams0'][:, j] + dlt

    def fun(x):
        return cpu_ref.ba_residual(x[:C * 7], x[C * 7:], p['cam_idx'], p['pt_idx'], p['uv'], intr,
                                   p['dist'])

    t0 = time.perf_counter()
    res = least_squares(fun, x0, jac_sparsity=A, verbose=0, x_scale='jac', ftol=1e-4,
                        method='trf', bounds=(lb, ub), max_nfev=4)
    dt = time.perf_counter() - t0
    its = max(int(res.njev) - 1, 1)
    return {"value": round(its / dt, 4), "unit": "TRF iterations/s", "cores": cpu_ref.num_threads(),
            "kind": "port",
            "sample": "scipy least_squares(trf, jac_sparsity, x_scale='jac') + oracle/cpu_ref.c "
                      "residual on %d cameras / %d points / %d observations, %d iterations in "
                      "%.1f s" % (C, P, O, its, dt),
            "extrapolated_to_full_config": round(its / dt * O / 1960898.0, 4)}


def cpu_baseline(sample_images):
    """Brute-force 2-NN on the host cores over a bounded sample of the same workload (4096 x 4096 x
    128 pairs, both directions, ~10 s of CPU work), OpenMP over all cores in one parallel region:
    the AVX-512 VNNI kernel of oracle/knn2_simd.c where the host has it (vpdpbusd, train rows
    tiled lane-wise, four query rows per tile load: ~30x the plain loop), else the plain C loop
    of oracle/cpu_ref.c.  Both are the oracle's arithmetic (exact integer d^2, same results):
    a port, cv2 is not installed."""
    from oracle import cpu_ref
    rng = np.random.default_rng(0)
    n_img = 16
    imgs = rng.integers(0, 256, (n_img, KPTS, DIM), dtype=np.uint8)
    if sample_images is not None:
        imgs[:len(sample_images)] = sample_images
    threads = cpu_ref.set_num_threads(cpu_ref.host_cores())
    unordered = [(i, j) for i in range(n_img) for j in range(i + 1, n_img)]       # 120 pairs
    ordered = np.array(unordered + [(j, i) for i, j in unordered], np.int32)
    simd = cpu_ref.knn2_simd_available()
    run = cpu_ref.knn2_l2_u8_batch_simd if simd else cpu_ref.knn2_l2_u8_batch
    run(imgs, ordered[:2 * threads // 64 + 2])                                     # warm
    n, t0 = 0, time.perf_counter()
    while True:
        run(imgs, ordered)
        n += len(unordered)
        el = time.perf_counter() - t0
        if el > 10.0 or n >= 1000000:
            break
    out = {"value": round(n / el, 3), "unit": "pairs/s", "cores": threads, "kind": "port",
           "sample": "%d pairs of 4096x4096x128 (both directions, exact top-2 only) in %.1f s with %s "
                     "(OpenMP, %d threads, one parallel region)"
                     % (n, el, "oracle/knn2_simd.c (AVX-512 VNNI)" if simd else "oracle/cpu_ref.c (plain C)",
                        threads)}
    if simd:
        # the plain loop beside it (round 1-2's baseline), a short sample
        t1 = time.perf_counter()
        cpu_ref.knn2_l2_u8_batch(imgs, ordered[:max(2 * (threads // 16), 8)])
        out["plain_c_pairs_per_sec"] = round(max(threads // 16, 4) / (time.perf_counter() - t1), 3)
    return out


if __name__ == '__main__':
    main()
