"""GPU: the named BASELINE configs at their own size.

configs[2] -- all 3 952 266 unordered pairs of the 2812-image x 4096-keypoint survey through the
shipped symmetric sweep + candidate test + exact stage + compaction on ONE GPU (the whole job
of bench.py --gpus N, about eight seconds of kernels): nothing unresolved, survivor counts that
follow the planted overlaps, and >= 200 ordered pairs spread over the schedule bit-equal to
oracle/cpu_ref.c."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_config2_all_pairs_job_on_one_gpu():
    import torch
    sys.path.insert(0, REPO)
    import bench
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    dev = kernels.require_gpu()
    n_img, kpts = bench.CONFIG2_IMAGES, bench.KPTS
    raw = bench.synth_descriptors(n_img, 0, n_img, dev)                  # 1.47 GB of uint8
    store = kernels.DescriptorStore([kpts] * n_img)
    L, _ptr, sp = kernels.lib(), kernels._ptr, kernels.stream_ptr()
    src_off = torch.arange(n_img + 1, dtype=torch.int64, device=dev) * kpts
    scratch = torch.empty(3 * n_img * kpts, dtype=torch.int32, device=dev)
    kernels.check(L.iamx_desc_pack_u8(_ptr(raw), n_img * kpts, _ptr(store.desc), _ptr(store.norm_q),
                                      _ptr(store.norm_t), sp), 'iamx_desc_pack_u8')
    kernels.check(L.iamx_desc3_pack_batch_u8(_ptr(raw), _ptr(src_off), _ptr(store.img_off3), n_img,
                                             n_img * kpts, kpts, _ptr(store.desc3), _ptr(store.sn2),
                                             _ptr(store.sct), _ptr(store.sperm), _ptr(store.sinv),
                                             _ptr(scratch), sp), 'iamx_desc3_pack_batch_u8')
    del scratch
    launches, n_pairs = bench.pair_schedule(n_img, 0, 1, 8192)
    assert n_pairs == n_img * (n_img - 1) // 2 == 3952266
    thresh = bench.MAX_DISTANCE * bench.MATCH_RATIO
    # the pairs checked against the oracle: 104 unordered (208 ordered), spread over the launches
    # -- the first and the last pair of the schedule, neighbours (planted overlap) and far pairs
    rng = np.random.default_rng(2812)
    pick = {}
    for li in np.unique(np.concatenate([[0, len(launches) - 1],
                                        rng.integers(0, len(launches), 60)])):
        half = len(launches[li]) // 2
        ks = set(int(k) for k in rng.integers(0, half, 2))
        adj = np.nonzero(np.abs(launches[li][:half, 0] - launches[li][:half, 1]) == 1)[0]
        if len(adj):
            ks.add(int(adj[0]))
        if li == 0:
            ks.add(0)
        if li == len(launches) - 1:
            ks.add(half - 1)
        pick[int(li)] = sorted(ks)
    ws = kernels.PairWorkspace(8192 * kpts, 8192)
    counts = torch.zeros(2 * n_pairs, dtype=torch.int32, device=dev)
    cands = torch.zeros(1, dtype=torch.int64, device=dev)
    unresolved = torch.zeros(1, dtype=torch.int64, device=dev)
    zero_div = torch.zeros(1, dtype=torch.int64, device=dev)
    got, base = {}, 0
    for li, ordered in enumerate(launches):
        pb = kernels.PairBatch(store, ordered, sym=True)
        assert pb.sym and pb.sym_form == 2
        pb.run(ws, thresh)
        n = pb.n_pairs
        counts[2 * base:2 * base + n] = ws.surv_cnt[:n]
        cands += ws.seg_count[:n].sum()
        unresolved += ws.unresolved
        zero_div += ws.zero_div
        if li in pick:
            first, cnt = ws.survivor_counts(n)
            half = n // 2
            for k in pick[li]:
                for p in (k, half + k):
                    lo, hi = int(first[p]), int(first[p] + cnt[p])
                    got[(int(ordered[p, 0]), int(ordered[p, 1]))] = (
                        ws.surv_q[lo:hi].cpu().numpy(), ws.surv_t[lo:hi].cpu().numpy(),
                        ws.surv_metric[lo:hi].cpu().numpy())
        base += n // 2
    torch.cuda.synchronize()
    assert base == n_pairs
    assert int(unresolved.item()) == 0 and int(zero_div.item()) == 0
    counts = counts.cpu().numpy().astype(np.int64)
    # per launch the layout is [fwd of its pairs..., rev of its pairs...]
    und = np.concatenate([o[:len(o) // 2] for o in launches])
    fwd = np.concatenate([counts[2 * s:2 * s + h] for s, h in
                          zip(np.cumsum([0] + [len(o) // 2 for o in launches[:-1]]),
                              [len(o) // 2 for o in launches])])
    rev = np.concatenate([counts[2 * s + h:2 * s + 2 * h] for s, h in
                          zip(np.cumsum([0] + [len(o) // 2 for o in launches[:-1]]),
                              [len(o) // 2 for o in launches])])
    adjacent = np.abs(und[:, 0] - und[:, 1]) == 1
    assert adjacent.sum() == n_img - 1
    # image j repeats 30 % of image j-1's rows (+-6 noise): ~1200 survivors in either direction of
    # a neighbouring pair, next to nothing everywhere else
    assert fwd[adjacent].min() > 600 and rev[adjacent].min() > 600
    assert fwd[~adjacent].max() < 80 and rev[~adjacent].max() < 80
    total = int(counts.sum())
    assert total <= int(cands.item()) <= total * 1.01          # the bound test is nearly exact
    # (gamma-distributed synthetic descriptors have no chance coincidences at all: every
    #  survivor of the job belongs to a planted overlap, found in both directions)
    assert total == int(fwd[adjacent].sum() + rev[adjacent].sum()) > 4.0e6
    # ---- the sampled pairs against the oracle (plain C, OpenMP)
    assert len(got) >= 200
    imgs = sorted({i for pr in got for i in pr})
    host = {i: raw[i].cpu().numpy() for i in imgs}
    for (a, b), (sq, st, sm) in got.items():
        ridx, rd2 = cpu_ref.knn2_l2_u8(host[a], host[b])
        d = np.sqrt(rd2.astype(np.float32)).astype(np.float64)
        metric = d[:, 0] * (d[:, 0] / d[:, 1])
        keep = np.nonzero(metric < thresh)[0]
        assert np.array_equal(sq, keep), (a, b)
        assert np.array_equal(st, ridx[keep, 0]), (a, b)
        assert np.array_equal(sm, metric[keep]), (a, b)


def test_config4_slice_at_the_real_frame_size():
    """BASELINE configs[4] (detect -> match -> link -> triangulate -> BA) on 24 rendered frames of
    the survey camera's own size, 5472 x 3648, at the reference's detector scale 0.4: the chain
    scripts/process.py:236-407 drives, through the drop-in entry points (the function bench.py's
    `e2e` record runs).  The project is handed poses that are off by ~1 m / ~1 deg."""
    sys.path.insert(0, REPO)
    import bench
    out = bench.e2e_bench(24, full_frame=True)
    assert out["image_size"] == [5472, 3648] and out["images"] >= 24 and out["detect_scale"] == 0.4
    assert out["keypoints_per_image"] > 20000                 # 20 MP frames at scale 0.4
    rows, cols = out["grid"]
    # every image overlaps its neighbours along and across the flight lines
    assert out["image_pairs_with_matches"] >= (rows - 1) * cols + rows * (cols - 1)
    assert out["chains"] > 20000
    assert out["groups"] == [out["images"]]                    # one connected block
    ba = out["ba"]
    assert ba["cameras"] == out["images"] and ba["observations"] > 3 * ba["points"] > 30000
    assert ba["mean_abs_residual_px_before"] > 10.0            # ~1 m / 1 deg at 2.7 cm per pixel
    assert ba["mean_abs_residual_px_after"] < 1.0
    # relative geometry recovered up to the one global scale BA cannot observe
    assert abs(out["baseline_scale"] - 1.0) < 0.03 and out["max_baseline_error_m"] < 0.15


def test_config4_at_128_frames_distance_schedule():
    """BASELINE configs[4] beyond a slice: 128 rendered 5472 x 3648 frames through the whole chain
    on the neighbour + distance-window schedule (scripts/lib/matcher.py:886-903 with its window
    enabled -- all-pairs is not the shape of a survey of thousands of frames): one connected
    block, sub-pixel residuals, the relative geometry of the truth."""
    sys.path.insert(0, REPO)
    import bench
    out = bench.e2e_bench(128, full_frame=True, schedule='distance')
    n = out["images"]
    assert n >= 128 and out["image_size"] == [5472, 3648] and out["schedule"] == 'distance'
    rows, cols = out["grid"]
    assert out["image_pairs_matched"] < n * (n - 1) // 4          # a window, not all pairs
    assert out["image_pairs_with_matches"] >= (rows - 1) * cols + rows * (cols - 1)
    assert out["groups"] == [n]
    ba = out["ba"]
    print(out)
    # (Optimizer keeps the chains seen by >= 3 images: min_chain_len, scripts/lib/optimizer.py:66-95)
    assert ba["cameras"] == n and ba["observations"] > 3 * ba["points"] > 3 * 15000
    assert ba["mean_abs_residual_px_before"] > 10.0 and ba["mean_abs_residual_px_after"] < 1.0
    assert abs(out["baseline_scale"] - 1.0) < 0.03 and out["max_baseline_error_m"] < 0.3
    assert 0 < out["peak_hbm_bytes"] < 200 * 2 ** 30


def test_config4_at_2048_frames_bounded_device_memory():
    """BASELINE configs[4] at 2048 rendered 20 MP frames, rendered / detected / deleted 1024 at a
    time like the 10 011-frame run of profiles/r6_e2e_full_10000.json (the whole survey's JPEGs
    need not fit the scratch disk): one connected block, sub-pixel residuals -- and the device
    memory the matching stage
    holds is what matcher.device_memory_model() says it is: the descriptor arena (the ONLY part
    that grows with the survey: two layouts x 140-144 B per row, no parity-partitioned copy) within
    5 % of the model, the pooled per-round workspaces inside the BATCH_BYTES budget, the peak of
    the whole chain below arena + three workspaces + the SIFT slots."""
    sys.path.insert(0, REPO)
    import bench
    from imageanalysis_amd import matcher
    out = bench.e2e_bench(2048, full_frame=True, schedule='distance', window=1024)
    n = out["images"]
    assert n >= 2048 and out["window_frames"] == 1024 and out["schedule"] == 'distance' and out["groups"] == [n]
    ba = out["ba"]
    assert ba["cameras"] == n and ba["mean_abs_residual_px_before"] > 10.0
    assert ba["mean_abs_residual_px_after"] < 1.0
    # (camera-to-camera distances up to 2.6 km across the block, against the truth the frames were
    #  rendered from, one gauge scale: 0.95 m at 2048 frames in round 5)
    assert abs(out["baseline_scale"] - 1.0) < 0.03 and out["max_baseline_error_m"] < 2.0
    rep, model = out["hbm_after_match"], out["hbm_model"]
    assert rep["images"] == n and rep["descriptor_rows"] == pytest.approx(n * out["keypoints_per_image"], rel=0.01)
    assert 0.9 * model["arena_bytes"] <= rep["descriptor_arena_bytes"] + rep["keypoint_arena_bytes"] \
        <= 1.05 * model["arena_bytes"]
    # per row of the arena: 128 + 12 (original order) + 128 + 16 (sorted order) bytes, padded; + 128
    # + 12 when a dense round of this run asked for the one-direction sweep's layout
    routed = out["route_rounds"]["one_direction"] > 0
    assert rep["descriptor_arena_bytes"] / float(rep["descriptor_rows"]) < (440.0 if routed else 300.0)
    assert rep["pooled_workspace_bytes"] <= 3 * matcher.BATCH_BYTES
    # (+ 8 SIFT slots, BA; the model is evaluated at the MEAN keypoint count, the run sizes its
    #  workspaces for 1.25 x the LARGEST count it knows: a few per cent of three 20 GB workspaces)
    assert out["peak_hbm_bytes"] <= 1.05 * model["peak_bytes"] + 24 * 2 ** 30
    print({k: out[k] for k in ("stage_seconds", "total_seconds", "peak_hbm_bytes", "hbm_after_match", "hbm_model")})
