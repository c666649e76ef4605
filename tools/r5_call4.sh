#!/bin/bash
# round 5, GPU call 4: where bench.py's single-stream SIFT time comes from; configs[4] at 4096 frames
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== sift bisect $(date +%T)"
timeout 400 python tools/sift_stream_bisect.py > "$OUT/r5_sift_stream_bisect.txt" 2>&1; grep -v amdgpu.ids "$OUT/r5_sift_stream_bisect.txt" | tail -20
echo "== consolidation host $(date +%T)"
IAMX_LINK_TIMING=1 timeout 300 python tools/consolidate_rate.py 16 32 6000 2>&1 | grep -E "setup|pass 1|total|^  [a-z_0-9]+ +[0-9.]+ s|consolidate" > "$OUT/r5_consolidate_rate.txt"; cat "$OUT/r5_consolidate_rate.txt"
echo "== e2e-full $(date +%T)"
N=${E2E_N:-4096}
timeout 1700 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full $N > "$OUT/r5_e2e_full_raw.json" 2> "$OUT/r5_e2e_full.err"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r5_e2e_full_raw.json').read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_%d.json' % e['images'], 'w'), indent=1)
    print(json.dumps({k: e.get(k) for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'host_peak_rss_bytes', 'hbm_after_match', 'hbm_model', 'ba', 'image_pairs_matched', 'image_pairs_with_matches', 'keypoints_per_image', 'render_seconds_untimed', 'max_baseline_error_m', 'route_rounds')}))
except Exception as ex:
    print('e2e-full: no result', ex)
PY
tail -n 5 "$OUT/r5_e2e_full.err" | grep -v amdgpu.ids
df -h /tmp | tail -1
echo "== done $(date +%T)"
