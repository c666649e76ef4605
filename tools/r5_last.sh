#!/bin/bash
# round 5, last GPU call: tests, consolidation (host) on the box, bench, configs[4] at 4186 frames -- final binaries
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== gpu tests $(date +%T)"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/r5_gpu_tests.txt" 2>&1; grep -E "passed|failed" "$OUT/r5_gpu_tests.txt" | tail -2; grep -E "^FAILED|^ERROR" "$OUT/r5_gpu_tests.txt" | head
echo "== consolidation host $(date +%T)"
IAMX_LINK_TIMING=1 timeout 300 python tools/consolidate_rate.py 16 32 6000 --dup=0.15 2>&1 | grep -E "setup|pass 1|total|^  [a-z_0-9]+ +[0-9.]+ s|consolidate" > "$OUT/r5_consolidate_rate.txt"; cat "$OUT/r5_consolidate_rate.txt"
echo "== entry points $(date +%T)"
timeout 600 python tools/find_matches_rate.py > "$OUT/r5_fm_dense_final.txt" 2>&1; tail -n 1 "$OUT/r5_fm_dense_final.txt"
timeout 600 python tools/find_matches_rate.py 38 74 4096 > "$OUT/r5_fm_config2_final.txt" 2>&1; tail -n 1 "$OUT/r5_fm_config2_final.txt"
echo "== bench $(date +%T)"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/r5_bench_latest.json" 2> "$OUT/r5_bench_latest.err"; tail -c 1700 "$OUT/r5_bench_latest.json"; echo
for N in 4096; do
echo "== e2e-full $N $(date +%T)"
IAMX_LINK_TIMING=1 IAMX_E2E_PROFILE=consolidate timeout 1700 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full $N > "$OUT/r5_e2e_final_$N.raw" 2> "$OUT/r5_e2e_final_$N.err"
python - "$N" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r5_e2e_final_%s.raw' % sys.argv[1]).read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_%d_final.json' % e['images'], 'w'), indent=1)
    print(json.dumps({k: e.get(k) for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'host_peak_rss_bytes', 'hbm_after_match', 'ba', 'image_pairs_matched', 'image_pairs_with_matches', 'keypoints_per_image', 'max_baseline_error_m', 'route_rounds')}))
except Exception as ex:
    print('no result', ex)
PY
grep iamx_link_matches "$OUT/r5_e2e_final_$N.err" > "$OUT/r5_link_passes_$N.txt"
grep -v "amdgpu.ids\|iamx_link_matches" "$OUT/r5_e2e_final_$N.err" | head -30 > "$OUT/r5_e2e_4186_consolidate_profile.txt"; cat "$OUT/r5_e2e_4186_consolidate_profile.txt"
done
echo "== done $(date +%T)"
