#!/bin/bash
# round 5, GPU call 2: all GPU tests, SIFT timing after the blur rework, consolidation THP A/B on this host,
# bench, configs[4] at 2048 frames
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== gpu tests $(date +%T)"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/r5_gpu_tests_2.txt" 2>&1; grep -E "passed|failed" "$OUT/r5_gpu_tests_2.txt" | tail -3; grep -E "^FAILED|^ERROR" "$OUT/r5_gpu_tests_2.txt" | head -20
echo "== sift $(date +%T)"
{ timeout 120 python tools/sift_stream_time.py 20; } > "$OUT/r5_sift_time_blur2.txt" 2>&1; cat "$OUT/r5_sift_time_blur2.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sift_stats -o s -- python "$OLDPWD/tools/sift_stream_time.py" 20 > /tmp/sift_stats.log 2>&1)
python tools/prof_summary.py /tmp/sift_stats "$OUT/r5_sift_kernel_stats.txt" > /dev/null 2>&1; head -30 "$OUT/r5_sift_kernel_stats.txt"
echo "== consolidation, host only: THP probe vs forced $(date +%T)"
for v in "" "IAMX_THP=1" "IAMX_THP=0"; do
  echo "-- ${v:-probe}"; env $v IAMX_LINK_TIMING=1 timeout 300 python tools/consolidate_rate.py 16 32 6000 2>&1 | grep -E "setup|pass 1|total|^  [a-z_0-9]+ +[0-9.]+ s|consolidate"
done > "$OUT/r5_consolidate_thp.txt" 2>&1; cat "$OUT/r5_consolidate_thp.txt"
echo "== bench $(date +%T)"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/r5_bench_2.json" 2> "$OUT/r5_bench_2.err"; tail -c 2600 "$OUT/r5_bench_2.json"; tail -n 3 "$OUT/r5_bench_2.err"
echo "== e2e-full $(date +%T)"
N=${E2E_N:-2048}
timeout 1500 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full $N > "$OUT/r5_e2e_full_raw.json" 2> "$OUT/r5_e2e_full.err"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r5_e2e_full_raw.json').read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_%d.json' % e['images'], 'w'), indent=1)
    print(json.dumps({k: e.get(k) for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'host_peak_rss_bytes', 'hbm_after_match', 'hbm_model', 'ba', 'image_pairs_matched', 'image_pairs_with_matches', 'keypoints_per_image', 'render_seconds_untimed', 'max_baseline_error_m')}))
except Exception as ex:
    print('e2e-full: no result', ex)
PY
tail -n 5 "$OUT/r5_e2e_full.err"
echo "== done $(date +%T)"
