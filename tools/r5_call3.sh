#!/bin/bash
# round 5, GPU call 3: tests, stream-clutter experiment, detect rate with the .feat record encoder (A/B),
# dense routing A/B on a 512-frame survey
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== gpu tests $(date +%T)"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/r5_gpu_tests_3.txt" 2>&1; grep -E "passed|failed" "$OUT/r5_gpu_tests_3.txt" | tail -3; grep -E "^FAILED|^ERROR" "$OUT/r5_gpu_tests_3.txt" | head -20
echo "== stream clutter $(date +%T)"
timeout 200 python tools/sift_stream_clutter.py > "$OUT/r5_sift_stream_clutter.txt" 2>&1; cat "$OUT/r5_sift_stream_clutter.txt"
echo "== detect rate $(date +%T)"
{ timeout 300 python tools/detect_rate.py 192 --no-serial; timeout 300 python tools/detect_rate.py 192 --no-serial --feat-zlib; } > "$OUT/r5_detect_rate.txt" 2>&1; grep -v amdgpu.ids "$OUT/r5_detect_rate.txt"
echo "== dense routing A/B, 512 frames $(date +%T)"
for mode in never auto; do
  IAMX_DENSE_ROUTE=$mode timeout 900 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full 512 > "$OUT/r5_e2e_512_$mode.raw" 2> "$OUT/r5_e2e_512_$mode.err"
  python - "$mode" <<'PY'
import json, sys
mode = sys.argv[1]
try:
    d = json.loads(open('gpurun_out/r5_e2e_512_%s.raw' % mode).read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_512_%s.json' % mode, 'w'), indent=1)
    print(mode, json.dumps({k: e.get(k) for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'route_rounds', 'image_pairs_matched', 'image_pairs_with_matches', 'ba', 'max_baseline_error_m')}))
except Exception as ex:
    print(mode, 'no result', ex)
PY
  tail -n 3 "$OUT/r5_e2e_512_$mode.err" | grep -v amdgpu.ids
done
echo "== done $(date +%T)"
