#!/bin/bash
# round 5: consolidation on the box + configs[4] at 4186 and 512 frames on the final binaries
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== consolidation host $(date +%T)"
IAMX_LINK_TIMING=1 timeout 300 python tools/consolidate_rate.py 16 32 6000 --dup=0.15 2>&1 | grep -E "setup|pass |total|^  [a-z_0-9]+ +[0-9.]+ s|consolidate" > "$OUT/r5_consolidate_rate.txt"; cat "$OUT/r5_consolidate_rate.txt"
echo "== cleanup + dropin + pipeline tests $(date +%T)"
timeout 600 python -m pytest tests/test_cleanup.py tests/test_dropin.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed"
for N in 512 4096; do
echo "== e2e-full $N $(date +%T)"
IAMX_LINK_TIMING=1 timeout 1700 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full $N > "$OUT/r5_e2e_final_$N.raw" 2> "$OUT/r5_e2e_final_$N.err"
python - "$N" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r5_e2e_final_%s.raw' % sys.argv[1]).read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_%d_final.json' % e['images'], 'w'), indent=1)
    print(json.dumps({k: e.get(k) for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'host_peak_rss_bytes', 'ba', 'image_pairs_matched', 'image_pairs_with_matches', 'max_baseline_error_m', 'route_rounds')}))
except Exception as ex:
    print('no result', ex)
PY
grep iamx_link_matches "$OUT/r5_e2e_final_$N.err" > "$OUT/r5_link_passes_$N.txt"; tail -7 "$OUT/r5_link_passes_$N.txt"
done
echo "== done $(date +%T)"
