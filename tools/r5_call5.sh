#!/bin/bash
# round 5, GPU call 5: the bench's SIFT section first vs in place; consolidation profile at 1024 frames; new tests
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== bench sift order $(date +%T)"
for tag in inplace first; do
  flag=""; [ $tag = first ] && flag="--sift-first"
  timeout 400 python bench.py --steps 5 --warmup 2 --no-ba --no-e2e --no-survey --no-cpu-baseline --no-sift-full $flag > "$OUT/r5_bench_sift_$tag.json" 2> "$OUT/r5_bench_sift_$tag.err"
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r5_bench_sift_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
s = d['sift']
print(sys.argv[1], {k: s.get(k) for k in ('ms_per_image_detector_kernels', 'host_enqueue_ms_per_image', 'value', 'keypoints_per_image')}, s['roofline']['frac'], s.get('concurrent_8', {}).get('frac'))
PY
done
echo "== tests $(date +%T)"
timeout 900 python -m pytest tests/test_config_sizes_gpu.py tests/test_mirror_gpu.py tests/test_cleanup.py tests/test_pipeline_gpu.py -m gpu -q -s > "$OUT/r5_gpu_tests_5.txt" 2>&1; grep -E "passed|failed" "$OUT/r5_gpu_tests_5.txt" | tail -2; grep -E "^FAILED|^ERROR" "$OUT/r5_gpu_tests_5.txt" | head; grep "stage_seconds" "$OUT/r5_gpu_tests_5.txt" | tail -2 | cut -c1-900
echo "== e2e 1024 with the consolidation stage profiled $(date +%T)"
IAMX_E2E_PROFILE=consolidate timeout 900 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full 1024 > "$OUT/r5_e2e_1024.raw" 2> "$OUT/r5_e2e_1024_consolidate_profile.txt"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r5_e2e_1024.raw').read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_%d.json' % e['images'], 'w'), indent=1)
    print(json.dumps({k: e.get(k) for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'route_rounds', 'ba')}))
except Exception as ex:
    print('no result', ex)
PY
grep -v amdgpu.ids "$OUT/r5_e2e_1024_consolidate_profile.txt" | head -45
echo "== done $(date +%T)"
