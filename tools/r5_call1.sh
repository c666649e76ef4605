#!/bin/bash
# round 5, GPU call 1: state of the box, the GPU tests, SIFT graph A/B, the default bench, configs[4] at scale
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
{ free -g | head -2; df -h /tmp | tail -1; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2; } > "$OUT/r5_box.txt" 2>&1
cat "$OUT/r5_box.txt"
echo "== gpu tests $(date +%T)"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/r5_gpu_tests_1.txt" 2>&1; tail -n 5 "$OUT/r5_gpu_tests_1.txt"
echo "== sift graph A/B $(date +%T)"
{ timeout 120 python tools/sift_stream_time.py 20; IAMX_SIFT_NO_GRAPH=1 timeout 120 python tools/sift_stream_time.py 20; } > "$OUT/r5_sift_graph_ab.txt" 2>&1
cat "$OUT/r5_sift_graph_ab.txt"
echo "== bench $(date +%T)"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/r5_bench_1.json" 2> "$OUT/r5_bench_1.err"; tail -c 2500 "$OUT/r5_bench_1.json"; tail -n 3 "$OUT/r5_bench_1.err"
echo "== e2e-full $(date +%T)"
AVAIL=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo)
N=2048; [ "$AVAIL" -lt 110 ] && N=1024
echo "MemAvailable ${AVAIL} GB -> --e2e-full $N"
timeout 1200 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full $N > "$OUT/r5_e2e_full_raw.json" 2> "$OUT/r5_e2e_full.err"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r5_e2e_full_raw.json').read().strip().splitlines()[-1])
    e = d.get('e2e_full')
    json.dump(e, open('gpurun_out/r5_e2e_full_%d.json' % e['images'], 'w'), indent=1)
    print(json.dumps({k: e[k] for k in ('images', 'stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes', 'host_peak_rss_bytes', 'hbm_after_match', 'hbm_model', 'ba', 'image_pairs_matched', 'image_pairs_with_matches', 'keypoints_per_image', 'render_seconds_untimed')}))
except Exception as ex:
    print('e2e-full: no result', ex)
PY
tail -n 5 "$OUT/r5_e2e_full.err"
echo "== done $(date +%T)"
