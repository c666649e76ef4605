#!/bin/bash
# final sanity on the GPU box: build() from scratch, smoke(), the GPU tests, the driver's bench command
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== build + smoke $(date +%T)"
IAMX_INCREMENTAL=1 timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
echo "== gpu tests $(date +%T)"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 > "$OUT/r5_gpu_tests.txt" 2>&1; grep -E "passed|failed" "$OUT/r5_gpu_tests.txt" | tail -2; grep -E "^FAILED|^ERROR" "$OUT/r5_gpu_tests.txt" | head
echo "== bench $(date +%T)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r5_bench_latest.json" 2> "$OUT/r5_bench_latest.err"; tail -c 1700 "$OUT/r5_bench_latest.json"; echo
echo "== done $(date +%T)"
