#!/bin/bash
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sift_gpu.py tests/test_image_gpu.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED"
timeout 120 python tools/sift_stream_time.py 20 2>&1 | grep -v amdgpu > "$OUT/r5_sift_time_tail3.txt"; cat "$OUT/r5_sift_time_tail3.txt"
bash tools/r5_sift_single.sh 2>&1 | grep -E "one stream|stream [0-9]" | tail -75 | cut -c1-120
