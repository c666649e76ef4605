#!/bin/bash
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "-- desc 8, ori 5 (as built)"
timeout 600 python -m pytest tests/test_sift_gpu.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED"
for i in 1 2; do IAMX_SIFT_SINGLE=1 timeout 120 python tools/sift_stream_time.py 30 2>&1 | grep "one stream"; done
for ow in 6 8; do
echo "-- desc 8, ori $ow"
touch imageanalysis_amd/csrc/sift.hip; IAMX_EXTRA_FLAGS=-DIAMX_ORI_WAVES=$ow bash imageanalysis_amd/csrc/build.sh > /dev/null 2>&1
for i in 1 2; do IAMX_SIFT_SINGLE=1 timeout 120 python tools/sift_stream_time.py 30 2>&1 | grep "one stream"; done
done
