#!/bin/bash
# the matching-related GPU tests on the final binaries
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_mirror_gpu.py tests/test_match_gpu.py tests/test_pipeline_gpu.py -q -m gpu --maxfail=3 2>&1 | tail -6 | tee gpurun_out/r5_recheck2.txt
