#!/bin/bash
# the remaining GPU test files on the final binaries
mkdir -p gpurun_out
timeout 80 python -m pytest tests/test_sift_gpu.py tests/test_image_gpu.py tests/test_jpeg_gpu.py tests/test_trf_helpers_gpu.py tests/test_comm_gpu.py -q -m gpu --maxfail=3 2>&1 | tail -6 | tee gpurun_out/r5_recheck3.txt
