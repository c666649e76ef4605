#!/bin/bash
mkdir -p gpurun_out
timeout 70 python -m pytest tests/test_config_sizes_gpu.py -q -m gpu --maxfail=3 --deselect "tests/test_config_sizes_gpu.py::test_config4_slice_at_the_real_frame_size" 2>&1 | tail -6 | tee gpurun_out/r5_recheck4.txt
