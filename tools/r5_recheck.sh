#!/bin/bash
# after the four-set mix-up: the tests that failed on the final run (default = two-set form now),
# the parity tests of all three exact-stage forms, and the timing A/B of two against four sets
mkdir -p gpurun_out
out=gpurun_out/r5_recheck.txt
: > $out
timeout 400 python -m pytest tests/test_match_sym_gpu.py tests/test_bench_gpu.py "tests/test_config_sizes_gpu.py::test_config4_slice_at_the_real_frame_size" -q -m gpu --maxfail=3 2>&1 | tail -6 | tee -a $out
for sets in 2 4; do
  echo "SETS=$sets" | tee -a $out
  IAMX_EXACT_SETS=$sets timeout 200 python tools/exact_stage_ab.py 2>&1 | grep "PRUNE=\|Error\|error" | tee -a $out
done
