#!/bin/bash
# narrow exact stage: parity tests, then the dense-overlap A/B against the full scan
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_match_sym_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_narrow_tests.txt
cat gpurun_out/r6_narrow_tests.txt
{
IAMX_EXACT_NARROW=1 timeout 600 python tools/exact_stage_ab.py 2>&1 | grep "rows:"
IAMX_EXACT_NARROW=0 timeout 600 python tools/exact_stage_ab.py 2>&1 | grep "rows:"
} > gpurun_out/r6_narrow_ab.txt
cat gpurun_out/r6_narrow_ab.txt
