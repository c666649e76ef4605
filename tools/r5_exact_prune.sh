#!/bin/bash
# exact stage forms: parity first, then the timing A/B
mkdir -p gpurun_out
out=gpurun_out/r5_exact_prune.txt
: > $out
timeout 900 python -m pytest tests/test_match_sym_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee -a $out
for cfg in "2 2" "4 1" "4 2"; do
  set -- $cfg
  echo "SETS=$1 SUB=$2" | tee -a $out
  IAMX_EXACT_SETS=$1 IAMX_EXACT_SUB=$2 timeout 300 python tools/exact_stage_ab.py 2>&1 | grep "PRUNE=\|Error\|error" | tee -a $out
done
