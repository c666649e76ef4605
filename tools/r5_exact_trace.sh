#!/bin/bash
# per-kernel times of the symmetric form on the dense-overlap workload
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for sub in 1 2; do
IAMX_EXACT_SUB=$sub timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t$sub -o b -- python $REPO/tools/exact_pmc_run.py > /tmp/t$sub.log 2>&1
python $REPO/tools/prof_summary.py /tmp/t$sub $REPO/gpurun_out/r5_exact_trace_sub$sub.txt | grep "sym\|knn2" | cut -c1-60,110-200
done
