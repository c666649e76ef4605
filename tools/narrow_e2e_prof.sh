#!/bin/bash
# kernel stats of the 512-frame end-to-end run (match stage split) + class counts on rendered frames
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r6b}
timeout 300 python tools/cand_rate.py 2>&1 | grep -v "^  \|keypoints" | tail -6 > gpurun_out/${TAG}_cand_rate.txt
cat gpurun_out/${TAG}_cand_rate.txt
rm -rf /tmp/p_e2e
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_e2e -o b --output-format csv -- python bench.py --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --images 64 --e2e-full 512 > gpurun_out/${TAG}_e2e512_prof.json 2> /tmp/e2e_prof.err
python tools/prof_summary.py /tmp/p_e2e gpurun_out/${TAG}_e2e512_kernel_stats.txt > /dev/null
head -14 gpurun_out/${TAG}_e2e512_kernel_stats.txt | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_e2e512_prof.json')); print(d['e2e_full']['stage_seconds'])"
