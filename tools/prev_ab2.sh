#!/bin/bash
cd "$(dirname "$0")/.."
FLAGS="--steps 2 --warmup 1 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --images 2812"
run() { (cd $1 && env $2 python bench.py $FLAGS $3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 $3', 'headline %.0f pairs/s (%.1f ms/step, frac %.4f)' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"); }
for i in 1 2; do
run _prev A=1
run . A=1
run . IAMX_EXACT_NARROW=0
done
run _prev A=1 --no-overlap
run . A=1 --no-overlap
