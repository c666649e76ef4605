#!/bin/bash
# same-box A/B of the 512-frame end-to-end run: _prev tree against this one
cd "$(dirname "$0")/.."
N=${1:-512}
FLAGS="--steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --images 64 --e2e-full $N"
for tree in ${TREES:-_prev . _prev .}; do
  (cd $tree && python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['e2e_full']
print('$tree', d['images'], 'frames:', d['stage_seconds'], 'total', d['total_seconds'], 'pairs', d['image_pairs_matched'], 'chains', d['chains'])")
done
