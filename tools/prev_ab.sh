#!/bin/bash
# same-box A/B against the tree of an earlier commit checked out and built under _prev/
# (git worktree add _prev <commit>; bash _prev/imageanalysis_amd/csrc/build.sh)
cd "$(dirname "$0")/.."
FLAGS="--steps 3 --warmup 1 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e"
for tree in _prev . _prev .; do
  (cd $tree && python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['summary']; o=d['dense_overlap']
print('$tree', 'headline %.0f pairs/s (%.1f ms/step, frac %.4f)' % (d['value'], d['ms_per_step'], d['roofline']['frac']), 'config1 %.0f' % s['config1_500_pairs_per_sec'], 'dense %.0f pairs/s (sweep %.3f ms, filter+exact %.3f ms)' % (o['pairs_per_sec'], o['sweep_ms'], o['filter_and_exact_ms']))")
done
