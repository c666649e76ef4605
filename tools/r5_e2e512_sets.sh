#!/bin/bash
# 512 rendered frames through the chain with the two workgroup forms of the exact stage
OUT=gpurun_out
mkdir -p $OUT
for sets in 4 2; do
  IAMX_EXACT_SETS=$sets timeout 600 python bench.py --images 64 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --e2e-full 512 > $OUT/r5_e2e_512_sets$sets.raw 2> $OUT/r5_e2e_512_sets$sets.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/r5_e2e_512_sets$sets.raw').read().strip().splitlines()[-1])
    e = d['e2e_full']
    json.dump(e, open('$OUT/r5_e2e_full_512_sets$sets.json', 'w'), indent=1)
    print('sets=$sets', json.dumps({k: e[k] for k in ('stage_seconds', 'total_seconds', 'images_per_sec_end_to_end', 'peak_hbm_bytes')}))
except Exception as ex:
    print('sets=$sets: no result', ex)
PY
done
