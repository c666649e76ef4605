#!/bin/bash
# find_matches meeting undetected images (process.py's call form), N frames: priority of the
# prefetch workers' streams A/B
cd "$(dirname "$0")/.."
N=${1:-1024}
FLAGS="--steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey --images 64"
for prio in ${PRIOS:--1 0}; do
  IAMX_WORKER_PRIO=$prio timeout 900 python bench.py $FLAGS --e2e-full $N --e2e-fused > /tmp/fused_$prio.json 2> /tmp/fused_$prio.err
  python - /tmp/fused_$prio.json $prio <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["e2e_full"]
print("IAMX_WORKER_PRIO=%s images %d:" % (sys.argv[2], d["images"]), d["stage_seconds"], "total", d["total_seconds"])
P
done
