#!/bin/bash
# per-kernel averages of the 500-image bench command, _prev tree against this one (same box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
FLAGS="${PROF_FLAGS:---images 500 --steps 3 --warmup 1 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey}"
for tree in _prev .; do
  rm -rf /tmp/pp
  (cd $tree && rocprofv3 --kernel-trace --stats -d /tmp/pp -o x --output-format csv -- python bench.py $FLAGS > /dev/null 2>&1)
  echo "== $tree"
  python - "$(find /tmp/pp -name '*kernel_stats.csv' | head -1)" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r['Name']
    if any(k in n for k in ('sym','narrow','compact','knn2','postfilter','pack','similarity','scan','metric','copyBuffer','fill')):
        print("%-60s calls=%5s total_ms=%8.2f avg_us=%9.1f" % (n[:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
P
done
