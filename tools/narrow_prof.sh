#!/bin/bash
# per-kernel times of the dense-overlap workload with the narrow exact stage on / off
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for nar in 1 0; do
  rm -rf /tmp/p_nar$nar
  IAMX_EXACT_NARROW=$nar timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_nar$nar -o x --output-format csv -- python tools/exact_stage_ab.py > /tmp/nar$nar.log 2>&1
  f=$(find /tmp/p_nar$nar -name "*kernel_stats.csv" | head -1)
  echo "== NARROW=$nar" >> gpurun_out/r6_narrow_prof.txt
  grep "rows:" /tmp/nar$nar.log >> gpurun_out/r6_narrow_prof.txt
  python - "$f" >> gpurun_out/r6_narrow_prof.txt <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('sym','knn2','narrow','compact')):
        print("%-70s calls=%s total_ms=%.3f avg_us=%.1f" % (n[:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
P
done
cat gpurun_out/r6_narrow_prof.txt
