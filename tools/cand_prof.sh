#!/bin/bash
# per-kernel times of tools/cand_rate.py (rendered-frame descriptors, 30 ordered pairs)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp IAMX_CAND_CACHE=/tmp/cand.npz
[ -f /tmp/cand.npz ] || python tools/cand_rate.py > /dev/null 2>&1
rm -rf /tmp/p_cand
rocprofv3 --kernel-trace --stats -d /tmp/p_cand -o x --output-format csv -- python tools/cand_rate.py > /tmp/cand.log 2>&1
grep "sweep " /tmp/cand.log
python - "$(find /tmp/p_cand -name '*kernel_stats.csv' | head -1)" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('sym','narrow','compact')):
        print("%-64s calls=%s avg_us=%.1f" % (n[:64], r['Calls'], float(r['AverageNs'])/1e3))
P
