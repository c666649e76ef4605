#!/bin/bash
# per-launch timeline of single-stream SIFT detections in steady state (rocprofv3 --kernel-trace)
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
(cd /tmp && IAMX_SIFT_SINGLE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sift1 -o s -- python "$OLDPWD/tools/sift_stream_time.py" 20 > /tmp/sift1.log 2>&1)
grep "one stream" /tmp/sift1.log
python tools/prof_summary.py /tmp/sift1 "$OUT/r5_sift_single_kernel_stats.txt" > /dev/null 2>&1; head -24 "$OUT/r5_sift_single_kernel_stats.txt" | cut -c1-200
python - <<'PY' > gpurun_out/r5_sift_single_timeline.txt
import csv, glob, os
rows=[]
for f in glob.glob('/tmp/sift1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id',''), r.get('Queue_Id','')))
rows.sort()
# the last complete detection: from the last gray_up2x to the descriptor kernel behind it
starts=[i for i,r in enumerate(rows) if 'gray_up2x' in r[2]]
i0=starts[-2]; i1=starts[-1]
t0=rows[i0][0]
for s,e,name,st,q in rows[i0:i1]:
    short=name.replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    print('%-34s stream %-4s queue %-3s start %9.1f us  dur %8.1f us  end %9.1f' % (short[:34], st, q, (s-t0)/1e3, (e-s)/1e3, (e-t0)/1e3))
PY
cat gpurun_out/r5_sift_single_timeline.txt
