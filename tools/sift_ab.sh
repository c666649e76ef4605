#!/bin/bash
# A/B of the SIFT chain on the GPU box: per-kernel time (rocprofv3 --kernel-trace --stats) and HBM
# traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) of tools/sift_time.py, once per setting of
# the environment switches given as arguments ("name:VAR=1" ...; "base:" = none).
#   bash tools/sift_ab.sh base: noxcd:IAMX_SIFT_NO_XCD=1      -> gpurun_out/r4_sift_ab_<name>.txt
#   PASSES=stats: times only (the report then shows no traffic column values)
OUT="$PWD/gpurun_out"; REPO="$PWD"; mkdir -p "$OUT"; export TMPDIR=/tmp
for spec in "$@"; do
    name="${spec%%:*}"; envs="${spec#*:}"
    for pass in ${PASSES:-stats FETCH_SIZE WRITE_SIZE}; do
        d=/tmp/ab_${name}_$pass; rm -rf $d
        if [ $pass = stats ]; then args="--kernel-trace --stats"; else args="--pmc $pass"; fi
        (cd /tmp && env $envs timeout 300 rocprofv3 $args --output-format csv -d $d -o s -- \
            python "$REPO/tools/sift_time.py" 0.4 > $d.log 2>&1)
        python "$REPO/tools/prof_summary.py" $d "$OUT/r4_sift_ab_${name}_$pass.txt" > /dev/null
    done
    python "$REPO/tools/sift_ab_report.py" "$name" | tee "$OUT/r4_sift_ab_${name}.txt"
    tail -n 2 /tmp/ab_${name}_stats.log
done
