#!/bin/bash
# Counter passes on the SIFT descriptor kernel (tools/sift_time.py, three detections), one run per
# counter set and setting of IAMX_DESC_FORM:   bash tools/sift_desc_pmc.sh 0 25   -> gpurun_out/sift_desc_pmc_<form>.txt
OUT="$PWD/gpurun_out"; REPO="$PWD"; mkdir -p "$OUT"; export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
      "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"
      )   # (a third set with TCC_HIT_sum / TCC_MISS_sum / FETCH_SIZE ran for minutes and returned nothing on this pool)
for form in "$@"; do
    : > "$OUT/sift_desc_pmc_$form.txt"
    k=0
    for set in "${SETS[@]}"; do
        d=/tmp/dp_${form}_$k; rm -rf $d
        (cd /tmp && IAMX_DESC_FORM=$form timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -o s -- \
            python "$REPO/tools/sift_time.py" 0.4 > $d.log 2>&1)
        python "$REPO/tools/prof_summary.py" $d /tmp/dp_sum.txt > /dev/null
        python - /tmp/dp_sum.txt >> "$OUT/sift_desc_pmc_$form.txt" <<'P'
import sys
show = False
for line in open(sys.argv[1]):
    if line.strip().startswith('kernel'):
        show = 'descriptor_kernel' in line
    if show:
        print(line.rstrip()[:200])
P
        k=$((k+1))
    done
    echo "== form $form"; cat "$OUT/sift_desc_pmc_$form.txt"
done
