#!/bin/bash
# SQ counters of the exact stage on the dense-overlap workload (two passes, 8 counters each)
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
    --output-format csv -d /tmp/x1 -o b -- python $REPO/tools/exact_pmc_run.py > /tmp/x1.log 2>&1
tail -2 /tmp/x1.log
python $REPO/tools/prof_summary.py /tmp/x1 $REPO/gpurun_out/r5_exact_pmc_1.txt > /dev/null
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM \
    SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d /tmp/x2 -o b -- python $REPO/tools/exact_pmc_run.py > /tmp/x2.log 2>&1
tail -2 /tmp/x2.log
python $REPO/tools/prof_summary.py /tmp/x2 $REPO/gpurun_out/r5_exact_pmc_2.txt > /dev/null
grep -A 10 "symexact_wg\|knn2sym_kernel" $REPO/gpurun_out/r5_exact_pmc_1.txt | head -40
grep -A 10 "symexact_wg\|knn2sym_kernel" $REPO/gpurun_out/r5_exact_pmc_2.txt | head -40
