#!/bin/bash
# After a gpurun call of tools/collect_evidence.sh: copy the judged summaries from gpurun_out/
# (scratch, merged back by gpurun) into profiles/ (tracked).   bash tools/pull_evidence.sh [tag]
TAG="${1:-r6}"
cd "$(dirname "$0")/.."
for f in knn2sym_pmc_fetch.txt knn2sym_pmc_write.txt knn2sym_pmc_sq.txt knn2sym_traffic.json \
         aux_pmc_fetch.txt aux_pmc_write.txt ba_sift_traffic.json kernel_stats.txt \
         sift_kernel_stats.txt sift_time.txt gpu_tests.txt bench_latest.json bench_under_rocprof.json \
         lsmr_gaps.txt fm_dense.txt detect_rate_final.txt fm_config2_run.txt e2e_full_512.json link_passes_512.txt; do
    [ -f "gpurun_out/${TAG}_$f" ] && cp "gpurun_out/${TAG}_$f" "profiles/${TAG}_$f" && echo "profiles/${TAG}_$f"
done
