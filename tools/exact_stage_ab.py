"""A/B of the exact stage (IAMX_EXACT_PRUNE=0: the unpruned scan): the dense-overlap workload of
bench.py at two image sizes; prints the sweep and the filter + exact stage times."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

import bench

dev = torch.device('cuda:0')
for n_img, rows in ((12, 16384), (8, 36864)):
    r = bench.dense_overlap_bench(dev, oracle_pairs=2, n_img=n_img, rows=rows)
    print("NARROW=%s PRUNE=%s %2d x %5d rows: sweep %.3f ms, filter+exact %.3f ms (%.0f TFLOP/s), candidate share %.3f, "
          "one-direction finish %.3f ms" % (os.environ.get('IAMX_EXACT_NARROW', '1'), os.environ.get('IAMX_EXACT_PRUNE', '1'), n_img, rows,
                                            r['sweep_ms'], r['filter_and_exact_ms'], r['exact_stage_tflops'],
                                            r['candidate_share'], r['one_direction_form']['filter_and_finish_ms']),
          flush=True)
