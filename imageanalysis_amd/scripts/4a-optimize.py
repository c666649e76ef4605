#!/usr/bin/env python3
"""CLI twin of process.py step 4 on the MI355X path (reference: scripts/4a-optimize.py).
Run from the reference's scripts/ directory: python <repo>/imageanalysis_amd/scripts/4a-optimize.py PROJECT
"""
import argparse
import os
import pickle

from lib import camera, groups, project

from imageanalysis_amd import optimizer

ap = argparse.ArgumentParser(description='Sparse bundle adjustment on MI355X.')
ap.add_argument('project', help='project directory')
ap.add_argument('--group', type=int, default=0, help='group number')
ap.add_argument('--refine', action='store_true', help='refine a previous optimization.')
ap.add_argument('--cam-calibration', action='store_true',
                help='include camera calibration in the optimization.')
ap.add_argument('--solver', default='device', choices=['device', 'device-lsmr', 'scipy'],
                help="'device' (default, like Optimizer.solver): GPU-resident TRF, Schur-complement subproblem solver; 'device-lsmr': the same with SciPy's LSMR formulation; 'scipy': SciPy TRF fed with the device residual/Jacobian")
args = ap.parse_args()

proj = project.ProjectMgr(args.project)
proj.load_images_info()

source_file = os.path.join(proj.analysis_dir, 'matches_grouped')
print('Match file:', source_file)
matches = pickle.load(open(source_file, "rb"))
print('Match features:', len(matches))
group_list = groups.load(proj.analysis_dir)

opt = optimizer.Optimizer(args.project)
opt.solver = args.solver
opt.setup(proj, group_list, args.group, matches, optimized=args.refine,
          cam_calib=args.cam_calibration)
cameras, features, cam_index_map, feat_index_map, fx_opt, fy_opt, cu_opt, cv_opt, distCoeffs_opt \
    = opt.run()
opt.update_camera_poses(proj)

camera.set_K(fx_opt, fy_opt, cu_opt, cv_opt, optimized=True)
camera.set_dist_coeffs(distCoeffs_opt.tolist(), optimized=True)
proj.save()

opt.refit(proj, matches, group_list, args.group)
print('Updating matches file:', len(matches), 'features')
pickle.dump(matches, open(source_file, 'wb'))
