#!/usr/bin/env python3
"""CLI twin of process.py step 3a on the MI355X path (reference: scripts/3a-matching.py, whose
find_matches call at :113 is stale -- this uses the scripts/process.py:290-292 call form).

Run from the reference's scripts/ directory (it provides lib.project / lib.camera / lib.smart):
    python -m torch.distributed.run --nproc-per-node 8 <repo>/imageanalysis_amd/scripts/3a-matching.py PROJECT
or single GPU:  python <repo>/imageanalysis_amd/scripts/3a-matching.py PROJECT
"""
import argparse
import os

from props import getNode

from lib import camera, project, smart, srtm
from lib.logger import log

from imageanalysis_amd import matcher

ap = argparse.ArgumentParser(description='Feature matching on MI355X.')
ap.add_argument('project', help='project directory')
ap.add_argument('--scale', type=float, default=0.4, help='image scale for feature detection')
ap.add_argument('--detector', default='SIFT', choices=['SIFT'])
ap.add_argument('--match-ratio', default=0.75, type=float)
ap.add_argument('--min-pairs', default=25, type=int)
ap.add_argument('--min-dist', default=0, type=float)
ap.add_argument('--max-dist', default=75, type=float)
ap.add_argument('--filter', default='gms', choices=['gms', 'homography', 'fundamental', 'essential', 'none'])
ap.add_argument('--min-chain-length', type=int, default=3)
ap.add_argument('--schedule', default='neighbours', choices=['neighbours', 'distance', 'all-pairs'],
                help="pair schedule: the reference at HEAD matches sequential neighbours only")
args = ap.parse_args()

if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group('nccl')

proj = project.ProjectMgr(args.project)
proj.load_images_info()
proj.load_match_pairs()

node = getNode('/config/detector', True)
node.setString('detector', args.detector)
node.setString('scale', args.scale)
node = getNode('/config/matcher', True)
node.setFloat('match_ratio', args.match_ratio)
node.setString('filter', args.filter)
node.setInt('min_pairs', args.min_pairs)
node.setFloat('min_dist', args.min_dist)
node.setFloat('max_dist', args.max_dist)
node.setInt('min_chain_len', args.min_chain_length)
node.setString('schedule', args.schedule)
proj.save()

ref_node = getNode('/config/ned_reference', True)
ref = [ref_node.getFloat('lat_deg'), ref_node.getFloat('lon_deg'), ref_node.getFloat('alt_m')]
log("NED reference location:", ref)
srtm.initialize(ref, 6000, 6000, 30)
smart.load(proj.analysis_dir)
smart.update_srtm_elevations(proj)
smart.set_yaw_error_estimates(proj)
proj.save_images_info()

K = camera.get_K()
matcher.configure()
matcher.find_matches(proj, K, strategy='traditional', transform=args.filter, sort=True,
                     review=False)

n = sum(image.num_features for image in proj.image_list)
log("Average # of features per image found = %.0f" % (n / max(len(proj.image_list), 1)))
