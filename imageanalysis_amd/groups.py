"""MI355X-path stand-in for the reference's scripts/lib/groups.py: connected groups of images
grown from the feature chains (scripts/process.py:343-344), same `compute / save / load`.

compute() keeps the reference's order-dependent greedy procedure (groups.py:25-133): per group
level a seed chain, then sweeps over all chains until none can be added; an image belongs to
the group once `min_connections` of its features are placed.  The sweeps over the chains run as
one native host routine on flat arrays (csrc/host_cleanup.hip iamx_group_level); the python
here does the set bookkeeping exactly like the reference so that the order of the names inside
a group (iteration order of a python set) is the same."""
import json
import os
import sys
from math import sqrt

import numpy as np

from . import _deps

min_group = 7                      # groups.py:14-16
min_connections = 25
max_wanted = 250                   # overridden in compute()


def _log(*a):
    _deps.logger().log(*a)


def my_add(placed_matches, matches, group_level, i):
    """groups.py:18-22, kept for callers that drive the grouping by hand: chain i joins
    `group_level`, and every image it is seen in has one more placed feature.  (compute() does
    this for whole sweeps inside iamx_group_level.)"""
    for obs in matches[i][2:]:
        placed_matches[obs[0]] += 1
    matches[i][1] = group_level


def compute(image_list, matches):
    """notice: matches are assumed sorted longest chain first (link_matches does that)."""
    from ._lib import c_void_p, lib
    _log("Start of grouping algorithm...")
    matcher_node = _deps.getNode('/config/matcher', True)
    min_chain_len = matcher_node.getInt("min_chain_len")
    if min_chain_len == 0:
        min_chain_len = 3
    _log("/config/matcher/min_chain_len:", min_chain_len)
    use_single_pairs = (min_chain_len == 2)
    n_img = len(image_list)
    wanted = int(8000 / sqrt(n_img))
    if wanted < 200:
        wanted = 200
    _log("max features desired per image:", wanted)
    print("Notice: I should really work on this formula ...")

    n = len(matches)
    from .match_cleanup import Chains
    fast = isinstance(matches, Chains) and matches.untouched()
    if fast:
        ptr, img = matches.ptr, matches.img           # the arrays link_matches() linked
    else:
        ptr = np.zeros(n + 1, np.int64)
        import gc
        gc_was = gc.isenabled()
        gc.disable()           # (millions of small lists alive: see match_cleanup.triangulate_smart)
        try:
            if n:
                np.cumsum([len(m) - 2 for m in matches], out=ptr[1:])
            img = np.fromiter((p[0] for m in matches for p in m[2:]), np.int32, int(ptr[-1]))
        finally:
            if gc_was:
                gc.enable()
    level = np.full(n, -1, np.int32)
    placed_images = set()
    placed_flag = np.zeros(max(n_img, 1), np.uint8)
    placed_matches = np.zeros(max(n_img, 1), np.int32)
    P = lambda a: c_void_p(a.ctypes.data)
    groups = []
    done = False
    while not done:
        group_level = len(groups)
        _log("Start of new group level:", group_level)
        seed = int(lib().iamx_group_level(P(img), P(ptr), n, max(n_img, 1), P(level), P(placed_flag),
                                          group_level, 1 if use_single_pairs else 0, wanted,
                                          min_connections, P(placed_matches)))
        if seed < -1:
            raise RuntimeError("iamx_group_level failed (%d): %s"
                               % (seed, (lib().iamx_last_error() or b'?').decode()))
        if seed == -1:
            break
        _log('Seeding group with:', image_list[int(img[ptr[seed] + 1])].name)
        group_images = set()
        for i in range(n_img):
            if placed_matches[i] >= min_connections:
                group_images.add(i)
        group_list = []
        for i in list(group_images):
            placed_images.add(i)
            placed_flag[i] = 1
            group_list.append(image_list[i].name)
        if len(group_images) >= min_group:
            _log(group_list)
            groups.append(group_list)
        if len(group_images) < 3:
            done = True
    if fast and matches.untouched():
        matches.group[:] = level
        return groups
    for m, lv in zip(matches, level.tolist()):
        m[1] = lv
    return groups


def save(path, groups):
    file = os.path.join(path, 'groups.json')
    try:
        with open(file, 'w') as fd:
            json.dump(groups, fd, indent=4, sort_keys=True)
    except Exception:                    # noqa: BLE001 (the reference logs and carries on)
        _log('{}: error saving file: {}'.format(file, str(sys.exc_info()[1])))


def load(path):
    file = os.path.join(path, 'groups.json')
    try:
        with open(file, 'r') as fd:
            groups = json.load(fd)
    except Exception:                    # noqa: BLE001
        _log('{}: error loading file: {}'.format(file, str(sys.exc_info()[1])))
        groups = []
    return groups
