#!/usr/bin/env python3
"""End-to-end rate of Image.detect_features() on 20 MP JPEGs: decode + CLAHE/resize + SIFT +
the reference's cache files, (a) one image after the other with synchronous writes -- what a
straight port of scripts/lib/image.py:287-350 does -- and (b) with imageanalysis_amd.cacheio:
decode prefetched on worker threads, cache files written in the background.

    python tools/detect_rate.py [n_images]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import cacheio, image as iimg, synth  # noqa: E402
from imageanalysis_amd._deps import getNode  # noqa: E402
from imageanalysis_amd.hostlib import camera  # noqa: E402


def main():
    from PIL import Image as PILImage
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if args else 12
    tmp = tempfile.mkdtemp(prefix='iamx_det_')
    os.makedirs(os.path.join(tmp, 'images'))
    getNode('/config/directories', True).setString('project_dir', tmp)
    getNode('/config/detector', True).setString('detector', 'SIFT')
    camera.set_image_params(5472, 3648)
    t0 = time.time()
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=12) as pool:           # (the encoder releases the interpreter)
        futs = []
        for k in range(n):
            bgr = synth.make_survey_image(seed=k).cpu().numpy()
            futs.append(pool.submit(PILImage.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save,
                                    os.path.join(tmp, 'images', 'D%04d.JPG' % k), quality=92))
            while len(futs) > 24:
                futs.pop(0).result()
        for f in futs:
            f.result()
    print('%d synthetic 5472x3648 JPEGs written in %.1f s' % (n, time.time() - t0))
    if '--feat-zlib' in sys.argv:
        iimg.FEAT_GZIP_STRATEGY = 1                            # round 4's .feat members: zlib level 4, Z_FILTERED
        print('.feat through zlib (level %d, Z_FILTERED)' % iimg.FEAT_GZIP_LEVEL)

    def project(tag):
        an = os.path.join(tmp, 'ImageAnalysis_' + tag)
        os.makedirs(os.path.join(an, 'cache'))
        os.makedirs(os.path.join(an, 'meta'))
        return [iimg.Image(an, 'D%04d' % k) for k in range(n)]

    warm = project('warm')[0]
    warm.detect_features(0.4)                               # first-touch costs (library, allocator)
    cacheio.wait()

    # (a) serial, synchronous cache writes
    if '--no-serial' not in sys.argv:
        imgs = project('serial')
        iimg.ASYNC_CACHE_WRITES = False
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for im in imgs:
            im.detect_features(0.4)
        ta = time.perf_counter() - t0
        nk = sum(len(im.kp_list) for im in imgs) / float(n)
        print('serial    : %.2f s = %.2f images/s (%.0f keypoints/image)' % (ta, n / ta, nk))

    # (b) prefetch + background writes; the files must be complete when the clock stops
    imgs = project('overlap')
    iimg.ASYNC_CACHE_WRITES = True
    t0 = time.perf_counter()
    pf = iimg.prefetch(imgs, scale=0.4)
    for im in imgs:
        im.detect_features(0.4)
    t_det = time.perf_counter() - t0
    cacheio.wait()
    tb = time.perf_counter() - t0
    pf.close()
    print('overlapped: %.2f s = %.2f images/s (detector loop done after %.2f s)' % (tb, n / tb, t_det))

    # (b2) the same without the reference's float32 .desc (uint8 sidecar only: a cache only this
    # package reads)
    imgs2 = project('sidecar_only')
    iimg.WRITE_REFERENCE_DESC = False
    prof = None
    if '--profile' in sys.argv:
        import cProfile
        prof = cProfile.Profile()
    t0 = time.perf_counter()
    pf = iimg.prefetch(imgs2, scale=0.4)
    if prof:
        prof.enable()
    for im in imgs2:
        im.detect_features(0.4)
    if prof:
        prof.disable()
        import pstats
        pstats.Stats(prof).sort_stats('cumulative').print_stats(30)
    t_det = time.perf_counter() - t0
    cacheio.wait()
    tb2 = time.perf_counter() - t0
    pf.close()
    iimg.WRITE_REFERENCE_DESC = True
    print('overlapped, no float32 .desc: %.2f s = %.2f images/s (detector loop done after %.2f s)'
          % (tb2, n / tb2, t_det))

    # (c) reload from the cache just written (what a second run of the pipeline does)
    for im in imgs:
        im.kp_list = im.des_list = None
    t0 = time.perf_counter()
    pf = iimg.prefetch(imgs, scale=0.4)
    for im in imgs:
        im.detect_features(0.4)
    tc = time.perf_counter() - t0
    pf.close()
    print('cache load: %.2f s = %.2f images/s' % (tc, n / tc))


if __name__ == '__main__':
    main()
