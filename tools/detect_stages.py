#!/usr/bin/env python3
"""Where a fresh detect_features() loop spends its time when the prefetch workers run the whole
detection: wall time per stage summed over the worker threads (so stages overlap) and what the
calling loop itself waits for.      python tools/detect_stages.py [n_images]"""
import collections
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import cacheio, image as iimg, kernels, synth  # noqa: E402
from imageanalysis_amd._deps import getNode  # noqa: E402
from imageanalysis_amd.hostlib import camera  # noqa: E402

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
cpu = collections.defaultdict(float)
lock = threading.Lock()


def timed(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name

    def wrap(*a, **k):
        t = time.perf_counter()
        c = time.thread_time()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t
            dc = time.thread_time() - c
            with lock:
                acc[label] += dt
                cpu[label] += dc
                cnt[label] += 1
    setattr(mod, name, wrap)


def main():
    from PIL import Image as PILImage
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    if len(sys.argv) > 2:
        iimg.PREFETCH_DEPTH = int(sys.argv[2])
    if len(sys.argv) > 3:
        kernels.DETECT_SLOTS = int(sys.argv[3])
    tmp = tempfile.mkdtemp(prefix='iamx_stage_')
    os.makedirs(os.path.join(tmp, 'images'))
    getNode('/config/directories', True).setString('project_dir', tmp)
    getNode('/config/detector', True).setString('detector', 'SIFT')
    camera.set_image_params(5472, 3648)
    for k in range(n):
        bgr = synth.make_survey_image(seed=k).cpu().numpy()
        PILImage.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(
            os.path.join(tmp, 'images', 'D%04d.JPG' % k), quality=92)

    def project(tag):
        an = os.path.join(tmp, 'ImageAnalysis_' + tag)
        os.makedirs(os.path.join(an, 'cache'))
        os.makedirs(os.path.join(an, 'meta'))
        return [iimg.Image(an, 'D%04d' % k) for k in range(n)]

    warm = project('warm')
    pf = iimg.prefetch(warm[:32], scale=0.4)
    for im in warm[:32]:
        im.detect_features(0.4)
    pf.close()
    cacheio.wait()

    timed(kernels, 'jpeg_host_decode')
    timed(kernels, 'jpeg_reconstruct')
    timed(kernels, 'equalize_resize')
    timed(kernels, 'sift_detect')
    timed(iimg, 'features_from_bgr')
    timed(iimg, '_to_float32')
    timed(iimg, '_desc_gzip_from_u8')
    timed(iimg, '_prefetch_job')
    timed(cacheio, '_write_job')
    timed(cacheio, '_member')
    ent = kernels.detector_slot.__enter__

    def enter(self):
        t = time.perf_counter()
        r = ent(self)
        with lock:
            acc['wait for a detector slot'] += time.perf_counter() - t
            cnt['wait for a detector slot'] += 1
        return r
    kernels.detector_slot.__enter__ = enter

    imgs = project('run')
    torch.cuda.synchronize()
    # GIL pressure: a thread that sleeps 1 ms has to take the GIL back when it wakes -- how late is it?
    # and where are the python threads (innermost python frame of every thread, sampled by it)
    probe = {'late': [], 'stop': False, 'where': collections.Counter()}

    def gil_probe():
        me = threading.get_ident()
        while not probe['stop']:
            t = time.perf_counter()
            time.sleep(0.001)
            probe['late'].append(time.perf_counter() - t - 0.001)
            for tid, fr in sys._current_frames().items():
                if tid != me:
                    probe['where']['%s:%d %s' % (os.path.basename(fr.f_code.co_filename), fr.f_lineno,
                                                  fr.f_code.co_name)] += 1
    th = threading.Thread(target=gil_probe, daemon=True)
    th.start()
    c0 = os.times()
    t0 = time.perf_counter()
    pf = iimg.prefetch(imgs, scale=0.4)
    t_take = 0.0
    for im in imgs:
        t = time.perf_counter()
        im.detect_features(0.4)
        t_take += time.perf_counter() - t
    t_loop = time.perf_counter() - t0
    cacheio.wait()
    t_all = time.perf_counter() - t0
    pf.close()
    print('%d images: loop %.2f s (%.1f ms / image), files complete after %.2f s = %.1f images/s'
          % (n, t_loop, t_loop / n * 1e3, t_all, n / t_all))
    c1 = os.times()
    probe['stop'] = True
    th.join()
    late = np.array(probe['late']) * 1e3
    print('GIL probe: %d wake-ups, late by mean %.2f ms, median %.2f, p90 %.2f, max %.1f'
          % (len(late), late.mean(), np.median(late), np.percentile(late, 90), late.max()))
    for k, v in probe['where'].most_common(25):
        print('    %6d  %s' % (v, k))
    print('slots %d, prefetch depth %d, cores %d; process CPU: user %.1f s, system %.1f s in %.2f s wall'
          % (kernels.DETECT_SLOTS, iimg.PREFETCH_DEPTH, os.cpu_count(), c1.user - c0.user,
             c1.system - c0.system, t_all))
    for k in sorted(acc, key=lambda k: -acc[k]):
        print('  %-28s %6d calls  %8.1f ms / image wall (sum over threads)  %7.2f ms / call  %7.1f ms CPU / image (calling thread)'
              % (k, cnt[k], acc[k] / n * 1e3, acc[k] / cnt[k] * 1e3, cpu[k] / n * 1e3))


if __name__ == '__main__':
    main()
