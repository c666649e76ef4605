"""CPU: the host I/O runtime around the detector (imageanalysis_amd/cacheio.py) -- the cache
files stay readable by the reference's loaders (scripts/lib/image.py:140-180: gzip.open +
pickle.load / np.load) and hold byte-identical payloads."""
import gzip
import io
import os
import pickle
import threading
import time

import numpy as np
import pytest

from imageanalysis_amd import cacheio, image as iimg
from imageanalysis_amd._deps import getNode


def _reference_style_read(feat_path, desc_path):
    with gzip.open(feat_path, 'rb') as fp:                   # image.py:143-145
        feats = pickle.load(fp)
    with gzip.open(desc_path, 'rb') as fp:                   # image.py:166-168
        des = np.load(fp)
    return feats, des


def test_multi_member_stream_is_a_plain_gzip_file(tmp_path):
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, 3 * cacheio.MEMBER_BYTES + 12345, dtype=np.uint8).tobytes()
    blob = cacheio.gzip_members(raw)
    assert blob[:2] == b'\x1f\x8b' and blob.count(b'\x1f\x8b\x08') >= 4   # several members
    assert gzip.decompress(blob) == raw
    p = tmp_path / 'x.gz'
    cacheio.write_gzip(str(p), raw)
    cacheio.wait(str(p))
    with gzip.open(str(p), 'rb') as fp:
        assert fp.read() == raw
    small = b'abc' * 100
    assert gzip.decompress(cacheio.gzip_members(small)) == small
    assert cacheio.gzip_members(b'') and gzip.decompress(cacheio.gzip_members(b'')) == b''


def test_feature_cache_roundtrip_and_reference_reader(tmp_path):
    rng = np.random.default_rng(1)
    getNode('/config/directories', True).setString('project_dir', str(tmp_path))
    an = tmp_path / 'ImageAnalysis'
    (an / 'cache').mkdir(parents=True)
    (an / 'meta').mkdir()
    im = iimg.Image(str(an), 'A0001')
    n = 20000                                               # > 1 MiB of descriptors: many members
    xy = rng.uniform(0, 5000, (n, 2)).astype(np.float32)
    im.kp_list = [iimg.make_keypoint(x, y, 3.5, 10.0, 0.02, 65793 + k % 3)
                  for k, (x, y) in enumerate(xy.tolist())]
    im.des_list = rng.integers(0, 256, (n, 128)).astype(np.float32)
    want_feat = [(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id) for kp in im.kp_list]
    want_des = im.des_list.copy()
    im.save_features()
    im.save_descriptors()
    im.des_list = None                                      # find_matches' flush does this
    im.kp_list = None
    # a reader in the same process waits for the background write ...
    assert im.load_features() and im.load_descriptors()
    assert np.array_equal(im.des_list, want_des) and im.des_list.dtype == np.float32
    assert [(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id)
            for kp in im.kp_list] == want_feat
    # ... and the files are what the reference's own loader expects (the descriptors above came
    # from the uint8 sidecar; the 25 MB float32 .desc may still be in the background writer)
    cacheio.wait()
    feats, des = _reference_style_read(im.features_file, im.desc_file)
    assert feats == want_feat and np.array_equal(des, want_des)
    # decompressed payloads are byte-identical to what the reference's writer produces
    buf = io.BytesIO()
    np.save(buf, want_des)
    with gzip.open(im.desc_file, 'rb') as fp:
        assert fp.read() == buf.getvalue()
    with gzip.open(im.features_file, 'rb') as fp:
        assert fp.read() == pickle.dumps(want_feat)
    assert not [f for f in os.listdir(str(an / 'cache')) if '.tmp' in f]


def test_write_errors_are_reported_not_raised(tmp_path, capsys):
    im = iimg.Image.__new__(iimg.Image)
    im.features_file = str(tmp_path / 'missing_dir' / 'x.feat')
    im.kp_list = [iimg.make_keypoint(1.0, 2.0, 3.0, 4.0, 0.5, 1)]
    im.save_features()
    cacheio.wait(im.features_file)
    assert 'save_features(): I/O error' in capsys.readouterr().out


def test_prefetch_window_and_order():
    started, lock = [], threading.Lock()

    def job(x):
        with lock:
            started.append(x)
        time.sleep(0.02)
        return x * x

    class Item(object):
        def __init__(self, v):
            self.v = v

    items = [Item(v) for v in range(10)]
    pf = cacheio.Prefetch(lambda it: job(it.v), items, depth=3)
    time.sleep(0.15)
    assert sorted(started) == [0, 1, 2]                     # never more than `depth` ahead
    assert pf.pending(items[0]) and not pf.pending(items[5])
    assert pf.take(items[0]) == 0                           # frees a slot: item 3 starts
    assert pf.take(items[7]) == 49                          # not scheduled yet: computed inline
    out = [pf.take(it) for it in items[1:7]]
    assert out == [1, 4, 9, 16, 25, 36]
    pf.close()


def test_prefetch_feeds_detect_features_from_cache(tmp_path):
    rng = np.random.default_rng(2)
    getNode('/config/directories', True).setString('project_dir', str(tmp_path))
    an = tmp_path / 'ImageAnalysis'
    (an / 'cache').mkdir(parents=True)
    (an / 'meta').mkdir()
    imgs = []
    for k in range(4):
        im = iimg.Image(str(an), 'B%04d' % k)
        im.kp_list = [iimg.make_keypoint(x, y, 2.0, 1.0, 0.1, 7) for x, y in rng.uniform(0, 99, (50, 2))]
        im.des_list = rng.integers(0, 256, (50, 128)).astype(np.float32)
        im.save_features()
        im.save_descriptors()
        im._want = im.des_list.copy()
        im.kp_list = im.des_list = None
        imgs.append(im)
    pf = iimg.prefetch(imgs, depth=2)
    for im in imgs:
        im.detect_features(0.4)                             # cache hit through the prefetched bytes
        assert np.array_equal(im.des_list, im._want) and len(im.kp_list) == 50
    pf.close()


@pytest.mark.parametrize('early_hook', [True, False])
def test_queued_files_survive_an_immediate_exit(tmp_path, early_hook):
    """a process that ends right after save_*() must still leave complete cache files: either
    the flush runs before concurrent.futures shuts down (threading hook) or, without that
    hook, the queued jobs compress their members themselves"""
    import subprocess
    import sys
    code = '''
import sys, threading
sys.path.insert(0, %r)
import concurrent.futures.thread
if not %r:
    del threading._register_atexit          # (after concurrent.futures took its own hook)
import numpy as np
import pytest
from imageanalysis_amd import cacheio
raw = np.random.default_rng(0).integers(0, 255, 6 << 20, dtype=np.uint8).tobytes()
for k in range(6):
    cacheio.write_gzip(%r + '/f%%d.gz' %% k, raw)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), early_hook, str(tmp_path))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert 'failed' not in r.stdout + r.stderr
    files = sorted(os.listdir(str(tmp_path)))
    assert files == ['f%d.gz' % k for k in range(6)]
    for f in files:
        with gzip.open(os.path.join(str(tmp_path), f), 'rb') as fp:
            assert len(fp.read()) == 6 << 20


def test_uint8_descriptor_sidecar(tmp_path):
    """<image>.desc.u8.npy: written beside the reference's .desc, preferred
    on reload, rejected when it cannot belong to the cached keypoints, never written for
    non-integer descriptors; the .desc stays what the reference's reader expects."""
    rng = np.random.default_rng(3)
    getNode('/config/directories', True).setString('project_dir', str(tmp_path))
    an = tmp_path / 'ImageAnalysis'
    (an / 'cache').mkdir(parents=True)
    (an / 'meta').mkdir()
    im = iimg.Image(str(an), 'S0001')
    n = 3000
    im.kp_list = [iimg.make_keypoint(float(k), 2.0 * k, 3.5, 10.0, 0.02, 65793) for k in range(n)]
    des = rng.integers(0, 256, (n, 128)).astype(np.float32)
    im.des_list = des.copy()
    im.save_features()
    im.save_descriptors()
    cacheio.wait()
    side = im.desc_file + '.u8.npy'
    assert os.path.exists(side) and os.path.getsize(side) < n * 128 + 256
    assert np.array_equal(np.load(side), des.astype(np.uint8))
    _feats, ref_des = _reference_style_read(im.features_file, im.desc_file)
    assert ref_des.dtype == np.float32 and np.array_equal(ref_des, des)
    # reload: the sidecar is used (make the .desc unreadable to prove it)
    with open(im.desc_file, 'wb') as f:
        f.write(b'not a gzip file')
    im.des_list = None
    assert im.load_descriptors() and im.des_list.dtype == np.float32
    assert np.array_equal(im.des_list, des)
    # a sidecar of another keypoint count is ignored
    np.save(side, des[:-5].astype(np.uint8))
    im.des_list = None
    assert not im.load_descriptors()
    # non-integer descriptors (another detector) get no sidecar
    im2 = iimg.Image(str(an), 'S0002')
    im2.kp_list = im.kp_list
    im2.des_list = rng.normal(size=(50, 128)).astype(np.float32)
    im2.save_descriptors()
    cacheio.wait()
    assert not os.path.exists(im2.desc_file + '.u8.npy') and os.path.exists(im2.desc_file)
    # reference-format file can be switched off for caches only this package reads
    iimg.WRITE_REFERENCE_DESC = False
    try:
        im3 = iimg.Image(str(an), 'S0003')
        im3.kp_list = im.kp_list
        im3.des_list = des.copy()
        im3.save_descriptors()
        cacheio.wait()
        assert not os.path.exists(im3.desc_file) and os.path.exists(im3.desc_file + '.u8.npy')
        im3.des_list = None
        assert im3.load_descriptors() and np.array_equal(im3.des_list, des)
    finally:
        iimg.WRITE_REFERENCE_DESC = True


def test_desc_file_from_uint8_is_the_reference_format(tmp_path):
    """iamx_gzip_f32_from_u8 (host code of libiamx): the .desc file written straight from the
    detector's uint8 descriptors decompresses to exactly the bytes np.save(float32) writes, for
    any content -- empty, constant, all 256 values, SIFT-like -- and the reference's reader
    (gzip.open + np.load, scripts/lib/image.py:166-177) loads it"""
    import gzip
    import io
    from imageanalysis_amd import image as iimg
    rng = np.random.default_rng(0)
    cases = [np.zeros((0, 128), np.uint8), np.zeros((3, 128), np.uint8), np.full((5, 128), 255, np.uint8),
             rng.integers(0, 256, (100, 128), dtype=np.uint8),
             np.arange(256, dtype=np.uint8).reshape(2, 128),
             (rng.gamma(0.6, 1, (3000, 128)) * 40).clip(0, 255).astype(np.uint8)]
    for u8 in cases:
        f32 = u8.astype(np.float32)
        buf = io.BytesIO()
        np.save(buf, f32)
        z = bytes(iimg._desc_gzip_from_u8(u8))
        assert gzip.decompress(z) == buf.getvalue()
        path = tmp_path / 'x.desc'
        path.write_bytes(z)
        with gzip.open(str(path), 'rb') as fp:
            back = np.load(fp)
        assert back.dtype == np.float32 and np.array_equal(back, f32)


def test_native_gzip_members_are_a_gzip_stream_of_the_buffers():
    """iamx_gzip_members (one call, zlib on its own threads): any gzip reader gets the buffers back"""
    rng = np.random.default_rng(5)
    text = bytes(rng.integers(97, 105, 3_000_000, dtype=np.uint8))
    arr = rng.integers(0, 4, (70000, 16)).astype(np.float32)
    cases = [([text], 1 << 20), ([text[:10], memoryview(arr.reshape(-1).view(np.uint8))], 384 << 10),
             ([b''], 1 << 20), ([b'x'], 1), ([text[:5000]], 4096), ([b'', text[:100], b''], 64)]
    for bufs, mb in cases:
        for level in (1, 6):
            members = cacheio.gzip_member_list(bufs if len(bufs) > 1 else bufs[0], level, mb)
            blob = b''.join(bytes(m) for m in members)
            want = b''.join(bytes(memoryview(b).cast('B')) for b in bufs)
            assert gzip.decompress(blob) == want
            if len(want) > 100000:
                assert len(blob) < len(want) // 2
    # the python members (interpreter shutdown path) give the same payload
    py = b''.join(cacheio._member(memoryview(text)[i:i + (1 << 20)], 6) for i in range(0, len(text), 1 << 20))
    assert gzip.decompress(py) == text


def test_native_float32_conversion_and_feat_records():
    import pickle
    from imageanalysis_amd.keypoints import KeyPointList
    rng = np.random.default_rng(6)
    for n in (0, 1, 77, 300000):
        u8 = rng.integers(0, 256, (n, 128), dtype=np.uint8)
        f = iimg._to_float32(u8)
        assert f.dtype == np.float32 and f.shape == u8.shape and np.array_equal(f, u8.astype(np.float32))
    n = 5000
    cols = [rng.uniform(0, 5000, n), rng.uniform(0, 3000, n), rng.uniform(1, 40, n), rng.uniform(0, 360, n),
            rng.uniform(0, 0.2, n)]
    octv = rng.integers(-(1 << 20), 1 << 24, n).astype(np.int32)
    kl = KeyPointList(*cols, octv)
    f32 = [np.asarray(c, np.float64).astype(np.float32) for c in cols]
    want = [((float(f32[0][i]), float(f32[1][i])), float(f32[2][i]), float(f32[3][i]), float(f32[4][i]),
             int(octv[i]), -1) for i in range(n)]
    blob = kl.feat_bytes()
    assert isinstance(blob, bytes) and pickle.loads(blob) == want
    assert bytes(kl.feat_bytes(as_view=True)) == blob


def test_record_deflate_members_are_plain_gzip_and_small():
    """iamx_gzip_records (the .feat writer since round 5): DEFLATE with matches one record back
    only.  Any gzip reader inflates the members to the payload -- python's gzip / zlib here, the
    reference's `pickle.load(gzip.open(...))` (scripts/lib/image.py:140-150) on a real .feat -- and
    the file is no larger than zlib's level 4 / Z_FILTERED output it replaces."""
    import pickle
    import zlib
    from imageanalysis_amd.keypoints import KeyPointList
    rng = np.random.default_rng(12)
    n = 30000
    kl = KeyPointList(rng.uniform(0, 5472, n), rng.uniform(0, 3648, n),
                      2.0 * 1.6 * 2 ** rng.uniform(0, 5, n), rng.uniform(0, 360, n), rng.uniform(0.02, 0.1, n),
                      (rng.integers(0, 6, n) | (rng.integers(1, 4, n) << 8) | (rng.integers(0, 256, n) << 16)))
    payload = bytes(kl.feat_bytes())
    members = cacheio.gzip_member_list(payload, 4, 384 << 10, ('records', 58))
    blob = b''.join(bytes(m) for m in members)
    assert gzip.decompress(blob) == payload
    assert pickle.load(gzip.GzipFile(fileobj=io.BytesIO(blob))) == pickle.loads(payload)
    zl = b''.join(bytes(m) for m in cacheio.gzip_member_list(payload, 4, 384 << 10, 1))
    assert len(blob) <= len(zl)
    # the first member alone is a complete gzip stream of the first 384 KiB
    d = zlib.decompressobj(47)
    assert d.decompress(blob) == payload[:384 << 10] and d.eof
    # degenerate inputs, record widths, member sizes
    cases = [b'', b'a', b'abc' * 5, bytes(range(256)) * 3, b'\x00' * 100000,
             rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), payload[:57], payload[:59]]
    for pl in cases:
        for mb in (1000, 384 << 10):
            for rec in (58, 1, 7, 32768):
                out = b''.join(bytes(m) for m in cacheio.gzip_member_list(pl, 4, mb, ('records', rec)))
                assert gzip.decompress(out) == pl, (len(pl), mb, rec)
