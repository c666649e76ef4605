"""GPU: the device half of the split JPEG decoder (csrc/jpeg.hip: dequantisation, islow IDCT, fancy
chroma upsampling, YCbCr -> BGR) against libjpeg-turbo itself (Pillow -- the decoder OpenCV's
cv2.imread of scripts/lib/image.py:99-104 uses as well): bit-identical pixels."""
import numpy as np
import pytest

from test_jpeg import CASES, encode, pillow_bgr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,sub,quality,extra', CASES)
def test_device_reconstruction_equals_libjpeg(shape, sub, quality, extra):
    from imageanalysis_amd import kernels
    data = encode(shape, sub, quality, extra, seed=quality)
    got = kernels.jpeg_decode(data)
    assert got is not None
    want = pillow_bgr(data)
    got = got.cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize('sub', [1, 2])
def test_full_frame_20mp(sub):
    """a 5472 x 3648 frame (the survey camera's): every pixel equal, and the frame that feeds
    the detector is the same whether it comes from the split decoder or from Pillow"""
    import io
    import torch
    from PIL import Image
    from imageanalysis_amd import kernels, synth
    img = synth.make_survey_image(seed=3).cpu().numpy()
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(buf, 'JPEG', quality=93, subsampling=sub)
    data = buf.getvalue()
    got = kernels.jpeg_decode(data)
    want = pillow_bgr(data)
    assert np.array_equal(got.cpu().numpy(), want)
    a = kernels.equalize_resize(got, 0.4)
    b = kernels.equalize_resize(np.ascontiguousarray(want), 0.4)
    assert torch.equal(a, b)


def test_unsupported_falls_back_to_none():
    import io
    from PIL import Image
    from imageanalysis_amd import kernels
    from test_jpeg import scene
    buf = io.BytesIO()
    Image.fromarray(scene(64, 64, 1)).save(buf, 'JPEG', quality=90, progressive=True)
    assert kernels.jpeg_decode(buf.getvalue()) is None


def test_detect_features_uses_the_split_decoder(tmp_path):
    """Image.detect_features on a JPEG file: same keypoints / descriptors with the device decode
    (default) and with the host decode"""
    from PIL import Image as PILImage
    from imageanalysis_amd import image as iimg, matcher, synth
    from imageanalysis_amd._deps import getNode
    from imageanalysis_amd.hostlib import camera
    from test_sift_gpu import texture
    proj = tmp_path / 'p'
    (proj / 'images').mkdir(parents=True)
    an = proj / 'ImageAnalysis'
    (an / 'cache').mkdir(parents=True)
    (an / 'meta').mkdir()
    tex = texture(480, 640, 4)
    PILImage.fromarray(np.ascontiguousarray(tex[:, :, ::-1])).save(str(proj / 'images' / 'A.JPG'),
                                                                   quality=92, subsampling=1)
    getNode('/config/directories', True).setString('project_dir', str(proj))
    matcher.detector_node.setString('detector', 'SIFT')
    camera.set_image_params(640, 480)
    out = []
    for dev_jpeg in (True, False):
        iimg.USE_DEVICE_JPEG = dev_jpeg
        im = iimg.Image(str(an), 'A')
        im.detect_features(1.0, use_cache=False)
        iimg.cacheio.wait()
        out.append((im.kp_list.xy().copy(), np.asarray(im.des_list).copy()))
    iimg.USE_DEVICE_JPEG = True
    assert len(out[0][0]) > 500
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
