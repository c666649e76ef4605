"""Synthetic survey generators for bench.py and tests: numpy only, data
generation -- never on a timed path."""
import numpy as np

from .hostlib import transforms as tf

W_PX, H_PX = 5472, 3648
FX = 3666.6665
K_FC6310S = np.array([[FX, 0.0, 2736.0], [0.0, FX, 1824.0], [0.0, 0.0, 1.0]])
BODY2CAM = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])


def _rotations(quats):
    """R (ned -> camera) for [C,4] w,x,y,z quaternions."""
    q = quats / np.linalg.norm(quats, axis=1, keepdims=True)
    w, x, y, z = q.T
    B = np.empty((len(q), 3, 3))
    B[:, 0, 0] = 1 - 2 * (y * y + z * z); B[:, 0, 1] = 2 * (x * y - z * w); B[:, 0, 2] = 2 * (x * z + y * w)
    B[:, 1, 0] = 2 * (x * y + z * w); B[:, 1, 1] = 1 - 2 * (x * x + z * z); B[:, 1, 2] = 2 * (y * z - x * w)
    B[:, 2, 0] = 2 * (x * z - y * w); B[:, 2, 1] = 2 * (y * z + x * w); B[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return np.einsum('ij,ckj->cik', BODY2CAM, B)          # body2cam . body2ned^T


def project(cams, pts, cam_idx, pt_idx, K, dist):
    R = _rotations(cams[:, 3:7])[cam_idx]
    Xc = np.einsum('oij,oj->oi', R, pts[pt_idx] - cams[cam_idx, :3])
    x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], 1), Xc[:, 2]


def make_ba_problem(rows=38, cols=74, n_points=300000, n_obs=2000000, spacing=20.0, agl=100.0,
                    dist=(0.0, 0.0, 0.0, 0.0, 0.0), seed=42, cam_sigma=1.0, pt_sigma=2.0,
                    px_sigma=0.5):
    """BASELINE config 4: nadir cameras on a lawn-mower grid, ground points, noisy projections
    capped at n_obs, initial guess = truth + noise (cameras inside the +-3 m bounds).
    Returns dict(cams0 [C,7], pts0 [P,3], cam_idx, pt_idx (camera-major int32), uv [O,2], K, dist)."""
    rng = np.random.default_rng(seed)
    C = rows * cols
    r, c = np.divmod(np.arange(C), cols)
    c = np.where(r % 2 == 1, cols - 1 - c, c)
    ned = np.stack([r * spacing, c * spacing, np.full(C, -agl)], 1) + rng.normal(0, 0.3, (C, 3))
    yaw = np.where(r % 2 == 0, 0.0, 180.0) + rng.normal(0, 3.0, C)
    pitch = -90.0 + rng.normal(0, 2.0, C)
    roll = rng.normal(0, 2.0, C)
    d2r = np.pi / 180
    quat = np.array([tf.quaternion_from_euler(y * d2r, p * d2r, q * d2r, 'rzyx')
                     for y, p, q in zip(yaw, pitch, roll)])
    cams = np.hstack([ned, quat])
    K = K_FC6310S
    pts = np.stack([rng.uniform(-40, (rows - 1) * spacing + 40, n_points),
                    rng.uniform(-60, (cols - 1) * spacing + 60, n_points),
                    rng.normal(0, 2.0, n_points)], 1)
    # candidate cameras of a point: the grid cells within the footprint half extents
    half_r = int(np.ceil(0.5 * H_PX / FX * agl / spacing)) + 1
    half_c = int(np.ceil(0.5 * W_PX / FX * agl / spacing)) + 1
    pr = np.rint(pts[:, 0] / spacing).astype(np.int64)
    pc = np.rint(pts[:, 1] / spacing).astype(np.int64)
    cand_c, cand_p = [], []
    grid = -np.ones((rows, cols), np.int64)
    grid[r, c] = np.arange(C)
    for dr in range(-half_r, half_r + 1):           # headings are 0/180 deg: wide axis = east
        for dc in range(-half_c, half_c + 1):
            rr, cc = pr + dr, pc + dc
            ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
            idx = np.nonzero(ok)[0]
            cand_c.append(grid[rr[idx], cc[idx]])
            cand_p.append(idx)
    cam_idx = np.concatenate(cand_c)
    pt_idx = np.concatenate(cand_p)
    uv, z = project(cams, pts, cam_idx, pt_idx, K, dist)
    vis = (z > 1.0) & (uv[:, 0] >= 0) & (uv[:, 0] < W_PX) & (uv[:, 1] >= 0) & (uv[:, 1] < H_PX)
    cam_idx, pt_idx, uv = cam_idx[vis], pt_idx[vis], uv[vis]
    if len(cam_idx) > n_obs:
        keep = np.sort(rng.permutation(len(cam_idx))[:n_obs])
        cam_idx, pt_idx, uv = cam_idx[keep], pt_idx[keep], uv[keep]
    # drop points with < 3 observations (min_chain_len, optimizer.py:79) and renumber
    cnt = np.bincount(pt_idx, minlength=n_points)
    good = cnt >= 3
    sel = good[pt_idx]
    cam_idx, pt_idx, uv = cam_idx[sel], pt_idx[sel], uv[sel]
    remap = -np.ones(n_points, np.int64)
    remap[good] = np.arange(int(good.sum()))
    pt_idx = remap[pt_idx]
    pts = pts[good]
    order = np.lexsort((pt_idx, cam_idx))              # camera-major, points ascending
    cam_idx, pt_idx, uv = cam_idx[order], pt_idx[order], uv[order]
    uv = uv + rng.normal(0, px_sigma, uv.shape)
    cams0 = cams.copy()
    cams0[:, :3] += rng.normal(0, cam_sigma, (C, 3))
    cams0[:, 3:] += rng.normal(0, 0.005, (C, 4))
    pts0 = pts + rng.normal(0, pt_sigma, pts.shape)
    return dict(cams0=cams0, pts0=pts0, cams_true=cams, pts_true=pts,
                cam_idx=cam_idx.astype(np.int32), pt_idx=pt_idx.astype(np.int32), uv=uv,
                K=K, dist=np.asarray(dist, np.float64))


def make_survey_image(h=3648, w=5472, seed=0, device='cuda'):
    """SURVEY.md 8d "20 MP" SIFT input: seeded procedural fractal noise, BGR uint8 [h,w,3] on the
    device, textured enough for several 10^4 SIFT keypoints at scale 0.4."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    img = torch.zeros((1, 1, h, w), device=device)
    for s in (2, 4, 8, 16, 32, 64):
        n = torch.randn((1, 1, h // s + 2, w // s + 2), generator=g, device=device)
        up = torch.nn.functional.interpolate(n, scale_factor=s, mode='bilinear',
                                             align_corners=False)[:, :, :h, :w]
        img += up * s ** 0.7
    img = (img - img.min()) / (img.max() - img.min()) * 255
    img = img[0, 0]
    return torch.stack([img, img * 0.9 + 10, img * 0.8 + 20], 2).clamp(0, 255).to(torch.uint8).contiguous()


def make_survey_gray_big(h, w, seed=0, device='cuda', band=2048):
    """The grey channel of make_survey_image() for ground textures beyond 2^31 pixels (a survey
    of thousands of 20 MP frames at 3 cm: 53 k x 91 k for 2048 frames), where one
    interpolate() call over the whole plane is past the framework's 32-bit indexing: the same
    seeded noise grids, upsampled band by band with the bilinear weights of
    interpolate(align_corners=False) written out, accumulated in float32, normalised by the
    global extrema like the small form.  uint8 [h, w] on the device."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    acc = torch.zeros((h, w), device=device)
    xs = torch.arange(w, device=device, dtype=torch.float32)
    for s in (2, 4, 8, 16, 32, 64):
        gh, gw = h // s + 2, w // s + 2
        if gh * gw < (1 << 30):
            n = torch.randn((1, 1, gh, gw), generator=g, device=device)[0, 0]    # (= the small form's grid)
        else:
            n = torch.empty((gh, gw), device=device)
            step = max(1, (1 << 30) // gw)
            for a in range(0, gh, step):
                n[a:a + step].normal_(generator=g)
        sx = ((xs + 0.5) / s - 0.5).clamp_(min=0)
        x0 = sx.floor().long().clamp_(max=gw - 1)
        x1 = (x0 + 1).clamp_(max=gw - 1)
        wx = (sx - x0.to(sx.dtype))[None, :]
        for a in range(0, h, band):
            b = min(a + band, h)
            sy = ((torch.arange(a, b, device=device, dtype=torch.float32) + 0.5) / s - 0.5).clamp_(min=0)
            y0 = sy.floor().long().clamp_(max=gh - 1)
            y1 = (y0 + 1).clamp_(max=gh - 1)
            wy = (sy - y0.to(sy.dtype))[:, None]
            top, bot = n[y0], n[y1]                           # [band, gw]
            up = (top[:, x0] * (1 - wx) + top[:, x1] * wx) * (1 - wy) + \
                 (bot[:, x0] * (1 - wx) + bot[:, x1] * wx) * wy
            acc[a:b] += up * s ** 0.7
            del top, bot, up
        del n
    lo, hi = float(acc.min()), float(acc.max())
    out = torch.empty((h, w), dtype=torch.uint8, device=device)
    for a in range(0, h, band):
        b = min(a + band, h)
        out[a:b] = ((acc[a:b] - lo) / (hi - lo) * 255).clamp_(0, 255).to(torch.uint8)
    return out


def render_view_device_gray(tex, M, ned, w, h, gsd, origin):
    """render_view_device() over the single-channel texture of make_survey_gray_big(): the texels
    under the frame are cut out first (indices inside the cut-out fit 32 bits), the grey value is
    interpolated and the three channels derived from it as make_survey_image() does."""
    import torch
    dev = tex.device
    f64 = torch.float64
    u = torch.arange(w, device=dev, dtype=f64)[None, :]
    v = torch.arange(h, device=dev, dtype=f64)[:, None]
    Mt = [[float(M[i, j]) for j in range(3)] for i in range(3)]
    ray = [Mt[i][0] * u + Mt[i][1] * v + Mt[i][2] for i in range(3)]
    t = -float(ned[2]) / ray[2]
    r = (float(ned[0]) + ray[0] * t + origin) / gsd
    c = (float(ned[1]) + ray[1] * t + origin) / gsd
    del ray, t
    r.clamp_(0, tex.shape[0] - 2)
    c.clamp_(0, tex.shape[1] - 2)
    r_lo, c_lo = int(r.min()), int(c.min())
    r_hi, c_hi = min(int(r.max()) + 2, tex.shape[0]), min(int(c.max()) + 2, tex.shape[1])
    sub = tex[r_lo:r_hi, c_lo:c_hi].contiguous()
    r -= r_lo
    c -= c_lo
    r0, c0 = torch.floor(r), torch.floor(c)
    fr, fc = r - r0, c - c0
    del r, c
    r0 = r0.long().clamp_(0, sub.shape[0] - 2)
    c0 = c0.long().clamp_(0, sub.shape[1] - 2)
    flat = sub.reshape(-1)
    i00 = r0 * sub.shape[1] + c0
    del r0, c0
    top = flat[i00].to(f64) * (1 - fc) + flat[i00 + 1].to(f64) * fc
    i00 += sub.shape[1]
    bot = flat[i00].to(f64) * (1 - fc) + flat[i00 + 1].to(f64) * fc
    gray = top * (1 - fr) + bot * fr
    img = torch.stack([gray, gray * 0.9 + 10, gray * 0.8 + 20], 2)
    return torch.round(img).clamp_(0, 255).to(torch.uint8).cpu().numpy()


# --------------------------------------------------------------------------------------
# a rendered survey on disk (bench.py --e2e, BASELINE configs[4] shape): a textured ground plane
# photographed by nadir cameras on a lawn-mower grid, every pixel ray-cast onto the plane with the
# camera model the pipeline uses; the project is handed poses that are off by ~1 m / ~1 deg
# --------------------------------------------------------------------------------------
def ground_texture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    for s in (2, 4, 8, 16, 32):
        g = rng.normal(size=(h // s + 2, w // s + 2))
        img += np.kron(g, np.ones((s, s)))[:h, :w] * s ** 0.7
    k = np.array([1, 4, 6, 4, 1.]) / 16
    img = np.apply_along_axis(lambda r: np.convolve(r, k, 'same'), 1, img)
    img = np.apply_along_axis(lambda r: np.convolve(r, k, 'same'), 0, img)
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return np.stack([img, img * 0.9 + 10, img * 0.8 + 20], 2).clip(0, 255).astype(np.uint8)


def render_view(tex, M, ned, w, h, gsd, origin):
    """image[v, u] = ground texture under the ray of pixel (u, v); ground plane z = 0 (NED)"""
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    ray = np.einsum('ij,jvu->ivu', M, np.stack([u, v, np.ones_like(u)]))
    t = -ned[2] / ray[2]
    n, e = ned[0] + ray[0] * t, ned[1] + ray[1] * t
    r, c = (n + origin) / gsd, (e + origin) / gsd
    r0, c0 = np.floor(r).astype(int), np.floor(c).astype(int)
    fr, fc = (r - r0)[..., None], (c - c0)[..., None]
    r0 = np.clip(r0, 0, tex.shape[0] - 2)
    c0 = np.clip(c0, 0, tex.shape[1] - 2)
    t00, t01 = tex[r0, c0].astype(float), tex[r0, c0 + 1].astype(float)
    t10, t11 = tex[r0 + 1, c0].astype(float), tex[r0 + 1, c0 + 1].astype(float)
    img = (t00 * (1 - fc) + t01 * fc) * (1 - fr) + (t10 * (1 - fc) + t11 * fc) * fr
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def render_view_device(tex, M, ned, w, h, gsd, origin):
    """render_view() with torch on the device that holds `tex` (uint8 [H,W,3]): 20 MP frames take
    a second each in numpy.  Data generation for tests / bench.py --e2e, never a timed path."""
    import torch
    dev = tex.device
    f64 = torch.float64
    u = torch.arange(w, device=dev, dtype=f64)[None, :]
    v = torch.arange(h, device=dev, dtype=f64)[:, None]
    Mt = [[float(M[i, j]) for j in range(3)] for i in range(3)]
    ray = [Mt[i][0] * u + Mt[i][1] * v + Mt[i][2] for i in range(3)]
    t = -float(ned[2]) / ray[2]
    r = (float(ned[0]) + ray[0] * t + origin) / gsd
    c = (float(ned[1]) + ray[1] * t + origin) / gsd
    del ray, t
    r0, c0 = torch.floor(r), torch.floor(c)
    fr, fc = (r - r0)[..., None], (c - c0)[..., None]
    del r, c
    r0 = r0.long().clamp_(0, tex.shape[0] - 2)
    c0 = c0.long().clamp_(0, tex.shape[1] - 2)
    flat = tex.reshape(-1, 3)
    i00 = r0 * tex.shape[1] + c0
    del r0, c0
    top = flat[i00].to(f64) * (1 - fc) + flat[i00 + 1].to(f64) * fc
    i00 += tex.shape[1]
    bot = flat[i00].to(f64) * (1 - fc) + flat[i00 + 1].to(f64) * fc
    img = top * (1 - fr) + bot * fr
    return torch.round(img).clamp_(0, 255).to(torch.uint8).cpu().numpy()


BIG_TEXTURE_PIXELS = 1500 * 1000 * 1000     # ground textures above this are built band by band (grey only)
FULL_FRAME = dict(w=W_PX, h=H_PX, focal=FX, gsd=0.03)       # the FC6310S frame of BASELINE configs[4]


def make_rendered_survey(project_dir, rows, cols, w=1368, h=912, focal=916.7, alt=100.0,
                         spacing=(45.0, 40.0), gsd=0.11, seed=2024, device=None, chunk=0,
                         on_frames=None):
    """Writes <project_dir>/images/Pnnn.JPG and returns (names, truth [(ned, ypr)], logged
    [(ned, ypr)], K).  Default camera = the FC6310S field of view at a quarter of its pixels;
    **FULL_FRAME = the 5472x3648 frame itself (pass `device`: texture and ray casting on the GPU).
    chunk / on_frames: after every `chunk` frames (all of them on disk) on_frames(names, truth,
    logged, K, first, stop) is called -- a survey larger than the scratch disk is rendered,
    consumed and deleted a window at a time (10 000 frames of 20 MP are 80 GB of JPEG)."""
    from PIL import Image as PILImage
    from . import match_cleanup
    rng = np.random.default_rng(seed)
    os_ = __import__('os')
    os_.makedirs(os_.path.join(project_dir, 'images'), exist_ok=True)
    origin = 80.0
    th, tw = int((rows * spacing[0] + 2 * origin) / gsd), int((cols * spacing[1] + 2 * origin) / gsd)
    big = device is not None and th * tw > BIG_TEXTURE_PIXELS
    if big:
        tex = make_survey_gray_big(th, tw, seed, device)     # (beyond one interpolate() call)
    elif device is not None:
        tex = make_survey_image(th, tw, seed, device)
    else:
        tex = ground_texture(th, tw, seed)
    K = np.array([[focal, 0, w / 2.0], [0, focal, h / 2.0], [0, 0, 1.0]])
    IK = np.linalg.inv(K)
    d2r = np.pi / 180.0
    names, truth, logged = [], [], []
    handed = 0
    # (the JPEG encoder releases the interpreter: frames are encoded on a few threads while the
    #  next ones are rendered -- 2048 frames of 20 MP take a minute instead of four)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=min(12, max(2, (os_.cpu_count() or 4) - 2)))
    saves = []

    def _save(rgb, path):
        PILImage.fromarray(rgb).save(path, quality=95)
    for row in range(rows):
        for col in range(cols):
            k = col if row % 2 == 0 else cols - 1 - col
            ned = np.array([20.0 + spacing[0] * row + rng.normal(0, 0.5),
                            25.0 + spacing[1] * k + rng.normal(0, 0.5), -alt + rng.normal(0, 0.5)])
            ypr = np.array([(0.0 if row % 2 == 0 else 180.0) + rng.normal(0, 2.0),
                            -90.0 + rng.normal(0, 1.5), rng.normal(0, 1.5)])
            name = ('P%03d' if rows * cols <= 9999 else 'P%05d') % len(names)
            q = tf.quaternion_from_euler(ypr[0] * d2r, ypr[1] * d2r, ypr[2] * d2r, 'rzyx')
            M = tf.quaternion_matrix(q)[:3, :3].dot(match_cleanup.CAM2BODY).dot(IK)
            if big:
                bgr = render_view_device_gray(tex, M, ned, w, h, gsd, origin)
            elif device is not None:
                bgr = render_view_device(tex, M, ned, w, h, gsd, origin)
            else:
                bgr = render_view(tex, M, ned, w, h, gsd, origin)
            saves.append(pool.submit(_save, np.ascontiguousarray(bgr[:, :, ::-1]),
                                     os_.path.join(project_dir, 'images', name + '.JPG')))
            while len(saves) > 24:                       # (bounded: 60 MB per frame in flight)
                saves.pop(0).result()
            names.append(name)
            truth.append((ned, ypr))
            logged.append((ned + rng.normal(0, 0.8, 3), ypr + rng.normal(0, 0.7, 3)))
            if on_frames is not None and chunk > 0 and len(names) - handed >= chunk:
                for f in saves:
                    f.result()
                saves = []
                on_frames(names, truth, logged, K, handed, len(names))
                handed = len(names)
    for f in saves:
        f.result()
    pool.shutdown()
    if on_frames is not None and len(names) > handed:
        on_frames(names, truth, logged, K, handed, len(names))
    return names, truth, logged, K
