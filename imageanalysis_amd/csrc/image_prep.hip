// Image preparation in front of SIFT (gfx950): what the reference does with OpenCV in
// Image.load_rgb(equalize=True) and detect_features (scripts/lib/image.py:105-112,313):
//   BGR -> HSV (8 bit), CLAHE(clipLimit 3.0, 8x8 tiles) on V, HSV -> BGR, bilinear resize by
//   `scale`.  HBM-bound byte streams; one 20 MP image is 60 MB in, 3 passes.
// The 8-bit algorithms are restated from their published definitions (fixed-point HSV with
// 12-bit division tables, clip + uniform redistribution, bilinear blending of the tile LUTs,
// 11-bit fixed-point resize); cv2 itself is absent, the oracle is oracle/image_oracle.py.
#include "iamx_common.h"

namespace {

constexpr int HSV_SHIFT = 12;
constexpr int TILES = 8;

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    p = p < 0 ? -p : p;
    p %= period;
    return p >= n ? period - p : p;
}

__device__ __forceinline__ int div_table(int num, double den_scale, int i)
{
    // saturate_cast<int>((num << shift) / (den_scale * i)), round half to even
    return i == 0 ? 0 : (int)rint((double)(num << HSV_SHIFT) / (den_scale * i));
}

__global__ __launch_bounds__(256) void hsv_kernel(const uint8_t *__restrict__ bgr, int64_t npx,
                                                  uint8_t *__restrict__ hsv)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
    const int s = (diff * div_table(255, 1.0, v) + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * div_table(180, 6.0, diff) + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT;
    if (h < 0) h += 180;
    hsv[3 * i] = (uint8_t)h;
    hsv[3 * i + 1] = (uint8_t)s;
    hsv[3 * i + 2] = (uint8_t)v;
}

// histogram of V per tile (padded image, reflect-101); several blocks per tile
__global__ __launch_bounds__(256) void clahe_hist_kernel(const uint8_t *__restrict__ hsv, int h, int w,
                                                         int th, int tw, int blocks_per_tile,
                                                         int *__restrict__ hist /*[64][256]*/)
{
    __shared__ int sh[256];
    const int tile = blockIdx.x / blocks_per_tile, part = blockIdx.x % blocks_per_tile;
    const int tj = tile / TILES, ti = tile % TILES;
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t area = (int64_t)th * tw;
    for (int64_t p = (int64_t)part * 256 + threadIdx.x; p < area; p += (int64_t)blocks_per_tile * 256) {
        const int y = reflect101(tj * th + (int)(p / tw), h), x = reflect101(ti * tw + (int)(p % tw), w);
        atomicAdd(&sh[hsv[((int64_t)y * w + x) * 3 + 2]], 1);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[tile * 256 + threadIdx.x], sh[threadIdx.x]);
}

// clip + redistribute + cumulative LUT: one block per tile
__global__ __launch_bounds__(256) void clahe_lut_kernel(const int *__restrict__ hist, int clip,
                                                        float lut_scale, uint8_t *__restrict__ lut)
{
    __shared__ int hs[256];
    __shared__ int tot;
    const int t = threadIdx.x, tile = blockIdx.x;
    int v = hist[tile * 256 + t];
    if (t == 0) tot = 0;
    __syncthreads();
    const int over = v > clip ? v - clip : 0;
    if (over) atomicAdd(&tot, over);
    v = v < clip ? v : clip;
    __syncthreads();
    const int clipped = tot;
    const int batch = clipped / 256;
    int residual = clipped - batch * 256;
    v += batch;
    if (residual) {
        const int step = 256 / residual > 1 ? 256 / residual : 1;
        // bins 0, step, 2*step, ... get +1 while residual lasts
        if (t % step == 0 && t / step < residual) v += 1;
    }
    hs[t] = v;
    __syncthreads();
    if (t == 0) {
        int sum = 0;
        for (int i = 0; i < 256; ++i) {
            sum += hs[i];
            float f = rintf((float)sum * lut_scale);
            f = f < 0.f ? 0.f : (f > 255.f ? 255.f : f);
            lut[tile * 256 + i] = (uint8_t)f;
        }
    }
}

// apply CLAHE to V and convert HSV -> BGR
__global__ __launch_bounds__(256) void clahe_apply_kernel(const uint8_t *__restrict__ hsv, int h, int w,
                                                          int th, int tw,
                                                          const uint8_t *__restrict__ lut,
                                                          uint8_t *__restrict__ bgr)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)h * w) return;
    const int x = (int)(i % w), y = (int)(i / w);
    const int H = hsv[3 * i], S = hsv[3 * i + 1], V = hsv[3 * i + 2];
    const float tyf = __fsub_rn(__fmul_rn((float)y, (float)(1.0 / (double)th)), 0.5f);
    const float txf = __fsub_rn(__fmul_rn((float)x, (float)(1.0 / (double)tw)), 0.5f);
    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
    const float ya = __fsub_rn(tyf, (float)ty1), xa = __fsub_rn(txf, (float)tx1);
    const int ty2 = min(ty1 + 1, TILES - 1), tx2 = min(tx1 + 1, TILES - 1);
    ty1 = max(ty1, 0);
    tx1 = max(tx1, 0);
    const float l11 = lut[(ty1 * TILES + tx1) * 256 + V], l12 = lut[(ty1 * TILES + tx2) * 256 + V];
    const float l21 = lut[(ty2 * TILES + tx1) * 256 + V], l22 = lut[(ty2 * TILES + tx2) * 256 + V];
    const float xa1 = __fsub_rn(1.f, xa), ya1 = __fsub_rn(1.f, ya);
    const float top = __fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa));
    const float bot = __fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa));
    float res = rintf(__fadd_rn(__fmul_rn(top, ya1), __fmul_rn(bot, ya)));
    res = res < 0.f ? 0.f : (res > 255.f ? 255.f : res);
    // HSV -> BGR (float)
    const float hh = __fmul_rn((float)H, (float)(6.0 / 180.0));
    const float s = __fmul_rn((float)S, (float)(1.0 / 255.0)), v = __fmul_rn(res, (float)(1.0 / 255.0));
    float b, g, r;
    if (S == 0) {
        b = g = r = v;
    } else {
        int sector = (int)floorf(hh);
        float f = __fsub_rn(hh, (float)sector);
        if (sector < 0 || sector >= 6) { sector = 0; f = 0.f; }
        const float tab[4] = {v, __fmul_rn(v, __fsub_rn(1.f, s)),
                              __fmul_rn(v, __fsub_rn(1.f, __fmul_rn(s, f))),
                              __fmul_rn(v, __fsub_rn(1.f, __fmul_rn(s, __fsub_rn(1.f, f))))};
        const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
        b = tab[sd[sector][0]];
        g = tab[sd[sector][1]];
        r = tab[sd[sector][2]];
    }
    auto q = [](float x) {
        float f = rintf(__fmul_rn(x, 255.f));
        return (uint8_t)(f < 0.f ? 0.f : (f > 255.f ? 255.f : f));
    };
    bgr[3 * i] = q(b);
    bgr[3 * i + 1] = q(g);
    bgr[3 * i + 2] = q(r);
}

__global__ __launch_bounds__(256) void resize_kernel(const uint8_t *__restrict__ src, int h, int w,
                                                     int ch, int dh, int dw,
                                                     uint8_t *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)dh * dw) return;
    const int dx = (int)(i % dw), dy = (int)(i / dw);
    auto tap = [](int d, int n_src, int n_dst, int &s0, int &s1, int &a0, int &a1) {
        const double sc = (double)n_src / n_dst;
        const double f = ((double)d + 0.5) * sc - 0.5;
        s0 = (int)floor(f);
        float t = (float)(f - s0);
        if (s0 < 0) { s0 = 0; t = 0.f; }
        if (s0 >= n_src - 1) { s0 = n_src - 1; t = 0.f; }
        s1 = s0 + 1 < n_src ? s0 + 1 : n_src - 1;
        a1 = (int)rintf(t * 2048.f);
        a0 = (int)rintf((1.f - t) * 2048.f);
    };
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    tap(dx, w, dw, x0, x1, ax0, ax1);
    tap(dy, h, dh, y0, y1, ay0, ay1);
    for (int c = 0; c < ch; ++c) {
        const int r0 = src[((int64_t)y0 * w + x0) * ch + c] * ax0 + src[((int64_t)y0 * w + x1) * ch + c] * ax1;
        const int r1 = src[((int64_t)y1 * w + x0) * ch + c] * ax0 + src[((int64_t)y1 * w + x1) * ch + c] * ax1;
        int o = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
        o = o < 0 ? 0 : (o > 255 ? 255 : o);
        dst[i * ch + c] = (uint8_t)o;
    }
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int64_t iamx_image_prep_workspace_bytes(int height, int width)
{
    if (height < 1 || width < 1) return 0;
    return (int64_t)height * width * 6 + 64 * 256 * 4 + 64 * 256 + 1024;
}

extern "C" int iamx_image_resized_dims(int height, int width, double scale, int *out_h, int *out_w)
{
    IAMX_REQUIRE(out_h && out_w && scale > 0, "bad argument");
    *out_h = (int)lrint(height * scale);
    *out_w = (int)lrint(width * scale);
    return IAMX_OK;
}

extern "C" int iamx_image_equalize_resize(const uint8_t *bgr, int height, int width, int equalize,
                                          float clip_limit, double scale, void *workspace,
                                          int64_t workspace_bytes, uint8_t *out, void *stream)
{
    IAMX_REQUIRE(bgr && workspace && out, "null pointer");
    IAMX_REQUIRE(height >= 8 && width >= 8 && scale > 0, "bad size / scale");
    IAMX_REQUIRE(workspace_bytes >= iamx_image_prep_workspace_bytes(height, width),
                 "workspace too small");
    hipStream_t st = iamx::as_stream(stream);
    const int64_t npx = (int64_t)height * width;
    uint8_t *hsv = static_cast<uint8_t *>(workspace);
    uint8_t *eq = hsv + npx * 3;
    int *hist = reinterpret_cast<int *>(eq + npx * 3 + (256 - (npx * 6) % 256) % 256);
    uint8_t *lut = reinterpret_cast<uint8_t *>(hist + 64 * 256);
    const uint8_t *src = bgr;
    if (equalize) {
        const int pw = width % TILES ? width + (TILES - width % TILES) : width;
        const int ph = height % TILES ? height + (TILES - height % TILES) : height;
        const int tw = pw / TILES, th = ph / TILES;
        const int area = tw * th;
        int clip = (int)(clip_limit * area / 256.0f);
        if (clip < 1) clip = 1;
        (void)hipMemsetAsync(hist, 0, 64 * 256 * 4, st);
        hipLaunchKernelGGL(hsv_kernel, dim3(nblk(npx)), dim3(256), 0, st, bgr, npx, hsv);
        const int bpt = 16;
        hipLaunchKernelGGL(clahe_hist_kernel, dim3(64 * bpt), dim3(256), 0, st, hsv, height, width,
                           th, tw, bpt, hist);
        hipLaunchKernelGGL(clahe_lut_kernel, dim3(64), dim3(256), 0, st, hist, clip,
                           255.0f / (float)area, lut);
        hipLaunchKernelGGL(clahe_apply_kernel, dim3(nblk(npx)), dim3(256), 0, st, hsv, height,
                           width, th, tw, lut, eq);
        src = eq;
    }
    int dh, dw;
    iamx_image_resized_dims(height, width, scale, &dh, &dw);
    IAMX_REQUIRE(dh >= 1 && dw >= 1, "scaled image is empty");
    hipLaunchKernelGGL(resize_kernel, dim3(nblk((int64_t)dh * dw)), dim3(256), 0, st, src, height,
                       width, 3, dh, dw, out);
    return iamx::check_launch("iamx_image_equalize_resize");
}
