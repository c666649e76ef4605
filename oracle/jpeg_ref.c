/* ORACLE / TEST INFRASTRUCTURE ONLY -- CPU restatement of the pixel reconstruction a JPEG
 * decoder performs after the Huffman stage, as libjpeg(-turbo) does it with its defaults (what
 * cv2.imread of scripts/lib/image.py:99-104 returns): dequantisation + jidctint.c's "islow"
 * integer inverse DCT, jdsample.c's fancy upsampling (h2v1 / h2v2), jdcolor.c's YCbCr -> RGB
 * tables.  It is the twin of csrc/jpeg.hip's kernels: tests/test_jpeg.py feeds it the
 * coefficients of iamx_jpeg_decode_coefficients and compares with Pillow (libjpeg-turbo) on the
 * CPU; tests/test_jpeg_gpu.py compares the kernels with Pillow directly.
 * Never linked or loaded by the product. */
#include <stdint.h>
#include <stdlib.h>

static int range_limit_idct(int x)
{
    const int i = x & 1023;
    return i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896));
}

static void idct_1d(const int in[8], int out[8], int shift)
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137);
    int tmp3 = z1 + z2 * 6270;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) * 8192;
    int tmp1 = (z2 - z3) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    out[0] = (tmp10 + tmp3 + rnd) >> shift; out[7] = (tmp10 - tmp3 + rnd) >> shift;
    out[1] = (tmp11 + tmp2 + rnd) >> shift; out[6] = (tmp11 - tmp2 + rnd) >> shift;
    out[2] = (tmp12 + tmp1 + rnd) >> shift; out[5] = (tmp12 - tmp1 + rnd) >> shift;
    out[3] = (tmp13 + tmp0 + rnd) >> shift; out[4] = (tmp13 - tmp0 + rnd) >> shift;
}

static int chroma_at(const uint8_t *pl, int pitch, int dw, int dh, int hmax, int vmax, int x, int y)
{
    if (hmax == 1) return pl[(size_t)y * pitch + x];
    if (vmax == 1) {
        const uint8_t *row = pl + (size_t)y * pitch;
        const int i = x >> 1, v = row[i];
        if (x & 1) return i == dw - 1 ? v : (v * 3 + row[i + 1] + 2) >> 2;
        return i == 0 ? v : (v * 3 + row[i - 1] + 1) >> 2;
    }
    const int r = y >> 1;
    int rn = (y & 1) ? r + 1 : r - 1;
    rn = rn < 0 ? 0 : (rn > dh - 1 ? dh - 1 : rn);
    const uint8_t *r0 = pl + (size_t)r * pitch, *r1 = pl + (size_t)rn * pitch;
    const int i = x >> 1, cur = r0[i] * 3 + r1[i];
    if (x & 1) return i == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + (r0[i + 1] * 3 + r1[i + 1]) + 7) >> 4;
    return i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + (r0[i - 1] * 3 + r1[i - 1]) + 8) >> 4;
}

static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* info as written by iamx_jpeg_info; coef [info[11]][64] int16, quant [3][64]; bgr [h][w][3] */
int oracle_jpeg_reconstruct(const int16_t *coef, const uint16_t *quant, const int32_t *info, uint8_t *bgr)
{
    const int width = info[0], height = info[1], ncomp = info[2], hmax = info[3], vmax = info[4];
    uint8_t *plane[3] = {0, 0, 0};
    size_t first = 0;
    for (int c = 0; c < ncomp; ++c) {
        const int bw = info[5 + 2 * c], bh = info[6 + 2 * c], pitch = bw * 8;
        plane[c] = (uint8_t *)malloc((size_t)bw * bh * 64);
        if (!plane[c]) return -2;
        for (int by = 0; by < bh; ++by)
            for (int bx = 0; bx < bw; ++bx) {
                const int16_t *src = coef + 64 * (first + (size_t)by * bw + bx);
                int cf[64], ws[64];
                for (int i = 0; i < 64; ++i) cf[i] = (int)src[i] * (int)quant[c * 64 + i];
                for (int col = 0; col < 8; ++col) {
                    int in[8], out[8];
                    for (int r = 0; r < 8; ++r) in[r] = cf[r * 8 + col];
                    idct_1d(in, out, 11);
                    for (int r = 0; r < 8; ++r) ws[r * 8 + col] = out[r];
                }
                for (int r = 0; r < 8; ++r) {
                    int out[8];
                    idct_1d(ws + r * 8, out, 18);
                    for (int k = 0; k < 8; ++k)
                        plane[c][((size_t)by * 8 + r) * pitch + bx * 8 + k] = (uint8_t)range_limit_idct(out[k]);
                }
            }
        first += (size_t)bw * bh;
    }
    const int dw = (width + hmax - 1) / hmax, dh = (height + vmax - 1) / vmax;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            uint8_t *o = bgr + ((size_t)y * width + x) * 3;
            const int yv = plane[0][(size_t)y * (info[5] * 8) + x];
            if (ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)yv; continue; }
            const int cp = info[7] * 8;
            const int cb = chroma_at(plane[1], cp, dw, dh, hmax, vmax, x, y) - 128;
            const int cr = chroma_at(plane[2], cp, dw, dh, hmax, vmax, x, y) - 128;
            o[2] = (uint8_t)clamp255(yv + ((91881 * cr + 32768) >> 16));
            o[1] = (uint8_t)clamp255(yv + ((-22554 * cb + 32768 + (-46802) * cr) >> 16));
            o[0] = (uint8_t)clamp255(yv + ((116130 * cb + 32768) >> 16));
        }
    for (int c = 0; c < ncomp; ++c) free(plane[c]);
    return 0;
}
