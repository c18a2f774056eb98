// Evaluation metrics of the reference's eval loops on the device (all fp32, single-channel luma planes):
//   BT.601 luma of the clipped RGB output          Super_resolution/code/train_adaptive_unet.py:144-157
//   per-image PSNR / MSE numerator                  :686-692, evaluate_model.py:117-123
//   tf.image.ssim: 11x11 Gaussian (sigma 1.5) VALID windows, K1 0.01, K2 0.03      :689, evaluate_model.py:119
//   tf.image.ssim_multiscale: 5 scales, 2x2 average pooling (symmetric padding of odd sizes)      :690, :120
// HBM-bound streaming kernels; the SSIM kernel keeps a (32+10) x (8+10) window of both planes in LDS and filters
// separably (horizontal pass into LDS, vertical pass per output pixel), so every input pixel is read ~1.7 times.
#include "common.h"

namespace {

struct Gauss11 {
    float g[11];
};

// y = clip((65.481 r + 128.553 g + 24.966 b + 16) / 255, 0, 1) of clip(rgb, 0, 1); one thread per pixel
__global__ __launch_bounds__(256) void luma_kernel(const float* __restrict__ rgb, float* __restrict__ y, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const float r = fminf(fmaxf(rgb[i * 3 + 0], 0.f), 1.f), g = fminf(fmaxf(rgb[i * 3 + 1], 0.f), 1.f),
                    b = fminf(fmaxf(rgb[i * 3 + 2], 0.f), 1.f);
        const float v = (r * 65.481f + g * 128.553f + b * 24.966f + 16.0f) / 255.0f;
        y[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// part[img][block] = sum over the block's pixels of (a - b)^2 inside the window [y0, y0+h) x [x0, x0+w) of planes with
// row stride ld and image stride is
__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ a, const float* __restrict__ b, int h, int w,
                                                    int64_t is, int ld, float* __restrict__ part) {
    __shared__ float sm[4];
    const int img = blockIdx.y;
    const float* pa = a + img * is;
    const float* pb = b + img * is;
    float s = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < h * w; i += gridDim.x * 256) {
        const int yy = i / w, xx = i - yy * w;
        const float d = pa[(int64_t)yy * ld + xx] - pb[(int64_t)yy * ld + xx];
        s += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)img * gridDim.x + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// out[img] = (sum of part[img][0..nb)) / count, fixed order.  A true division (tf.reduce_mean divides): a sum of `count` ones
// is exactly 1, which sum * (1 / count) is not -- the reference's degenerate row `inf, 1.0, 1.0, 0.0`
// (exp1_depth3_scale0.20_eval/per_image_metrics.csv:1388) is reproduced exactly (tests/test_metrics_gpu.py).
__global__ void rows_finish_kernel(const float* __restrict__ part, int nb, int ncol, float count, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (img, col)
    if (i >= n * ncol) return;
    const int img = i / ncol, col = i - img * ncol;
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += part[((size_t)img * nb + k) * ncol + col];
    out[i] = s / count;
}

constexpr int ST_W = 32, ST_H = 8, SK = 11, SHW = ST_W + SK - 1, SHH = ST_H + SK - 1;

// part[img][block][2] = { sum ssim, sum cs } over the block's 32 x 8 output pixels (VALID windows)
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int h, int w,
                                                   int64_t is, int ld, Gauss11 gk, float c1, float c2,
                                                   float* __restrict__ part, int tiles_x) {
    __shared__ float ta[SHH][SHW + 1], tb[SHH][SHW + 1];
    __shared__ float hb[5][SHH][ST_W + 1];
    __shared__ float red[2][4];
    const int img = blockIdx.y, tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W;
    const int oh = h - SK + 1, ow = w - SK + 1;
    const float* pa = a + img * is;
    const float* pb = b + img * is;
    const int tid = threadIdx.x;
    for (int i = tid; i < SHH * SHW; i += 256) {
        const int r = i / SHW, c = i - r * SHW;
        const int yy = min(oy0 + r, h - 1), xx = min(ox0 + c, w - 1);      // clamped reads feed only discarded outputs
        ta[r][c] = pa[(int64_t)yy * ld + xx];
        tb[r][c] = pb[(int64_t)yy * ld + xx];
    }
    __syncthreads();
    for (int i = tid; i < SHH * ST_W; i += 256) {
        const int r = i / ST_W, c = i - r * ST_W;
        float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
#pragma unroll
        for (int k = 0; k < SK; ++k) {
            const float va = ta[r][c + k], vb = tb[r][c + k], g = gk.g[k];
            sa += g * va; sb += g * vb; saa += g * va * va; sbb += g * vb * vb; sab += g * va * vb;
        }
        hb[0][r][c] = sa; hb[1][r][c] = sb; hb[2][r][c] = saa; hb[3][r][c] = sbb; hb[4][r][c] = sab;
    }
    __syncthreads();
    const int r = tid / ST_W, c = tid - r * ST_W;
    float ssim = 0.f, cs = 0.f;
    if (oy0 + r < oh && ox0 + c < ow) {
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < SK; ++k) {
            const float g = gk.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += g * hb[q][r + k][c];
        }
        // tf.image.ssim's own arrangement (_ssim_helper), every product and sum rounded by itself (no fused multiply-add):
        //   lum = (2 mu_a mu_b + c1) / (mu_a^2 + mu_b^2 + c1),  cs = (2 E[ab] - 2 mu_a mu_b + c2) / (E[a^2] + E[b^2] - mu_a^2 - mu_b^2 + c2)
        // On identical planes numerator and denominator are then the SAME floats and both ratios are exactly 1, as TensorFlow
        // reports for the all-black patch of the reference's evaluation; a contracted a*b+c rounds the two sides differently.
        const float num0 = __fmul_rn(2.f, __fmul_rn(m[0], m[1]));
        const float den0 = __fadd_rn(__fmul_rn(m[0], m[0]), __fmul_rn(m[1], m[1]));
        const float lum = __fdiv_rn(__fadd_rn(num0, c1), __fadd_rn(den0, c1));
        const float num1 = __fmul_rn(2.f, m[4]), den1 = __fadd_rn(m[2], m[3]);
        cs = __fdiv_rn(__fadd_rn(__fsub_rn(num1, num0), c2), __fadd_rn(__fsub_rn(den1, den0), c2));
        ssim = __fmul_rn(lum, cs);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ssim += __shfl_xor(ssim, o, 64); cs += __shfl_xor(cs, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ssim; red[1][tid >> 6] = cs; }
    __syncthreads();
    if (tid == 0) {
        float* p = part + ((size_t)img * gridDim.x + tile) * 2;
        p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// 2x2 average pooling; an odd extent is padded by repeating its last row / column (symmetric padding by one)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, int h, int w, int64_t is, int ld,
                                                       float* __restrict__ y, int oh, int ow) {
    const int img = blockIdx.y;
    const float* px = x + img * is;
    float* py = y + (int64_t)img * oh * ow;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < oh * ow; i += gridDim.x * 256) {
        const int oy = i / ow, ox = i - oy * ow;
        const int y0 = 2 * oy, y1 = min(2 * oy + 1, h - 1), x0 = 2 * ox, x1 = min(2 * ox + 1, w - 1);
        py[i] = 0.25f * ((px[(int64_t)y0 * ld + x0] + px[(int64_t)y0 * ld + x1]) + (px[(int64_t)y1 * ld + x0] + px[(int64_t)y1 * ld + x1]));
    }
}

}  // namespace

extern "C" int ad_luma_bt601(const float* rgb, float* y, int64_t npix, void* stream) {
    AD_REQUIRE(rgb && y && npix > 0, "ad_luma_bt601: bad arguments");
    const int blocks = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    luma_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(rgb, y, npix);
    AD_LAUNCH_CHECK("ad_luma_bt601");
    return AD_OK;
}

extern "C" size_t ad_metrics_ws_bytes(int n, int h, int w) {
    const size_t tiles = (size_t)((w + ST_W - 1) / ST_W) * ((h + ST_H - 1) / ST_H);
    return (size_t)n * (tiles > 64 ? tiles : 64) * 2 * sizeof(float);
}

extern "C" int ad_mse_per_image(const float* a, const float* b, int n, int h, int w, int64_t image_stride, int row_stride,
                                float* mse, void* ws, size_t ws_bytes, void* stream) {
    AD_REQUIRE(a && b && mse && n > 0 && h > 0 && w > 0 && row_stride >= w, "ad_mse_per_image: bad arguments");
    const int nb = 64;
    if (!ws || ws_bytes < (size_t)n * nb * sizeof(float)) return ad_set_error(AD_ERR_WS, "ad_mse_per_image: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    sqerr_kernel<<<dim3(nb, n), 256, 0, s>>>(a, b, h, w, image_stride, row_stride, (float*)ws);
    rows_finish_kernel<<<(n + 63) / 64, 64, 0, s>>>((const float*)ws, nb, 1, (float)h * (float)w, mse, n);
    AD_LAUNCH_CHECK("ad_mse_per_image");
    return AD_OK;
}

extern "C" int ad_ssim_per_image(const float* a, const float* b, int n, int h, int w, int64_t image_stride, int row_stride,
                                 float max_val, float* ssim_cs, void* ws, size_t ws_bytes, void* stream) {
    AD_REQUIRE(a && b && ssim_cs && n > 0 && row_stride >= w, "ad_ssim_per_image: bad arguments");
    AD_REQUIRE(h >= SK && w >= SK, "ad_ssim_per_image: %dx%d is smaller than the 11x11 window", h, w);
    const int oh = h - SK + 1, ow = w - SK + 1;
    const int tiles_x = (ow + ST_W - 1) / ST_W, tiles_y = (oh + ST_H - 1) / ST_H;
    const int tiles = tiles_x * tiles_y;
    AD_REQUIRE(tiles <= 1 << 20 && n <= 65535, "ad_ssim_per_image: grid too large");
    if (!ws || ws_bytes < (size_t)n * tiles * 2 * sizeof(float)) return ad_set_error(AD_ERR_WS, "ad_ssim_per_image: workspace too small");
    Gauss11 gk;
    double sum = 0.0, gd[SK];
    for (int i = 0; i < SK; ++i) { const double x = i - (SK - 1) / 2.0; gd[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); sum += gd[i]; }
    for (int i = 0; i < SK; ++i) gk.g[i] = (float)(gd[i] / sum);
    const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
    hipStream_t s = (hipStream_t)stream;
    ssim_kernel<<<dim3(tiles, n), 256, 0, s>>>(a, b, h, w, image_stride, row_stride, gk, c1, c2, (float*)ws, tiles_x);
    rows_finish_kernel<<<(2 * n + 63) / 64, 64, 0, s>>>((const float*)ws, tiles, 2, (float)oh * (float)ow, ssim_cs, n);
    AD_LAUNCH_CHECK("ad_ssim_per_image");
    return AD_OK;
}

extern "C" int ad_avgpool2_plane(const float* x, int n, int h, int w, int64_t image_stride, int row_stride, float* y,
                                 void* stream) {
    AD_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && row_stride >= w && n <= 65535, "ad_avgpool2_plane: bad arguments");
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    const int blocks = (oh * ow + 255) / 256 < 1024 ? (oh * ow + 255) / 256 : 1024;
    avgpool2_kernel<<<dim3(blocks, n), 256, 0, (hipStream_t)stream>>>(x, h, w, image_stride, row_stride, y, oh, ow);
    AD_LAUNCH_CHECK("ad_avgpool2_plane");
    return AD_OK;
}
