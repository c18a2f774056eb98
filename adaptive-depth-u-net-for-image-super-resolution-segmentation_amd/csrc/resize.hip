// Antialiased bilinear resize (tf.image.resize(..., "bilinear", antialias=True)) and its gradient as
// one separable banded linear map with host-built per-axis tap tables.  fp32 arithmetic regardless of
// the storage dtype, as ResizeByScale / ResizeToMatch do (shared/custom_layers.py:102,124).
// HBM/L2-bound gather: one thread per (output pixel, 16-byte channel vector).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const int* __restrict__ sy, const float* __restrict__ wy, int ky,
                                                       const int* __restrict__ sx, const float* __restrict__ wx, int kx,
                                                       int n, int h, int w, int oh, int ow, int c, int accumulate) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT;
    const int64_t total = (int64_t)n * oh * ow * vecs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int v = (int)(i % vecs);
        int64_t p = i / vecs;
        int ox = (int)(p % ow);
        int64_t r = p / ow;
        int oy = (int)(r % oh);
        int nn = (int)(r / oh);
        float acc[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
        const int y0 = sy[oy], x0 = sx[ox];
        for (int a = 0; a < ky; ++a) {
            const float fy = wy[oy * ky + a];
            if (fy == 0.f) continue;
            const int iy = min(y0 + a, h - 1);
            const T* row = x + ((int64_t)nn * h + iy) * w * c + v * EPT;
            for (int b = 0; b < kx; ++b) {
                const float f = fy * wx[ox * kx + b];
                if (f == 0.f) continue;
                const int ix = min(x0 + b, w - 1);
                Vec16<T> ld;
                float t[EPT];
                ld.load(row + (int64_t)ix * c);
                ld.to_f32(t);
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[e] += f * t[e];
            }
        }
        T* dst = y + p * c + v * EPT;
        if (accumulate) {
            Vec16<T> old;
            float t[EPT];
            old.load(dst);
            old.to_f32(t);
#pragma unroll
            for (int e = 0; e < EPT; ++e) acc[e] += t[e];
        }
        Vec16<T> st;
        st.from_f32(acc);
        st.store(dst);
    }
}

}  // namespace

extern "C" int ad_resample(const void* x, void* y, const int* sy, const float* wy, int ky, const int* sx,
                           const float* wx, int kx, int n, int h, int w, int oh, int ow, int c, int accumulate,
                           int dtype, void* stream) {
    AD_REQUIRE(dtype == AD_BF16 || dtype == AD_F32, "ad_resample: bad dtype %d", dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ky > 0 && kx > 0, "ad_resample: bad shape");
    const int ept = dtype == AD_BF16 ? 8 : 4;
    AD_REQUIRE(c > 0 && c % ept == 0, "ad_resample: c=%d must be a multiple of %d", c, ept);
    hipStream_t s = (hipStream_t)stream;
    int64_t total = (int64_t)n * oh * ow * (c / ept);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == AD_BF16)
        resample_kernel<bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, sy, wy, ky, sx, wx, kx, n, h, w, oh,
                                                        ow, c, accumulate);
    else
        resample_kernel<float><<<blocks, 256, 0, s>>>((const float*)x, (float*)y, sy, wy, ky, sx, wx, kx, n, h, w, oh, ow,
                                                       c, accumulate);
    AD_LAUNCH_CHECK("ad_resample");
    return AD_OK;
}
