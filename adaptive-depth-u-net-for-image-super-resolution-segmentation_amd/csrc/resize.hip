// Antialiased bilinear resize (tf.image.resize(..., "bilinear", antialias=True)) and its gradient as
// one separable banded linear map with host-built per-axis tap tables.  fp32 arithmetic regardless of
// the storage dtype, as ResizeByScale / ResizeToMatch do (shared/custom_layers.py:102,124).
// HBM/L2-bound gather: one thread per (output pixel, 16-byte channel vector).
#include "common.h"

namespace {

// grid: x = blocks over one output row (ox, channel vector), y = output row, z = image.  No 64-bit divisions: the
// row/image come from the block index, (ox, vector) from one 32-bit division by the vectors-per-pixel count.
template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const int* __restrict__ sy, const float* __restrict__ wy, int ky,
                                                       const int* __restrict__ sx, const float* __restrict__ wx, int kx,
                                                       int h, int w, int oh, int ow, int c, int accumulate) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT;
    const int oy = blockIdx.y, nn = blockIdx.z;
    const int y0 = sy[oy];
    const float* wyr = wy + oy * ky;
    const T* xn = x + (size_t)nn * h * w * c;
    T* yrow = y + ((size_t)nn * oh + oy) * ow * c;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ow * vecs; i += gridDim.x * 256) {
        const int ox = i / vecs, v = i - ox * vecs;
        const int x0 = sx[ox];
        const float* wxr = wx + ox * kx;
        float acc[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
        for (int a = 0; a < ky; ++a) {
            const float fy = wyr[a];
            if (fy == 0.f) continue;
            const T* row = xn + (size_t)min(y0 + a, h - 1) * w * c + v * EPT;
            for (int b = 0; b < kx; ++b) {
                const float f = fy * wxr[b];
                if (f == 0.f) continue;
                Vec16<T> ld;
                float t[EPT];
                ld.load(row + (size_t)min(x0 + b, w - 1) * c);
                ld.to_f32(t);
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[e] += f * t[e];
            }
        }
        T* dst = yrow + (size_t)i * EPT;
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
    AD_REQUIRE(oh <= 65535 && n <= 65535, "ad_resample: oh=%d n=%d exceed the grid limits", oh, n);
    hipStream_t s = (hipStream_t)stream;
    const int row_items = ow * (c / ept);
    dim3 grid((row_items + 255) / 256, oh, n);
    if (dtype == AD_BF16)
        resample_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, sy, wy, ky, sx, wx, kx, h, w, oh, ow, c,
                                                      accumulate);
    else
        resample_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)y, sy, wy, ky, sx, wx, kx, h, w, oh, ow, c,
                                                     accumulate);
    AD_LAUNCH_CHECK("ad_resample");
    return AD_OK;
}
