// Antialiased bilinear resize (tf.image.resize(..., "bilinear", antialias=True)) and its gradient as
// one separable banded linear map with host-built per-axis tap tables.  fp32 arithmetic regardless of
// the storage dtype, as ResizeByScale / ResizeToMatch do (shared/custom_layers.py:102,124).
// HBM-bound: resample_march_kernel reads every input row once per group of R output rows and keeps the horizontal
// overlap in the L1; resample_kernel (one gather per output vector) is the fallback for very wide tap tables.
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

// March kernel: a thread owns one (output column, 16-byte channel vector) for R consecutive output rows and walks
// down the union of their input rows once.  Per input row it forms the horizontal sum (KX independent 16-byte loads
// in flight, tap weights in registers) and adds it to each of the R row accumulators with that row's vertical weight
// (block-uniform).  Against the gather kernel this cuts the loads per output from ky*kx to (rows walked)*kx/R
// and the L2 traffic of a x4 reduction from 4x the input to 1.25x.  KX >= kx; surplus taps carry weight 0 and
// re-read the last valid tap.
template <typename T, int R, int KX>
__global__ __launch_bounds__(256) void resample_march_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             const int* __restrict__ sy, const float* __restrict__ wy,
                                                             int ky, const int* __restrict__ sx,
                                                             const float* __restrict__ wx, int kx, int h, int w, int oh,
                                                             int ow, int c, int accumulate) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT;
    const int oy0 = blockIdx.y * R, nn = blockIdx.z;
    int syr[R];
    int ylo = h, yhi = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (oy0 + r < oh) {
            syr[r] = sy[oy0 + r];
            ylo = min(ylo, syr[r]);
            yhi = max(yhi, min(syr[r] + ky, h));
        } else {
            syr[r] = 1 << 29;      // never in range
        }
    }
    const T* xn = x + (size_t)nn * h * w * c;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ow * vecs; i += gridDim.x * 256) {
        const int ox = i / vecs, v = i - ox * vecs;
        const int x0 = sx[ox];
        float fx[KX];
        int off[KX];
#pragma unroll
        for (int b = 0; b < KX; ++b) {
            fx[b] = b < kx ? wx[ox * kx + b] : 0.f;
            off[b] = min(x0 + min(b, kx - 1), w - 1) * c + v * EPT;
        }
        float acc[R][EPT];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < EPT; ++e) acc[r][e] = 0.f;
#pragma unroll 2
        for (int iy = ylo; iy < yhi; ++iy) {
            const T* row = xn + (size_t)iy * w * c;
            Vec16<T> ld[KX];
#pragma unroll
            for (int b = 0; b < KX; ++b) ld[b].load(row + off[b]);
            float hs[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) hs[e] = 0.f;
#pragma unroll
            for (int b = 0; b < KX; ++b) {
                float t[EPT];
                ld[b].to_f32(t);
#pragma unroll
                for (int e = 0; e < EPT; ++e) hs[e] += fx[b] * t[e];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int a = iy - syr[r];
                const float fy = (a >= 0 && a < ky) ? wy[(oy0 + r) * ky + a] : 0.f;
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[r][e] += fy * hs[e];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (oy0 + r >= oh) break;
            T* dst = y + (((size_t)nn * oh + oy0 + r) * ow) * c + (size_t)i * EPT;
            if (accumulate) {
                Vec16<T> old;
                float t[EPT];
                old.load(dst);
                old.to_f32(t);
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[r][e] += t[e];
            }
            Vec16<T> st;
            st.from_f32(acc[r]);
            st.store(dst);
        }
    }
}

template <typename T, int R>
void launch_march(const void* x, void* y, const int* sy, const float* wy, int ky, const int* sx, const float* wx, int kx,
                  int n, int h, int w, int oh, int ow, int c, int accumulate, hipStream_t s) {
    const int ept = ElemTraits<T>::EPT;
    dim3 grid((ow * (c / ept) + 255) / 256, (oh + R - 1) / R, n);
#define AD_MARCH(KX)                                                                                                   \
    resample_march_kernel<T, R, KX><<<grid, 256, 0, s>>>((const T*)x, (T*)y, sy, wy, ky, sx, wx, kx, h, w, oh, ow, c,     \
                                                          accumulate)
    if (kx <= 2) AD_MARCH(2);
    else if (kx <= 4) AD_MARCH(4);
    else if (kx <= 8) AD_MARCH(8);
    else if (kx <= 12) AD_MARCH(12);
    else AD_MARCH(16);
#undef AD_MARCH
}

}  // namespace

extern "C" int ad_resample(const void* x, void* y, const int* sy, const float* wy, int ky, const int* sx,
                           const float* wx, int kx, int n, int h, int w, int oh, int ow, int c, int accumulate,
                           int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_resample: bad dtype %d", dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ky > 0 && kx > 0, "ad_resample: bad shape");
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(c > 0 && c % ept == 0, "ad_resample: c=%d must be a multiple of %d", c, ept);
    AD_REQUIRE(oh <= 65535 && n <= 65535, "ad_resample: oh=%d n=%d exceed the grid limits", oh, n);
    hipStream_t s = (hipStream_t)stream;
    if (kx <= 16) {
        // R output rows per thread: 4 where there are rows enough to keep every CU busy, else 1 (bottleneck maps)
        const bool r4 = (long)oh * n >= 4096;
#define AD_RS(T, R) launch_march<T, R>(x, y, sy, wy, ky, sx, wx, kx, n, h, w, oh, ow, c, accumulate, s)
        AD_DISPATCH_DTYPE(dtype, T_, if (r4) AD_RS(T_, 4); else AD_RS(T_, 1);)
#undef AD_RS
        AD_LAUNCH_CHECK("ad_resample");
        return AD_OK;
    }
    const int row_items = ow * (c / ept);
    dim3 grid((row_items + 255) / 256, oh, n);
    AD_DISPATCH_DTYPE(dtype, T_, resample_kernel<T_><<<grid, 256, 0, s>>>((const T_*)x, (T_*)y, sy, wy, ky, sx, wx, kx, h, w, oh, ow,
                                                                        c, accumulate);)
    AD_LAUNCH_CHECK("ad_resample");
    return AD_OK;
}
