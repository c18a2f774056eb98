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

// The skip connection's gradient junction fused with the LayerNorm + ReLU backward that follows it in the encoder:
//   d(act) = dskip + R^T d(low)          (gradient of enc_down's input: the resample above with accumulate)
//   dz     = LayerNorm/ReLU backward of d(act) with the block's saved z, mean, rstd, gamma, beta
// d(act) stays in fp32 registers: one read of dskip / z and one write of dz replace "read dskip, write d(act), read
// d(act), read z, write dz".  The c / EPT lanes that hold a pixel's channels are adjacent (i = ox * vecs + v), so the
// LayerNorm reductions are xor-shuffles over that lane group (vecs a power of two <= 64).  A block walks several row
// groups (grid.y is capped) so that the per-block partial sums {dgamma, dbeta, dbias}[c] stay few.
// (r05: the <4, 2> instantiation -- the x4 skip junction of every level -- took 170 registers, two over the step to three waves
// per SIMD; asked for three the bf16 instantiation fits 168 without a spill (fp16 would spill four: it keeps two waves):
// 10.94 -> 10.89 ms per K2' step.  Bounds that make hipcc spill lose:
// resample_march_kernel<4, 8> at four waves +0.08 ms, head_ln_bwd_kernel at four +0.65 ms.)
template <typename T, int R, int KX>
__global__ __launch_bounds__(256, (R == 4 && KX == 2 && __is_same(T, bf16_t)) ? 3 : 1) void resample_ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dskip,
                                                              const T* __restrict__ z, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, T* __restrict__ dz,
                                                              float* __restrict__ part, const int* __restrict__ sy,
                                                              const float* __restrict__ wy, int ky,
                                                              const int* __restrict__ sx, const float* __restrict__ wx, int kx,
                                                              int h, int w, int oh, int ow, int c) {
    constexpr int EPT = ElemTraits<T>::EPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);       // [256 / vecs][3][c]
    const int vecs = c / EPT;
    const int nn = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;       // (ox, v); ow * vecs is a multiple of 256 (launcher)
    const int ox = i / vecs, v = i - ox * vecs;
    const T* xn = x + (size_t)nn * h * w * c;
    const int x0 = sx[ox];
    float fx[KX];
    int off[KX];
#pragma unroll
    for (int b = 0; b < KX; ++b) {
        fx[b] = b < kx ? wx[ox * kx + b] : 0.f;
        off[b] = min(x0 + min(b, kx - 1), w - 1) * c + v * EPT;
    }
    float gam[EPT], bet[EPT], a_g[EPT], a_b[EPT], a_z[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        gam[e] = gamma[v * EPT + e]; bet[e] = beta[v * EPT + e];
        a_g[e] = a_b[e] = a_z[e] = 0.f;
    }
    const float inv_c = 1.0f / (float)c;
    const int ngroups = (oh + R - 1) / R;
    for (int yg = blockIdx.y; yg < ngroups; yg += gridDim.y) {
        const int oy0 = yg * R;
        int syr[R];
        int ylo = h, yhi = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (oy0 + r < oh) {
                syr[r] = sy[oy0 + r];
                ylo = min(ylo, syr[r]);
                yhi = max(yhi, min(syr[r] + ky, h));
            } else {
                syr[r] = 1 << 29;
            }
        }
        // this thread's pixel of each of the R rows: skip gradient, saved conv output and statistics, fetched up front
        Vec16<T> lsk[R], lz[R];
        float mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int oy = min(oy0 + r, oh - 1);
            const size_t pix = ((size_t)nn * oh + oy) * ow + ox;
            lsk[r].load(dskip + pix * c + v * EPT);
            lz[r].load(z + pix * c + v * EPT);
            mu[r] = mean[pix];
            rs[r] = rstd[pix];
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
            const bool live = oy0 + r < oh;                     // uniform over the block
            float sk[EPT], zz[EPT], hh[EPT], gg[EPT];
            lsk[r].to_f32(sk);
            lz[r].to_f32(zz);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float da = acc[r][e] + sk[e];              // gradient of the block's ReLU output
                hh[e] = (zz[e] - mu[r]) * rs[r];
                const float yv = hh[e] * gam[e] + bet[e];
                const float dl = (live && yv > 0.f) ? da : 0.f;
                a_g[e] += dl * hh[e];
                a_b[e] += dl;
                gg[e] = dl * gam[e];
                s1 += gg[e];
                s2 += gg[e] * hh[e];
            }
            for (int o = 1; o < vecs; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            s1 *= inv_c; s2 *= inv_c;
            float o8[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) o8[e] = rs[r] * (gg[e] - s1 - hh[e] * s2);
            if (live) {
                Vec16<T> st;
                st.from_f32(o8);
                st.store(dz + (((size_t)nn * oh + oy0 + r) * ow + ox) * c + v * EPT);
                float back[EPT];
                st.to_f32(back);
#pragma unroll
                for (int e = 0; e < EPT; ++e) a_z[e] += back[e];
            }
        }
    }
    // block reduction over the 256 / vecs pixels of the block (fixed order => deterministic)
    const int gp = threadIdx.x / vecs;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ch = v * EPT + e;
        red[(gp * 3 + 0) * c + ch] = a_g[e];
        red[(gp * 3 + 1) * c + ch] = a_b[e];
        red[(gp * 3 + 2) * c + ch] = a_z[e];
    }
    __syncthreads();
    const int block = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int ppb = 256 / vecs;
    for (int k = threadIdx.x; k < 3 * c; k += 256) {
        float s = 0.f;
        for (int p = 0; p < ppb; ++p) s += red[p * 3 * c + k];
        part[(size_t)block * 3 * c + k] = s;
    }
}

// sums the per-block partials [nblocks][3][c] in a fixed order (as colsum_reduce_kernel of norm.hip)
__global__ __launch_bounds__(256) void resample_ln_partials_kernel(const float* __restrict__ part, int nblocks, int c,
                                                                   float* __restrict__ o0, float* __restrict__ o1,
                                                                   float* __restrict__ o2) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int i = blockIdx.x * 4 + cl;
    float s = 0.f;
    if (i < 3 * c)
    {
#pragma unroll 8
        for (int b = rg; b < nblocks; b += 64) s += part[(size_t)b * 3 * c + i];
    }
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < 3 * c) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        const int which = i / c, ch = i % c;
        (which == 0 ? o0 : which == 1 ? o1 : o2)[ch] = t;
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


// ---- skip-gradient junction + LayerNorm/ReLU backward (see resample_ln_bwd_kernel)
static bool resample_ln_ok(int n, int oh, int ow, int c, int kx, int dtype, int* gy) {
    const int ept = ad_is_half(dtype) ? 8 : 4;
    if (c <= 0 || c % ept) return false;
    const int vecs = c / ept;
    if (vecs > 64 || (vecs & (vecs - 1)) || (ow * vecs) % 256 || kx > 8 || n > 65535) return false;
    const int ngroups = (oh + 3) / 4;
    *gy = ngroups < 8 ? ngroups : 8;
    return true;
}

extern "C" int ad_resample_ln_bwd_supported(int n, int oh, int ow, int c, int kx, int dtype) {
    int gy;
    return ad_dtype_ok(dtype) && n > 0 && oh > 0 && ow > 0 && resample_ln_ok(n, oh, ow, c, kx, dtype, &gy);
}

extern "C" size_t ad_resample_ln_bwd_ws_bytes(int n, int oh, int ow, int c, int dtype) {
    int gy;
    if (!resample_ln_ok(n, oh, ow, c, 1, dtype, &gy)) return 0;
    const int ept = ad_is_half(dtype) ? 8 : 4;
    return (size_t)n * gy * (ow * (c / ept) / 256) * 3 * c * sizeof(float);
}

extern "C" int ad_resample_ln_bwd(const void* d_low, const void* dskip, const void* z, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta,
                                  float* dbias, const int* sy, const float* wy, int ky, const int* sx, const float* wx, int kx,
                                  int n, int h, int w, int oh, int ow, int c, void* ws, size_t ws_bytes, int dtype,
                                  void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_resample_ln_bwd: bad dtype %d", dtype);
    int gy;
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ky > 0 && resample_ln_ok(n, oh, ow, c, kx, dtype, &gy),
               "ad_resample_ln_bwd: unsupported shape n=%d %dx%d -> %dx%d c=%d kx=%d (ask ad_resample_ln_bwd_supported)", n, h, w,
               oh, ow, c, kx);
    AD_REQUIRE(d_low && dskip && z && mean && rstd && gamma && beta && dz && dgamma && dbeta && dbias, "ad_resample_ln_bwd: NULL operand");
    const int ept = ad_is_half(dtype) ? 8 : 4;
    const int gx = ow * (c / ept) / 256;
    const int nblocks = n * gy * gx;
    const size_t need = (size_t)nblocks * 3 * c * sizeof(float);
    if (!ws || ws_bytes < need) return ad_set_error(AD_ERR_WS, "ad_resample_ln_bwd: workspace %zu < %zu", ws_bytes, need);
    const size_t lds = (size_t)(256 / (c / ept)) * 3 * c * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(gx, gy, n);
#define AD_RLN(T, KX)                                                                                                   \
    resample_ln_bwd_kernel<T, 4, KX><<<grid, 256, lds, s>>>((const T*)d_low, (const T*)dskip, (const T*)z, mean, rstd, gamma,  \
                                                            beta, (T*)dz, (float*)ws, sy, wy, ky, sx, wx, kx, h, w, oh, ow, c)
    AD_DISPATCH_DTYPE(dtype, T_, if (kx <= 2) AD_RLN(T_, 2); else if (kx <= 4) AD_RLN(T_, 4); else AD_RLN(T_, 8);)
#undef AD_RLN
    AD_LAUNCH_CHECK("ad_resample_ln_bwd");
    resample_ln_partials_kernel<<<(3 * c + 3) / 4, 256, 0, s>>>((const float*)ws, nblocks, c, dgamma, dbeta, dbias);
    AD_LAUNCH_CHECK("resample_ln partials");
    return AD_OK;
}
