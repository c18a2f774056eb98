// LayerNormalization(axis=-1, eps=1e-3) + ReLU, forward and backward, and the ReLU-only backward.
// HBM-bound row kernels: G = C / (EPT * NV) lanes cooperate on one pixel with 16-byte loads,
// statistics in fp32 via xor-shuffles.  Replaces keras LayerNormalization + Activation("relu")
// at Super_resolution/code/train_adaptive_unet.py:203-204,208-209.
#include "common.h"

namespace {

constexpr int NB_MAX = 1024;  // blocks used by the backward kernels (partials per block in ws)

template <int G>
__device__ __forceinline__ float group_sum(float v) { return ad_group_sum<G>(v); }

// ----------------------------------------------------------------------------- forward
template <typename T, int NV, int G>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ z, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t npix,
                                                     int c, float eps, int relu) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;  // pixels per block pass
    const int tid = threadIdx.x;
    const int gl = tid % G, gp = tid / G;
    float gam[NV][EPT], bet[NV][EPT];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int ch = (v * G + gl) * EPT + e;
            gam[v][e] = gamma[ch];
            bet[v][e] = beta[ch];
        }
    const float inv_c = 1.0f / (float)c;
    for (int64_t pix = (int64_t)blockIdx.x * PPB + gp; pix < npix; pix += (int64_t)gridDim.x * PPB) {
        float x[NV][EPT];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            Vec16<T> ld;
            ld.load(z + pix * c + (v * G + gl) * EPT);
            ld.to_f32(x[v]);
#pragma unroll
            for (int e = 0; e < EPT; ++e) s += x[v][e];
        }
        const float mu = group_sum<G>(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                float d = x[v][e] - mu;
                q += d * d;
            }
        const float rs = rsqrtf(group_sum<G>(q) * inv_c + eps);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float o[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                float t = (x[v][e] - mu) * rs * gam[v][e] + bet[v][e];
                o[e] = relu ? fmaxf(t, 0.f) : t;
            }
            Vec16<T> st;
            st.from_f32(o);
            st.store(y + pix * c + (v * G + gl) * EPT);
        }
        if (gl == 0) {
            mean[pix] = mu;
            rstd[pix] = rs;
        }
    }
}

// ----------------------------------------------------------------------------- backward
// MODE 0: LayerNorm+ReLU, 1: LayerNorm only, 2: ReLU only (z holds the ReLU output y).
// Per-block partial column sums go to part[block][3][c] = {dgamma, dbeta, dbias}.
template <typename T, int NV, int G, int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ dz, float* __restrict__ part, int64_t npix, int c) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    // G = 128 (4 096 channels in 16 bits: the bottleneck of a depth-6 model): TWO waves own a pixel, so that a thread keeps the
    // 32 channels of the 2 048-channel case (with one wave per pixel and NV = 8 the five per-channel register arrays alone
    // are 320 registers: that instantiation spilled 84-92 registers in r03).  The per-pixel sums then cross the wave pair through
    // LDS, which needs the same number of barriers in every thread: the pixel loop runs a uniform trip count for G = 128.
    constexpr bool PAIR = G == 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [PPB][3][c]
    __shared__ float xch[2][4];
    const int tid = threadIdx.x;
    const int gl = tid % G, gp = tid / G;
    float gam[NV][EPT], bet[NV][EPT];
    float a_g[NV][EPT], a_b[NV][EPT], a_z[NV][EPT];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int ch = (v * G + gl) * EPT + e;
            gam[v][e] = MODE == 2 ? 1.f : gamma[ch];
            bet[v][e] = MODE == 2 ? 0.f : beta[ch];
            a_g[v][e] = a_b[v][e] = a_z[v][e] = 0.f;
        }
    const float inv_c = 1.0f / (float)c;
    // U pixels per thread and pass with all their loads issued before the first use.  Measured on the full-resolution
    // 64-channel layers (K2', batch 64): U = 2 is SLOWER than U = 1 (0.345 vs 0.319 ms per launch) -- the four resident
    // blocks per CU already keep the memory system busy and the extra registers cost occupancy -- so U stays 1.
    constexpr int U = 1;
    const int64_t step = (int64_t)gridDim.x * PPB;
    for (int64_t pix0 = (int64_t)blockIdx.x * PPB + gp; PAIR ? pix0 - gp < npix : pix0 < npix; pix0 += U * step) {
        Vec16<T> lz[U][NV], ld[U][NV];
        float mu_u[U], rs_u[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // past the end: re-read, never stored (PAIR: the second pixel group of the last pass re-reads the first group's pixel)
            const int64_t pix = pix0 + u * step < npix ? pix0 + u * step : (PAIR ? pix0 - gp : pix0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                lz[u][v].load(z + pix * c + (v * G + gl) * EPT);
                ld[u][v].load(dy + pix * c + (v * G + gl) * EPT);
            }
            mu_u[u] = MODE != 2 ? mean[pix] : 0.f;
            rs_u[u] = MODE != 2 ? rstd[pix] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pix = pix0 + u * step;
            const bool live = pix < npix;
            if (!PAIR && !live) break;
            float xh[NV][EPT], g[NV][EPT];
            const float mu = mu_u[u], rs = rs_u[u];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float zz[EPT], dd[EPT];
                lz[u][v].to_f32(zz);
                ld[u][v].to_f32(dd);
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    if (MODE == 2) {
                        g[v][e] = zz[e] > 0.f ? dd[e] : 0.f;  // dz directly
                    } else {
                        float h = (zz[e] - mu) * rs;
                        float yv = h * gam[v][e] + bet[v][e];
                        float dl = (MODE == 0 && !(yv > 0.f)) ? 0.f : dd[e];
                        if (PAIR && !live) dl = 0.f;
                        xh[v][e] = h;
                        a_g[v][e] += dl * h;
                        a_b[v][e] += dl;
                        float gg = dl * gam[v][e];
                        g[v][e] = gg;
                        s1 += gg;
                        s2 += gg * h;
                    }
                }
            }
            if (MODE != 2) {
                if constexpr (PAIR) {
                    s1 = group_sum<64>(s1);
                    s2 = group_sum<64>(s2);
                    if ((tid & 63) == 0) { xch[0][tid >> 6] = s1; xch[1][tid >> 6] = s2; }
                    __syncthreads();
                    s1 = (xch[0][2 * gp] + xch[0][2 * gp + 1]) * inv_c;
                    s2 = (xch[1][2 * gp] + xch[1][2 * gp + 1]) * inv_c;
                    __syncthreads();
                } else {
                    s1 = group_sum<G>(s1) * inv_c;
                    s2 = group_sum<G>(s2) * inv_c;
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float o[EPT];
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    float d = MODE == 2 ? g[v][e] : rs * (g[v][e] - s1 - xh[v][e] * s2);
                    o[e] = d;
                }
                Vec16<T> st;
                st.from_f32(o);
                if (!PAIR || live) st.store(dz + pix * c + (v * G + gl) * EPT);
                float back[EPT];  // dbias sums the values as stored (what the conv wgrad sees)
                st.to_f32(back);
#pragma unroll
                for (int e = 0; e < EPT; ++e) a_z[v][e] += (PAIR && !live) ? 0.f : back[e];
            }
        }
    }
    // block reduction over the PPB pixel groups (fixed order => deterministic)
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int ch = (v * G + gl) * EPT + e;
            red[(gp * 3 + 0) * c + ch] = a_g[v][e];
            red[(gp * 3 + 1) * c + ch] = a_b[v][e];
            red[(gp * 3 + 2) * c + ch] = a_z[v][e];
        }
    __syncthreads();
    for (int i = tid; i < 3 * c; i += 256) {
        float s = 0.f;
        for (int p = 0; p < PPB; ++p) s += red[p * 3 * c + i];
        part[(size_t)blockIdx.x * 3 * c + i] = s;
    }
}

// Sum the per-block partials [nblocks][3][c] in a fixed order: 4 columns x 64 row-groups per block (the job is
// latency bound -- a few hundred KB -- so the rows are spread over many lanes and folded through LDS).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int nblocks, int c,
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
        int which = i / c, ch = i % c;
        float* dst = which == 0 ? o0 : (which == 1 ? o1 : o2);
        if (dst) dst[ch] = t;
    }
}

struct RowCfg {
    int nv, g;
};

static bool row_cfg(int c, int ept, RowCfg* r, bool bwd = false) {
    if (c <= 0 || c % ept) return false;
    int vecs = c / ept;
    if (bwd && ept == 8 && vecs == 512) { r->nv = 4; r->g = 128; return true; }     // two waves per pixel (ln_bwd_kernel, PAIR)
    int nv = (vecs + 63) / 64;
    if (nv != 1 && nv != 2 && nv != 4 && nv != 8) return false;
    if (vecs % nv) return false;
    int g = vecs / nv;
    if (g != 1 && g != 2 && g != 4 && g != 8 && g != 16 && g != 32 && g != 64) return false;
    r->nv = nv;
    r->g = g;
    return true;
}

static int bwd_blocks(int64_t npix, int g) {
    int64_t ppb = 256 / g;
    int64_t nb = (npix + ppb - 1) / ppb;
    return (int)(nb < NB_MAX ? nb : NB_MAX);
}

#define DISPATCH_G(...)                                   \
    switch (cfg.g) {                                      \
        case 1: { constexpr int G_ = 1; __VA_ARGS__ } break;   \
        case 2: { constexpr int G_ = 2; __VA_ARGS__ } break;   \
        case 4: { constexpr int G_ = 4; __VA_ARGS__ } break;   \
        case 8: { constexpr int G_ = 8; __VA_ARGS__ } break;   \
        case 16: { constexpr int G_ = 16; __VA_ARGS__ } break; \
        case 32: { constexpr int G_ = 32; __VA_ARGS__ } break; \
        default: { constexpr int G_ = 64; __VA_ARGS__ } break; \
    }

#define DISPATCH_NVG(...)                                                          \
    switch (cfg.nv) {                                                              \
        case 1: { constexpr int NV_ = 1; DISPATCH_G(__VA_ARGS__) } break;          \
        case 2: { constexpr int NV_ = 2; constexpr int G_ = 64; __VA_ARGS__ } break; \
        case 4: { constexpr int NV_ = 4; constexpr int G_ = 64; __VA_ARGS__ } break; \
        default: { constexpr int NV_ = 8; constexpr int G_ = 64; __VA_ARGS__ } break; \
    }

// backward: 8 vectors per thread exist for fp32 only (32 channels per thread; 16-bit rows of that width take g = 128, nv = 4)
#define DISPATCH_NVG_BWD(...)                                                      \
    switch (cfg.nv) {                                                              \
        case 1: { constexpr int NV_ = 1; DISPATCH_G(__VA_ARGS__) } break;          \
        case 2: { constexpr int NV_ = 2; constexpr int G_ = 64; __VA_ARGS__ } break; \
        case 4: { constexpr int NV_ = 4;                                           \
                  if (cfg.g == 128) { if constexpr (sizeof(T) == 2) { constexpr int G_ = 128; __VA_ARGS__ } }       \
                  else { constexpr int G_ = 64; __VA_ARGS__ } } break;             \
        default: { if constexpr (sizeof(T) == 4) { constexpr int NV_ = 8; constexpr int G_ = 64; __VA_ARGS__ } } break; \
    }

template <typename T>
int ln_fwd_launch(const void* z, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                  int64_t npix, int c, float eps, int relu, hipStream_t s) {
    RowCfg cfg;
    if (!row_cfg(c, ElemTraits<T>::EPT, &cfg)) return ad_set_error(AD_ERR_ARG, "layernorm: unsupported c=%d", c);
    int64_t ppb = 256 / cfg.g;
    int64_t nb = (npix + ppb - 1) / ppb;
    int blocks = (int)(nb < 8192 ? nb : 8192);
    DISPATCH_NVG(ln_fwd_kernel<T, NV_, G_><<<blocks, 256, 0, s>>>((const T*)z, gamma, beta, (T*)y, mean, rstd, npix, c,
                                                                 eps, relu);)
    AD_LAUNCH_CHECK("ad_layernorm_relu_fwd");
    return AD_OK;
}

template <typename T, int MODE>
int ln_bwd_launch(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, void* dz, float* dgamma, float* dbeta, float* dbias, int64_t npix, int c,
                  void* ws, size_t ws_bytes, hipStream_t s) {
    RowCfg cfg;
    if (!row_cfg(c, ElemTraits<T>::EPT, &cfg, true)) return ad_set_error(AD_ERR_ARG, "layernorm bwd: unsupported c=%d", c);
    int blocks = bwd_blocks(npix, cfg.g);
    size_t need = (size_t)blocks * 3 * c * sizeof(float);
    if (!ws || ws_bytes < need) return ad_set_error(AD_ERR_WS, "layernorm bwd: workspace %zu < %zu", ws_bytes, need);
    size_t lds = (size_t)(256 / cfg.g) * 3 * c * sizeof(float);
    if (lds > 160 * 1024 - 64) return ad_set_error(AD_ERR_ARG, "layernorm bwd: c=%d needs %zu B LDS", c, lds);
    DISPATCH_NVG_BWD(
        auto kern = ln_bwd_kernel<T, NV_, G_, MODE>;
        if (lds > 64 * 1024)       // (dynamic + the kernel's 32 static bytes must stay within the CU's 160 KB)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        kern<<<blocks, 256, lds, s>>>((const T*)dy, (const T*)z, mean, rstd, gamma, beta, (T*)dz, (float*)ws, npix, c);)
    AD_LAUNCH_CHECK("layernorm bwd");
    colsum_reduce_kernel<<<(3 * c + 3) / 4, 256, 0, s>>>((const float*)ws, blocks, c, dgamma, dbeta, dbias);
    AD_LAUNCH_CHECK("colsum_reduce");
    return AD_OK;
}

}  // namespace

extern "C" int ad_layernorm_relu_fwd(const void* z, const float* gamma, const float* beta, void* y, float* mean,
                                     float* rstd, int64_t npix, int c, float eps, int relu, int dtype, void* stream) {
    if (npix <= 0) return AD_OK;
    hipStream_t s = (hipStream_t)stream;
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_layernorm_relu_fwd: bad dtype %d", dtype);
    AD_DISPATCH_DTYPE(dtype, T_, return ln_fwd_launch<T_>(z, gamma, beta, y, mean, rstd, npix, c, eps, relu, s);)
    return AD_OK;
}

extern "C" size_t ad_layernorm_bwd_ws_bytes(int64_t npix, int c) {
    (void)npix;
    return (size_t)NB_MAX * 3 * c * sizeof(float);
}

extern "C" int ad_layernorm_relu_bwd(const void* dy, const void* z, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta,
                                     float* dbias, int64_t npix, int c, int relu, void* ws, size_t ws_bytes, int dtype,
                                     void* stream) {
    AD_REQUIRE(npix > 0, "ad_layernorm_relu_bwd: npix=%ld", (long)npix);
    hipStream_t s = (hipStream_t)stream;
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_layernorm_relu_bwd: bad dtype %d", dtype);
    AD_DISPATCH_DTYPE(dtype, T_,
        return relu ? ln_bwd_launch<T_, 0>(dy, z, mean, rstd, gamma, beta, dz, dgamma, dbeta, dbias, npix, c, ws, ws_bytes, s)
                    : ln_bwd_launch<T_, 1>(dy, z, mean, rstd, gamma, beta, dz, dgamma, dbeta, dbias, npix, c, ws, ws_bytes, s);)
    return AD_OK;
}

extern "C" int ad_relu_bwd(const void* dy, const void* y, void* dz, float* dbias, int64_t npix, int c, void* ws,
                           size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(npix > 0, "ad_relu_bwd: npix=%ld", (long)npix);
    hipStream_t s = (hipStream_t)stream;
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_relu_bwd: bad dtype %d", dtype);
    AD_DISPATCH_DTYPE(dtype, T_,
        return ln_bwd_launch<T_, 2>(dy, y, nullptr, nullptr, nullptr, nullptr, dz, nullptr, nullptr, dbias, npix, c, ws, ws_bytes, s);)
    return AD_OK;
}
