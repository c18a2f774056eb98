// Fused residual head: 1x1 Conv2D (64 -> 3, "residual_rgb", train_adaptive_unet.py:267-274)
// + ClippedResidualAdd (shared/custom_layers.py:136-139) + Charbonnier / L1 loss and the per-image
// squared error behind tf.image.psnr (train_adaptive_unet.py:308-334), forward and backward.
// Arithmetic intensity ~3 FLOP/B => HBM-bound: one pass over x_head, G = ch/EPT lanes per pixel.
#include "common.h"

namespace {

constexpr int BPI_MAX = 32;  // blocks per image

template <int G>
__device__ __forceinline__ float gsum(float v) { return ad_group_sum<G>(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int G>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ xh, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* __restrict__ inp,
                                                       const float* __restrict__ target, float* __restrict__ out,
                                                       float* __restrict__ part, int64_t ppi, int ch, int loss_kind,
                                                       float eps) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    __shared__ float sm[2][4];
    const int tid = threadIdx.x, gl = tid % G, gp = tid / G;
    const int img = blockIdx.y;
    float wl[EPT][3];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
#pragma unroll
        for (int o = 0; o < 3; ++o) wl[e][o] = w[(gl * EPT + e) * 3 + o];
    const float bo = gl < 3 ? b[gl] : 0.f;
    float lsum = 0.f, qsum = 0.f;
    // U pixels per thread and pass with their 16-byte loads issued together: measured neutral-to-worse in this kernel
    // (0.195 -> 0.234 ms at U = 4; the backward kernel gains: 0.363 -> 0.287 ms), so the forward keeps U = 1
    constexpr int U = 1;
    const int64_t step = (int64_t)gridDim.x * PPB;
    for (int64_t q0 = (int64_t)blockIdx.x * PPB + gp; q0 < ppi; q0 += U * step) {
        Vec16<T> ld[U];
        float inp_v[U], tgt_v[U];     // fetched with the activations, not after the reductions that need them last
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * step;
            if (q < ppi) ld[u].load(xh + ((int64_t)img * ppi + q) * ch + gl * EPT); else ld[u].zero();
            const int64_t pixl = (int64_t)img * ppi + (q < ppi ? q : q0);
            inp_v[u] = gl < 3 ? inp[pixl * 3 + gl] : 0.f;
            tgt_v[u] = gl < 3 && target ? target[pixl * 3 + gl] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * step;
            if (q >= ppi) break;
            const int64_t pix = (int64_t)img * ppi + q;
            float x[EPT];
            ld[u].to_f32(x);
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                r0 += x[e] * wl[e][0];
                r1 += x[e] * wl[e][1];
                r2 += x[e] * wl[e][2];
            }
            r0 = gsum<G>(r0); r1 = gsum<G>(r1); r2 = gsum<G>(r2);
            if (gl < 3) {
                float r = (gl == 0 ? r0 : (gl == 1 ? r1 : r2)) + bo;
                float pre = inp_v[u] + r;
                float o = fminf(fmaxf(pre, 0.f), 1.f);
                out[pix * 3 + gl] = o;
                if (target) {
                    float d = tgt_v[u] - o;
                    lsum += loss_kind == 0 ? sqrtf(d * d + eps * eps) : fabsf(d);
                    qsum += d * d;
                }
            }
        }
    }
    lsum = wave_sum(lsum);
    qsum = wave_sum(qsum);
    if ((tid & 63) == 0) { sm[0][tid >> 6] = lsum; sm[1][tid >> 6] = qsum; }
    __syncthreads();
    if (tid == 0 && part) {
        size_t slot = (size_t)img * gridDim.x + blockIdx.x;
        part[slot * 2 + 0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        part[slot * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    }
}

// stats[0] = sum of all loss partials, stats[2] = their mean over all elements; stats[1] = mean over the images of tf.image.psnr(max_val = 1) = -float32(10 / ln 10) ln(MSE)
// (+inf at MSE 0, as TensorFlow returns); sqerr[img] = sum of that image's squared-error partials
__global__ __launch_bounds__(256) void head_stats_kernel(const float* __restrict__ part, int n, int bpi,
                                                         float* __restrict__ stats, float* __restrict__ sqerr,
                                                         float elems_per_img) {
    __shared__ float sm[256];
    __shared__ float sp[256];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < n * bpi; i += 256) s += part[(size_t)i * 2];
    float ps = 0.f;
    for (int img = tid; img < n; img += 256) {
        float q = 0.f;
        for (int k = 0; k < bpi; ++k) q += part[((size_t)img * bpi + k) * 2 + 1];
        if (sqerr) sqerr[img] = q;
        ps += -4.3429448190325175f * logf(q / elems_per_img);       // tf.image.psnr: -float32(10 / ln 10) * ln(mse)
    }
    sm[tid] = s;
    sp[tid] = ps;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sm[tid] += sm[tid + o]; sp[tid] += sp[tid + o]; }
        __syncthreads();
    }
    if (tid == 0 && stats) { stats[0] = sm[0]; stats[1] = sp[0] / (float)n; stats[2] = sm[0] / ((float)n * elems_per_img); }
}

// head_stats_kernel for the partials head_ln_bwd_kernel leaves in columns col0, col0 + 1 of its [n * bpi][ncol] table
__global__ __launch_bounds__(256) void head_stats_strided_kernel(const float* __restrict__ part, int n, int bpi, int ncol, int col0,
                                                                 float* __restrict__ stats, float* __restrict__ sqerr,
                                                                 float elems_per_img) {
    __shared__ float sm[256];
    __shared__ float sp[256];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < n * bpi; i += 256) s += part[(size_t)i * ncol + col0];
    float ps = 0.f;
    for (int img = tid; img < n; img += 256) {
        float q = 0.f;
        for (int k = 0; k < bpi; ++k) q += part[((size_t)img * bpi + k) * ncol + col0 + 1];
        if (sqerr) sqerr[img] = q;
        ps += -4.3429448190325175f * logf(q / elems_per_img);       // tf.image.psnr: -float32(10 / ln 10) * ln(mse)
    }
    sm[tid] = s;
    sp[tid] = ps;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sm[tid] += sm[tid + o]; sp[tid] += sp[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { stats[0] = sm[0]; stats[1] = sp[0] / (float)n; stats[2] = sm[0] / ((float)n * elems_per_img); }
}

template <typename T, int G>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ xh, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* __restrict__ inp,
                                                       const float* __restrict__ target, T* __restrict__ dxh,
                                                       float* __restrict__ part, int64_t ppi, int ch, int loss_kind,
                                                       float eps, float gscale_host, const float* __restrict__ loss_scale) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [PPB][ch*3+3]
    const float gscale = loss_scale ? gscale_host * loss_scale[0] : gscale_host;   // dynamic loss scale (fp16), device resident
    const int ncol = ch * 3 + 3;
    const int tid = threadIdx.x, gl = tid % G, gp = tid / G;
    const int img = blockIdx.y;
    float wl[EPT][3], aw[EPT][3], ab[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < EPT; ++e)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            wl[e][o] = w[(gl * EPT + e) * 3 + o];
            aw[e][o] = 0.f;
        }
    const float b0 = b[0], b1 = b[1], b2 = b[2];
    constexpr int U = 4;                         // pixels in flight per thread (see head_fwd_kernel)
    const int64_t step = (int64_t)gridDim.x * PPB;
    for (int64_t q0 = (int64_t)blockIdx.x * PPB + gp; q0 < ppi; q0 += U * step) {
        Vec16<T> ldv[U];
        float in3[U][3], tg3[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * step;
            const int64_t pix = (int64_t)img * ppi + (q < ppi ? q : q0);
            ldv[u].load(xh + pix * ch + gl * EPT);
#pragma unroll
            for (int o = 0; o < 3; ++o) { in3[u][o] = inp[pix * 3 + o]; tg3[u][o] = target[pix * 3 + o]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * step;
            if (q >= ppi) break;
            const int64_t pix = (int64_t)img * ppi + q;
            float x[EPT];
            ldv[u].to_f32(x);
            float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                r[0] += x[e] * wl[e][0];
                r[1] += x[e] * wl[e][1];
                r[2] += x[e] * wl[e][2];
            }
            r[0] = gsum<G>(r[0]) + b0; r[1] = gsum<G>(r[1]) + b1; r[2] = gsum<G>(r[2]) + b2;
            float g[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float pre = in3[u][o] + r[o];
                float ov = fminf(fmaxf(pre, 0.f), 1.f);
                float d = tg3[u][o] - ov;
                float dl = loss_kind == 0 ? -d * rsqrtf(d * d + eps * eps) : (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f));
                g[o] = (pre >= 0.f && pre <= 1.f) ? dl * gscale : 0.f;
            }
            float dx[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                dx[e] = g[0] * wl[e][0] + g[1] * wl[e][1] + g[2] * wl[e][2];
#pragma unroll
                for (int o = 0; o < 3; ++o) aw[e][o] += x[e] * g[o];
            }
#pragma unroll
            for (int o = 0; o < 3; ++o) ab[o] += g[o];
            Vec16<T> st;
            st.from_f32(dx);
            st.store(dxh + pix * ch + gl * EPT);
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e)
#pragma unroll
        for (int o = 0; o < 3; ++o) red[gp * ncol + (gl * EPT + e) * 3 + o] = aw[e][o];
    if (gl == 0) {
#pragma unroll
        for (int o = 0; o < 3; ++o) red[gp * ncol + ch * 3 + o] = ab[o];
    }
    __syncthreads();
    for (int i = tid; i < ncol; i += 256) {
        float s = 0.f;
        for (int p = 0; p < PPB; ++p) s += red[p * ncol + i];
        part[((size_t)img * gridDim.x + blockIdx.x) * ncol + i] = s;
    }
}

// head_bwd_kernel and the LayerNorm + ReLU backward of the layer that feeds the head (conv_block's last
// Conv2D -> LayerNormalization -> ReLU, train_adaptive_unet.py:265) in ONE pass: the gradient of the head activations
// never goes to memory (it stays fp32 in registers), so the pair "write dxh, read dxh + z, write dz" becomes "read z,
// write dz".  Same lane layout as both kernels: G = ch / EPT lanes own a pixel's channels.
// part[block][ch*3 + 3 + 3*ch] = { dW head, db head, dgamma, dbeta, dbias of the conv in front of the LayerNorm }
#ifndef AD_HEAD_LN_U
#define AD_HEAD_LN_U 1
#endif
template <typename T, int G, bool REDERIVE>        // REDERIVE: xh == NULL, the head's input is re-derived from z
__global__ __launch_bounds__(256) void head_ln_bwd_kernel(const T* __restrict__ xh, const float* __restrict__ w,
                                                          const float* __restrict__ b, const float* __restrict__ inp,
                                                          const float* __restrict__ target, const T* __restrict__ z,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ dz, float* __restrict__ part, int64_t ppi, int ch,
                                                          int loss_kind, float eps, float gscale_host,
                                                          const float* __restrict__ loss_scale, int want_stats) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [PPB][ncol]
    const float gscale = loss_scale ? gscale_host * loss_scale[0] : gscale_host;
    // want_stats: two more columns, the block's loss and squared-error partials (what head_fwd_kernel reports) -- the kernel
    // re-derives the head's output anyway, so a train step needs no forward pass over the head at all
    const int ncol = ch * 6 + 3 + (want_stats ? 2 : 0);
    float a_l = 0.f, a_q = 0.f;
    const int tid = threadIdx.x, gl = tid % G, gp = tid / G;
    const int img = blockIdx.y;
    float wl[EPT][3], aw[EPT][3], ab[3] = {0.f, 0.f, 0.f};
    float gam[EPT], bet[EPT], a_g[EPT], a_b[EPT], a_z[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            wl[e][o] = w[(gl * EPT + e) * 3 + o];
            aw[e][o] = 0.f;
        }
        gam[e] = gamma[gl * EPT + e]; bet[e] = beta[gl * EPT + e];
        a_g[e] = a_b[e] = a_z[e] = 0.f;
    }
    const float b0 = b[0], b1 = b[1], b2 = b[2];
    const float inv_c = 1.0f / (float)ch;
    // U pixels per thread and pass with every load issued before the first use: at one pixel the kernel's 166 registers
    // allow three waves per SIMD = 6 MB in flight on the chip, which bounds it at ~3.9 TB/s (latency x bandwidth)
#ifndef AD_HEAD_LN_UR
#define AD_HEAD_LN_UR AD_HEAD_LN_U
#endif
    constexpr int U = REDERIVE ? AD_HEAD_LN_UR : AD_HEAD_LN_U;
    const int64_t qstep = (int64_t)gridDim.x * PPB;
    for (int64_t q0 = (int64_t)blockIdx.x * PPB + gp; q0 < ppi; q0 += U * qstep) {
        Vec16<T> lxs[U], lzs[U];
        float mus[U], rss[U], inps[U][3], tgts[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t qq = q0 + u * qstep < ppi ? q0 + u * qstep : q0;        // past the end: re-read, never stored
            const int64_t pix = (int64_t)img * ppi + qq;
            if constexpr (!REDERIVE) lxs[u].load(xh + pix * ch + gl * EPT);
            lzs[u].load(z + pix * ch + gl * EPT);
            mus[u] = mean[pix]; rss[u] = rstd[pix];
#pragma unroll
            for (int o = 0; o < 3; ++o) { inps[u][o] = inp[pix * 3 + o]; tgts[u][o] = target[pix * 3 + o]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
        const int64_t q = q0 + u * qstep;
        if (q >= ppi) break;
        const int64_t pix = (int64_t)img * ppi + q;
        const float mu = mus[u], rs = rss[u];
        float x[EPT], zz[EPT], h[EPT];
        unsigned pos = 0;                    // REDERIVE: bit e = the LayerNorm output of channel e is positive (the ReLU mask)
        lzs[u].to_f32(zz);
        if constexpr (REDERIVE) {            // the head's input as a stored activation tensor would hold it
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                h[e] = (zz[e] - mu) * rs;
                const float yv = h[e] * gam[e] + bet[e];
                pos |= yv > 0.f ? 1u << e : 0u;
                x[e] = yv > 0.f ? yv : 0.f;
            }
            Vec16<T> ra;
            ra.from_f32(x);
            ra.to_f32(x);
        } else {
            lxs[u].to_f32(x);
        }
        // the head's three outputs of this lane's channels on float pairs (even / odd channels in the two halves, folded at the
        // end): 12 packed multiply-adds where hipcc left 24 scalar ones (r05: the kernel is bound by its vector instructions)
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        f32x2_ rp[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int e = 0; e < EPT; e += 2) {
            const f32x2_ xp = {x[e], x[e + 1]};
#pragma unroll
            for (int o = 0; o < 3; ++o) rp[o] = __builtin_elementwise_fma(xp, f32x2_{wl[e][o], wl[e + 1][o]}, rp[o]);
        }
        float r[3];
        r[0] = gsum<G>(rp[0].x + rp[0].y) + b0; r[1] = gsum<G>(rp[1].x + rp[1].y) + b1; r[2] = gsum<G>(rp[2].x + rp[2].y) + b2;
        float g[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float pre = inps[u][o] + r[o];
            float ov = fminf(fmaxf(pre, 0.f), 1.f);
            float d = tgts[u][o] - ov;
            float dl = loss_kind == 0 ? -d * rsqrtf(d * d + eps * eps) : (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f));
            g[o] = (pre >= 0.f && pre <= 1.f) ? dl * gscale : 0.f;
            if (want_stats) {        // (every lane of the pixel's group holds the same three values: lane 0 reports them)
                // v_sqrt_f32 (1 ulp; the argument is >= eps^2, far from the denormals whose handling makes sqrtf() a dozen
                // instructions): the reported loss moves by ~1e-7 relative, the gradient does not use it
                a_l += loss_kind == 0 ? __builtin_amdgcn_sqrtf(d * d + eps * eps) : fabsf(d);
                a_q += d * d;
            }
        }
        float gg[EPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float da = g[0] * wl[e][0] + g[1] * wl[e][1] + g[2] * wl[e][2];       // gradient of the head activation
#pragma unroll
            for (int o = 0; o < 3; ++o) aw[e][o] += x[e] * g[o];
            bool on;
            if constexpr (REDERIVE) {
                on = (pos >> e) & 1u;
            } else {
                h[e] = (zz[e] - mu) * rs;
                on = h[e] * gam[e] + bet[e] > 0.f;
            }
            const float dl = on ? da : 0.f;                                              // ReLU
            a_g[e] += dl * h[e];
            a_b[e] += dl;
            gg[e] = dl * gam[e];
            s1 += gg[e];
            s2 += gg[e] * h[e];
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) ab[o] += g[o];
        s1 = gsum<G>(s1) * inv_c;
        s2 = gsum<G>(s2) * inv_c;
        float o8[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) o8[e] = rs * (gg[e] - s1 - h[e] * s2);
        Vec16<T> st;
        st.from_f32(o8);
        st.store(dz + pix * ch + gl * EPT);
        float back[EPT];                     // the conv's bias gradient sums dz as stored (what its wgrad sees)
        st.to_f32(back);
#pragma unroll
        for (int e = 0; e < EPT; ++e) a_z[e] += back[e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = gl * EPT + e;
#pragma unroll
        for (int o = 0; o < 3; ++o) red[gp * ncol + c * 3 + o] = aw[e][o];
        red[gp * ncol + ch * 3 + 3 + c] = a_g[e];
        red[gp * ncol + ch * 4 + 3 + c] = a_b[e];
        red[gp * ncol + ch * 5 + 3 + c] = a_z[e];
    }
    if (gl == 0) {
#pragma unroll
        for (int o = 0; o < 3; ++o) red[gp * ncol + ch * 3 + o] = ab[o];
        if (want_stats) { red[gp * ncol + ch * 6 + 3] = a_l; red[gp * ncol + ch * 6 + 4] = a_q; }
    }
    __syncthreads();
    for (int i = tid; i < ncol; i += 256) {
        float s = 0.f;
        for (int p = 0; p < PPB; ++p) s += red[p * ncol + i];
        part[((size_t)img * gridDim.x + blockIdx.x) * ncol + i] = s;
    }
}

// out_k[col - start_k] = sum over rows of part[row][col] for the column ranges of up to five outputs, fixed order
struct HeadLnOuts {
    float* ptr[5];
    int end[5];      // exclusive column end of each output
};
__global__ __launch_bounds__(256) void rows_reduce5_kernel(const float* __restrict__ part, int nrows, int stride, int ncols,
                                                           HeadLnOuts o) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int i = blockIdx.x * 4 + cl;
    float s = 0.f;
    if (i < ncols)                       // (the first `ncols` of `stride` columns per row)
    {
#pragma unroll 8
        for (int r = rg; r < nrows; r += 64) s += part[(size_t)r * stride + i];
    }
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < ncols) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        int k = 0, start = 0;
        while (i >= o.end[k]) { start = o.end[k]; ++k; }
        o.ptr[k][i - start] = t;
    }
}

// out[col] = sum over rows of part[row][col], fixed order (4 columns x 64 row-groups per block)
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ part, int nrows, int ncols,
                                                          float* __restrict__ o0, int n0, float* __restrict__ o1) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int i = blockIdx.x * 4 + cl;
    float s = 0.f;
    if (i < ncols)
    {
#pragma unroll 8
        for (int r = rg; r < nrows; r += 64) s += part[(size_t)r * ncols + i];
    }
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < ncols) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        if (i < n0) o0[i] = t; else o1[i - n0] = t;
    }
}

static int head_bpi(int64_t ppi, int g) {
    int64_t ppb = 256 / g;
    int64_t nb = (ppi + ppb - 1) / ppb;
    return (int)(nb < BPI_MAX ? nb : BPI_MAX);
}

#define HEAD_DISPATCH(...)                                         \
    switch (g) {                                                   \
        case 4: { constexpr int G_ = 4; __VA_ARGS__ } break;       \
        case 8: { constexpr int G_ = 8; __VA_ARGS__ } break;       \
        case 16: { constexpr int G_ = 16; __VA_ARGS__ } break;     \
        case 32: { constexpr int G_ = 32; __VA_ARGS__ } break;     \
        default: { constexpr int G_ = 64; __VA_ARGS__ } break;     \
    }

static bool head_group(int ch, int ept, int* g) {
    if (ch <= 0 || ch % ept) return false;
    int v = ch / ept;
    if (v != 4 && v != 8 && v != 16 && v != 32 && v != 64) return false;
    *g = v;
    return true;
}

}  // namespace

extern "C" size_t ad_head_ws_bytes(int n, int ch) {
    return (size_t)n * BPI_MAX * (ch * 3 + 3) * sizeof(float);
}

extern "C" int ad_head_fwd(const void* xh, const float* w, const float* b, const float* inp, const float* target,
                           float* out, float* stats, float* sqerr, int n, int64_t pix_per_img, int ch, int loss_kind,
                           float eps, void* ws, size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_head_fwd: bad dtype %d", dtype);
    AD_REQUIRE(n > 0 && pix_per_img > 0, "ad_head_fwd: bad shape");
    AD_REQUIRE(loss_kind == 0 || loss_kind == 1, "ad_head_fwd: loss_kind=%d", loss_kind);
    int g;
    AD_REQUIRE(head_group(ch, ad_is_half(dtype) ? 8 : 4, &g), "ad_head_fwd: unsupported ch=%d", ch);
    const int bpi = head_bpi(pix_per_img, g);
    float* part = nullptr;
    if (target) {
        size_t need = (size_t)n * bpi * 2 * sizeof(float);
        if (!ws || ws_bytes < need) return ad_set_error(AD_ERR_WS, "ad_head_fwd: workspace %zu < %zu", ws_bytes, need);
        part = (float*)ws;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bpi, n);
    AD_DISPATCH_DTYPE(dtype, T_,
        HEAD_DISPATCH(head_fwd_kernel<T_, G_><<<grid, 256, 0, s>>>((const T_*)xh, w, b, inp, target, out, part, pix_per_img, ch,
                                                                   loss_kind, eps);))
    AD_LAUNCH_CHECK("ad_head_fwd");
    if (target) {
        head_stats_kernel<<<1, 256, 0, s>>>(part, n, bpi, stats, sqerr, (float)pix_per_img * 3.f);
        AD_LAUNCH_CHECK("head_stats");
    }
    return AD_OK;
}

extern "C" int ad_head_bwd(const void* xh, const float* w, const float* b, const float* inp, const float* target,
                           void* dxh, float* dw, float* db, int n, int64_t pix_per_img, int ch, int loss_kind, float eps,
                           float grad_scale, const float* loss_scale, void* ws, size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_head_bwd: bad dtype %d", dtype);
    AD_REQUIRE(n > 0 && pix_per_img > 0 && target, "ad_head_bwd: bad shape / missing target");
    AD_REQUIRE(loss_kind == 0 || loss_kind == 1, "ad_head_bwd: loss_kind=%d", loss_kind);
    int g;
    AD_REQUIRE(head_group(ch, ad_is_half(dtype) ? 8 : 4, &g), "ad_head_bwd: unsupported ch=%d", ch);
    const int bpi = head_bpi(pix_per_img, g);
    const int ncol = ch * 3 + 3;
    size_t need = (size_t)n * bpi * ncol * sizeof(float);
    if (!ws || ws_bytes < need) return ad_set_error(AD_ERR_WS, "ad_head_bwd: workspace %zu < %zu", ws_bytes, need);
    size_t lds = (size_t)(256 / g) * ncol * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bpi, n);
    AD_DISPATCH_DTYPE(dtype, T_,
        HEAD_DISPATCH(head_bwd_kernel<T_, G_><<<grid, 256, lds, s>>>((const T_*)xh, w, b, inp, target, (T_*)dxh, (float*)ws,
                                                                     pix_per_img, ch, loss_kind, eps, grad_scale, loss_scale);))
    AD_LAUNCH_CHECK("ad_head_bwd");
    rows_reduce_kernel<<<(ncol + 3) / 4, 256, 0, s>>>((const float*)ws, n * bpi, ncol, dw, ch * 3, db);
    AD_LAUNCH_CHECK("head rows_reduce");
    return AD_OK;
}


extern "C" size_t ad_head_ln_bwd_ws_bytes(int n, int ch) { return (size_t)n * BPI_MAX * (ch * 6 + 5) * sizeof(float); }

extern "C" int ad_head_ln_bwd(const void* xh, const float* w, const float* b, const float* inp, const float* target,
                              const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              void* dz, float* dw, float* db, float* dgamma, float* dbeta, float* dbias_conv, int n,
                              int64_t pix_per_img, int ch, int loss_kind, float eps, float grad_scale,
                              const float* loss_scale, float* stats, float* sqerr, void* ws, size_t ws_bytes, int dtype,
                              void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_head_ln_bwd: bad dtype %d", dtype);
    AD_REQUIRE(stats || !sqerr, "ad_head_ln_bwd: sqerr without stats");
    AD_REQUIRE(n > 0 && pix_per_img > 0 && target && z && mean && rstd && gamma && beta && dz && dgamma && dbeta && dbias_conv,
               "ad_head_ln_bwd: bad shape / missing operand");
    AD_REQUIRE(loss_kind == 0 || loss_kind == 1, "ad_head_ln_bwd: loss_kind=%d", loss_kind);
    int g;
    AD_REQUIRE(head_group(ch, ad_is_half(dtype) ? 8 : 4, &g), "ad_head_ln_bwd: unsupported ch=%d", ch);
    const int bpi = head_bpi(pix_per_img, g);
    const int ncol = ch * 6 + 3 + (stats ? 2 : 0);
    size_t need = (size_t)n * bpi * ncol * sizeof(float);
    if (!ws || ws_bytes < need) return ad_set_error(AD_ERR_WS, "ad_head_ln_bwd: workspace %zu < %zu", ws_bytes, need);
    size_t lds = (size_t)(256 / g) * ncol * sizeof(float);
    if (lds > 64 * 1024) return ad_set_error(AD_ERR_ARG, "ad_head_ln_bwd: ch=%d needs %zu B of LDS", ch, lds);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bpi, n);
    if (xh) {
        AD_DISPATCH_DTYPE(dtype, T_,
            HEAD_DISPATCH(head_ln_bwd_kernel<T_, G_, false><<<grid, 256, lds, s>>>((const T_*)xh, w, b, inp, target, (const T_*)z, mean,
                                                                                   rstd, gamma, beta, (T_*)dz, (float*)ws, pix_per_img, ch,
                                                                                   loss_kind, eps, grad_scale, loss_scale, stats ? 1 : 0);))
    } else {
        AD_DISPATCH_DTYPE(dtype, T_,
            HEAD_DISPATCH(head_ln_bwd_kernel<T_, G_, true><<<grid, 256, lds, s>>>((const T_*)nullptr, w, b, inp, target, (const T_*)z, mean,
                                                                                  rstd, gamma, beta, (T_*)dz, (float*)ws, pix_per_img, ch,
                                                                                  loss_kind, eps, grad_scale, loss_scale, stats ? 1 : 0);))
    }
    AD_LAUNCH_CHECK("ad_head_ln_bwd");
    HeadLnOuts o;
    o.ptr[0] = dw; o.end[0] = ch * 3;
    o.ptr[1] = db; o.end[1] = ch * 3 + 3;
    o.ptr[2] = dgamma; o.end[2] = ch * 4 + 3;
    o.ptr[3] = dbeta; o.end[3] = ch * 5 + 3;
    o.ptr[4] = dbias_conv; o.end[4] = ch * 6 + 3;
    rows_reduce5_kernel<<<(ch * 6 + 3 + 3) / 4, 256, 0, s>>>((const float*)ws, n * bpi, ncol, ch * 6 + 3, o);
    AD_LAUNCH_CHECK("head_ln rows_reduce");
    if (stats) {
        head_stats_strided_kernel<<<1, 256, 0, s>>>((const float*)ws, n, bpi, ncol, ch * 6 + 3, stats, sqerr, (float)pix_per_img * 3.f);
        AD_LAUNCH_CHECK("head_ln stats");
    }
    return AD_OK;
}
