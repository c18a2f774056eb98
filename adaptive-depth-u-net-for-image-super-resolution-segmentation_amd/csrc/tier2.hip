// Tier-2 ops of the segmentation models: BatchNormalization (+ReLU), MaxPooling2D(2),
// the pixel shuffles behind Conv2DTranspose(k=2, s=2), and the sigmoid / BCE / Dice head.
//   Segmenation/code/train_adaptive_unet.py:258-362 (BN conv_block, MaxPool, bilinear x2, sigmoid head, losses)
//   Segmenation/code/unet_vinillia.py:66-69 (Conv2DTranspose(nf, 2, strides=2))
// All HBM-bound row kernels over NHWC tensors; statistics and reductions in fp32, deterministic two-stage sums.
#include "common.h"

namespace {

constexpr int NBLK = 512;   // blocks of the column-reduction kernels (partials per block in ws)

// ------------------------------------------------------------------ per-channel column statistics
// part[block][2][c] = { sum_p (x[p][ch] - shift[ch]), sum_p (x[p][ch] - shift[ch])^2 }
// FIRST: shift[ch] = x[0][ch], the first pixel's value (r05).  Both moments then come out of ONE pass over x without the
// cancellation of E[x^2] - mean^2: var = E[(x - s)^2] - (E[x - s])^2 loses (mean - s)^2 / var * 2^-24 of the variance, and a
// sample of the channel lies a few standard deviations from its mean whatever the mean's offset is (the r04 defect of the
// fused LayerNorm epilogue was the UNSHIFTED form at mean / std = 1 000).  Until r04: two passes (mean, then centred moment).
template <typename T, bool FIRST = false>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ x, const float* __restrict__ shift,
                                                       float* __restrict__ part, int64_t npix, int c, int ld) {
    constexpr int EPT = ElemTraits<T>::EPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [rows][2][c]
    const int vecs = c / EPT;                       // threads per pixel row
    const int rows = 256 / vecs > 0 ? 256 / vecs : 1;
    const int tid = threadIdx.x;
    const int v = tid % vecs, r = tid / vecs;
    float s1[EPT], s2[EPT], sh[EPT];
    const int64_t ldx = ld;                       // row stride in elements (>= c: the launch may cover a channel slice)
#pragma unroll
    for (int e = 0; e < EPT; ++e) { s1[e] = s2[e] = 0.f; sh[e] = shift ? shift[v * EPT + e] : 0.f; }
    if (FIRST && r < rows) {
        Vec16<T> l0;
        l0.load(x + v * EPT);
        l0.to_f32(sh);
    }
    if (r < rows)
        for (int64_t p = (int64_t)blockIdx.x * rows + r; p < npix; p += (int64_t)gridDim.x * rows) {
            Vec16<T> ld;
            float f[EPT];
            ld.load(x + p * ldx + v * EPT);
            ld.to_f32(f);
#pragma unroll
            for (int e = 0; e < EPT; ++e) { float d = f[e] - sh[e]; s1[e] += d; s2[e] += d * d; }
        }
    if (r < rows) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            red[(r * 2 + 0) * c + v * EPT + e] = s1[e];
            red[(r * 2 + 1) * c + v * EPT + e] = s2[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * c; i += 256) {
        float s = 0.f;
        for (int q = 0; q < rows; ++q) s += red[q * 2 * c + i];
        part[(size_t)blockIdx.x * 2 * c + i] = s;
    }
}

// out0[ch] = sum_b part[b][0][ch] * scale0 (+ add0[ch]), out1[ch] = sum_b part[b][1][ch] * scale1.
// The job is latency bound (nblocks x 2c floats): a block folds 4 of the 2c columns with 64 row groups through LDS in a
// fixed order (deterministic); one thread per channel walking all the blocks took 50-140 us per call.
__global__ __launch_bounds__(256) void colstats_finish_kernel(const float* __restrict__ part, int nblocks, int c, float scale0,
                                                              float scale1, const float* __restrict__ add0,
                                                              float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int i = blockIdx.x * 4 + cl;              // column of the [nblocks][2c] partial matrix
    float s = 0.f;
    if (i < 2 * c)
        for (int b = rg; b < nblocks; b += 64) s += part[(size_t)b * 2 * c + i];
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < 2 * c) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        if (i < c) { if (out0) out0[i] = t * scale0 + (add0 ? add0[i] : 0.f); }
        else if (out1) out1[i - c] = t * scale1;
    }
}

// One-pass BatchNorm statistics: folds the [nblocks][2][c] partials of colstats_kernel<T, true> in the fixed order of
// colstats_finish_kernel (64 row groups, then a serial sum of the 64), then mean = x0 + S1 / m, var = S2 / m - (S1 / m)^2,
// rstd and the Keras moving averages.  One block per 2 channels (both moments of a channel meet in one block).
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ part, int nblocks, int c, float inv_m,
                                                              const T* __restrict__ x0, float* __restrict__ mean,
                                                              float* __restrict__ var, float* __restrict__ rstd,
                                                              float* __restrict__ mmean, float* __restrict__ mvar,
                                                              float momentum, float eps) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int ch = blockIdx.x * 2 + (cl & 1);
    const int i = (cl >> 1) * c + ch;                  // column of the [nblocks][2c] partial matrix: moment cl >> 1 of channel ch
    float s = 0.f;
    if (ch < c)
        for (int b = rg; b < nblocks; b += 64) s += part[(size_t)b * 2 * c + i];
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        sm[0][cl] = t;
    }
    __syncthreads();
    if (tid < 2 && blockIdx.x * 2 + tid < c) {
        const int k = blockIdx.x * 2 + tid;
        const float d1 = sm[0][tid] * inv_m, d2 = sm[0][2 + tid] * inv_m;
        const float mu = (float)x0[k] + d1;
        const float vv = fmaxf(d2 - d1 * d1, 0.f);
        mean[k] = mu; var[k] = vv; rstd[k] = rsqrtf(vv + eps);
        if (mmean) mmean[k] = mmean[k] * momentum + mu * (1.f - momentum);
        if (mvar) mvar[k] = mvar[k] * momentum + vv * (1.f - momentum);
    }
}

// rstd = rsqrt(var + eps); moving <- moving * m + batch * (1 - m)   (Keras BatchNormalization, momentum 0.99)
__global__ void bn_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ var, float* __restrict__ rstd,
                                   float* __restrict__ mmean, float* __restrict__ mvar, float momentum, float eps, int c) {
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    rstd[ch] = rsqrtf(var[ch] + eps);
    if (mmean) mmean[ch] = mmean[ch] * momentum + mean[ch] * (1.f - momentum);
    if (mvar) mvar[ch] = mvar[ch] * momentum + var[ch] * (1.f - momentum);
}

// y = relu?((x - mean) * rstd * gamma + beta).  The grid stride (gridDim.x * 256) is a multiple of the vectors per pixel
// (c / EPT divides 256: chan_ok), so a thread always sees the same EPT channels and keeps their parameters in registers.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ y, int64_t npix, int c,
                                                       int relu, int ld) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT;
    const int64_t total = npix * vecs;
    const int v = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % vecs);
    float mu[EPT], rs[EPT], ga[EPT], be[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ch = v * EPT + e;
        mu[e] = mean[ch]; rs[e] = rstd[ch]; ga[e] = gamma[ch]; be[e] = beta[ch];
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        Vec16<T> ld16, st;
        float f[EPT], o[EPT];
        const int64_t off = (i / vecs) * ld + v * EPT;        // ld: row stride in elements (channel-slice launches)
        ld16.load(x + off);
        ld16.to_f32(f);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float t = (f[e] - mu[e]) * rs[e] * ga[e] + be[e];
            o[e] = relu ? fmaxf(t, 0.f) : t;
        }
        st.from_f32(o);
        st.store(y + off);
    }
}

// bn_apply_kernel over 2 x 2 windows: writes the activation y AND its MaxPooling2D(2) (the encoder's conv_block ->
// MaxPooling2D, Segmenation/code/train_adaptive_unet.py:349-351) in one pass, so the pooling never re-reads y.  The maximum is
// taken of the ROUNDED activations (what a separate pooling kernel would read).  h, w even.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            T* __restrict__ pooled, int n, int h, int w, int c, int relu) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT, oh = h / 2, ow = w / 2;
    const int64_t total = (int64_t)n * oh * ow * vecs;
    const int v = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % vecs);
    float mu[EPT], rs[EPT], ga[EPT], be[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ch = v * EPT + e;
        mu[e] = mean[ch]; rs[e] = rstd[ch]; ga[e] = gamma[ch]; be[e] = beta[ch];
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / vecs;
        const int ox = (int)(p % ow);
        const int64_t r = p / ow;
        const int oy = (int)(r % oh), nn = (int)(r / oh);
        Vec16<T> in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            in[k].load(x + (((int64_t)nn * h + 2 * oy + (k >> 1)) * w + 2 * ox + (k & 1)) * c + v * EPT);
        float m[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[EPT], o[EPT];
            in[k].to_f32(f);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float t = (f[e] - mu[e]) * rs[e] * ga[e] + be[e];
                o[e] = relu ? fmaxf(t, 0.f) : t;
            }
            Vec16<T> st;
            st.from_f32(o);
            st.store(y + (((int64_t)nn * h + 2 * oy + (k >> 1)) * w + 2 * ox + (k & 1)) * c + v * EPT);
            st.to_f32(o);
#pragma unroll
            for (int e = 0; e < EPT; ++e) m[e] = fmaxf(m[e], o[e]);
        }
        Vec16<T> sp;
        sp.from_f32(m);
        sp.store(pooled + p * c + v * EPT);
    }
}

// backward pass 1: part[block][2][c] = { sum dyl * xhat, sum dyl }  with dyl = dy * [y > 0]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ part, int64_t npix, int c, int relu, int ld) {
    constexpr int EPT = ElemTraits<T>::EPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int vecs = c / EPT;
    const int rows = 256 / vecs > 0 ? 256 / vecs : 1;
    const int tid = threadIdx.x;
    const int v = tid % vecs, r = tid / vecs;
    float s1[EPT], s2[EPT], mu[EPT], rs[EPT], ga[EPT], be[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ch = v * EPT + e;
        s1[e] = s2[e] = 0.f; mu[e] = mean[ch]; rs[e] = rstd[ch]; ga[e] = gamma[ch]; be[e] = beta[ch];
    }
    if (r < rows)
        for (int64_t p = (int64_t)blockIdx.x * rows + r; p < npix; p += (int64_t)gridDim.x * rows) {
            Vec16<T> l0, l1;
            float f[EPT], d[EPT];
            l0.load(x + p * (int64_t)ld + v * EPT);
            l1.load(dy + p * (int64_t)ld + v * EPT);
            l0.to_f32(f);
            l1.to_f32(d);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                float h = (f[e] - mu[e]) * rs[e];
                float dl = (relu && !(h * ga[e] + be[e] > 0.f)) ? 0.f : d[e];
                s1[e] += dl * h;
                s2[e] += dl;
            }
        }
    if (r < rows) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            red[(r * 2 + 0) * c + v * EPT + e] = s1[e];
            red[(r * 2 + 1) * c + v * EPT + e] = s2[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * c; i += 256) {
        float s = 0.f;
        for (int q = 0; q < rows; ++q) s += red[q * 2 * c + i];
        part[(size_t)blockIdx.x * 2 * c + i] = s;
    }
}

// backward pass 2: dx = gamma * rstd * (dyl - dbeta/m - xhat * dgamma/m); per-thread channel parameters as bn_apply_kernel
// COLSUM (r05): dsum_part[block][c] = column sums of dx AS STORED (the BiasAddGrad of the convolution in front, which until r04
// was a separate ad_colsum pass over dx): a thread always sees the same EPT channels, so its sums stay in registers and the
// block folds the threads of a channel vector through LDS in a fixed order.
template <typename T, bool COLSUM = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           T* __restrict__ dx, int64_t npix, int c, int relu, int ld,
                                                           float* __restrict__ dsum_part = nullptr) {
    constexpr int EPT = ElemTraits<T>::EPT;
    float acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
    const int vecs = c / EPT;
    const int64_t total = npix * vecs;
    const float inv_m = 1.0f / (float)npix;
    const int v = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % vecs);
    float mu[EPT], rs[EPT], ga[EPT], be[EPT], k0[EPT], k1[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ch = v * EPT + e;
        mu[e] = mean[ch]; rs[e] = rstd[ch]; ga[e] = gamma[ch]; be[e] = beta[ch];
        k0[e] = dbeta[ch] * inv_m; k1[e] = dgamma[ch] * inv_m;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        Vec16<T> l0, l1, st;
        float f[EPT], d[EPT], o[EPT];
        const int64_t off = (i / vecs) * ld + v * EPT;
        l0.load(x + off);
        l1.load(dy + off);
        l0.to_f32(f);
        l1.to_f32(d);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float h = (f[e] - mu[e]) * rs[e];
            const float dl = (relu && !(h * ga[e] + be[e] > 0.f)) ? 0.f : d[e];
            o[e] = ga[e] * rs[e] * (dl - k0[e] - h * k1[e]);
        }
        st.from_f32(o);
        st.store(dx + off);
        if constexpr (COLSUM) {
            st.to_f32(o);
#pragma unroll
            for (int e = 0; e < EPT; ++e) acc[e] += o[e];
        }
    }
    if constexpr (COLSUM) {
        __shared__ float red[256 * EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) red[threadIdx.x * EPT + e] = acc[e];
        __syncthreads();
        // threads t with t % vecs == v hold channel vector v (256 % vecs == 0: chan_ok)
        for (int ch = threadIdx.x; ch < c; ch += 256) {
            const int vv = ch / EPT, e = ch - vv * EPT;
            float t = 0.f;
            for (int q = vv; q < 256; q += vecs) t += red[q * EPT + e];
            dsum_part[(size_t)blockIdx.x * c + ch] = t;
        }
    }
}

// out[ch] = sum_b part[b][ch], fixed order (64 row groups, then a serial sum): the dbias partials of bn_bwd_apply_kernel
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nblocks, int c, float* __restrict__ out) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int i = blockIdx.x * 4 + cl;
    float s = 0.f;
    if (i < c)
        for (int b = rg; b < nblocks; b += 64) s += part[(size_t)b * c + i];
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < c) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        out[i] = t;
    }
}

// ------------------------------------------------------------------ MaxPooling2D(2)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w,
                                                           int c) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT, oh = h / 2, ow = w / 2;
    const int64_t total = (int64_t)n * oh * ow * vecs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int v = (int)(i % vecs);
        int64_t p = i / vecs;
        int ox = (int)(p % ow);
        int64_t r = p / ow;
        int oy = (int)(r % oh), nn = (int)(r / oh);
        float m[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                Vec16<T> ld;
                float f[EPT];
                ld.load(x + (((int64_t)nn * h + 2 * oy + a) * w + 2 * ox + b) * c + v * EPT);
                ld.to_f32(f);
#pragma unroll
                for (int e = 0; e < EPT; ++e) m[e] = fmaxf(m[e], f[e]);
            }
        Vec16<T> st;
        st.from_f32(m);
        st.store(y + p * c + v * EPT);
    }
}

// gradient goes to the FIRST maximal element of each window (TF MaxPoolGrad); untouched rows/cols get zero.
// add != NULL (r05): dx = pooling gradient + add, summed in fp32 and rounded once -- the encoder junction "gradient through
// MaxPooling2D + gradient of the skip connection" (Segmenation/code/train_adaptive_unet.py:349-351,358) in the pass that
// writes dx (until r04 an accumulating identity ad_resample re-read and re-wrote dx).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           T* __restrict__ dx, int n, int h, int w, int c,
                                                           const T* __restrict__ add) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT, oh = h / 2, ow = w / 2;
    const int64_t total = (int64_t)n * h * w * vecs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int v = (int)(i % vecs);
        int64_t p = i / vecs;
        int xx = (int)(p % w);
        int64_t r = p / w;
        int yy = (int)(r % h), nn = (int)(r / h);
        float o[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) o[e] = 0.f;
        const int oy = yy / 2, ox = xx / 2;
        if (oy < oh && ox < ow) {
            float win[4][EPT], g[EPT];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Vec16<T> ld;
                ld.load(x + (((int64_t)nn * h + 2 * oy + (k >> 1)) * w + 2 * ox + (k & 1)) * c + v * EPT);
                ld.to_f32(win[k]);
            }
            Vec16<T> lg;
            lg.load(dy + (((int64_t)nn * oh + oy) * ow + ox) * c + v * EPT);
            lg.to_f32(g);
            const int me = (yy & 1) * 2 + (xx & 1);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                int arg = 0;
                float best = win[0][e];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (win[k][e] > best) { best = win[k][e]; arg = k; }
                o[e] = arg == me ? g[e] : 0.f;
            }
        }
        if (add) {
            Vec16<T> la;
            float a8[EPT];
            la.load(add + p * c + v * EPT);
            la.to_f32(a8);
#pragma unroll
            for (int e = 0; e < EPT; ++e) o[e] += a8[e];
        }
        Vec16<T> st;
        st.from_f32(o);
        st.store(dx + p * c + v * EPT);
    }
}

// ------------------------------------------------------------------ pixel shuffles for Conv2DTranspose(2, s=2)
// depth_to_space: x[n,h,w,4*c] (blocks ordered a*2+b) -> y[n,2h,2w,c];  TO_SPACE = false is the inverse.
template <typename T, bool TO_SPACE>
__global__ __launch_bounds__(256) void shuffle2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = c / EPT;
    const int64_t total = (int64_t)n * h * w * 4 * vecs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int v = (int)(i % vecs);
        int64_t r = i / vecs;
        int ab = (int)(r % 4);
        int64_t p = r / 4;                                   // low-res pixel
        int xx = (int)(p % w);
        int64_t q = p / w;
        int yy = (int)(q % h), nn = (int)(q / h);
        const int64_t lo = (p * 4 + ab) * c + v * EPT;
        const int64_t hi = ((((int64_t)nn * 2 * h + 2 * yy + (ab >> 1)) * 2 * w) + 2 * xx + (ab & 1)) * c + v * EPT;
        if (TO_SPACE) *reinterpret_cast<uint4*>(y + hi) = *reinterpret_cast<const uint4*>(x + lo);
        else *reinterpret_cast<uint4*>(y + lo) = *reinterpret_cast<const uint4*>(x + hi);
    }
}

// ------------------------------------------------------------------ segmentation head
// p = sigmoid(xh @ w[ch] + b); per-sample sums for BCE / Dice / IoU:
//   sums[n][0] = sum BCE(y, clip(p)),  [1] = sum y*pc,  [2] = sum (y + pc)     (pc = clip(p, 1e-7, 1-1e-7))
template <int G>
__device__ __forceinline__ float gsum2(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// COUNTS: six more per-sample sums for the vanilla baseline's Keras metrics (Segmenation/code/unet_vinillia.py:266-271:
// BinaryAccuracy / Precision / Recall at threshold 0.5 and its global dice_coefficient on the UNCLIPPED probability, :94-99):
//   [3] = sum [p > .5] y (true positives), [4] = sum [p > .5], [5] = sum y, [6] = sum [(p > .5) == (y > .5)], [7] = sum y p, [8] = sum (y + p)
template <typename T, int G, bool COUNTS = false>
__global__ __launch_bounds__(256) void seg_head_fwd_kernel(const T* __restrict__ xh, const float* __restrict__ w,
                                                           const float* __restrict__ b, const float* __restrict__ target,
                                                           float* __restrict__ prob, float* __restrict__ part, int64_t ppi,
                                                           int ch) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    constexpr int NS = COUNTS ? 9 : 3;
    __shared__ float sm[NS][4];
    float cnt[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x, gl = tid % G, gp = tid / G;
    const int img = blockIdx.y;
    float wl[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) wl[e] = w[gl * EPT + e];
    const float b0 = b[0];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t q = (int64_t)blockIdx.x * PPB + gp; q < ppi; q += (int64_t)gridDim.x * PPB) {
        const int64_t pix = (int64_t)img * ppi + q;
        Vec16<T> ld;
        float x[EPT];
        ld.load(xh + pix * ch + gl * EPT);
        ld.to_f32(x);
        float r = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) r += x[e] * wl[e];
        r = gsum2<G>(r) + b0;
        if (gl == 0) {
            const float p = 1.f / (1.f + __expf(-r));
            prob[pix] = p;
            if (target) {
                const float y = target[pix];
                const float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
                s0 += -(y * __logf(pc) + (1.f - y) * __logf(1.f - pc));
                s1 += y * pc;
                s2 += y + pc;
                if (COUNTS) {
                    const bool pos = p > 0.5f, truth = y > 0.5f;
                    cnt[0] += pos && truth ? 1.f : 0.f;
                    cnt[1] += pos ? 1.f : 0.f;
                    cnt[2] += truth ? 1.f : 0.f;
                    cnt[3] += pos == truth ? 1.f : 0.f;
                    cnt[4] += y * p;
                    cnt[5] += y + p;
                }
            }
        }
    }
    float v[NS];
    v[0] = s0; v[1] = s1; v[2] = s2;
    if (COUNTS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[3 + (k < NS - 3 ? k : 0)] = cnt[k];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if ((tid & 63) == 0) sm[k][tid >> 6] = v[k];
    }
    __syncthreads();
    if (tid < NS && part)
        part[((size_t)img * gridDim.x + blockIdx.x) * NS + tid] = sm[tid][0] + sm[tid][1] + sm[tid][2] + sm[tid][3];
}

// sums[img][0..3) (and counts[img][0..6) when the partial rows have 9 columns) = column sums of part[img][0..bpi)
__global__ void seg_sums_kernel(const float* __restrict__ part, int n, int bpi, float* __restrict__ sums, int ncol,
                                float* __restrict__ counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ncol) return;
    int img = i / ncol, k = i % ncol;
    float s = 0.f;
    for (int j = 0; j < bpi; ++j) s += part[((size_t)img * bpi + j) * ncol + k];
    if (k < 3) sums[img * 3 + k] = s;
    else counts[img * 6 + k - 3] = s;
}

// out[0..3) = { loss, dice, iou } of one batch from the per-sample sums of ad_seg_head_fwd (Segmenation/code/train_adaptive_unet.py:
// 258-304): bce = sum_n sums[n][0] / count, dice = mean_n (2 I_n + s) / (U_n + s), iou = mean_n (I_n + s) / (U_n - I_n + s),
// loss = wb * bce + wd * (1 - dice).  One wave, samples summed in index order (deterministic).
__global__ __launch_bounds__(64) void seg_metrics_kernel(const float* __restrict__ sums, int n, float count, float wb, float wd,
                                                         float smooth, float* __restrict__ out) {
    __shared__ float sm[3][64];
    float b = 0.f, d = 0.f, u = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) {
        const float ce = sums[i * 3], inter = sums[i * 3 + 1], tot = sums[i * 3 + 2];
        b += ce;
        d += (2.0f * inter + smooth) / (tot + smooth);
        u += (inter + smooth) / (tot - inter + smooth);
    }
    sm[0][threadIdx.x] = b; sm[1][threadIdx.x] = d; sm[2][threadIdx.x] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tb = 0.f, td = 0.f, tu = 0.f;
        for (int i = 0; i < 64; ++i) { tb += sm[0][i]; td += sm[1][i]; tu += sm[2][i]; }
        const float bce = tb / count, dice = td / (float)n, iou = tu / (float)n;
        out[0] = wb * bce + wd * (1.0f - dice);
        out[1] = dice;
        out[2] = iou;
    }
}

// backward: dL/dp = wb * dBCE/dp / count + wd * (-(1/n) * d dice_n / dp); dlogit = dL/dp * p (1 - p)
template <typename T, int G>
__global__ __launch_bounds__(256) void seg_head_bwd_kernel(const T* __restrict__ xh, const float* __restrict__ w,
                                                           const float* __restrict__ target, const float* __restrict__ prob,
                                                           const float* __restrict__ sums, T* __restrict__ dxh,
                                                           float* __restrict__ part, int64_t ppi, int ch, int n, float wb,
                                                           float wd, float smooth, const float* __restrict__ loss_scale) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [PPB][ch + 1]
    const int ncol = ch + 1;
    const int tid = threadIdx.x, gl = tid % G, gp = tid / G;
    const int img = blockIdx.y;
    float wl[EPT], aw[EPT], ab = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { wl[e] = w[gl * EPT + e]; aw[e] = 0.f; }
    const float inter = sums[img * 3 + 1], uni = sums[img * 3 + 2];
    const float den = uni + smooth;
    const float inv_cnt = 1.0f / ((float)n * (float)ppi);
    const float lscale = loss_scale ? loss_scale[0] : 1.f;        // dynamic loss scale (fp16), device resident
    for (int64_t q = (int64_t)blockIdx.x * PPB + gp; q < ppi; q += (int64_t)gridDim.x * PPB) {
        const int64_t pix = (int64_t)img * ppi + q;
        const float p = prob[pix], y = target[pix];
        const bool inside = p >= 1e-7f && p <= 1.f - 1e-7f;     // clip_by_value passes the gradient inside only
        const float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
        float dp = 0.f;
        if (inside) {
            dp = wb * (-(y / pc) + (1.f - y) / (1.f - pc)) * inv_cnt;
            const float ddice = (2.f * y * den - (2.f * inter + smooth)) / (den * den);
            dp += -wd * ddice / (float)n;
        }
        const float g = dp * p * (1.f - p) * lscale;
        Vec16<T> ld, st;
        float x[EPT], dx[EPT];
        ld.load(xh + pix * ch + gl * EPT);
        ld.to_f32(x);
#pragma unroll
        for (int e = 0; e < EPT; ++e) { dx[e] = g * wl[e]; aw[e] += x[e] * g; }
        ab += g;
        st.from_f32(dx);
        st.store(dxh + pix * ch + gl * EPT);
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) red[gp * ncol + gl * EPT + e] = aw[e];
    if (gl == 0) red[gp * ncol + ch] = ab;
    __syncthreads();
    for (int i = tid; i < ncol; i += 256) {
        float s = 0.f;
        for (int p = 0; p < PPB; ++p) s += red[p * ncol + i];
        part[((size_t)img * gridDim.x + blockIdx.x) * ncol + i] = s;
    }
}

__global__ void rows_sum_kernel(const float* __restrict__ part, int nrows, int ncols, float* __restrict__ o0, int n0,
                                float* __restrict__ o1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    float s = 0.f;
    for (int r = 0; r < nrows; ++r) s += part[(size_t)r * ncols + i];
    if (i < n0) o0[i] = s; else o1[i - n0] = s;
}

// rows_sum_kernel with 64 row groups per column (fixed order: deterministic)
__global__ __launch_bounds__(256) void rows_sum_par_kernel(const float* __restrict__ part, int nrows, int ncols, float* __restrict__ o0,
                                                           int n0, float* __restrict__ o1) {
    __shared__ float sm[64][5];
    const int tid = threadIdx.x, cl = tid & 3, rg = tid >> 2;
    const int i = blockIdx.x * 4 + cl;
    float s = 0.f;
    if (i < ncols)
        for (int r = rg; r < nrows; r += 64) s += part[(size_t)r * ncols + i];
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < ncols) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sm[r][cl];
        if (i < n0) o0[i] = t; else o1[i - n0] = t;
    }
}

static int ew_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b < 16384 ? b : 16384);
}

// Channel counts: a multiple of the 16-byte vector; per launch at most 256 vectors per pixel (one thread each) with
// 256 % vectors == 0.  Wider tensors (c = 2048 fp32 at the depth-5 bottleneck) run as channel slices of 256 vectors: the
// kernels take the row stride separately, the per-channel statistics of different slices are independent.
static bool chan_ok(int c, int ept) {
    if (c <= 0 || c % ept) return false;
    const int vecs = c / ept;
    return vecs <= 256 ? 256 % vecs == 0 : vecs % 256 == 0;
}
static int slice_channels(int c, int ept) { return c / ept <= 256 ? c : 256 * ept; }

template <typename T>
int colstats(const void* x, const float* shift, float* part, int64_t npix, int c, int ld, hipStream_t s, int* nblocks) {
    const int vecs = c / ElemTraits<T>::EPT;
    const int rows = 256 / vecs > 0 ? 256 / vecs : 1;
    int64_t nb = (npix + rows - 1) / rows;
    *nblocks = (int)(nb < NBLK ? nb : NBLK);
    size_t lds = (size_t)rows * 2 * c * sizeof(float);
    if (lds > 64 * 1024) return ad_set_error(AD_ERR_ARG, "batchnorm: c=%d too wide", c);
    colstats_kernel<T><<<*nblocks, 256, lds, s>>>((const T*)x, shift, part, npix, c, ld);
    AD_LAUNCH_CHECK("colstats");
    return AD_OK;
}

// pool_{n,h,w} > 0 (whole-width launches only: ld == c): the activation's MaxPooling2D(2) is written to `pooled` in the same pass
template <typename T>
int bn_fwd_train(const T* z, const float* gamma, const float* beta, T* y, float* save_mean, float* save_rstd, float* save_var,
                 float* moving_mean, float* moving_var, float momentum, int64_t npix, int c, int ld, float eps, int relu,
                 float* part, hipStream_t s, T* pooled = nullptr, int pool_n = 0, int pool_h = 0, int pool_w = 0) {
    int nb = 0;
    // ONE pass over z for both moments: deviations from the first pixel's row (colstats_kernel<T, true>), biased variance as Keras
    const int vecs = c / ElemTraits<T>::EPT;
    const int rows = 256 / vecs > 0 ? 256 / vecs : 1;
    int64_t nbl = (npix + rows - 1) / rows;
    nb = (int)(nbl < NBLK ? nbl : NBLK);
    size_t lds = (size_t)rows * 2 * c * sizeof(float);
    if (lds > 64 * 1024) return ad_set_error(AD_ERR_ARG, "batchnorm: c=%d too wide", c);
    colstats_kernel<T, true><<<nb, 256, lds, s>>>(z, nullptr, part, npix, c, ld);
    bn_stats_finish_kernel<T><<<(c + 1) / 2, 256, 0, s>>>(part, nb, c, 1.0f / (float)npix, z, save_mean, save_var, save_rstd,
                                                         moving_mean, moving_var, momentum, eps);
    if (pooled)
        bn_apply_pool_kernel<T><<<ew_blocks((int64_t)pool_n * (pool_h / 2) * (pool_w / 2) * vecs), 256, 0, s>>>(
            z, save_mean, save_rstd, gamma, beta, y, pooled, pool_n, pool_h, pool_w, c, relu);
    else
        bn_apply_kernel<T><<<ew_blocks(npix * vecs), 256, 0, s>>>(z, save_mean, save_rstd, gamma, beta, y, npix, c, relu, ld);
    AD_LAUNCH_CHECK("ad_batchnorm_relu_fwd_train");
    return AD_OK;
}

template <typename T>
int bn_bwd(const T* dy, const T* z, const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
           T* dz, float* dgamma, float* dbeta, int64_t npix, int c, int ld, int relu, float* part, hipStream_t s,
           float* dbias = nullptr) {
    const int vecs = c / ElemTraits<T>::EPT;
    const int rows = 256 / vecs > 0 ? 256 / vecs : 1;
    int64_t nbl = (npix + rows - 1) / rows;
    const int nb = (int)(nbl < NBLK ? nbl : NBLK);
    size_t lds = (size_t)rows * 2 * c * sizeof(float);
    bn_bwd_reduce_kernel<T><<<nb, 256, lds, s>>>(dy, z, save_mean, save_rstd, gamma, beta, part, npix, c, relu, ld);
    colstats_finish_kernel<<<(2 * c + 3) / 4, 256, 0, s>>>(part, nb, c, 1.f, 1.f, nullptr, dgamma, dbeta);
    if (dbias) {
        // the partials of the reduce above have been folded: the workspace now takes [blocks][c] column sums of dz as stored
        int64_t bl = (npix * vecs + 255) / 256;
        const int nba = (int)(bl < 2 * NBLK ? bl : 2 * NBLK);
        bn_bwd_apply_kernel<T, true><<<nba, 256, 0, s>>>(dy, z, save_mean, save_rstd, gamma, beta, dgamma, dbeta, dz, npix, c, relu,
                                                         ld, part);
        colsum_finish_kernel<<<(c + 3) / 4, 256, 0, s>>>(part, nba, c, dbias);
    } else {
        bn_bwd_apply_kernel<T><<<ew_blocks(npix * vecs), 256, 0, s>>>(dy, z, save_mean, save_rstd, gamma, beta, dgamma, dbeta, dz,
                                                                      npix, c, relu, ld);
    }
    AD_LAUNCH_CHECK("ad_batchnorm_relu_bwd");
    return AD_OK;
}

}  // namespace

extern "C" size_t ad_batchnorm_ws_bytes(int c) { return (size_t)NBLK * 2 * c * sizeof(float); }

extern "C" int ad_batchnorm_relu_fwd_train(const void* z, const float* gamma, const float* beta, void* y, float* save_mean,
                                           float* save_rstd, float* save_var, float* moving_mean, float* moving_var,
                                           float momentum, int64_t npix, int c, float eps, int relu, void* ws,
                                           size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_batchnorm_relu_fwd_train: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(npix > 0 && chan_ok(c, ept), "ad_batchnorm_relu_fwd_train: unsupported npix=%ld c=%d", (long)npix, c);
    if (!ws || ws_bytes < ad_batchnorm_ws_bytes(c)) return ad_set_error(AD_ERR_WS, "ad_batchnorm: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int cs = slice_channels(c, ept);
    for (int c0 = 0; c0 < c; c0 += cs) {
        int rc = AD_OK;
        AD_DISPATCH_DTYPE(dtype, T_,
            rc = bn_fwd_train<T_>((const T_*)z + c0, gamma + c0, beta + c0, (T_*)y + c0, save_mean + c0, save_rstd + c0,
                                  save_var + c0, moving_mean ? moving_mean + c0 : nullptr, moving_var ? moving_var + c0 : nullptr,
                                  momentum, npix, cs, c, eps, relu, (float*)ws, s);)
        if (rc) return rc;
    }
    return AD_OK;
}

extern "C" int ad_batchnorm_relu_pool_fwd_train(const void* z, const float* gamma, const float* beta, void* y, void* pooled,
                                                float* save_mean, float* save_rstd, float* save_var, float* moving_mean,
                                                float* moving_var, float momentum, int n, int h, int w, int c, float eps,
                                                int relu, void* ws, size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_batchnorm_relu_pool_fwd_train: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(n > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && pooled, "ad_batchnorm_relu_pool_fwd_train: bad shape n=%d %dx%d", n, h, w);
    const int64_t npix = (int64_t)n * h * w;
    AD_REQUIRE(chan_ok(c, ept) && c / ept <= 256, "ad_batchnorm_relu_pool_fwd_train: unsupported c=%d (at most %d channels)", c, 256 * ept);
    if (!ws || ws_bytes < ad_batchnorm_ws_bytes(c)) return ad_set_error(AD_ERR_WS, "ad_batchnorm: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int rc = AD_OK;
    AD_DISPATCH_DTYPE(dtype, T_,
        rc = bn_fwd_train<T_>((const T_*)z, gamma, beta, (T_*)y, save_mean, save_rstd, save_var, moving_mean, moving_var, momentum,
                              npix, c, c, eps, relu, (float*)ws, s, (T_*)pooled, n, h, w);)
    return rc;
}

extern "C" int ad_batchnorm_relu_fwd_infer(const void* z, const float* gamma, const float* beta, const float* moving_mean,
                                           const float* moving_var, void* y, float* rstd_tmp, int64_t npix, int c, float eps,
                                           int relu, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_batchnorm_relu_fwd_infer: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(npix > 0 && chan_ok(c, ept), "ad_batchnorm_relu_fwd_infer: unsupported npix=%ld c=%d", (long)npix, c);
    hipStream_t s = (hipStream_t)stream;
    bn_finalize_kernel<<<(c + 255) / 256, 256, 0, s>>>(moving_mean, moving_var, rstd_tmp, nullptr, nullptr, 0.f, eps, c);
    const int cs = slice_channels(c, ept);
    const int blocks = ew_blocks(npix * (cs / ept));
    for (int c0 = 0; c0 < c; c0 += cs) {
        AD_DISPATCH_DTYPE(dtype, T_, bn_apply_kernel<T_><<<blocks, 256, 0, s>>>((const T_*)z + c0, moving_mean + c0, rstd_tmp + c0,
                                                                              gamma + c0, beta + c0, (T_*)y + c0, npix, cs, relu, c);)
    }
    AD_LAUNCH_CHECK("ad_batchnorm_relu_fwd_infer");
    return AD_OK;
}

extern "C" int ad_batchnorm_relu_bwd_dbias(const void* dy, const void* z, const float* save_mean, const float* save_rstd,
                                           const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta,
                                           float* dbias, int64_t npix, int c, int relu, void* ws, size_t ws_bytes, int dtype,
                                           void* stream);

extern "C" int ad_batchnorm_relu_bwd(const void* dy, const void* z, const float* save_mean, const float* save_rstd,
                                     const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta,
                                     int64_t npix, int c, int relu, void* ws, size_t ws_bytes, int dtype, void* stream) {
    return ad_batchnorm_relu_bwd_dbias(dy, z, save_mean, save_rstd, gamma, beta, dz, dgamma, dbeta, nullptr, npix, c, relu, ws,
                                       ws_bytes, dtype, stream);
}

extern "C" int ad_batchnorm_relu_bwd_dbias(const void* dy, const void* z, const float* save_mean, const float* save_rstd,
                                           const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta,
                                           float* dbias, int64_t npix, int c, int relu, void* ws, size_t ws_bytes, int dtype,
                                           void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_batchnorm_relu_bwd: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(npix > 0 && chan_ok(c, ept), "ad_batchnorm_relu_bwd: unsupported npix=%ld c=%d", (long)npix, c);
    if (!ws || ws_bytes < ad_batchnorm_ws_bytes(c)) return ad_set_error(AD_ERR_WS, "ad_batchnorm: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int cs = slice_channels(c, ept);
    for (int c0 = 0; c0 < c; c0 += cs) {
        int rc = AD_OK;
        AD_DISPATCH_DTYPE(dtype, T_,
            rc = bn_bwd<T_>((const T_*)dy + c0, (const T_*)z + c0, save_mean + c0, save_rstd + c0, gamma + c0, beta + c0,
                            (T_*)dz + c0, dgamma + c0, dbeta + c0, npix, cs, c, relu, (float*)ws, s, dbias ? dbias + c0 : nullptr);)
        if (rc) return rc;
    }
    return AD_OK;
}

extern "C" int ad_colsum(const void* x, float* out, int64_t npix, int c, void* ws, size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_colsum: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(npix > 0 && chan_ok(c, ept), "ad_colsum: unsupported npix=%ld c=%d", (long)npix, c);
    if (!ws || ws_bytes < ad_batchnorm_ws_bytes(c)) return ad_set_error(AD_ERR_WS, "ad_colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int cs = slice_channels(c, ept);
    for (int c0 = 0; c0 < c; c0 += cs) {
        int nb = 0;
        int rc = AD_OK;
        AD_DISPATCH_DTYPE(dtype, T_, rc = colstats<T_>((const T_*)x + c0, nullptr, (float*)ws, npix, cs, c, s, &nb);)
        if (rc) return rc;
        colstats_finish_kernel<<<(2 * cs + 3) / 4, 256, 0, s>>>((const float*)ws, nb, cs, 1.f, 0.f, nullptr, out + c0, nullptr);
    }
    AD_LAUNCH_CHECK("ad_colsum");
    return AD_OK;
}

extern "C" int ad_maxpool2_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_maxpool2_fwd: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % ept == 0, "ad_maxpool2_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = ew_blocks((int64_t)n * (h / 2) * (w / 2) * (c / ept));
    AD_DISPATCH_DTYPE(dtype, T_, maxpool2_fwd_kernel<T_><<<blocks, 256, 0, s>>>((const T_*)x, (T_*)y, n, h, w, c);)
    AD_LAUNCH_CHECK("ad_maxpool2_fwd");
    return AD_OK;
}

extern "C" int ad_maxpool2_bwd_add(const void* dy, const void* x, const void* add, void* dx, int n, int h, int w, int c, int dtype,
                                   void* stream);
extern "C" int ad_maxpool2_bwd(const void* dy, const void* x, void* dx, int n, int h, int w, int c, int dtype, void* stream) {
    return ad_maxpool2_bwd_add(dy, x, nullptr, dx, n, h, w, c, dtype, stream);
}

extern "C" int ad_maxpool2_bwd_add(const void* dy, const void* x, const void* add, void* dx, int n, int h, int w, int c, int dtype,
                                   void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_maxpool2_bwd: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % ept == 0, "ad_maxpool2_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = ew_blocks((int64_t)n * h * w * (c / ept));
    AD_DISPATCH_DTYPE(dtype, T_, maxpool2_bwd_kernel<T_><<<blocks, 256, 0, s>>>((const T_*)dy, (const T_*)x, (T_*)dx, n, h, w, c,
                                                                                  (const T_*)add);)
    AD_LAUNCH_CHECK("ad_maxpool2_bwd");
    return AD_OK;
}

extern "C" int ad_pixel_shuffle2(const void* x, void* y, int n, int h, int w, int c, int to_space, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_pixel_shuffle2: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % ept == 0, "ad_pixel_shuffle2: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = ew_blocks((int64_t)n * h * w * 4 * (c / ept));
    AD_DISPATCH_DTYPE(dtype, T_,
        if (to_space) shuffle2_kernel<T_, true><<<blocks, 256, 0, s>>>((const T_*)x, (T_*)y, n, h, w, c);
        else shuffle2_kernel<T_, false><<<blocks, 256, 0, s>>>((const T_*)x, (T_*)y, n, h, w, c);)
    AD_LAUNCH_CHECK("ad_pixel_shuffle2");
    return AD_OK;
}

// Conv2D(num_classes, 1, activation="softmax") head (Segmenation/code/unet_vinillia.py:89-90, num_classes > 1): the G
// lanes of a pixel each hold EPT channels; class k's logit is a dot product folded over the group, lane 0 writes it to
// prob[pix][k], re-reads its own K logits for the running maximum and normalises in place.  Inference-only path (the
// reference defines no loss for it), so one pass with K group reductions per pixel is all it needs.
template <typename T, int G>
__global__ __launch_bounds__(256) void softmax_head_fwd_kernel(const T* __restrict__ xh, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ prob,
                                                               int64_t npix, int ch, int k_classes) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int PPB = 256 / G;
    const int tid = threadIdx.x, gl = tid % G, gp = tid / G;
    for (int64_t pix = (int64_t)blockIdx.x * PPB + gp; pix < npix; pix += (int64_t)gridDim.x * PPB) {
        Vec16<T> ld;
        float x[EPT];
        ld.load(xh + pix * ch + gl * EPT);
        ld.to_f32(x);
        float* out = prob + pix * k_classes;
        float mx = -__builtin_inff();
        for (int k = 0; k < k_classes; ++k) {
            float r = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) r += x[e] * w[(size_t)(gl * EPT + e) * k_classes + k];     // kernel [ch][K]
            r = gsum2<G>(r) + b[k];
            mx = fmaxf(mx, r);
            if (gl == 0) out[k] = r;
        }
        if (gl == 0) {
            float sum = 0.f;
            for (int k = 0; k < k_classes; ++k) { const float e = __expf(out[k] - mx); out[k] = e; sum += e; }
            const float inv = 1.f / sum;
            for (int k = 0; k < k_classes; ++k) out[k] *= inv;
        }
    }
}

#define SEG_DISPATCH(...)                                          \
    switch (g) {                                                   \
        case 4: { constexpr int G_ = 4; __VA_ARGS__ } break;       \
        case 8: { constexpr int G_ = 8; __VA_ARGS__ } break;       \
        case 16: { constexpr int G_ = 16; __VA_ARGS__ } break;     \
        case 32: { constexpr int G_ = 32; __VA_ARGS__ } break;     \
        default: { constexpr int G_ = 64; __VA_ARGS__ } break;     \
    }

static bool seg_group(int ch, int ept, int* g) {
    if (ch <= 0 || ch % ept) return false;
    int v = ch / ept;
    if (v != 4 && v != 8 && v != 16 && v != 32 && v != 64) return false;
    *g = v;
    return true;
}

static int seg_bpi(int64_t ppi, int g) {
    int64_t ppb = 256 / g;
    int64_t nb = (ppi + ppb - 1) / ppb;
    return (int)(nb < 64 ? nb : 64);
}

extern "C" size_t ad_seg_head_ws_bytes(int n, int ch) { return (size_t)n * 64 * (ch + 3) * sizeof(float); }

extern "C" int ad_seg_head_fwd_counts(const void* xh, const float* w, const float* b, const float* target, float* prob,
                                      float* sums, float* counts, int n, int64_t pix_per_img, int ch, void* ws, size_t ws_bytes,
                                      int dtype, void* stream);

extern "C" int ad_seg_head_fwd(const void* xh, const float* w, const float* b, const float* target, float* prob, float* sums,
                               int n, int64_t pix_per_img, int ch, void* ws, size_t ws_bytes, int dtype, void* stream) {
    return ad_seg_head_fwd_counts(xh, w, b, target, prob, sums, nullptr, n, pix_per_img, ch, ws, ws_bytes, dtype, stream);
}

extern "C" int ad_seg_head_fwd_counts(const void* xh, const float* w, const float* b, const float* target, float* prob,
                                      float* sums, float* counts, int n, int64_t pix_per_img, int ch, void* ws, size_t ws_bytes,
                                      int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_seg_head_fwd: bad dtype %d", dtype);
    int g;
    AD_REQUIRE(n > 0 && pix_per_img > 0 && seg_group(ch, ad_is_half(dtype) ? 8 : 4, &g), "ad_seg_head_fwd: unsupported shape ch=%d", ch);
    AD_REQUIRE(!counts || (target && sums), "ad_seg_head_fwd_counts: counts need a target and sums");
    const int bpi = seg_bpi(pix_per_img, g);
    const int ncol = counts ? 9 : 3;
    float* part = nullptr;
    if (target) {
        if (!ws || ws_bytes < (size_t)n * bpi * ncol * sizeof(float)) return ad_set_error(AD_ERR_WS, "ad_seg_head_fwd: workspace too small");
        part = (float*)ws;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bpi, n);
    if (counts) {
        AD_DISPATCH_DTYPE(dtype, T_,
            SEG_DISPATCH((seg_head_fwd_kernel<T_, G_, true><<<grid, 256, 0, s>>>((const T_*)xh, w, b, target, prob, part, pix_per_img, ch));))
    } else {
        AD_DISPATCH_DTYPE(dtype, T_,
            SEG_DISPATCH((seg_head_fwd_kernel<T_, G_><<<grid, 256, 0, s>>>((const T_*)xh, w, b, target, prob, part, pix_per_img, ch));))
    }
    AD_LAUNCH_CHECK("ad_seg_head_fwd");
    if (target) {
        seg_sums_kernel<<<(n * ncol + 255) / 256, 256, 0, s>>>(part, n, bpi, sums, ncol, counts);
        AD_LAUNCH_CHECK("seg_sums");
    }
    return AD_OK;
}

extern "C" int ad_seg_metrics(const float* sums, int n, float count, float bce_weight, float dice_weight, float smooth,
                              float* out3, void* stream) {
    AD_REQUIRE(sums && out3 && n > 0 && count > 0.f, "ad_seg_metrics: bad arguments n=%d count=%g", n, (double)count);
    seg_metrics_kernel<<<1, 64, 0, (hipStream_t)stream>>>(sums, n, count, bce_weight, dice_weight, smooth, out3);
    AD_LAUNCH_CHECK("ad_seg_metrics");
    return AD_OK;
}

extern "C" int ad_softmax_head_fwd(const void* xh, const float* w, const float* b, float* prob, int64_t npix, int ch,
                                   int num_classes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_softmax_head_fwd: bad dtype %d", dtype);
    int g;
    AD_REQUIRE(npix > 0 && num_classes >= 2 && num_classes <= 1024 && seg_group(ch, ad_is_half(dtype) ? 8 : 4, &g),
               "ad_softmax_head_fwd: unsupported shape ch=%d classes=%d", ch, num_classes);
    const int ppb = 256 / g;
    const int64_t nb = (npix + ppb - 1) / ppb;
    const int blocks = (int)(nb < 4096 ? nb : 4096);
    hipStream_t s = (hipStream_t)stream;
    AD_DISPATCH_DTYPE(dtype, T_,
        SEG_DISPATCH(softmax_head_fwd_kernel<T_, G_><<<blocks, 256, 0, s>>>((const T_*)xh, w, b, prob, npix, ch, num_classes);))
    AD_LAUNCH_CHECK("ad_softmax_head_fwd");
    return AD_OK;
}

extern "C" int ad_seg_head_bwd(const void* xh, const float* w, const float* target, const float* prob, const float* sums,
                               void* dxh, float* dw, float* db, int n, int64_t pix_per_img, int ch, float bce_weight,
                               float dice_weight, float smooth, const float* loss_scale, void* ws, size_t ws_bytes, int dtype,
                               void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_seg_head_bwd: bad dtype %d", dtype);
    int g;
    AD_REQUIRE(n > 0 && pix_per_img > 0 && seg_group(ch, ad_is_half(dtype) ? 8 : 4, &g), "ad_seg_head_bwd: unsupported shape ch=%d", ch);
    const int bpi = seg_bpi(pix_per_img, g);
    const int ncol = ch + 1;
    if (!ws || ws_bytes < (size_t)n * bpi * ncol * sizeof(float)) return ad_set_error(AD_ERR_WS, "ad_seg_head_bwd: workspace too small");
    size_t lds = (size_t)(256 / g) * ncol * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bpi, n);
    AD_DISPATCH_DTYPE(dtype, T_,
        SEG_DISPATCH(seg_head_bwd_kernel<T_, G_><<<grid, 256, lds, s>>>((const T_*)xh, w, target, prob, sums, (T_*)dxh, (float*)ws,
                                                                        pix_per_img, ch, n, bce_weight, dice_weight, smooth, loss_scale);))
    AD_LAUNCH_CHECK("ad_seg_head_bwd");
    // (until r04 one thread per column walked all n * bpi rows: 1 024 dependent loads, most of the launch's 0.3 ms)
    rows_sum_par_kernel<<<(ncol + 3) / 4, 256, 0, s>>>((const float*)ws, n * bpi, ncol, dw, ch, db);
    AD_LAUNCH_CHECK("seg rows_sum");
    return AD_OK;
}
