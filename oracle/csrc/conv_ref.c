/* CPU restatement (test infrastructure only; never linked or loaded by the product) of Keras Conv2D, kernel 3x3 or 1x1, stride 1,
 * padding "same", NHWC float64 -- Super_resolution/code/train_adaptive_unet.py:202,207,259,267-274 (the reference's call sites;
 * the arithmetic lives in TensorFlow) -- and of its filter gradient.  Same sums as oracle/ops.py::conv2d_same_fwd / _bwd (which
 * remain the definition and the fallback); this file exists because the layer-wise audits convolve whole BASELINE-size batches
 * in float64 and NumPy's strided copies made that ten of the GPU suite's thirteen minutes.
 *
 *   y[n,oy,ox,co] = b[co] + sum_{dy,dx,ci} xp[n, oy+dy, ox+dx, ci] * w[dy,dx,ci,co]       (xp = x zero-padded by kh/2, kw/2)
 *   dw[dy,dx,ci,co] = sum_{n,oy,ox} xp[n, oy+dy, ox+dx, ci] * dy_[n,oy,ox,co]
 * The dgrad is the forward function on dy with the kernel rotated by 180 degrees and its channel axes swapped (done in NumPy).
 *
 * For a fixed dy the kw * Cin values a pixel needs from padded row oy+dy are CONTIGUOUS (NHWC), so the inner contraction runs
 * over kw * Cin consecutive doubles per input row: no im2col copy.  Accumulation order: dy, then (dx, ci) ascending, in double.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* zero-padded copy [n, h + kh - 1, w + kw - 1, c] of x [n, h, w, c], written in parallel (first touch by the thread that fills it:
 * np.pad of a 268 MB batch took over a second on one core) */
static double* pad_copy(const double* x, long n, long h, long w, long c, long kh, long kw) {
    const long ph = kh / 2, pw = kw / 2, hp = h + kh - 1, wp = w + kw - 1;
    double* xp = (double*)malloc((size_t)n * hp * wp * c * sizeof(double));
    if (!xp) return NULL;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < n * hp; ++r) {
        const long img = r / hp, yy = r - img * hp - ph;
        double* dst = xp + (size_t)r * wp * c;
        if (yy < 0 || yy >= h) { memset(dst, 0, (size_t)wp * c * sizeof(double)); continue; }
        memset(dst, 0, (size_t)pw * c * sizeof(double));
        memcpy(dst + pw * c, x + ((size_t)img * h + yy) * w * c, (size_t)w * c * sizeof(double));
        memset(dst + (pw + w) * c, 0, (size_t)(wp - pw - w) * c * sizeof(double));
    }
    return xp;
}

#define PB 4   /* pixels per register block */
#define CB 8   /* output channels per register block */

/* x: [n, h, w, cin], wt: [kh, kw, cin, cout], y: [n, h, w, cout]; returns 0, or -1 when the padded copy cannot be allocated */
int oracle_conv_fwd(const double* x, const double* wt, const double* bias, double* y, long n, long h, long w, long cin, long cout,
                    long kh, long kw) {
    const long wp = w + kw - 1, hp = h + kh - 1, kc = kw * cin;
    const long rows = n * h;
    double* xp = pad_copy(x, n, h, w, cin, kh, kw);
    if (!xp) return -1;
#pragma omp parallel for schedule(dynamic, 4)
    for (long r = 0; r < rows; ++r) {
        const long img = r / h, oy = r - img * h;
        double* yrow = y + (size_t)r * w * cout;
        for (long x0 = 0; x0 < w; x0 += PB) {
            const long np_ = w - x0 < PB ? w - x0 : PB;
            for (long c0 = 0; c0 < cout; c0 += CB) {
                const long nc = cout - c0 < CB ? cout - c0 : CB;
                double acc[PB][CB];
                for (long p = 0; p < PB; ++p)
                    for (long c = 0; c < CB; ++c) acc[p][c] = bias && c < nc ? bias[c0 + c] : 0.0;
                for (long dy = 0; dy < kh; ++dy) {
                    const double* xrow = xp + ((size_t)(img * hp + oy + dy) * wp + x0) * cin;
                    const double* wk = wt + (size_t)dy * kc * cout + c0;
                    if (np_ == PB && nc == CB) {
                        for (long k = 0; k < kc; ++k) {
                            const double* wv = wk + (size_t)k * cout;
                            const double a0 = xrow[k], a1 = xrow[cin + k], a2 = xrow[2 * cin + k], a3 = xrow[3 * cin + k];
#pragma omp simd
                            for (long c = 0; c < CB; ++c) {
                                acc[0][c] += a0 * wv[c]; acc[1][c] += a1 * wv[c]; acc[2][c] += a2 * wv[c]; acc[3][c] += a3 * wv[c];
                            }
                        }
                    } else {
                        for (long k = 0; k < kc; ++k)
                            for (long p = 0; p < np_; ++p)
                                for (long c = 0; c < nc; ++c) acc[p][c] += xrow[p * cin + k] * wk[(size_t)k * cout + c];
                    }
                }
                for (long p = 0; p < np_; ++p)
                    for (long c = 0; c < nc; ++c) yrow[(x0 + p) * cout + c0 + c] = acc[p][c];
            }
        }
    }
    free(xp);
    return 0;
}

/* dw: [kh, kw, cin, cout] = sum over pixels; one private copy per thread, added in thread order (deterministic for a fixed
 * thread count; the audits compare at 1e-4 of the tensor's maximum, double rounding is 1e-16) */
#include <omp.h>
int oracle_conv_wgrad(const double* x, const double* dyv, double* dw, long n, long h, long w, long cin, long cout, long kh, long kw) {
    const long wp = w + kw - 1, hp = h + kh - 1, kc = kw * cin;
    const size_t nel = (size_t)kh * kc * cout;
    const long rows = n * h;
    const int nt = omp_get_max_threads();
    double* xp = pad_copy(x, n, h, w, cin, kh, kw);
    double* priv = (double*)calloc((size_t)nt * nel, sizeof(double));
    if (!xp || !priv) { free(xp); free(priv); return -1; }
#pragma omp parallel
    {
        double* mine = priv + (size_t)omp_get_thread_num() * nel;
#pragma omp for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const long img = r / h, oy = r - img * h;
            const double* drow = dyv + (size_t)r * w * cout;
            for (long dy = 0; dy < kh; ++dy) {
                const double* xrow = xp + (size_t)(img * hp + oy + dy) * wp * cin;
                double* dwk = mine + (size_t)dy * kc * cout;
                for (long k0 = 0; k0 < kc; k0 += PB) {
                    const long nk = kc - k0 < PB ? kc - k0 : PB;
                    for (long c0 = 0; c0 < cout; c0 += CB) {
                        const long nc = cout - c0 < CB ? cout - c0 : CB;
                        double acc[PB][CB];
                        for (long p = 0; p < PB; ++p)
                            for (long c = 0; c < CB; ++c) acc[p][c] = 0.0;
                        if (nk == PB && nc == CB) {
                            for (long x = 0; x < w; ++x) {
                                const double* a = xrow + x * cin + k0;
                                const double* d = drow + x * cout + c0;
                                const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
#pragma omp simd
                                for (long c = 0; c < CB; ++c) {
                                    acc[0][c] += a0 * d[c]; acc[1][c] += a1 * d[c]; acc[2][c] += a2 * d[c]; acc[3][c] += a3 * d[c];
                                }
                            }
                        } else {
                            for (long x = 0; x < w; ++x)
                                for (long p = 0; p < nk; ++p)
                                    for (long c = 0; c < nc; ++c) acc[p][c] += xrow[x * cin + k0 + p] * drow[x * cout + c0 + c];
                        }
                        for (long p = 0; p < nk; ++p)
                            for (long c = 0; c < nc; ++c) dwk[(size_t)(k0 + p) * cout + c0 + c] += acc[p][c];
                    }
                }
            }
        }
    }
    memset(dw, 0, nel * sizeof(double));
    for (int t = 0; t < nt; ++t)
        for (size_t i = 0; i < nel; ++i) dw[i] += priv[(size_t)t * nel + i];
    free(priv);
    free(xp);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * LayerNormalization(axis=-1, epsilon) forward and backward over the last axis of a [npix, c] float64 matrix: the same sums as
 * oracle/ops.py::layernorm_fwd / layernorm_bwd (Super_resolution/code/train_adaptive_unet.py:203,208; Keras: mean, then the
 * mean of squared deviations, biased), one fused pass per pixel instead of a dozen whole-tensor NumPy temporaries (on a
 * 268 MB batch the NumPy form took 11 s per layer for the forward and the three bracketed backward evaluations of an audit).
 * ------------------------------------------------------------------------------------------------------------------------- */
#include <math.h>
void oracle_ln_fwd(const double* x, const double* gamma, const double* beta, double eps, double* y, double* xhat, double* rstd,
                   long npix, long c) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        const double* xr = x + (size_t)p * c;
        double s = 0.0;
        for (long i = 0; i < c; ++i) s += xr[i];
        const double mu = s / (double)c;
        double q = 0.0;
        for (long i = 0; i < c; ++i) { const double d = xr[i] - mu; q += d * d; }
        const double rs = 1.0 / sqrt(q / (double)c + eps);
        rstd[p] = rs;
        double* yr = y + (size_t)p * c;
        double* hr = xhat + (size_t)p * c;
        for (long i = 0; i < c; ++i) { const double h = (xr[i] - mu) * rs; hr[i] = h; yr[i] = h * gamma[i] + beta[i]; }
    }
}

/* dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; dgamma = sum_p dy xhat; dbeta = sum_p dy (per-thread partial sums,
 * added in thread order) */
int oracle_ln_bwd(const double* dy, const double* gamma, const double* xhat, const double* rstd, double* dx, double* dgamma,
                  double* dbeta, long npix, long c) {
    const int nt = omp_get_max_threads();
    double* part = (double*)calloc((size_t)nt * 2 * c, sizeof(double));
    if (!part) return -1;
#pragma omp parallel
    {
        double* pg = part + (size_t)omp_get_thread_num() * 2 * c;
        double* pb = pg + c;
#pragma omp for schedule(static)
        for (long p = 0; p < npix; ++p) {
            const double* dr = dy + (size_t)p * c;
            const double* hr = xhat + (size_t)p * c;
            double m1 = 0.0, m2 = 0.0;
            for (long i = 0; i < c; ++i) {
                const double g = dr[i] * gamma[i];
                m1 += g; m2 += g * hr[i];
                pg[i] += dr[i] * hr[i]; pb[i] += dr[i];
            }
            m1 /= (double)c; m2 /= (double)c;
            const double rs = rstd[p];
            double* xr = dx + (size_t)p * c;
            for (long i = 0; i < c; ++i) xr[i] = rs * (dr[i] * gamma[i] - m1 - hr[i] * m2);
        }
    }
    for (long i = 0; i < c; ++i) { dgamma[i] = 0.0; dbeta[i] = 0.0; }
    for (int t = 0; t < nt; ++t)
        for (long i = 0; i < c; ++i) { dgamma[i] += part[(size_t)t * 2 * c + i]; dbeta[i] += part[(size_t)t * 2 * c + c + i]; }
    free(part);
    return 0;
}

/* oracle/ops.py::bf16_round for float64 arrays: nearest bfloat16 (ties to even) of the value's float32 rounding, non-finite
 * values passed through -- the same integer arithmetic, one parallel pass */
#include <stdint.h>
void oracle_bf16_round(const double* x, double* out, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        const float f = (float)x[i];
        if (!isfinite(f)) { out[i] = (double)f; continue; }
        uint32_t u;
        memcpy(&u, &f, 4);
        const uint32_t r = (uint32_t)(((uint64_t)u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
        float g;
        memcpy(&g, &r, 4);
        out[i] = (double)g;
    }
}

/* Threads of the parallel loops above.  The GPU boxes show 256 logical CPUs to a process whose CPU share is 16: one OpenMP thread
 * per visible CPU then spends the share spinning at barriers (the audits got SLOWER with the C loops until this was capped). */
void oracle_set_threads(int n) {
    if (n < 1) n = 1;
    omp_set_num_threads(n);
}
