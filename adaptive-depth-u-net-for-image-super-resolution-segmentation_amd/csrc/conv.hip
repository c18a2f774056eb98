// 3x3 "same" convolution on gfx950 as an LDS-tiled implicit GEMM on the matrix cores.
//
//   forward / dgrad : M = pixels of a spatial tile (256), N = 64 output channels, K = 9 * Cin
//   wgrad           : M = input channels, N = 64 output channels, K = pixels (split over workgroups)
//
// Replaces the TensorFlow Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter ops reached from
// L.Conv2D(nf, 3, padding="same") at Super_resolution/code/train_adaptive_unet.py:202,207,259.
//
// Data layout: activations NHWC; per workgroup a halo tile [(TH+2)x(TW+2) pixels][64-byte channel
// chunk] lives in LDS with an 80-byte pixel stride (conflict-free ds_read_b128 for 16 consecutive
// pixels), so one HBM read of the tile feeds all nine taps.  Weights are pre-packed to
// [tap][Cin/KV][Cout][KV] (KV = 16 B) so that every MFMA B fragment is one 16-byte LDS read.
//
// Tile geometry is a runtime (TI images x TH x TW) split of 256 pixels so that tiny feature maps
// (4x4, 2x2, 1x1 at the bottleneck) pack many images into one tile instead of wasting the MFMA.
//
// Two element policies: bf16 (v_mfma_f32_16x16x32_bf16, throughput path) and f32
// (v_mfma_f32_16x16x4_f32, exact fp32 parity path).  fp32 accumulation in both.
#include "common.h"

namespace {

constexpr int PIXB = 80;        // LDS bytes per halo pixel: 64-byte chunk + 16-byte pad
constexpr int TM = 256;         // pixels per workgroup tile
constexpr int BN = 64;          // output channels per workgroup
constexpr int WT_BYTES = 9 * 4 * BN * 16;  // one channel chunk of packed weights, all taps

struct Geo {
    int lti, lth, ltw;  // log2 of images / rows / cols per tile
    int ph, pw;         // halo present along h / w (0 when that extent is 1)
    int HH, HW, NPH;    // halo rows, cols, pixels
    int tiles_x, tiles_y, tiles_i;
};

static bool pick_geo(int n, int h, int w, Geo* g) {
    static const int cand[5][3] = {{0, 4, 4}, {2, 3, 3}, {4, 2, 2}, {6, 1, 1}, {8, 0, 0}};
    long best = -1;
    for (int i = 0; i < 5; ++i) {
        int ti = 1 << cand[i][0], th = 1 << cand[i][1], tw = 1 << cand[i][2];
        if (i == 4 && !(h == 1 && w == 1)) continue;
        long cnt = (long)((n + ti - 1) / ti) * ((h + th - 1) / th) * ((w + tw - 1) / tw);
        if (best < 0 || cnt < best) {
            best = cnt;
            g->lti = cand[i][0]; g->lth = cand[i][1]; g->ltw = cand[i][2];
        }
    }
    int ti = 1 << g->lti, th = 1 << g->lth, tw = 1 << g->ltw;
    g->ph = h > 1; g->pw = w > 1;
    g->HH = th + 2 * g->ph; g->HW = tw + 2 * g->pw;
    g->NPH = ti * g->HH * g->HW;
    g->tiles_x = (w + tw - 1) / tw; g->tiles_y = (h + th - 1) / th; g->tiles_i = (n + ti - 1) / ti;
    return true;
}

// ------------------------------------------------------------------ element policies
struct PolBF16 {
    typedef bf16_t T;
    static constexpr int CK = 32;   // channels per 64-byte chunk
    static constexpr int KV = 8;    // channels per 16 bytes
    // forward: one 16x16x32 MFMA per (m-tile, n-tile, tap, chunk)
    static __device__ __forceinline__ int a_lane_off(int lane) { return (lane >> 4) * 16; }
    static __device__ __forceinline__ void mma_tap(f32x4 (&acc)[4][4], const char* xt, const int (&abase)[4],
                                                   int toff, const char* wtap, int lane) {
        bf16x8 bfr[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            bfr[nt] = *reinterpret_cast<const bf16x8*>(wtap + (((lane >> 4) * BN) + nt * 16 + (lane & 15)) * 16);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            bf16x8 af = *reinterpret_cast<const bf16x8*>(xt + abase[mt] + toff);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[nt], acc[mt][nt], 0, 0, 0);
        }
    }
};

struct PolF32 {
    typedef float T;
    static constexpr int CK = 16;
    static constexpr int KV = 4;
    static __device__ __forceinline__ int a_lane_off(int lane) { return (lane >> 4) * 4; }
    static __device__ __forceinline__ void mma_tap(f32x4 (&acc)[4][4], const char* xt, const int (&abase)[4],
                                                   int toff, const char* wtap, int lane) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float bfr[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                bfr[nt] = *reinterpret_cast<const float*>(wtap + ((ks * BN) + nt * 16 + (lane & 15)) * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float af = *reinterpret_cast<const float*>(xt + abase[mt] + toff + ks * 16);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfr[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
};

// ------------------------------------------------------------------ shared tile helpers
struct TileCtx {
    int n0, y0, x0;
};

__device__ __forceinline__ TileCtx decode_tile(int tile, const Geo& g) {
    TileCtx t;
    int tx = tile % g.tiles_x;
    int r = tile / g.tiles_x;
    int ty = r % g.tiles_y;
    int ti = r / g.tiles_y;
    t.x0 = tx << g.ltw; t.y0 = ty << g.lth; t.n0 = ti << g.lti;
    return t;
}

// gtab[hp] = flat pixel index (n*H + y)*W + x of halo pixel hp, or -1 outside the image / batch.
__device__ __forceinline__ void build_gtab(int* gtab, const Geo& g, const TileCtx& t, int n, int h, int w, int tid) {
    for (int hp = tid; hp < g.NPH; hp += 256) {
        int hx = hp % g.HW;
        int r = hp / g.HW;
        int hy = r % g.HH;
        int img = r / g.HH;
        int nn = t.n0 + img, y = t.y0 + hy - g.ph, x = t.x0 + hx - g.pw;
        bool ok = nn < n && y >= 0 && y < h && x >= 0 && x < w;
        gtab[hp] = ok ? (nn * h + y) * w + x : -1;
    }
}

// Halo-pixel index (centre tap) of tile pixel m.
__device__ __forceinline__ int halo_of(int m, const Geo& g) {
    int tx = m & ((1 << g.ltw) - 1);
    int ty = (m >> g.ltw) & ((1 << g.lth) - 1);
    int img = m >> (g.ltw + g.lth);
    return (img * g.HH + ty + g.ph) * g.HW + tx + g.pw;
}

// Stage one 64-byte channel chunk of the halo tile: global -> LDS, zero outside the image.
__device__ __forceinline__ void stage_halo(char* xt, const int* gtab, const char* src, int row_bytes, int off_bytes,
                                           int nph, int tid) {
    for (int s = tid; s < nph * 4; s += 256) {
        int hp = s >> 2, part = s & 3;
        int gp = gtab[hp];
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gp >= 0) v = *reinterpret_cast<const uint4*>(src + (size_t)gp * row_bytes + off_bytes + part * 16);
        *reinterpret_cast<uint4*>(xt + hp * PIXB + part * 16) = v;
    }
}

struct ConvArgs {
    const char* x1; const char* x2; int c1, c2;
    const char* wp; const float* bias;
    char* y1; char* y2; int cy1;
    int n, h, w, cout, epilogue;
    Geo g;
};

template <typename P>
__global__ __launch_bounds__(256) void conv3x3_fwd_kernel(ConvArgs a) {
    typedef typename P::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gtab_bytes = (g.NPH * 4 + 15) & ~15;
    int* gtab = reinterpret_cast<int*>(smem);
    char* xt = smem + gtab_bytes;
    char* wt = xt + g.NPH * PIXB;

    const TileCtx t = decode_tile(blockIdx.x, g);
    const int nb = blockIdx.y;
    build_gtab(gtab, g, t, a.n, a.h, a.w, tid);

    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = wave * 64 + mt * 16 + (lane & 15);
        abase[mt] = halo_of(m, g) * PIXB + P::a_lane_off(lane);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cin = a.c1 + a.c2;
    const int nchunks = cin / P::CK;
    const int kc_total = cin / P::KV;
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * P::CK;
        const char* src; int row_bytes, off_bytes;
        if (c0 < a.c1) { src = a.x1; row_bytes = a.c1 * (int)sizeof(T); off_bytes = c0 * (int)sizeof(T); }
        else { src = a.x2; row_bytes = a.c2 * (int)sizeof(T); off_bytes = (c0 - a.c1) * (int)sizeof(T); }
        stage_halo(xt, gtab, src, row_bytes, off_bytes, g.NPH, tid);
        for (int s = tid; s < 9 * 4 * BN; s += 256) {
            int co = s & 63, kc = (s >> 6) & 3, tap = s >> 8;
            const char* p = a.wp + ((size_t)(tap * kc_total + ch * 4 + kc) * a.cout + nb * BN + co) * 16;
            *reinterpret_cast<uint4*>(wt + s * 16) = *reinterpret_cast<const uint4*>(p);
        }
        __syncthreads();
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            if (!g.ph && kh != 1) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                if (!g.pw && kw != 1) continue;
                const int toff = ((kh - 1) * g.HW + (kw - 1)) * PIXB;
                P::mma_tap(acc, xt, abase, toff, wt + (kh * 3 + kw) * (4 * BN * 16), lane);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias (+ReLU), convert, transpose through LDS, coalesced 16-byte stores
    constexpr int OS = BN * (int)sizeof(T) + 16;
    char* ot = smem;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = nt * 16 + (lane & 15);
        const float bv = a.bias ? a.bias[nb * BN + col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int pix = wave * 64 + mt * 16 + (lane >> 4) * 4 + r;
                float v = acc[mt][nt][r] + bv;
                if (a.epilogue == AD_EPI_RELU) v = fmaxf(v, 0.f);
                *reinterpret_cast<T*>(ot + pix * OS + col * (int)sizeof(T)) = (T)v;
            }
    }
    __syncthreads();
    constexpr int PARTS = BN * (int)sizeof(T) / 16;
    char* yp; int cy, coff;
    if (nb * BN < a.cy1) { yp = a.y1; cy = a.cy1; coff = nb * BN; }
    else { yp = a.y2; cy = a.cout - a.cy1; coff = nb * BN - a.cy1; }
    for (int s = tid; s < TM * PARTS; s += 256) {
        int pix = s / PARTS, part = s % PARTS;
        int tx = pix & ((1 << g.ltw) - 1);
        int ty = (pix >> g.ltw) & ((1 << g.lth) - 1);
        int img = pix >> (g.ltw + g.lth);
        int nn = t.n0 + img, y = t.y0 + ty, x = t.x0 + tx;
        if (nn < a.n && y < a.h && x < a.w) {
            size_t gp = ((size_t)nn * a.h + y) * a.w + x;
            *reinterpret_cast<uint4*>(yp + (gp * cy + coff) * sizeof(T) + part * 16) =
                *reinterpret_cast<const uint4*>(ot + pix * OS + part * 16);
        }
    }
}

// ------------------------------------------------------------------ wgrad
struct WgradArgs {
    const char* x1; const char* x2; int c1, c2;
    const char* dz;
    float* ws;
    int n, h, w, cout;
    int ntiles, tiles_per_split, ncib, ncob;
    Geo g;
};

template <typename P> struct WgradPol;

template <> struct WgradPol<PolBF16> {
    static constexpr int NACC = 2;   // n-tiles per wave (one m-tile of 16 input channels)
    static constexpr int DZS = BN * 2 + 16;
    static __device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
        typedef __attribute__((address_space(3))) short4_t* lds_p;
        short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
        short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        short8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, r);
    }
    // one tile (256 pixels) of K for this workgroup: 8 k-steps of 32 pixels
    static __device__ __forceinline__ void tile(f32x4 (&acc)[9][NACC], const char* xt, const char* dzt, const int* hbase,
                                                const Geo& g, int lane, int wave) {
        const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int mt = wave & 1, nt0 = (wave >> 1) * 2;
        for (int ks = 0; ks < TM / 32; ++ks) {
            const int m = ks * 32 + grp * 8 + q;
            const int hb0 = hbase[m] * PIXB + mt * 32 + p * 8;
            const int hb1 = hbase[m + 4] * PIXB + mt * 32 + p * 8;
            bf16x8 bfr[NACC];
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                bfr[j] = tr_pair(dzt + m * DZS + (nt0 + j) * 32 + p * 8, dzt + (m + 4) * DZS + (nt0 + j) * 32 + p * 8);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    if ((!g.ph && kh != 1) || (!g.pw && kw != 1)) continue;
                    const int toff = ((kh - 1) * g.HW + (kw - 1)) * PIXB;
                    bf16x8 af = tr_pair(xt + hb0 + toff, xt + hb1 + toff);
#pragma unroll
                    for (int j = 0; j < NACC; ++j)
                        acc[kh * 3 + kw][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[kh * 3 + kw][j], 0, 0, 0);
                }
        }
    }
    // slab element (ci_local, co_local) held by (wave, lane, j, r)
    static __device__ __forceinline__ void coords(int wave, int lane, int j, int r, int* ci, int* co) {
        *ci = (wave & 1) * 16 + (lane >> 4) * 4 + r;
        *co = ((wave >> 1) * 2 + j) * 16 + (lane & 15);
    }
};

template <> struct WgradPol<PolF32> {
    static constexpr int NACC = 1;
    static constexpr int DZS = BN * 4 + 16;
    static __device__ __forceinline__ void tile(f32x4 (&acc)[9][NACC], const char* xt, const char* dzt, const int* hbase,
                                                const Geo& g, int lane, int wave) {
        const int kk = lane >> 4, i = lane & 15;
        for (int ks = 0; ks < TM / 4; ++ks) {
            const int m = ks * 4 + kk;
            const int hb = hbase[m] * PIXB + i * 4;
            const float bfr = *reinterpret_cast<const float*>(dzt + m * DZS + (wave * 16 + i) * 4);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    if ((!g.ph && kh != 1) || (!g.pw && kw != 1)) continue;
                    const int toff = ((kh - 1) * g.HW + (kw - 1)) * PIXB;
                    const float af = *reinterpret_cast<const float*>(xt + hb + toff);
                    acc[kh * 3 + kw][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfr, acc[kh * 3 + kw][0], 0, 0, 0);
                }
        }
    }
    static __device__ __forceinline__ void coords(int wave, int lane, int j, int r, int* ci, int* co) {
        *ci = (lane >> 4) * 4 + r;
        *co = wave * 16 + (lane & 15);
    }
};

// grid: x = K split, y = input-channel block (P::CK channels), z = output-channel block (64)
// ws slab layout: [split][cib][cob][tap][P::CK][64] fp32
template <typename P>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradArgs a) {
    typedef typename P::T T;
    typedef WgradPol<P> WP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gtab_bytes = (g.NPH * 4 + 15) & ~15;
    int* gtab = reinterpret_cast<int*>(smem);
    int* hbase = reinterpret_cast<int*>(smem + gtab_bytes);
    char* xt = smem + gtab_bytes + TM * 4;
    char* dzt = xt + ((g.NPH * PIXB + 15) & ~15);

    const int split = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;
    const int c0 = cib * P::CK;
    const char* src; int row_bytes, off_bytes;
    if (c0 < a.c1) { src = a.x1; row_bytes = a.c1 * (int)sizeof(T); off_bytes = c0 * (int)sizeof(T); }
    else { src = a.x2; row_bytes = a.c2 * (int)sizeof(T); off_bytes = (c0 - a.c1) * (int)sizeof(T); }

    for (int m = tid; m < TM; m += 256) hbase[m] = halo_of(m, g);

    f32x4 acc[9][WP::NACC];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < WP::NACC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
    constexpr int PARTS = BN * (int)sizeof(T) / 16;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const TileCtx t = decode_tile(tile, g);
        __syncthreads();  // previous tile's LDS reads done (also orders hbase on the first pass)
        build_gtab(gtab, g, t, a.n, a.h, a.w, tid);
        __syncthreads();
        stage_halo(xt, gtab, src, row_bytes, off_bytes, g.NPH, tid);
        for (int s = tid; s < TM * PARTS; s += 256) {
            int pix = s / PARTS, part = s % PARTS;
            int gp = gtab[hbase[pix]];
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gp >= 0)
                v = *reinterpret_cast<const uint4*>(a.dz + ((size_t)gp * a.cout + cob * BN) * sizeof(T) + part * 16);
            *reinterpret_cast<uint4*>(dzt + pix * WP::DZS + part * 16) = v;
        }
        __syncthreads();
        WP::tile(acc, xt, dzt, hbase, g, lane, wave);
    }

    float* slab = a.ws + ((size_t)(split * a.ncib + cib) * a.ncob + cob) * (9 * P::CK * BN);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < WP::NACC; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ci, co;
                WP::coords(wave, lane, j, r, &ci, &co);
                slab[(tap * P::CK + ci) * BN + co] = acc[tap][j][r];
            }
}

// dw_hwio[tap][ci][co] = sum over splits of the slabs (fixed order => deterministic)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ncib, int ncob,
                                    int ck, int cin_real, int cout) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int total = 9 * cin_real * cout;
    if (idx >= total) return;
    int co = idx % cout;
    int r = idx / cout;
    int ci = r % cin_real;
    int tap = r / cin_real;
    int cib = ci / ck, cil = ci % ck, cob = co / BN, col = co % BN;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp)
        s += ws[(((size_t)(sp * ncib + cib) * ncob + cob) * 9 + tap) * (ck * BN) + cil * BN + col];
    dw[idx] = s;
}

// ------------------------------------------------------------------ weight packing
// w_fwd[tap][kc][co][kv] = W[tap][kc*KV+kv][co]; w_dgrad[tap'][kc][ci][kv] = W[8-tap'][ci][kc*KV+kv]
template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, int cin, int cout, int cin_pad, T* __restrict__ wf,
                            T* __restrict__ wd) {
    constexpr int KV = 16 / (int)sizeof(T);
    const int total_f = 9 * cin_pad * cout;
    const int cout_pad = cout;  // dgrad contraction axis (validated multiple of KV by the launcher)
    const int total_d = wd ? 9 * cout_pad * cin_pad : 0;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int stride = gridDim.x * blockDim.x;
    for (int i = idx; i < total_f; i += stride) {
        int kv = i % KV;
        int r = i / KV;
        int co = r % cout;
        r /= cout;
        int kc = r % (cin_pad / KV);
        int tap = r / (cin_pad / KV);
        int ci = kc * KV + kv;
        wf[i] = (T)(ci < cin ? w[((size_t)tap * cin + ci) * cout + co] : 0.f);
    }
    for (int i = idx; i < total_d; i += stride) {
        int kv = i % KV;
        int r = i / KV;
        int ci = r % cin_pad;
        r /= cin_pad;
        int kc = r % (cout_pad / KV);
        int tap = r / (cout_pad / KV);
        int co = kc * KV + kv;
        wd[i] = (T)(ci < cin ? w[((size_t)(8 - tap) * cin + ci) * cout + co] : 0.f);
    }
}

template <typename P>
int launch_fwd(const ConvArgs& a, hipStream_t s) {
    const Geo& g = a.g;
    size_t stage = ((g.NPH * 4 + 15) & ~15) + (size_t)g.NPH * PIXB + WT_BYTES;
    size_t outb = (size_t)TM * (BN * sizeof(typename P::T) + 16);
    size_t lds = stage > outb ? stage : outb;
    if (lds > 160 * 1024) return ad_set_error(AD_ERR_ARG, "conv3x3_fwd: LDS %zu too large", lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_fwd_kernel<P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(g.tiles_x * g.tiles_y * g.tiles_i, a.cout / BN);
    conv3x3_fwd_kernel<P><<<grid, 256, lds, s>>>(a);
    AD_LAUNCH_CHECK("conv3x3_fwd");
    return AD_OK;
}

struct WgradPlan {
    Geo g;
    int ntiles, nsplit, tiles_per_split, ncib, ncob, ck;
    size_t ws_bytes;
};

static void plan_wgrad(int n, int h, int w, int cin, int cout, int dtype, WgradPlan* p) {
    pick_geo(n, h, w, &p->g);
    p->ck = dtype == AD_BF16 ? PolBF16::CK : PolF32::CK;
    p->ncib = cin / p->ck;
    p->ncob = cout / BN;
    p->ntiles = p->g.tiles_x * p->g.tiles_y * p->g.tiles_i;
    int want = 1024 / (p->ncib * p->ncob);
    if (want < 1) want = 1;
    if (want > p->ntiles) want = p->ntiles;
    p->tiles_per_split = (p->ntiles + want - 1) / want;
    p->nsplit = (p->ntiles + p->tiles_per_split - 1) / p->tiles_per_split;
    p->ws_bytes = (size_t)p->nsplit * p->ncib * p->ncob * 9 * p->ck * BN * sizeof(float);
}

template <typename P>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t s) {
    typedef WgradPol<P> WP;
    const Geo& g = a.g;
    size_t lds = ((g.NPH * 4 + 15) & ~15) + TM * 4 + (((size_t)g.NPH * PIXB + 15) & ~15) + (size_t)TM * WP::DZS;
    if (lds > 160 * 1024) return ad_set_error(AD_ERR_ARG, "conv3x3_wgrad: LDS %zu too large", lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(p.nsplit, p.ncib, p.ncob);
    conv3x3_wgrad_kernel<P><<<grid, 256, lds, s>>>(a);
    AD_LAUNCH_CHECK("conv3x3_wgrad");
    return AD_OK;
}

}  // namespace

extern "C" int ad_conv3x3_pack(const float* w_hwio, int cin, int cout, int cin_pad, void* w_fwd, void* w_dgrad,
                               int dtype, void* stream) {
    AD_REQUIRE(dtype == AD_BF16 || dtype == AD_F32, "ad_conv3x3_pack: bad dtype %d", dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(cin > 0 && cout > 0 && cin_pad >= cin && cin_pad % gran == 0,
               "ad_conv3x3_pack: cin=%d cin_pad=%d must be a multiple of %d", cin, cin_pad, gran);
    AD_REQUIRE(w_fwd != nullptr, "ad_conv3x3_pack: w_fwd is NULL");
    if (w_dgrad) AD_REQUIRE(cout % gran == 0, "ad_conv3x3_pack: dgrad layout needs cout %% %d == 0 (got %d)", gran, cout);
    hipStream_t s = (hipStream_t)stream;
    int total = 9 * cin_pad * cout;
    int blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == AD_BF16)
        pack_kernel<bf16_t><<<blocks, 256, 0, s>>>(w_hwio, cin, cout, cin_pad, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
    else
        pack_kernel<float><<<blocks, 256, 0, s>>>(w_hwio, cin, cout, cin_pad, (float*)w_fwd, (float*)w_dgrad);
    AD_LAUNCH_CHECK("ad_conv3x3_pack");
    return AD_OK;
}

extern "C" int ad_conv3x3_fwd(const void* x1, int c1, const void* x2, int c2, const void* w_packed, const float* bias,
                              void* y1, int cy1, void* y2, int n, int h, int w, int cout, int epilogue, int dtype,
                              void* stream) {
    AD_REQUIRE(dtype == AD_BF16 || dtype == AD_F32, "ad_conv3x3_fwd: bad dtype %d", dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0, "ad_conv3x3_fwd: bad shape n=%d h=%d w=%d", n, h, w);
    AD_REQUIRE((long)n * h * w < (1L << 31), "ad_conv3x3_fwd: more than 2^31 pixels");
    AD_REQUIRE(x1 && c1 > 0 && c1 % gran == 0, "ad_conv3x3_fwd: c1=%d must be a positive multiple of %d", c1, gran);
    AD_REQUIRE((x2 == nullptr) == (c2 == 0) && c2 % gran == 0, "ad_conv3x3_fwd: c2=%d / x2 mismatch", c2);
    AD_REQUIRE(cout > 0 && cout % BN == 0, "ad_conv3x3_fwd: cout=%d must be a multiple of %d", cout, BN);
    AD_REQUIRE(cy1 > 0 && cy1 <= cout && cy1 % BN == 0 && ((cy1 == cout) == (y2 == nullptr)),
               "ad_conv3x3_fwd: bad output split cy1=%d cout=%d", cy1, cout);
    AD_REQUIRE(epilogue == AD_EPI_NONE || epilogue == AD_EPI_RELU, "ad_conv3x3_fwd: bad epilogue %d", epilogue);
    ConvArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.c1 = c1; a.c2 = c2;
    a.wp = (const char*)w_packed; a.bias = bias;
    a.y1 = (char*)y1; a.y2 = (char*)y2; a.cy1 = cy1;
    a.n = n; a.h = h; a.w = w; a.cout = cout; a.epilogue = epilogue;
    pick_geo(n, h, w, &a.g);
    hipStream_t s = (hipStream_t)stream;
    return dtype == AD_BF16 ? launch_fwd<PolBF16>(a, s) : launch_fwd<PolF32>(a, s);
}

extern "C" size_t ad_conv3x3_wgrad_ws_bytes(int n, int h, int w, int cin, int cout, int dtype) {
    WgradPlan p;
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || cout % BN) return 0;
    plan_wgrad(n, h, w, cin, cout, dtype, &p);
    return p.ws_bytes;
}

extern "C" int ad_conv3x3_wgrad(const void* x1, int c1, const void* x2, int c2, const void* dz, float* dw_hwio,
                                int cin_real, int n, int h, int w, int cout, void* ws, size_t ws_bytes, int dtype,
                                void* stream) {
    AD_REQUIRE(dtype == AD_BF16 || dtype == AD_F32, "ad_conv3x3_wgrad: bad dtype %d", dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0, "ad_conv3x3_wgrad: bad shape");
    AD_REQUIRE((long)n * h * w < (1L << 31), "ad_conv3x3_wgrad: more than 2^31 pixels");
    AD_REQUIRE(x1 && c1 > 0 && c1 % gran == 0, "ad_conv3x3_wgrad: c1=%d must be a positive multiple of %d", c1, gran);
    AD_REQUIRE((x2 == nullptr) == (c2 == 0) && c2 % gran == 0, "ad_conv3x3_wgrad: c2=%d / x2 mismatch", c2);
    AD_REQUIRE(cout > 0 && cout % BN == 0, "ad_conv3x3_wgrad: cout=%d must be a multiple of %d", cout, BN);
    const int cin = c1 + c2;
    AD_REQUIRE(cin_real > 0 && cin_real <= cin, "ad_conv3x3_wgrad: cin_real=%d", cin_real);
    WgradPlan p;
    plan_wgrad(n, h, w, cin, cout, dtype, &p);
    if (ws == nullptr || ws_bytes < p.ws_bytes)
        return ad_set_error(AD_ERR_WS, "ad_conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
    WgradArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.c1 = c1; a.c2 = c2;
    a.dz = (const char*)dz; a.ws = (float*)ws;
    a.n = n; a.h = h; a.w = w; a.cout = cout;
    a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split; a.ncib = p.ncib; a.ncob = p.ncob;
    a.g = p.g;
    hipStream_t s = (hipStream_t)stream;
    int rc = dtype == AD_BF16 ? launch_wgrad<PolBF16>(a, p, s) : launch_wgrad<PolF32>(a, p, s);
    if (rc) return rc;
    int total = 9 * cin_real * cout;
    wgrad_reduce_kernel<<<(total + 255) / 256, 256, 0, s>>>((const float*)ws, dw_hwio, p.nsplit, p.ncib, p.ncob, p.ck,
                                                           cin_real, cout);
    AD_LAUNCH_CHECK("wgrad_reduce");
    return AD_OK;
}
