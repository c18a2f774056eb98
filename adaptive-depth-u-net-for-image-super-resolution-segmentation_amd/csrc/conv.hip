// 3x3 "same" convolution on gfx950 as an LDS-tiled implicit GEMM on the matrix cores.
//
//   forward / dgrad : rows = 64 output channels, cols = pixels of a 256-pixel spatial tile, K = 9 * Cin
//   wgrad           : rows = input channels, cols = 64 output channels, K = pixels (split over workgroups)
//
// Replaces the TensorFlow Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter ops reached from
// L.Conv2D(nf, 3, padding="same") at Super_resolution/code/train_adaptive_unet.py:202,207,259.
//
// Data layout: activations NHWC; per workgroup a halo tile [(TH+2)x(TW+2) pixels][64-byte channel
// chunk] lives in LDS with a 96-byte pixel stride (conflict-free ds_read_b128 for 16 consecutive
// pixels), so one read of the tile feeds all nine taps.  Weights are pre-packed to
// [tap][Cin/KV][Cout][KV] (KV = 16 B) so that every MFMA weight fragment is one 16-byte LDS read.
// The MFMA is oriented D[cout][pixel] = W^T * X^T: each lane then owns 4 CONSECUTIVE output channels
// of one pixel, so the epilogue writes 8/16-byte pieces of NHWC rows (and per-pixel reductions over
// channels stay inside 4 registers x 4 lane groups).
//
// Kernel variants (all persistent, software pipelined through registers):
//   conv3x3_fwd_wres_kernel   bf16, 16x16 tiles, Cin = 64: weights resident in LDS, 4 MFMA + 4 loader waves
//   conv3x3_fwd_ws_kernel     bf16, 16x16 tiles, Cin = 64 k: weights streamed with the halo, same organisation
//   conv3x3_fwd_kernel        every other shape (tiny maps, fp32): 256 threads, split-K when there are few tiles
//   conv3x3_wgrad_ws_kernel   bf16, 16x16 tiles, Cin multiple of 64: 4 MFMA + 4 loader waves
//   conv3x3_wgrad_kernel      every other shape
// The wave-specialised variants use buffer loads / stores with out-of-range offsets for the zero padding and the
// ragged tile edges, which keeps every s_waitcnt vmcnt exact (see the comments above them).
//
// Tile geometry is a runtime (TI images x TH x TW) split of 256 pixels so that tiny feature maps
// (4x4, 2x2, 1x1 at the bottleneck) pack many images into one tile instead of wasting the MFMA.
//
// Two element policies: bf16 (v_mfma_f32_16x16x32_bf16, throughput path) and f32
// (v_mfma_f32_16x16x4_f32, exact fp32 parity path).  fp32 accumulation in both.
#include "common.h"

// cache policy of the big output streams of the wave-specialised kernels (buffer_store aux bits: 1 sc0, 2 nt, 16 sc1).  0: plain
// stores.  -DAD_STORE_AUX=2 / 16 / 17: the A/B builds of tools/build_variant.sh (r05, measured: see DESIGN 6)
#ifndef AD_STORE_AUX
#define AD_STORE_AUX 0
#endif

namespace {

// LDS bytes per halo pixel: 64-byte chunk + 32-byte pad.  ds_read_b128 serves a wave in 4 NON-contiguous
// 16-lane groups ({0-3,12-15,20-27}, ...), which mix two k-slots of the fragment; a 96-byte stride keeps all
// four groups conflict-free for 16 consecutive pixels (80 bytes is 2-way), and 8 consecutive pixels x 32 B
// tile the 256-byte bank row exactly for the transposed wgrad reads.
constexpr int PIXB = 96;
constexpr int TM = 256;         // pixels per workgroup tile
constexpr int BN = 64;          // output channels per workgroup
constexpr int WSLOTS = 9 * 4 * BN / 256;   // 16-byte weight slots per thread and channel chunk (= 9)
constexpr int WT_BYTES = 9 * 4 * BN * 16;  // one channel chunk of packed weights, all taps
#define NUM_CU ad_num_cu()      // host code only: compute units of the device, queried once (api.hip)

struct Geo {
    int lti, lth, ltw;  // log2 of images / rows / cols per tile
    int ph, pw;         // halo present along h / w (0 when that extent is 1)
    int HH, HW, NPH;    // halo rows, cols, pixels
    int NPHP;           // NPH rounded up to the staging granule of the kernel variant (slots/4)
    int tiles_x, tiles_y, tiles_i;
    // Image mosaic (wave-specialised 16 x 16 kernels only; plan_mosaic): the tiles walk ONE virtual map in which the n images
    // stand side by side, `mix` per mosaic row, with a single zero line between neighbours.  A map whose extent is not a
    // multiple of 16 (the 154 / 93 / 56 / 34-wide levels of the scale-0.6 pyramid, run_experiment_adaptive_depth.sh:47-55)
    // then wastes one line in 35 instead of 14 pixels in 48.  Only addresses change: a mosaic pixel maps to (image, y, x) or
    // to "zero" (loads) / "dropped" (stores); every output pixel sees the same operands in the same order as without it.
    int mos;            // 0: off
    int mix;            // images per mosaic row
    int mpw, mph;       // pitch of an image inside the mosaic: w + 1, h + 1
    unsigned mdw, mdh;  // ceil(2^32 / pitch): v / pitch == umulhi(v, md) for v, pitch < 65 536
    int mw, mh;         // mosaic extent in pixels
};

// max_nph: most halo pixels a tile may have (the fp32 wgrad kernel's dz tile leaves room for 576, not for the 1024 of
// 64 images x 4 x 4)
static bool pick_geo(int n, int h, int w, Geo* g, int max_nph = 1 << 30) {
    static const int cand[5][3] = {{0, 4, 4}, {2, 3, 3}, {4, 2, 2}, {6, 1, 1}, {8, 0, 0}};
    // Cost of a geometry = tiles x halo pixels per tile (what a tile loads and stages; its 256 output pixels cost the same
    // MFMA work in every geometry), with the 16 x 16 tiles -- the only ones the wave-specialised kernels take, at about twice
    // the generic kernel's rate -- weighted 0.6.  (Until r03 the rule was "fewest tiles": a 154-wide map then ran as 16
    // images x 4 x 4 tiles, 3 042 tiles against 3 200, on the generic kernel with 2.25 halo pixels per output pixel: 0.17 of
    // peak where the 16 x 16 geometry gives 0.5; likewise the 56- and 34-wide levels of the 0.6 pyramid.)
    double best = -1.0;
    const int pad = (h > 1 || w > 1) ? 2 : 0;
    for (int i = 0; i < 5; ++i) {
        int ti = 1 << cand[i][0], th = 1 << cand[i][1], tw = 1 << cand[i][2];
        if (i == 4 && !(h == 1 && w == 1)) continue;
        const int nph = ti * (th + pad) * (tw + pad);
        if (nph > max_nph) continue;
        long cnt = (long)((n + ti - 1) / ti) * ((h + th - 1) / th) * ((w + tw - 1) / tw);
        const double cost = (double)cnt * nph * (i == 0 ? 0.6 : 1.0);
        if (best < 0 || cost < best) {
            best = cost;
            g->lti = cand[i][0]; g->lth = cand[i][1]; g->ltw = cand[i][2];
        }
    }
    int ti = 1 << g->lti, th = 1 << g->lth, tw = 1 << g->ltw;
    g->ph = g->pw = h > 1 || w > 1;     // a 1 x W / H x 1 map keeps both halos: the rows / columns that do not exist are zero-filled
    g->HH = th + 2 * g->ph; g->HW = tw + 2 * g->pw;
    g->NPH = ti * g->HH * g->HW;
    g->NPHP = g->NPH;
    g->tiles_x = (w + tw - 1) / tw; g->tiles_y = (h + th - 1) / th; g->tiles_i = (n + ti - 1) / ti;
    g->mos = 0; g->mix = 0; g->mpw = g->mph = 0; g->mdw = g->mdh = 0; g->mw = g->mh = 0;
    return true;
}

// The mosaic of n images of h x w pixels that takes the fewest ROUNDS of `per_round` 16 x 16 tiles (the persistent kernels deal
// `per_round` tiles to the chip at a time, so a launch lasts ceil(tiles / per_round) rounds; among equals: the fewest tiles).
// false (g untouched) unless that is at least one round less than the per-image tiling needs (r04: on the scale-0.7 pyramid at
// batch 8 every level keeps its 8 or 9 rounds whatever the mosaic saves in tiles), or when the extents leave the range of
// the reciprocal division.
static bool plan_mosaic(int n, int h, int w, int per_round, Geo* g) {
    if (ad_option(AD_OPT_NO_MOSAIC) || n < 2 || h < 2 || w < 2 || h >= 32768 || w >= 32768 || per_round < 1) return false;
    const long plain = (long)n * ((h + 15) / 16) * ((w + 15) / 16);
    const long plain_rounds = (plain + per_round - 1) / per_round;
    long best = plain, best_rounds = plain_rounds;
    int best_ix = 0;
    for (int ix = 1; ix <= n; ++ix) {
        const int iy = (n + ix - 1) / ix;
        const long mw = (long)ix * (w + 1) - 1, mh = (long)iy * (h + 1) - 1;
        if (mw >= 65536 || mh >= 65536) continue;
        const long cnt = ((mw + 15) / 16) * ((mh + 15) / 16);
        const long rounds = (cnt + per_round - 1) / per_round;
        if (rounds < best_rounds || (rounds == best_rounds && best_ix && cnt < best)) { best = cnt; best_rounds = rounds; best_ix = ix; }
    }
    if (!best_ix) return false;
    const int iy = (n + best_ix - 1) / best_ix;
    g->mos = 1; g->mix = best_ix; g->mpw = w + 1; g->mph = h + 1;
    g->mdw = (unsigned)((0x100000000ULL + (unsigned)g->mpw - 1) / (unsigned)g->mpw);
    g->mdh = (unsigned)((0x100000000ULL + (unsigned)g->mph - 1) / (unsigned)g->mph);
    g->mw = best_ix * (w + 1) - 1; g->mh = iy * (h + 1) - 1;
    g->tiles_x = (g->mw + 15) / 16; g->tiles_y = (g->mh + 15) / 16; g->tiles_i = 1;
    return true;
}

// ------------------------------------------------------------------ element policies
// mma_chunk: acc[mt][nt] += sum over the taps of W_tap^T (rows: 16 couts of n-tile nt) x X (cols: 16 pixels of
// m-tile mt) for one channel chunk.  HALO = false: the feature map is 1x1, only the centre tap exists.
// bf16: the fragments of tap t+1 are read from LDS while the 16 MFMAs of tap t issue (two register sets,
// selected by the compile-time parity of t), so LDS latency is hidden inside a single wave.
template <typename E>
struct Pol16 {
    typedef E T;
    typedef typename Half16<E>::v8 v8;
    static constexpr int CK = 32;   // channels per 64-byte chunk
    static constexpr int KV = 8;    // channels per 16 bytes
    static __device__ __forceinline__ int a_lane_off(int lane) { return (lane >> 4) * 16; }
    template <int MT, bool HALO>
    static __device__ __forceinline__ void mma_chunk(f32x4 (&acc)[MT][4], const char* xt, const int (&abase)[MT],
                                                     int row_bytes, const char* wt, int lane) {
        constexpr int NTAP = HALO ? 9 : 1;
        v8 wf[2][4], xf[2][MT];
        const char* wl = wt + (((lane >> 4) * BN) + (lane & 15)) * 16;
#define AD_LOAD_TAP(BUF, TAP)                                                                             \
    {                                                                                                     \
        const int tap_ = HALO ? (TAP) : 4;                                                                \
        const int toff_ = HALO ? ((TAP) / 3 - 1) * row_bytes + ((TAP) % 3 - 1) * PIXB : 0;                \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                  \
            wf[BUF][nt] = *reinterpret_cast<const v8*>(wl + tap_ * (4 * BN * 16) + nt * 256);         \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                 \
            xf[BUF][mt] = *reinterpret_cast<const v8*>(xt + abase[mt] + toff_);                       \
    }
        AD_LOAD_TAP(0, 0)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            if (t + 1 < NTAP) AD_LOAD_TAP((t + 1) & 1, t + 1)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = Half16<E>::mfma(wf[t & 1][nt], xf[t & 1][mt], acc[mt][nt]);
        }
#undef AD_LOAD_TAP
    }
    // 16-pixel-wide tiles with 4 m-tiles per wave: m-tile mt is tile row 4*wave + mt, so tap (dy, dx) of m-tile mt
    // reads halo row mt + dy at column offset dx.  The six halo rows a wave touches are fetched once per dx and
    // reused by the three dy taps: 9*4 weight + 3*6 pixel fragment reads per chunk instead of 9*4 + 9*4 -- the LDS
    // pipe, not the MFMA, is what saturates first in this kernel.  abase0 = centre-tap offset of m-tile 0.
    // hook(st) is called after the MFMAs of step st (compile-time st): the caller spreads other work (the global
    // stores of the previous item's results) through the MFMA stream.
    // INIT: the first tap step starts the accumulators from c0[nt] (the bias of the lane's four channels of n-tile nt)
    // instead of reading acc, which saves the 64 v_mov of a separate initialisation per item.
    template <bool INIT = false, typename Hook>
    static __device__ __forceinline__ void mma_chunk_rows(f32x4 (&acc)[4][4], const char* xt, int abase0, int row_bytes,
                                                          const char* wt, int lane, const Hook& hook,
                                                          const f32x4* c0 = nullptr) {
        v8 wf[2][4], xr[2][6];
        const char* wl = wt + (((lane >> 4) * BN) + (lane & 15)) * 16;
        const char* xl = xt + abase0 - row_bytes - PIXB;       // halo row -1, column -1 of m-tile 0
#define AD_LOAD_W(BUF, TAP)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                      \
        wf[BUF][nt] = *reinterpret_cast<const v8*>(wl + (TAP) * (4 * BN * 16) + nt * 256);
#define AD_LOAD_R(BUF, DX)                                                                                \
    _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                         \
        xr[BUF][r] = *reinterpret_cast<const v8*>(xl + r * row_bytes + (DX) * PIXB);
        AD_LOAD_R(0, 0)
        AD_LOAD_W(0, 0)
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int dx = st / 3, dy = st % 3;
            if (st + 1 < 9) {
                const int ndx = (st + 1) / 3, ndy = (st + 1) % 3;
                AD_LOAD_W((st + 1) & 1, ndy * 3 + ndx)
                if (ndy == 0) AD_LOAD_R(ndx & 1, ndx)
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = Half16<E>::mfma(wf[st & 1][nt], xr[dx & 1][mt + dy], INIT && st == 0 ? c0[nt] : acc[mt][nt]);
            hook(st);
        }
#undef AD_LOAD_W
#undef AD_LOAD_R
    }
};
typedef Pol16<bf16_t> PolBF16;
typedef Pol16<f16_t> PolF16;

struct PolF32 {
    typedef float T;
    static constexpr int CK = 16;
    static constexpr int KV = 4;
    static __device__ __forceinline__ int a_lane_off(int lane) { return (lane >> 4) * 4; }
    template <int MT, bool HALO>
    static __device__ __forceinline__ void mma_chunk(f32x4 (&acc)[MT][4], const char* xt, const int (&abase)[MT],
                                                     int row_bytes, const char* wt, int lane) {
        constexpr int NTAP = HALO ? 9 : 1;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int tap = HALO ? t : 4;
            const int toff = HALO ? (t / 3 - 1) * row_bytes + (t % 3 - 1) * PIXB : 0;
            const char* wtap = wt + tap * (4 * BN * 16);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float wf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    wf[nt] = *reinterpret_cast<const float*>(wtap + ((ks * BN) + nt * 16 + (lane & 15)) * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float xf = *reinterpret_cast<const float*>(xt + abase[mt] + toff + ks * 16);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt], xf, acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }
};

// ------------------------------------------------------------------ shared tile helpers
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0) -- on gfx9 stores and
// loads share that counter -- which would expose the full latency of the epilogue's global stores and of the
// prefetch loads at every barrier.  The LDS hand-offs in these kernels need only lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// gtab[hp] = flat pixel index (n*H + y)*W + x of halo pixel hp.  Outside the image / batch it holds the
// bitwise NOT (negative) of the nearest valid pixel's index: the loader always reads a valid, well spread
// address (no branch, no hot "zero page" line) and the zero fill is applied when the registers go to LDS.
template <int NTHR = 256>
__device__ __forceinline__ void build_gtab(int* gtab, const Geo& g, const TileCtx& t, int n, int h, int w, int tid) {
    for (int hp = tid; hp < g.NPHP; hp += NTHR) {
        int hx = hp % g.HW;
        int r = hp / g.HW;
        int hy = r % g.HH;
        int img = r / g.HH;
        int nn = t.n0 + img, y = t.y0 + hy - g.ph, x = t.x0 + hx - g.pw;
        bool ok = hp < g.NPH && nn < n && y >= 0 && y < h && x >= 0 && x < w;
        int idx = (min(nn, n - 1) * h + min(max(y, 0), h - 1)) * w + min(max(x, 0), w - 1);
        gtab[hp] = ok ? idx : ~idx;
    }
}

// Halo-pixel index (centre tap) of tile pixel m.
__device__ __forceinline__ int halo_of(int m, const Geo& g) {
    int tx = m & ((1 << g.ltw) - 1);
    int ty = (m >> g.ltw) & ((1 << g.lth) - 1);
    int img = m >> (g.ltw + g.lth);
    return (img * g.HH + ty + g.ph) * g.HW + tx + g.pw;
}

// Register-staged halo chunk: XS 16-byte slots per thread (slot s = tid + NTHR*i -> pixel s>>2, part s&3).
// Branch-free; gtab is sized for XS*NTHR/4 entries (-1 beyond the halo), so no bounds test is needed.
template <int XS, int NTHR = 256>
__device__ __forceinline__ void load_halo(uint4 (&xr)[XS], const int* gtab, const char* src, int row_bytes,
                                          int off_bytes, int tid) {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
        const int s = tid + NTHR * i;
        const int gp = gtab[s >> 2];
        const int idx = gp >= 0 ? gp : ~gp;
        xr[i] = *reinterpret_cast<const uint4*>(src + (size_t)idx * row_bytes + off_bytes + (s & 3) * 16);
    }
}

// Scalar-slot forms of the two helpers above: the forward kernel keeps its staged slots in NAMED registers,
// because hipcc leaves a `uint4 xr[XS]` that is live across its multi-exit pipeline loop in scratch memory
// (every load then drains with vmcnt(0) into a scratch store, which serialises the prefetch).
__device__ __forceinline__ uint4 load_halo_slot(const int* gtab, const char* src, int row_bytes, int off_bytes, int s) {
    const int gp = gtab[s >> 2];
    const int idx = gp >= 0 ? gp : ~gp;
    return *reinterpret_cast<const uint4*>(src + (size_t)idx * row_bytes + off_bytes + (s & 3) * 16);
}

__device__ __forceinline__ void store_halo_slot(const uint4& v, char* xt, const int* gtab, int nph, int s) {
    const bool ok = gtab[s >> 2] >= 0;
    if (s < nph * 4) *reinterpret_cast<uint4*>(xt + (s >> 2) * PIXB + (s & 3) * 16) = ok ? v : make_uint4(0, 0, 0, 0);
}

template <int XS, int NTHR = 256>
__device__ __forceinline__ void store_halo(const uint4 (&xr)[XS], char* xt, const int* gtab, int nph, int tid) {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
        const int s = tid + NTHR * i;
        const bool ok = gtab[s >> 2] >= 0;
        if (s < nph * 4)
            *reinterpret_cast<uint4*>(xt + (s >> 2) * PIXB + (s & 3) * 16) = ok ? xr[i] : make_uint4(0, 0, 0, 0);
    }
}

// One 16-byte slot of a packed weight chunk (9 taps x 4 kc x 64 co x 16 B); slots past the end re-read slot 0
// and land in the LDS padding.  Slots are held in named scalars (not an array) so they stay in registers.
__device__ __forceinline__ uint4 load_w_slot(const char* wp, int kc_total, int ch, int cout, int nb, int s) {
    s = s < 9 * 4 * BN ? s : 0;
    const int co = s & 63, kc = (s >> 6) & 3, tap = s >> 8;
    return *reinterpret_cast<const uint4*>(wp + ((size_t)(tap * kc_total + ch * 4 + kc) * cout + nb * BN + co) * 16);
}

#define AD_EPI_LN_RELU 2        // internal: ad_conv3x3_ln_relu_fwd
#define AD_EPI_MASK 3           // internal: ad_conv3x3_dgrad_relu (ReLU-grad of the producer fused into this dgrad)
#define AD_EPI_LNBWD 4          // internal: ad_conv3x3_dgrad_ln_bwd (LayerNorm + ReLU backward of the producer fused into this dgrad)
#define AD_EPI_LN_STATS 5        // internal: ad_conv3x3_ln_relu_fwd without an activation tensor: z and the LayerNorm statistics only
#define AD_EPI_LN_ACT 6          // internal: ad_conv3x3_ln_relu_fwd without z (inference): the activation only (y1 = act)
#define AD_ERR_UNFUSED 1000     // internal: no fused kernel for this shape, run the two launches

struct ConvArgs {
    const char* x1; const char* x2; int c1, c2;
    const char* wp; const float* bias;
    char* y1; char* y2; int cy1;
    int n, h, w, cout, epilogue;   // cout: output channels rounded up to whole 64-channel blocks (weight pack / block maths)
    int cout_real;                 // channels that exist in y1 (+ y2); a ragged last block stores only those
    int ntiles;
    int ksplit;                // > 1: the channel chunks of an item are split over ksplit workgroups (tiny maps)
    float* slab;               // split-K partial sums [ksplit][n*h*w][cout] fp32
    unsigned long long* dbg;   // diagnostic builds only (-DAD_STAMP): per-workgroup phase cycle sums
    // fused LayerNorm + ReLU epilogue (epilogue == AD_EPI_LN_RELU, cout == 64): y1 receives z, a_out the activation
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    char* a_out; float* ln_mean; float* ln_rstd;
    // AD_EPI_MASK: outputs of y1's blocks are zeroed where mask1 (the producer's stored ReLU output, laid out like y1)
    // is not positive; dbias_part[workgroup][64] receives the column sums of what the workgroup stored into y1
    const char* mask1; float* dbias_part;
    // AD_EPI_LNBWD: the dgrad's 64 output channels are the gradient of a LayerNorm + ReLU output; lnb_z is that layer's
    // stored conv output (laid out like y1), ln_mean / ln_rstd / ln_gamma / ln_beta its statistics and parameters (inputs
    // here); y1 receives dz of that layer, dbias_part[workgroup][wave][3][64] the column sums for dgamma / dbeta / dbias
    const char* lnb_z;
    // weight pack row length in output channels when the launch computes a slice of the pack's output blocks (wp then
    // points at the slice's first block); 0: the pack is exactly a.cout wide.  Wave-specialised 16-bit kernels only.
    int wcout = 0;
    Geo g;
};

// -DAD_CLOCK (diagnostic build, tools/inkernel_clock.py): the shader clock a launch really runs at = cycles (s_memtime) over wall
// time (s_memrealtime, 100 MHz) across the life of MFMA wave 0 of every workgroup (MI355X_MICROARCH.md, "DVFS give-back" item 6).
// The stamps go to a buffer of their own that no kernel reads; the shipped build has none of this.
#ifdef AD_CLOCK
__device__ unsigned long long g_ad_clock[2 * 4096];
#define AD_CLOCK_BEGIN                                                                                   \
    const unsigned long long ck_c0_ = __builtin_amdgcn_s_memtime(), ck_w0_ = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);
#define AD_CLOCK_END(WAVE, LANE, BID)                                                                    \
    if ((WAVE) == 0 && (LANE) == 0 && (BID) < 4096) {                                                    \
        g_ad_clock[2 * (BID)] = __builtin_amdgcn_s_memtime() - ck_c0_;                                   \
        g_ad_clock[2 * (BID) + 1] = __builtin_amdgcn_s_memrealtime() - ck_w0_;                           \
    }
#else
#define AD_CLOCK_BEGIN
#define AD_CLOCK_END(WAVE, LANE, BID)
#endif

#ifdef AD_STAMP
#define STAMP(slot)                                            \
    do {                                                       \
        unsigned long long now_ = clock64();                   \
        if (tid == 0) st[slot] += now_ - t_last;               \
        t_last = now_;                                         \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

// Persistent forward / dgrad kernel.  Work item = (tile, 64-channel output block); the item's channel
// chunks form the K loop.  LDS: [gtab0][gtab1][halo chunk][weight chunk]; the output tile aliases the
// halo+weight region during the epilogue.
// 256 threads: 4 waves, each 64 pixels x 64 output channels (4 x 4 accumulator tiles of 16x16): 8 LDS
// fragment reads feed 16 MFMAs per tap, which keeps the LDS pipe at ~50 % of the MFMA time.
constexpr int FT = 256;                        // threads of the forward kernel
constexpr int FWS = (9 * 4 * BN + FT - 1) / FT;  // weight slots per thread (= 9)
constexpr int FMT = TM / (FT / 64) / 16;       // m-tiles per wave (= 4)

template <typename P, int XS, bool HALO>
__global__ __launch_bounds__(FT, XS <= 6 ? 2 : 1) void conv3x3_fwd_kernel(ConvArgs a) {
    typedef typename P::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gtab_bytes = g.NPHP * 4;
    int* gtab0 = reinterpret_cast<int*>(smem);
    int* gtab1 = reinterpret_cast<int*>(smem + gtab_bytes);
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * gtab_bytes);   // 2 x 64 biases (item parity)
    char* xt = smem + 2 * gtab_bytes + 2 * BN * 4;
    char* wt = xt + ((g.NPH * PIXB + 15) & ~15);

    const int nblk = a.cout / BN;
    const int ksplit = a.ksplit;
    const int nitems = a.ntiles * nblk * ksplit;     // item = (tile, output block, K slice)
    const int cin = a.c1 + a.c2;
    const int nch = cin / P::CK;
    const int kc_total = cin / P::KV;
    constexpr int TSZ = (int)sizeof(T);

    int abase[FMT];
#pragma unroll
    for (int mt = 0; mt < FMT; ++mt) {
        int m = wave * (16 * FMT) + mt * 16 + (lane & 15);
        abase[mt] = halo_of(m, g) * PIXB + P::a_lane_off(lane);
    }

    static_assert(XS <= 16, "halo slots are held in up to sixteen named registers");
    uint4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
    static_assert(FWS == 9, "weight slots are held in nine named registers");
    uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
    float bq = 0.f;   // bias of channel (tid & 63), fetched with the item's first chunk (a load in the epilogue
                      // would sit behind the prefetch in the in-order vmcnt queue and drain it)

    // issue the global loads of (tile described by gt, channel chunk ch, output block nb) into registers
#define FWD_ISSUE(GT, CH, NB, FIRST)                                                                      \
    do {                                                                                              \
        const int c0_ = (CH) * P::CK;                                                                 \
        const bool first_ = c0_ < a.c1;                                                               \
        const char* src_ = first_ ? a.x1 : a.x2;                                                      \
        const int rb_ = (first_ ? a.c1 : a.c2) * TSZ;                                                 \
        const int ob_ = (first_ ? c0_ : c0_ - a.c1) * TSZ;                                            \
        if ((FIRST) && a.bias) bq = (NB) * BN + (tid & 63) < a.cout_real ? a.bias[(NB) * BN + (tid & 63)] : 0.f; \
        if constexpr (XS > 0) x0 = load_halo_slot((GT), src_, rb_, ob_, tid + 0 * FT);                                  \
        if constexpr (XS > 1) x1 = load_halo_slot((GT), src_, rb_, ob_, tid + 1 * FT);                                  \
        if constexpr (XS > 2) x2 = load_halo_slot((GT), src_, rb_, ob_, tid + 2 * FT);                                  \
        if constexpr (XS > 3) x3 = load_halo_slot((GT), src_, rb_, ob_, tid + 3 * FT);                                  \
        if constexpr (XS > 4) x4 = load_halo_slot((GT), src_, rb_, ob_, tid + 4 * FT);                                  \
        if constexpr (XS > 5) x5 = load_halo_slot((GT), src_, rb_, ob_, tid + 5 * FT);                                  \
        if constexpr (XS > 6) x6 = load_halo_slot((GT), src_, rb_, ob_, tid + 6 * FT);                                  \
        if constexpr (XS > 7) x7 = load_halo_slot((GT), src_, rb_, ob_, tid + 7 * FT);                                  \
        if constexpr (XS > 8) x8 = load_halo_slot((GT), src_, rb_, ob_, tid + 8 * FT);                                  \
        if constexpr (XS > 9) x9 = load_halo_slot((GT), src_, rb_, ob_, tid + 9 * FT);                                  \
        if constexpr (XS > 10) x10 = load_halo_slot((GT), src_, rb_, ob_, tid + 10 * FT);                               \
        if constexpr (XS > 11) x11 = load_halo_slot((GT), src_, rb_, ob_, tid + 11 * FT);                               \
        if constexpr (XS > 12) x12 = load_halo_slot((GT), src_, rb_, ob_, tid + 12 * FT);                               \
        if constexpr (XS > 13) x13 = load_halo_slot((GT), src_, rb_, ob_, tid + 13 * FT);                               \
        if constexpr (XS > 14) x14 = load_halo_slot((GT), src_, rb_, ob_, tid + 14 * FT);                               \
        if constexpr (XS > 15) x15 = load_halo_slot((GT), src_, rb_, ob_, tid + 15 * FT);                               \
        w0 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid);                                    \
        w1 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + FT);                               \
        w2 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 2 * FT);                           \
        w3 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 3 * FT);                           \
        w4 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 4 * FT);                           \
        w5 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 5 * FT);                           \
        w6 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 6 * FT);                           \
        w7 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 7 * FT);                           \
        w8 = load_w_slot(a.wp, kc_total, (CH), a.cout, (NB), tid + 8 * FT);                           \
    } while (0)

#ifdef AD_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = clock64();
    const unsigned long long t_begin = t_last;
#endif
    int item = blockIdx.x;
    int cur = 0;
    if (item < nitems) {
        const TileCtx t0 = decode_tile(item / (nblk * ksplit), g);
        build_gtab<FT>(gtab0, g, t0, a.n, a.h, a.w, tid);
        lds_barrier();
        FWD_ISSUE(gtab0, (item % ksplit) * nch / ksplit, (item / ksplit) % nblk, true);
    }
    STAMP(0);

    for (; item < nitems; item += gridDim.x) {
        const int tile = item / (nblk * ksplit), nb = (item / ksplit) % nblk, ksl = item % ksplit;
        const int ch_lo = ksl * nch / ksplit, ch_hi = (ksl + 1) * nch / ksplit;
        const TileCtx t = decode_tile(tile, g);
        const int next = item + gridDim.x;
        const bool has_next = next < nitems;
        int* gt_cur = cur ? gtab1 : gtab0;
        int* gt_nxt = cur ? gtab0 : gtab1;

        f32x4 acc[FMT][4];
#pragma unroll
        for (int i = 0; i < FMT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            // registers hold (item, ch): publish them to LDS (previous readers are past their barrier)
            if constexpr (XS > 0) store_halo_slot(x0, xt, gt_cur, g.NPH, tid + 0 * FT);
            if constexpr (XS > 1) store_halo_slot(x1, xt, gt_cur, g.NPH, tid + 1 * FT);
            if constexpr (XS > 2) store_halo_slot(x2, xt, gt_cur, g.NPH, tid + 2 * FT);
            if constexpr (XS > 3) store_halo_slot(x3, xt, gt_cur, g.NPH, tid + 3 * FT);
            if constexpr (XS > 4) store_halo_slot(x4, xt, gt_cur, g.NPH, tid + 4 * FT);
            if constexpr (XS > 5) store_halo_slot(x5, xt, gt_cur, g.NPH, tid + 5 * FT);
            if constexpr (XS > 6) store_halo_slot(x6, xt, gt_cur, g.NPH, tid + 6 * FT);
            if constexpr (XS > 7) store_halo_slot(x7, xt, gt_cur, g.NPH, tid + 7 * FT);
            if constexpr (XS > 8) store_halo_slot(x8, xt, gt_cur, g.NPH, tid + 8 * FT);
            if constexpr (XS > 9) store_halo_slot(x9, xt, gt_cur, g.NPH, tid + 9 * FT);
            if constexpr (XS > 10) store_halo_slot(x10, xt, gt_cur, g.NPH, tid + 10 * FT);
            if constexpr (XS > 11) store_halo_slot(x11, xt, gt_cur, g.NPH, tid + 11 * FT);
            if constexpr (XS > 12) store_halo_slot(x12, xt, gt_cur, g.NPH, tid + 12 * FT);
            if constexpr (XS > 13) store_halo_slot(x13, xt, gt_cur, g.NPH, tid + 13 * FT);
            if constexpr (XS > 14) store_halo_slot(x14, xt, gt_cur, g.NPH, tid + 14 * FT);
            if constexpr (XS > 15) store_halo_slot(x15, xt, gt_cur, g.NPH, tid + 15 * FT);
            *reinterpret_cast<uint4*>(wt + tid * 16) = w0;
            *reinterpret_cast<uint4*>(wt + (tid + FT) * 16) = w1;
            *reinterpret_cast<uint4*>(wt + (tid + 2 * FT) * 16) = w2;
            *reinterpret_cast<uint4*>(wt + (tid + 3 * FT) * 16) = w3;
            *reinterpret_cast<uint4*>(wt + (tid + 4 * FT) * 16) = w4;
            *reinterpret_cast<uint4*>(wt + (tid + 5 * FT) * 16) = w5;
            *reinterpret_cast<uint4*>(wt + (tid + 6 * FT) * 16) = w6;
            *reinterpret_cast<uint4*>(wt + (tid + 7 * FT) * 16) = w7;
            *reinterpret_cast<uint4*>(wt + (tid + 8 * FT) * 16) = w8;
            if (ch == ch_lo && tid < BN) bias_lds[(cur ? BN : 0) + tid] = bq;
            STAMP(1);   // wait for prefetched loads + LDS stores
            const bool last = ch == ch_hi - 1;
            if (last && has_next) {
                const TileCtx tn = decode_tile(next / (nblk * ksplit), g);
                build_gtab<FT>(gt_nxt, g, tn, a.n, a.h, a.w, tid);
            }
            STAMP(2);   // gtab build
            lds_barrier();
            STAMP(3);   // barrier A
            // prefetch the next chunk (or the next item's first chunk) while this one is multiplied
            if (!last || has_next) {
                const int* gt_p = last ? gt_nxt : gt_cur;
                const int ch_p = last ? (next % ksplit) * nch / ksplit : ch + 1;
                const int nb_p = last ? (next / ksplit) % nblk : nb;
                FWD_ISSUE(gt_p, ch_p, nb_p, last);
            }
            STAMP(4);   // issue prefetch
            P::template mma_chunk<FMT, HALO>(acc, xt, abase, g.HW * PIXB, wt, lane);
            STAMP(5);   // MFMA phase
            lds_barrier();
            STAMP(6);   // barrier B
        }

        // ---- epilogue straight from the accumulators: each lane owns 4 consecutive couts (8 B bf16 / 16 B f32)
        // of pixel (lane & 15) of every m-tile; the four lane groups of a wave complete 32/64-byte row pieces.
        // No LDS round trip and no barrier: barrier B already fenced the staging buffers.
        {
            // the block's four 16-channel n-tiles each go to y1 (channels below cy1) or y2, or nowhere (padding of a
            // ragged last block): the output split and the channel count need only be multiples of 16
            const float* bl = bias_lds + (cur ? BN : 0);
            float4 bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const float4*>(bl + nt * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int mt = 0; mt < FMT; ++mt) {
                const int pix = wave * (16 * FMT) + mt * 16 + (lane & 15);
                const int tx = pix & ((1 << g.ltw) - 1);
                const int ty = (pix >> g.ltw) & ((1 << g.lth) - 1);
                const int img = pix >> (g.ltw + g.lth);
                const int nn = t.n0 + img, y = t.y0 + ty, x = t.x0 + tx;
                if (nn < a.n && y < a.h && x < a.w && ksplit > 1) {
                    // split-K: raw fp32 partial sums; bias / ReLU / conversion happen in splitk_finalize_kernel
                    const size_t gp = ((size_t)nn * a.h + y) * a.w + x;
                    float* dst = a.slab + ((size_t)ksl * a.n * a.h * a.w + gp) * a.cout + nb * BN + (lane >> 4) * 4;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        *reinterpret_cast<float4*>(dst + nt * 16) =
                            make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
                } else if (nn < a.n && y < a.h && x < a.w) {
                    const size_t gp = ((size_t)nn * a.h + y) * a.w + x;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int cg = nb * BN + nt * 16;                       // first channel of this n-tile
                        if (cg >= a.cout_real) continue;
                        T* dst = (cg < a.cy1 ? reinterpret_cast<T*>(a.y1) + gp * a.cy1 + cg
                                             : reinterpret_cast<T*>(a.y2) + gp * (a.cout_real - a.cy1) + (cg - a.cy1)) +
                                 (lane >> 4) * 4;
                        float v[4] = {acc[mt][nt][0] + bv[nt].x, acc[mt][nt][1] + bv[nt].y, acc[mt][nt][2] + bv[nt].z,
                                      acc[mt][nt][3] + bv[nt].w};
                        if (a.epilogue == AD_EPI_RELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        if constexpr (sizeof(T) == 2) {
                            typedef typename Half16<T>::v4 h4;
                            h4 pk = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                            *reinterpret_cast<h4*>(dst) = pk;
                        } else {
                            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            }
        }
        STAMP(7);   // epilogue
        cur ^= 1;
    }
#undef FWD_ISSUE
#ifdef AD_STAMP
    if (tid == 0 && a.dbg) {
        for (int i = 0; i < 8; ++i) a.dbg[blockIdx.x * 9 + i] = st[i];
        a.dbg[blockIdx.x * 9 + 8] = clock64() - t_begin;
    }
#endif
}

// y = convert(relu?(bias + sum over K slices of the fp32 partials)), fixed summation order; 4 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                              T* __restrict__ y1, T* __restrict__ y2, int cy1, int64_t npix,
                                                              int cout, int ksplit, int relu, int cout_real) {
    const int vecs = cout / 4;
    const int64_t total = npix * vecs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / vecs;
        const int co = (int)(i - p * vecs) * 4;
        if (co >= cout_real) continue;             // padding of a ragged last 64-channel block
        float4 s = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ksplit; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(slab + ((size_t)k * npix + p) * cout + co);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        T* dst = co < cy1 ? y1 + p * cy1 + co : y2 + p * (cout_real - cy1) + (co - cy1);
        dst[0] = (T)s.x; dst[1] = (T)s.y; dst[2] = (T)s.z; dst[3] = (T)s.w;
    }
}

// K slices for a launch with few (tile, output block) items: aim at ~2 workgroups per CU, at least 2 chunks a slice
// Shape queries may name any integers; the launch paths (and every kernel's 32-bit pixel index) need the whole batch's pixel
// count below 2^31.  Queries answer "not supported / no workspace" beyond that instead of planning with overflowed ints
// (found by the host-side sanitizer build, tests/test_host_sanitizers.py).
static inline bool pixels_ok(int n, int h, int w) { return n > 0 && h > 0 && w > 0 && (long long)n * h * w < (1LL << 31); }

static int pick_ksplit(int nitems, int nch) {
    if (nitems >= NUM_CU || nch < 4) return 1;
    int ks = (2 * NUM_CU + nitems - 1) / nitems;
    if (ks > nch / 2) ks = nch / 2;
    return ks < 2 ? 1 : ks;
}

// ------------------------------------------------------------------ forward / dgrad, weights resident
// Cin = 64 (two channel chunks) and 16x16 tiles: the 64 -> 64 layers that dominate the full-resolution levels
// (and the dgrads of 128 -> 64).  One 512-thread workgroup owns a CU and ONE 64-channel output block for the whole
// launch; its 9 x 64 x 64 weights (73.7 KB) are staged in LDS once.  The eight waves are specialised:
//   * waves 0-3 (one per SIMD) only read fragments, issue MFMAs and store their 64 pixels x 64 channels straight
//     from the accumulators (each lane owns 4 consecutive channels of a pixel: 8-byte pieces of NHWC rows);
//   * waves 4-7 (the second wave of each SIMD) only move data: they compute the halo addresses, keep the next
//     item's two channel chunks in flight in registers a full item ahead, and write them to the two LDS halo
//     buffers in the shadow of the MFMA phases.
// Two workgroup barriers per item hand the buffers over: B0 (X0 = chunk 0 of this item is complete, X1 may be
// refilled) and B1 (X1 = chunk 1 is complete, X0 may be refilled).
// Every global access is a BUFFER load/store issued unconditionally: out-of-image halo pixels and out-of-tile
// output pixels get the offset 0x80000000, which the hardware range check turns into "load zeros" / "drop the
// store".  That gives the zero padding for free and keeps the number of memory operations per iteration a
// compile-time constant: with loads or stores under a branch hipcc can no longer count what is in flight and
// falls back to s_waitcnt vmcnt(0), which drains the prefetch at every stage.  (Hence the launcher's 2 GiB limit
// on each tensor for this path.)
// LDS: [X0][X1][W chunk0][W chunk1] = 135.9 KB.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WR_T = 512;                      // threads: 4 MFMA waves + 4 loader waves
constexpr int WR_XS = 6;                       // halo slots per loader thread (324 pixels x 4 parts / 256)
constexpr int WR_XB = (324 * PIXB + 15) & ~15; // one halo buffer (31,104 B)
constexpr size_t WR_LDS = 2 * WR_XB + 2 * WT_BYTES + 3 * BN * 4;   // + [gamma][beta][bias] of the fused LayerNorm epilogue
constexpr unsigned WR_OOB = 0x80000000u;
constexpr long long WR_MAX_BYTES = 0x7fffffffLL;

// Raw buffer descriptor over [p, p + bytes) held in SGPRs.  The readfirstlane pins it as wave-uniform: without it a
// descriptor that depends on blockIdx arithmetic is treated as divergent and every access gets a waterfall loop.
__device__ __forceinline__ auto wave_uniform_rsrc(const void* p, int bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Flat pixel index (image, y, x) of mosaic pixel (Y, X), or -1 on a separating line, outside the mosaic or past the last image
__device__ __forceinline__ int mosaic_pix(const Geo& g, int n, int h, int w, int Y, int X) {
    if ((unsigned)Y >= (unsigned)g.mh || (unsigned)X >= (unsigned)g.mw) return -1;
    const int iy = (int)__umulhi((unsigned)Y, g.mdh), ix = (int)__umulhi((unsigned)X, g.mdw);
    const int y = Y - iy * g.mph, x = X - ix * g.mpw;
    const int img = iy * g.mix + ix;
    return (y < h && x < w && img < n) ? (img * h + y) * w + x : -1;
}

// Sum of v over the four 16-lane groups of a wave, returned in every lane.  v_permlane16_swap(a, b) leaves rows
// (a0,b0,a2,b2) / (a1,b1,a3,b3), v_permlane32_swap the lower / upper halves side by side, so with a = b = v two swaps
// and two adds do the butterfly.  Written as inline asm with two distinct registers: extracting BOTH results of the
// __builtin_amdgcn_permlane*_swap builtins as scalars is miscompiled by this hipcc (the second result aliases the first).
__device__ __forceinline__ float sum_lane_groups(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float s = a + b, c = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "+v"(c));
    return s + c;
}

// XCD-aware work order of the wave-specialised forward kernels.  Workgroup b runs on XCD b % 8 (round-robin dispatch
// over the 8 XCDs, each with its own L2).  In every round the 32 workgroups of one XCD take 32 / nblk CONSECUTIVE
// tiles (for a 256-wide map and nblk = 1: two full tile rows) and, when cout has several 64-channel blocks, all the
// blocks of a tile: the halo rows / columns that neighbouring tiles share and the input a tile's output blocks share
// are then fetched into ONE L2 instead of up to eight.  tile(k) = slot0 + k * stride for the k-th item of a workgroup.
struct WsOrder {
    int nb, slot0, stride, nloc;
};
__device__ __forceinline__ WsOrder ws_order(int ntiles, int nblk) {
    const int b = blockIdx.x, xcd = b & 7, r = b >> 3;
    const int tpx = (int)(gridDim.x >> 3) / nblk;   // tiles per XCD and round: one workgroup per CU, nblk divides CUs / 8 (launcher)
    WsOrder o;
    o.nb = r % nblk;
    o.slot0 = xcd * tpx + r / nblk;
    o.stride = 8 * tpx;
    o.nloc = o.slot0 < ntiles ? (ntiles - o.slot0 + o.stride - 1) / o.stride : 0;
    return o;
}

// Packs one 256-pixel x 64-channel accumulator tile of an MFMA wave set for its stores (and, EPI 2, applies the fused
// LayerNorm + ReLU).  A lane owns 4 consecutive channels (8 B) of pixel (mt, lane & 15) per n-tile; v_permlane16_swap
// trades the odd 16-lane rows of one n-tile with the even rows of the next, after which every lane holds 8 consecutive
// channels: 16-byte stores, two instructions per 128-byte NHWC row.  pend[0..7]: z (or relu(z), EPI 1) pieces
// (mt, n-tile pair), pend[8..15]: activation pieces (EPI 2); pvo[mt]: byte offset of the piece, out of range past the
// image edge.  mean / rstd are stored here (lane group 0).
__device__ __forceinline__ f32x2 relu2(f32x2 v) {
    return f32x2{__builtin_amdgcn_fmed3f(v.x, 0.f, __builtin_inff()), __builtin_amdgcn_fmed3f(v.y, 0.f, __builtin_inff())};
}

template <int EPI, typename E, typename RS>
__device__ __forceinline__ void ws_pack_tile(const f32x4 (&acc)[4][4], const float* gb, float eps, int wave, int lane,
                                             int img_h, int img_w, int nn, int y0, int x0, int cy, const int (&soff)[4],
                                             RS rsm, RS rsr, u32x4 (&pend)[EPI == 2 ? 16 : 8], unsigned (&pvo)[4]) {
    constexpr int TSZ = 2;
    constexpr int NPEND = EPI == 2 ? 16 : 8;
    const int grp = lane >> 4;
    const int pixbase = (nn * img_h + y0) * img_w + x0;
    const int tbase = pixbase * cy * TSZ;
    const bool xok = (lane & 15) < img_w - x0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#ifdef AD_DROP_STORES      // diagnostic: how fast is the kernel when nothing is written?
        const bool ok = false;
#else
        const bool ok = xok && wave * 4 + mt < img_h - y0;
#endif
        pvo[mt] = ok ? (unsigned)(tbase + soff[mt]) : WR_OOB;
        // EPI 2 arithmetic is written on float pairs: v_pk_add / v_pk_mul / v_pk_fma_f32 do two channels per instruction,
        // and this VALU work sits between the MFMA phases of two items (it is not hidden behind anything)
        float mean = 0.f, rstd = 0.f;
        if (EPI == 2 || EPI == 5 || EPI == 6) {
            // Two passes, as LayerNormalization itself (mean, then the mean of squared DEVIATIONS) and ln_fwd_kernel: until r03 this
            // was E[x^2] - mean^2 in one pass, whose cancellation costs (mean / std)^2 * 2^-24 of the variance -- 10 % at
            // mean / std = 1 000 (test_fused_layernorm_epilogue_with_a_large_mean_offset; found through ADVICE r03).  Price: eight
            // packed subtracts and a second, dependent lane-group sum per m-tile.
            f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                s1 += acc[mt][nt].xy; s1 += acc[mt][nt].zw;
            }
            mean = sum_lane_groups(s1.x + s1.y) * (1.f / 64.f);
            const f32x2 nm2 = {-mean, -mean};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x2 lo = acc[mt][nt].xy + nm2, hi = acc[mt][nt].zw + nm2;
                s2 = __builtin_elementwise_fma(lo, lo, s2);
                s2 = __builtin_elementwise_fma(hi, hi, s2);
            }
            rstd = rsqrtf(sum_lane_groups(s2.x + s2.y) * (1.f / 64.f) + eps);
            if (EPI != 6) {       // (6: inference, nobody reads the statistics)
                const unsigned so = ok && grp == 0 ? (unsigned)((pixbase + (wave * 4 + mt) * img_w + (lane & 15)) * 4) : WR_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mean), rsm, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd), rsr, so, 0, 0);
            }
        }
        // normalisation as two fused multiply-adds per pair, xhat = v * rstd + (-mean * rstd), y = xhat * gamma + beta
        // (r03; subtract / multiply / fma before: one packed instruction per pair less), ReLU on the PACKED 16-bit result
        // (v_pk_max_i16 against 0: a set sign bit is a negative int16 for bf16 and half alike) instead of one v_med3 per
        // element: 64 of ~436 VALU instructions per item
        const f32x2 rstd2 = {rstd, rstd}, nmr2 = {-mean * rstd, -mean * rstd};
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef E e16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int np = 0; np < 2; ++np) {
            union { typename Half16<E>::v4 h; u32x2 u; } pa, pb, qa, qb;
            f32x4 ga = {}, gb4 = {}, ba = {}, bb = {};
            if (EPI == 2 || EPI == 6) {
                ga = *reinterpret_cast<const f32x4*>(gb + (2 * np) * 16 + grp * 4);
                gb4 = *reinterpret_cast<const f32x4*>(gb + (2 * np + 1) * 16 + grp * 4);
                ba = *reinterpret_cast<const f32x4*>(gb + 64 + (2 * np) * 16 + grp * 4);
                bb = *reinterpret_cast<const f32x4*>(gb + 64 + (2 * np + 1) * 16 + grp * 4);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 va = h ? acc[mt][2 * np].zw : acc[mt][2 * np].xy;
                f32x2 vb = h ? acc[mt][2 * np + 1].zw : acc[mt][2 * np + 1].xy;
                if (EPI == 1) { va = relu2(va); vb = relu2(vb); }
                if (EPI != 6) {
                    const e16x2 za = __builtin_convertvector(va, e16x2), zb = __builtin_convertvector(vb, e16x2);
                    pa.u[h] = __builtin_bit_cast(unsigned, za);
                    pb.u[h] = __builtin_bit_cast(unsigned, zb);
                }
                if (EPI == 2 || EPI == 6) {
                    f32x2 ya = __builtin_elementwise_fma(__builtin_elementwise_fma(va, rstd2, nmr2), h ? ga.zw : ga.xy, h ? ba.zw : ba.xy);
                    f32x2 yb = __builtin_elementwise_fma(__builtin_elementwise_fma(vb, rstd2, nmr2), h ? gb4.zw : gb4.xy, h ? bb.zw : bb.xy);
                    // (built as values, not through the union: writing .h elements and reading .u back in the same iteration
                    // was folded to the h = 0 pair by hipcc)
                    // one v_cvt_pk per pair (r04: as element-wise casts the pairs were converted singly and joined by v_perm_b32,
                    // 96 instructions per item where 32 do)
                    asm volatile("" : "+v"(ya), "+v"(yb));
                    const e16x2 ta = __builtin_convertvector(ya, e16x2), tb = __builtin_convertvector(yb, e16x2);
                    qa.u[h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ta), s16x2{0, 0}));
                    qb.u[h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, tb), s16x2{0, 0}));
                }
            }
            if (EPI != 6) {
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pa.u[0], pb.u[0], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pa.u[1], pb.u[1], false, false);
                pend[mt * 2 + np] = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
            if (EPI == 2 || EPI == 6) {
                const u32x2 r0 = __builtin_amdgcn_permlane16_swap(qa.u[0], qb.u[0], false, false);
                const u32x2 r1 = __builtin_amdgcn_permlane16_swap(qa.u[1], qb.u[1], false, false);
                pend[(NPEND - 8) + mt * 2 + np] = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
        }
    }
}

// ---- the MFMA-wave role shared by the two wave-specialised forward kernels ------------------------------------
// EPI: 0 = bias, 1 = bias + ReLU, 2 = bias + LayerNorm(eps) + ReLU fused (Cout == 64: a pixel's 64 channels sit in
// 16 registers x the 4 lane groups of one wave, so the statistics are 2 lane-swap steps).  EPI 2 writes the conv
// output z, the activation a = relu(gamma * (z - mean) * rstd + beta) and mean / rstd per pixel; bias, gamma and
// beta are read from LDS (gb: [gamma 64][beta 64][bias 64] floats) to keep the wave under 256 registers.
// Results leave the registers during the MFMA phases of the NEXT item: packed to bf16 they wait in pend[] and one
// 16-byte store is pinned between two tap steps, so the store queue never backs up into the MFMA issue.
// Measured and withdrawn (r02): deferring the EPI 2 ARITHMETIC as well (raw fp32 results kept in a second register
// set, statistics / normalisation / packs done in twelve pieces between the tap steps of the next item, 233 VGPRs):
// 397 -> 394 us per full-resolution launch.  The VALU work is not hidden by the MFMAs of the same SIMD (an MFMA holds
// the vector issue port for half of its cycles; 590 VALU instructions per item need the other half entirely), so the
// launch costs MFMA + arithmetic + what the two store streams add either way: 304 us with the stores dropped
// (-DAD_DROP_STORES), 234 us for the plain epilogue.
template <typename P, int EPI, bool MOS = false>
__device__ __forceinline__ void ws_mma_role(const ConvArgs& a, const char* xb0, const char* xb1, const char* wt0,
                                            const char* wt1, const float* gb, int wave, int lane, const WsOrder& o,
                                            int nch) {
    static_assert(!MOS || EPI <= 1, "the image mosaic exists for the bias / bias + ReLU epilogues");
    const int nb = o.nb, nloc = o.nloc;
    constexpr int TSZ = 2;
    constexpr int HWB = 18 * PIXB;
    const Geo& g = a.g;
    const int nblk = a.cout / BN;
    const int npix = a.n * a.h * a.w;
    const int grp = lane >> 4;
    // 5: z and the LayerNorm statistics, no activation (ad_conv3x3_ln_relu_fwd, act == NULL); 6: the activation alone, neither z nor
    // statistics (z == NULL: inference -- one output stream instead of two); its pieces take z's place in pend[0..7] and y1 is `act`
    constexpr bool LN = EPI == 2 || EPI == 5 || EPI == 6;
    float4 bv[4];
    if (!LN) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bv[nt] = a.bias ? *reinterpret_cast<const float4*>(a.bias + nb * BN + nt * 16 + grp * 4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
            // consumed here, so that the wait for it is placed before the loop and not (as vmcnt(0)) inside it
            asm volatile("" : "+v"(bv[nt].x), "+v"(bv[nt].y), "+v"(bv[nt].z), "+v"(bv[nt].w));
        }
    }
    char* yp; int cy, coff;
    if (nb * BN < a.cy1) { yp = a.y1; cy = a.cy1; coff = nb * BN; }
    else { yp = a.y2; cy = a.cout_real - a.cy1; coff = nb * BN - a.cy1; }
    const auto rsy = wave_uniform_rsrc(yp, npix * cy * TSZ);
    const auto rsa = wave_uniform_rsrc(EPI == 2 ? a.a_out : yp, npix * cy * TSZ);
    const auto rsm = wave_uniform_rsrc(LN && EPI != 6 ? (const void*)a.ln_mean : (const void*)yp, npix * 4);
    const auto rsr = wave_uniform_rsrc(LN && EPI != 6 ? (const void*)a.ln_rstd : (const void*)yp, npix * 4);
    const int abase0 = ((wave * 4 + 1) * 18 + (lane & 15) + 1) * PIXB + P::a_lane_off(lane);
    int soff[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)   // after the row swap, lane group g holds channels (g&1)*16 + (g>>1)*8 .. +7 of an n-tile pair
        soff[mt] = (((wave * 4 + mt) * a.w + (lane & 15)) * cy + coff + (grp & 1) * 16 + (grp >> 1) * 8) * TSZ;
    constexpr int NPEND = EPI == 2 ? 16 : 8;
    u32x4 pend[NPEND];
    unsigned pvo[4] = {WR_OOB, WR_OOB, WR_OOB, WR_OOB};   // before the first item: out of range, stores dropped
#pragma unroll
    for (int i = 0; i < NPEND; ++i) pend[i] = u32x4{0u, 0u, 0u, 0u};
    // EPI 3: the eight 16-byte pieces of the producer's ReLU output that line up with pend[0..7] (fetched right after
    // the tile is packed, used a whole chunk later), and this lane's column sums of what it stored: after the row swap
    // a lane holds channels (grp&1)*16 + (grp>>1)*8 .. +7 of n-tile pair np, i.e. 2 x 8 distinct channels.
    const bool masked = EPI == 3 && nb * BN < a.cy1 && a.mask1 != nullptr;        // wave uniform
    const auto rsk = wave_uniform_rsrc(masked ? (const void*)a.mask1 : (const void*)yp, masked ? npix * cy * TSZ : 0);
    u32x4 mkv[EPI == 3 ? 8 : 1];
    float dbs[EPI == 3 ? 16 : 1];
    if (EPI == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) mkv[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i) dbs[i] = 0.f;
    }
    // stores pend[IDX] after zeroing the elements whose mask element is not positive (|x| bits != 0: the mask is a ReLU
    // output, never negative) and adds what is stored to the column sums
#define WS_MASK_STORE(IDX)                                                                                    \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        u32x4 v_ = pend[IDX];                                                                                 \
        if (masked) {                                                                                         \
            const u32x4 m_ = mkv[IDX];                                                                        \
            _Pragma("unroll") for (int d_ = 0; d_ < 4; ++d_) {                                                \
                const unsigned keep_ = ((m_[d_] & 0x7fffu) ? 0xffffu : 0u) | ((m_[d_] & 0x7fff0000u) ? 0xffff0000u : 0u); \
                v_[d_] &= keep_;                                                                              \
            }                                                                                                 \
            typedef typename Half16<typename P::T>::v8 hv8_;                                                  \
            const hv8_ h_ = __builtin_bit_cast(hv8_, v_);                                                     \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dbs[((IDX) & 1) * 8 + j_] += (float)h_[j_];      \
        }                                                                                                     \
        __builtin_amdgcn_raw_buffer_store_b128(v_, rsy, pvo[(IDX) >> 1], ((IDX) & 1) * 32 * TSZ, AD_STORE_AUX);          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#define WS_PEND_STORE(IDX, RS)                                                                                \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);   /* pin the store between two tap steps (hipcc would bunch them) */ \
        __builtin_amdgcn_raw_buffer_store_b128(pend[IDX], RS, pvo[((IDX) & 7) >> 1], ((IDX) & 1) * 32 * TSZ, AD_STORE_AUX); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
    auto hook0 = [&](int st) {
        if (EPI == 2) { if (st < 8) WS_PEND_STORE(st, rsy); }
        else if (EPI == 3) {}                             // the mask pieces are still in flight: all stores in chunk 1
        else if ((st & 1) && st < 8) WS_PEND_STORE(st >> 1, rsy);
    };
    auto hook1 = [&](int st) {
        if (EPI == 2) { if (st < 8) WS_PEND_STORE(8 + st, rsa); }
        else if (EPI == 3) { if (st < 8) WS_MASK_STORE(st); }
        else if ((st & 1) && st < 8) WS_PEND_STORE(4 + (st >> 1), rsy);
    };
#ifdef AD_STAMP
    unsigned long long wst[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long wt_last = clock64();
    const unsigned long long wt_begin = wt_last, wall_begin = wall_clock64();
#define WSTAMP(slot) do { unsigned long long now_ = clock64(); wst[slot] += now_ - wt_last; wt_last = now_; } while (0)
#else
#define WSTAMP(slot) do {} while (0)
#endif
    // tile coordinates advance by the (constant) stride without a division per item: (tx, ty, nn) += (sx, sy, sn) with carries
    int tx, ty, nn, sx, sy, sn;
    {
        const int r0 = o.slot0 / g.tiles_x, rs = o.stride / g.tiles_x;
        tx = o.slot0 - r0 * g.tiles_x; nn = r0 / g.tiles_y; ty = r0 - nn * g.tiles_y;
        sx = o.stride - rs * g.tiles_x; sn = rs / g.tiles_y; sy = rs - sn * g.tiles_y;
    }
    for (int k = 0; k < nloc; ++k) {
        const int x0 = tx << 4, y0 = ty << 4;
        const int nn_k = nn;
        tx += sx;
        if (tx >= g.tiles_x) { tx -= g.tiles_x; ++ty; }
        ty += sy;
        if (ty >= g.tiles_y) { ty -= g.tiles_y; ++nn; }
        nn += sn;
        f32x4 acc[4][4];
        f32x4 c0[4];                                    // bias: the accumulators start from it in the first tap step
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b4 = LN ? *reinterpret_cast<const float4*>(gb + 128 + j * 16 + grp * 4) : bv[j];
            c0[j] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
        WSTAMP(0);                                      // item set-up
        lds_barrier();                                  // even stage (chunk 0) ready
        WSTAMP(1);                                      // barrier waits
        P::template mma_chunk_rows<true>(acc, xb0, abase0, HWB, wt0, lane, hook0, c0);
        WSTAMP(2);                                      // MFMA phases (with the deferred stores of the previous item)
        lds_barrier();                                  // odd stage (chunk 1) ready
        WSTAMP(1);
        P::mma_chunk_rows(acc, xb1, abase0, HWB, wt1, lane, hook1);
        WSTAMP(2);
        for (int cp = 2; cp < nch; cp += 2) {
            lds_barrier();
            WSTAMP(1);
            P::mma_chunk_rows(acc, xb0, abase0, HWB, wt0, lane, [](int) {});
            WSTAMP(2);
            lds_barrier();
            WSTAMP(1);
            P::mma_chunk_rows(acc, xb1, abase0, HWB, wt1, lane, [](int) {});
            WSTAMP(2);
        }
        ws_pack_tile<(EPI == 3 ? 0 : EPI), typename P::T>(acc, gb, a.ln_eps, wave, lane, a.h, a.w, nn_k, y0, x0, cy, soff, rsm, rsr,
                                                          pend, pvo);
        if constexpr (MOS) {    // (y0, x0) are mosaic coordinates: the pieces go to the image pixel under them, or nowhere
            const int csoff = (coff + (grp & 1) * 16 + (grp >> 1) * 8) * TSZ;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int px = mosaic_pix(g, a.n, a.h, a.w, y0 + wave * 4 + mt, x0 + (lane & 15));
                pvo[mt] = px >= 0 ? (unsigned)(px * cy * TSZ + csoff) : WR_OOB;
            }
        }
        WSTAMP(3);                                      // pack (+ fused LayerNorm arithmetic)
        if (EPI == 3) {        // unconditional loads (zero records when this block is not masked): exact vmcnt bookkeeping
#pragma unroll
            for (int i = 0; i < 8; ++i)
                mkv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsk, pvo[i >> 1], (i & 1) * 32 * TSZ, 0);
        }
    }
    if (EPI == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) WS_MASK_STORE(i);
        // column sums: fold the 16 pixel lanes of each lane group; one row of 64 per MFMA wave goes to
        // dbias_part[workgroup][wave][64] (zeros from workgroups whose block is not masked) and
        // mask_dbias_reduce_kernel adds the rows of a block in a fixed order
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = dbs[j];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
            dbs[j] = v;
        }
        if ((lane & 15) == 0 && a.dbias_part) {
            float* row = a.dbias_part + ((size_t)blockIdx.x * 4 + wave) * 64;
#pragma unroll
            for (int j = 0; j < 16; ++j)                                         // channel of slot j: see dbs above
                row[(j >> 3) * 32 + (grp & 1) * 16 + (grp >> 1) * 8 + (j & 7)] = dbs[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) WS_PEND_STORE(i, rsy);
        if (EPI == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) WS_PEND_STORE(8 + i, rsa);
        }
    }
#undef WS_PEND_STORE
#undef WS_MASK_STORE
#ifdef AD_STAMP
    WSTAMP(4);                                          // drain of the last item
    if (wave == 0 && lane == 0 && a.dbg) {
        for (int i = 0; i < 5; ++i) a.dbg[blockIdx.x * 9 + i] = wst[i];
        a.dbg[blockIdx.x * 9 + 5] = nloc;
        a.dbg[blockIdx.x * 9 + 7] = wall_clock64() - wall_begin;      // 100 MHz
        a.dbg[blockIdx.x * 9 + 8] = clock64() - wt_begin;
    }
#endif
#undef WSTAMP
}

// ---- the MFMA-wave role of a dgrad whose output is the gradient of a Conv2D -> LayerNorm -> ReLU layer (EPI 4) ------------
// conv_block's second conv (and the layer after the decoder) takes the previous LayerNorm's activation as input, so the
// dgrad's 64 output channels are d(activation) of that layer and the very next thing is its LayerNorm / ReLU backward
// (norm.hip ln_bwd_kernel: read d and z, write dz: 1.6 GB per full-resolution layer).  Here the wave that holds a pixel's
// 64 gradient values in its fp32 accumulators does that arithmetic itself: z (8-byte pieces in accumulator layout),
// mean and rstd of the item's pixels are fetched during the MFMA phases, the masks are re-derived from z as the
// stand-alone kernel does, the two per-pixel sums are lane-swap butterflies, dz leaves as 16-byte pieces and the three
// per-channel sums (dgamma, dbeta and the conv bias gradient = column sums of dz AS STORED) stay in 48 registers per
// lane until the end of the launch.  The gradient of the activation never goes to memory.
template <typename P>
__device__ __forceinline__ void ws_mma_role_lnb(const ConvArgs& a, const char* xb0, const char* xb1, const char* wt0,
                                                const char* wt1, const float* gb, int wave, int lane, const WsOrder& o) {
    typedef typename P::T E;
    typedef typename Half16<E>::v4 h4;
    constexpr int TSZ = 2;
    constexpr int HWB = 18 * PIXB;
    constexpr int CY = BN;
    const Geo& g = a.g;
    const int nloc = o.nloc;
    const int npix = a.n * a.h * a.w;
    const int grp = lane >> 4;
    const auto rsy = wave_uniform_rsrc(a.y1, npix * CY * TSZ);
    const auto rsz = wave_uniform_rsrc(a.lnb_z, npix * CY * TSZ);
    const auto rsm = wave_uniform_rsrc(a.ln_mean, npix * 4);
    const auto rsr = wave_uniform_rsrc(a.ln_rstd, npix * 4);
    const int abase0 = ((wave * 4 + 1) * 18 + (lane & 15) + 1) * PIXB + P::a_lane_off(lane);
    // byte offsets inside y1 / lnb_z of this lane's pixel of m-tile mt: px(mt) = (4 wave + mt) * w + (lane & 15);
    //   store layout (after the row swap, ws_pack_tile):  px * 128 + scst,  scst = ((grp & 1) * 16 + (grp >> 1) * 8) * 2
    //   accumulator layout (z pieces of n-tile nt):       px * 128 + grp * 8 (+ nt * 32)
    // kept as one register each (px0s, zdiff) and re-derived per item: this role is register bound
    const int scst = ((grp & 1) * 16 + (grp >> 1) * 8) * TSZ;
    const int px0s = ((wave * 4) * a.w + (lane & 15)) * CY * TSZ + scst;
    const int zdiff = grp * 4 * TSZ - scst;
    // per-channel sums as float PAIRS (channels q, q + 1 of an n-tile): the epilogue below is written on pairs so that its
    // multiply-adds and adds are v_pk_* instructions (r04: 144 scalar v_add_f32 per item were the cb / cz updates)
    f32x2 cg2[8], cb2[8], cz2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cg2[j] = cb2[j] = cz2[j] = f32x2{0.f, 0.f};
    u32x2 zq[16];
    float mu[4], rs[4];
    int pixbase = 0, ylim = 0;                  // of the current item (wave uniform)
    bool xok = false;
    // store offset of this lane's pixel of m-tile mt, out of range past the image edge; re-derived where it is needed
    // instead of being kept through the MFMA phases
    auto pvo_of = [&](int mt) -> unsigned {
        return xok && wave * 4 + mt < ylim ? (unsigned)((pixbase + mt * a.w) * CY * TSZ + px0s) : WR_OOB;
    };
    // mean / rstd / z of m-tiles mt0, mt0 + 1 (out of range where the pixel is: zeros, rstd = 0 -> dz = 0)
    auto fetch = [&](int mt0) {
#pragma unroll
        for (int mt = mt0; mt < mt0 + 2; ++mt) {
            const unsigned pv = pvo_of(mt);
            const unsigned so = pv != WR_OOB ? (pv - scst) >> 5 : WR_OOB;            // (pixel index) * 4
            mu[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsm, so, 0, 0));
            rs[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, so, 0, 0));
        }
#pragma unroll
        for (int mt = mt0; mt < mt0 + 2; ++mt) {
            const unsigned pv = pvo_of(mt);
            const unsigned zo = pv != WR_OOB ? pv + zdiff : WR_OOB;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) zq[mt * 4 + nt] = __builtin_amdgcn_raw_buffer_load_b64(rsz, zo, nt * 16 * TSZ, 0);
        }
    };
    // the first two m-tiles' operands are fetched during the MFMA phases, the other two at the start of the epilogue,
    // behind the arithmetic of the first two: everything in flight through the MFMA phases would need ~20 more live
    // registers than there are
    auto hook0 = [&](int st) {
        if (st != 0) return;
        __builtin_amdgcn_sched_barrier(0);
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int tx, ty, nn, sx, sy, sn;
    {
        const int r0 = o.slot0 / g.tiles_x, rs0 = o.stride / g.tiles_x;
        tx = o.slot0 - r0 * g.tiles_x; nn = r0 / g.tiles_y; ty = r0 - nn * g.tiles_y;
        sx = o.stride - rs0 * g.tiles_x; sn = rs0 / g.tiles_y; sy = rs0 - sn * g.tiles_y;
    }
#ifdef AD_STAMP     // diagnostic build (tools/stamps_lnb.py): cycles of MFMA wave 0 by segment; slot 8 = total, 7 = wall (10 ns)
    unsigned long long lst[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long lt_last = clock64();
    const unsigned long long lt_begin = lt_last, lwall_begin = wall_clock64();
#define LSTAMP(slot) do { unsigned long long now_ = clock64(); lst[slot] += now_ - lt_last; lt_last = now_; } while (0)
#else
#define LSTAMP(slot) do {} while (0)
#endif
    for (int k = 0; k < nloc; ++k) {
        const int x0 = tx << 4, y0 = ty << 4;
        pixbase = (nn * a.h + y0) * a.w + x0;
        ylim = a.h - y0;
        xok = (lane & 15) < a.w - x0;
        tx += sx;
        if (tx >= g.tiles_x) { tx -= g.tiles_x; ++ty; }
        ty += sy;
        if (ty >= g.tiles_y) { ty -= g.tiles_y; ++nn; }
        nn += sn;
        f32x4 acc[4][4];
        f32x4 c0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        LSTAMP(0);                                      // item set-up
        lds_barrier();                                  // even stage (chunk 0) ready
        LSTAMP(1);                                      // barrier waits
        P::template mma_chunk_rows<true>(acc, xb0, abase0, HWB, wt0, lane, hook0, c0);
        LSTAMP(2);                                      // MFMA phases
        lds_barrier();                                  // odd stage (chunk 1) ready
        LSTAMP(1);
#ifndef AD_LNB_FETCH_ST
#define AD_LNB_FETCH_ST 99
#endif
        auto hook1 = [&](int st) {
            if (st != AD_LNB_FETCH_ST) return;
            __builtin_amdgcn_sched_barrier(0);
            fetch(2);
            __builtin_amdgcn_sched_barrier(0);
        };
        P::mma_chunk_rows(acc, xb1, abase0, HWB, wt1, lane, hook1);
        LSTAMP(2);
        // ---- LayerNorm + ReLU backward of the 64 values per pixel this wave holds
        // bf16: the second half's operands are fetched here, behind the arithmetic of the first two m-tiles (471 us per
        // full-resolution launch; fetched after the first m-tile: 488 us).  fp16 needs more conversion temporaries and
        // spills with that many registers live: it fetches after the first m-tile (234 registers, no spill).
        // r03: with the normalisation / dz arithmetic folded into fused multiply-adds (below) the early fetch no longer fits
        // bf16 either (256 registers + 60 bytes of scratch: 566 us); fetched after the first m-tile, both types run without
        // spills, bf16 at 458 us against 465 us for the r02 arithmetic with the early fetch (same box, tools/time_lnb.py)
        constexpr bool FETCH_EARLY = false;
        if (FETCH_EARLY) {
            __builtin_amdgcn_sched_barrier(0);
            fetch(2);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#ifdef AD_LNB_SB
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (AD_LNB_FETCH_ST == 99 && !FETCH_EARLY && mt == 1) {
                __builtin_amdgcn_sched_barrier(0);
                fetch(2);
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned pvo_mt = pvo_of(mt);
            const bool valid = pvo_mt != WR_OOB;
            const float rstd = rs[mt], nmr = -mu[mt] * rstd;     // xhat = z * rstd + (-mean * rstd): one fma per element (r03)
            const f32x2 rstd2 = {rstd, rstd}, nmr2 = {nmr, nmr};
            f32x2 xh2[8], gg2[8];
            f32x2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gb + nt * 16 + grp * 4);
                const f32x4 be = *reinterpret_cast<const f32x4*>(gb + 64 + nt * 16 + grp * 4);
                union { h4 h; u32x2 u; } zz;
                zz.u = zq[mt * 4 + nt];
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {          // channel pair (2 hp, 2 hp + 1) of this lane's four
                    const f32x2 ga2 = hp ? ga.zw : ga.xy, be2 = hp ? be.zw : be.xy;
                    const f32x2 zf = {(float)zz.h[2 * hp], (float)zz.h[2 * hp + 1]};
                    const f32x2 h = __builtin_elementwise_fma(zf, rstd2, nmr2);
                    const f32x2 yv = __builtin_elementwise_fma(h, ga2, be2);
                    const f32x2 av = hp ? acc[mt][nt].zw : acc[mt][nt].xy;
                    const f32x2 dl = {(valid && yv.x > 0.f) ? av.x : 0.f, (valid && yv.y > 0.f) ? av.y : 0.f};
                    const int j = nt * 2 + hp;
                    cg2[j] = __builtin_elementwise_fma(dl, h, cg2[j]);
                    cb2[j] += dl;
                    const f32x2 gv = dl * ga2;
                    s1p += gv;
                    s2p = __builtin_elementwise_fma(gv, h, s2p);
                    xh2[j] = h;
                    gg2[j] = gv;
                }
            }
            // dz = rstd (g - mean(g) - xhat mean(g xhat)) as two fused multiply-adds per element on per-pixel products (r03)
            const float s1 = sum_lane_groups(s1p.x + s1p.y) * (-rstd / 64.f);
            const float s2 = sum_lane_groups(s2p.x + s2p.y) * (-rstd / 64.f);
            const f32x2 s1v = {s1, s1}, s2v = {s2, s2};
            typedef E e16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                union { u32x2 u; } pa, pb;
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    const int ja = (2 * np) * 2 + hp, jb = (2 * np + 1) * 2 + hp;
                    const f32x2 da = __builtin_elementwise_fma(xh2[ja], s2v, __builtin_elementwise_fma(gg2[ja], rstd2, s1v));
                    const f32x2 db = __builtin_elementwise_fma(xh2[jb], s2v, __builtin_elementwise_fma(gg2[jb], rstd2, s1v));
                    const e16x2 ta = {(E)da.x, (E)da.y}, tb = {(E)db.x, (E)db.y};
                    pa.u[hp] = __builtin_bit_cast(unsigned, ta);
                    pb.u[hp] = __builtin_bit_cast(unsigned, tb);
                    // the conv bias gradient sums dz as stored (what wgrad sees); bf16: the two stored values back as floats are a
                    // shift and a mask of the packed word (hipcc converts each element a second time when asked for (float)ta.x)
                    if constexpr (sizeof(E) == 2 && !ad_same_type<E, f16_t>::value) {
                        asm volatile("" : "+v"(pa.u[hp]), "+v"(pb.u[hp]));      // opaque: else the low half is converted once more
                        cz2[ja] += f32x2{__builtin_bit_cast(float, pa.u[hp] << 16), __builtin_bit_cast(float, pa.u[hp] & 0xffff0000u)};
                        cz2[jb] += f32x2{__builtin_bit_cast(float, pb.u[hp] << 16), __builtin_bit_cast(float, pb.u[hp] & 0xffff0000u)};
                    } else {
                        cz2[ja] += f32x2{(float)ta.x, (float)ta.y};
                        cz2[jb] += f32x2{(float)tb.x, (float)tb.y};
                    }
                }
                const u32x2 w0 = __builtin_amdgcn_permlane16_swap(pa.u[0], pb.u[0], false, false);
                const u32x2 w1 = __builtin_amdgcn_permlane16_swap(pa.u[1], pb.u[1], false, false);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rsy, pvo_mt, np * 32 * TSZ, AD_STORE_AUX);
            }
            LSTAMP(3 + mt);                             // epilogue of m-tile mt (mt == 1 includes issuing the second fetch)
        }
    }
#ifdef AD_STAMP
    if (wave == 0 && lane == 0 && a.dbg) {
        for (int i = 0; i < 7; ++i) a.dbg[blockIdx.x * 10 + i] = lst[i];
        a.dbg[blockIdx.x * 10 + 7] = wall_clock64() - lwall_begin;
        a.dbg[blockIdx.x * 10 + 8] = clock64() - lt_begin;
        a.dbg[blockIdx.x * 10 + 9] = nloc;
    }
#endif
#undef LSTAMP
    // column sums: fold the 16 pixel lanes of each lane group; one row of 3 x 64 per MFMA wave goes to
    // dbias_part[workgroup][wave][dgamma | dbeta | dbias][64], lnb_reduce_kernel adds the rows in a fixed order
    float* row = a.dbias_part + ((size_t)blockIdx.x * 4 + wave) * (3 * BN);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v0 = cg2[j >> 1][j & 1], v1 = cb2[j >> 1][j & 1], v2 = cz2[j >> 1][j & 1];
#pragma unroll
        for (int of = 1; of < 16; of <<= 1) {
            v0 += __shfl_xor(v0, of, 64);
            v1 += __shfl_xor(v1, of, 64);
            v2 += __shfl_xor(v2, of, 64);
        }
        if ((lane & 15) == 0) {
            const int ch = (j >> 2) * 16 + grp * 4 + (j & 3);
            row[ch] = v0; row[BN + ch] = v1; row[2 * BN + ch] = v2;
        }
    }
}

template <typename P, int EPI>
__global__ __launch_bounds__(WR_T, 1) void conv3x3_fwd_wres_kernel(ConvArgs a) {
    typedef typename P::T T;
    static_assert(sizeof(T) == 2, "the weights-resident kernel is the bf16 throughput path");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;      // geometry is (1, 16, 16): HW = HH = 18, NPH = 324
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* xb0 = smem;
    char* xb1 = xb0 + WR_XB;
    char* wt = xb1 + WR_XB;
    constexpr int TSZ = 2;

    const int nblk = a.cout / BN;
    const WsOrder o = ws_order(a.ntiles, nblk);
    const int nb = o.nb;                            // one 64-channel output block per workgroup for the whole launch
    const int kc_total = (a.c1 + a.c2) / P::KV;
    const int npix = a.n * a.h * a.w;
    AD_CLOCK_BEGIN

    // weights of this output block: both chunks, once (all 512 threads)
#pragma unroll
    for (int i = 0; i < 2 * 9 * 4 * BN / WR_T; ++i) {
        const int s = tid + i * WR_T;               // 0 .. 4607: chunk s / 2304, slot s % 2304
        const int ch = s >= 9 * 4 * BN;
        *reinterpret_cast<uint4*>(wt + s * 16) = load_w_slot(a.wp, kc_total, ch, a.wcout ? a.wcout : a.cout, nb, s - ch * 9 * 4 * BN);
    }

    float* gb = reinterpret_cast<float*>(wt + 2 * WT_BYTES);
    if (EPI == 2 || EPI == 4 || EPI == 5 || EPI == 6) {       // [gamma][beta][bias] of the 64 output channels; visible to the MFMA waves after this barrier
        if (tid < 64) {
            gb[tid] = a.ln_gamma[tid]; gb[64 + tid] = a.ln_beta[tid]; gb[128 + tid] = a.bias ? a.bias[tid] : 0.f;
        }
        lds_barrier();
    }
    if (wave >= 4) {
        // ------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        const bool one_src = P::CK < a.c1;          // virtual concat: chunk 1 may come from x2
        const int rb0 = a.c1 * TSZ;
        const int rb1 = (one_src ? a.c1 : a.c2) * TSZ;
        const int ob1 = one_src ? P::CK * TSZ : 0;
        const auto rs0 = wave_uniform_rsrc(a.x1, npix * rb0);
        const auto rs1 = wave_uniform_rsrc(one_src ? a.x1 : a.x2, npix * rb1);
        // this thread's six halo slots: pixel hp = lt/4 + 64 i (row hy, column hx of the 18 x 18 halo), part lt&3
        const int part16 = (lt & 3) * 16;
        int hyx[WR_XS];
#pragma unroll
        for (int i = 0; i < WR_XS; ++i) {
            const int hp = (lt >> 2) + 64 * i;
            const int hy = hp / 18;
            hyx[i] = hp < 324 ? (hy << 8) | (hp - hy * 18) : (64 << 8);   // row 64: never inside an image
        }
        const int lds_slot = (lt >> 2) * PIXB + part16;                  // + 64 i * PIXB
        int pix0, pix1, pix2, pix3, pix4, pix5;     // flat pixel index of each slot for the tile being fetched, or -1
#define WR_PIX(I, NN, Y0, X0)                                                                    \
    {                                                                                            \
        const int y_ = (Y0) - 1 + (hyx[I] >> 8), x_ = (X0) - 1 + (hyx[I] & 255);                 \
        const bool ok_ = (unsigned)y_ < (unsigned)a.h && (unsigned)x_ < (unsigned)a.w;           \
        pix##I = ok_ ? ((NN) * a.h + y_) * a.w + x_ : -1;                                        \
    }
#define WR_PIXELS(K)                                                                             \
    {                                                                                            \
        const int tile_ = o.slot0 + min((K), o.nloc - 1) * o.stride;   /* past the end: re-fetch the last tile, loads stay unconditional */ \
        const int r_ = tile_ / g.tiles_x;                                                        \
        const int x0_ = (tile_ - r_ * g.tiles_x) << 4;                                           \
        const int nn_ = r_ / g.tiles_y;                                                          \
        const int y0_ = (r_ - nn_ * g.tiles_y) << 4;                                             \
        WR_PIX(0, nn_, y0_, x0_) WR_PIX(1, nn_, y0_, x0_) WR_PIX(2, nn_, y0_, x0_)               \
        WR_PIX(3, nn_, y0_, x0_) WR_PIX(4, nn_, y0_, x0_) WR_PIX(5, nn_, y0_, x0_)               \
    }
#define WR_LD(RS, RB, OB, PIXV) \
    __builtin_amdgcn_raw_buffer_load_b128((RS), (PIXV) >= 0 ? (unsigned)((PIXV) * (RB) + (OB) + part16) : WR_OOB, 0, 0)
        u32x4 xa0, xa1, xa2, xa3, xa4, xa5, xb_0, xb_1, xb_2, xb_3, xb_4, xb_5;
#define WR_ISSUE_A                                                                               \
    xa0 = WR_LD(rs0, rb0, 0, pix0); xa1 = WR_LD(rs0, rb0, 0, pix1); xa2 = WR_LD(rs0, rb0, 0, pix2); \
    xa3 = WR_LD(rs0, rb0, 0, pix3); xa4 = WR_LD(rs0, rb0, 0, pix4); xa5 = WR_LD(rs0, rb0, 0, pix5); \
    asm volatile("" ::: "memory");
#define WR_ISSUE_B                                                                               \
    xb_0 = WR_LD(rs1, rb1, ob1, pix0); xb_1 = WR_LD(rs1, rb1, ob1, pix1); xb_2 = WR_LD(rs1, rb1, ob1, pix2); \
    xb_3 = WR_LD(rs1, rb1, ob1, pix3); xb_4 = WR_LD(rs1, rb1, ob1, pix4); xb_5 = WR_LD(rs1, rb1, ob1, pix5); \
    asm volatile("" ::: "memory");
#define WR_STORE(XT, V0, V1, V2, V3, V4, V5)                                                     \
    *reinterpret_cast<u32x4*>((XT) + lds_slot) = V0;                                             \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 64 * PIXB) = V1;                                 \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 128 * PIXB) = V2;                                \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 192 * PIXB) = V3;                                \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 256 * PIXB) = V4;                                \
    if (lt < 16) *reinterpret_cast<u32x4*>((XT) + lds_slot + 320 * PIXB) = V5;

        // load order (the vmcnt arithmetic depends on it): A(0) B(0) A(1) | B(1) A(2) | B(2) A(3) | ...
        int k = 0;
        WR_PIXELS(k)
        WR_ISSUE_A
        WR_ISSUE_B
        WR_STORE(xb0, xa0, xa1, xa2, xa3, xa4, xa5)
        WR_PIXELS(k + 1)                             // pix* now describe item k + 1 until B(k + 1) has been issued
        WR_ISSUE_A
        for (; k < o.nloc; ++k) {
            lds_barrier();                           // B0: X0 = chunk 0 of this item is complete; X1 is free
            WR_STORE(xb1, xb_0, xb_1, xb_2, xb_3, xb_4, xb_5)
            WR_ISSUE_B                               // chunk 1 of the next item
            lds_barrier();                           // B1: X1 complete; X0 is free
            WR_STORE(xb0, xa0, xa1, xa2, xa3, xa4, xa5)
            WR_PIXELS(k + 2)
            WR_ISSUE_A                               // chunk 0 of the item after next
        }
#undef WR_PIX
#undef WR_PIXELS
#undef WR_LD
#undef WR_ISSUE_A
#undef WR_ISSUE_B
#undef WR_STORE
    } else {
        if constexpr (EPI == 4) ws_mma_role_lnb<P>(a, xb0, xb1, wt, wt + WT_BYTES, gb, wave, lane, o);
        else ws_mma_role<P, EPI>(a, xb0, xb1, wt, wt + WT_BYTES, gb, wave, lane, o, 2);
        AD_CLOCK_END(wave, lane, blockIdx.x)
    }
}

// ------------------------------------------------------------------ forward / dgrad, wave specialised, streamed weights
// The same organisation as conv3x3_fwd_wres_kernel for any EVEN number of 32-channel chunks (Cin = 128 ... 1024 on
// 16x16 tiles): the loader waves stream the weight chunk (36.9 KB, L2 resident) together with the halo chunk, one
// stage = one channel chunk, stages alternate between two [X][W] buffer pairs, one barrier per stage, loads issued
// two stages ahead of their LDS store.  The pending output stores of an item drain during the first two stages of
// the next one.  LDS: as above, 135.9 KB.
template <typename P, int EPI, bool MOS = false>
__global__ __launch_bounds__(WR_T, 1) void conv3x3_fwd_ws_kernel(ConvArgs a) {
    typedef typename P::T T;
    static_assert(sizeof(T) == 2, "bf16 throughput path");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;      // geometry is (1, 16, 16)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* xb0 = smem;
    char* xb1 = xb0 + WR_XB;
    char* wt0 = xb1 + WR_XB;
    char* wt1 = wt0 + WT_BYTES;
    constexpr int TSZ = 2;

    const int nblk = a.cout / BN;
    const WsOrder o = ws_order(a.ntiles, nblk);
    const int nb = o.nb;                            // one 64-channel output block per workgroup for the whole launch
    const int cin = a.c1 + a.c2;
    const int kc_total = cin / P::KV;
    const int nch = cin / P::CK;                    // even
    const int npix = a.n * a.h * a.w;
    const int nloc = o.nloc;                        // items of this workgroup (>= 1)
    float* gb = reinterpret_cast<float*>(wt1 + WT_BYTES);
    if (EPI == 2 || EPI == 4 || EPI == 6) {       // [gamma][beta][bias] of the 64 output channels; visible to the MFMA waves after this barrier
        if (tid < 64) {
            gb[tid] = a.ln_gamma[tid]; gb[64 + tid] = a.ln_beta[tid]; gb[128 + tid] = a.bias ? a.bias[tid] : 0.f;
        }
        lds_barrier();
    }

    if (wave >= 4) {
        // ------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        const auto rsx1 = wave_uniform_rsrc(a.x1, npix * a.c1 * TSZ);
        const auto rsx2 = wave_uniform_rsrc(a.c2 ? a.x2 : a.x1, npix * (a.c2 ? a.c2 : a.c1) * TSZ);
        const int wrow = a.wcout ? a.wcout : a.cout;   // pack row length (>= cout: a launch may take a slice of the blocks)
        const auto rsw = wave_uniform_rsrc(a.wp, 9 * cin * wrow * TSZ);
        const int part16 = (lt & 3) * 16;
        int hyx[WR_XS];
#pragma unroll
        for (int i = 0; i < WR_XS; ++i) {
            const int hp = (lt >> 2) + 64 * i;
            const int hy = hp / 18;
            hyx[i] = hp < 324 ? (hy << 8) | (hp - hy * 18) : (64 << 8);   // row 64 beyond the tile: never stored
        }
        const int lds_slot = (lt >> 2) * PIXB + part16;
        // weight slot i: s = lt + 256 i -> (tap s>>8 = i, kc (s>>6)&3, co s&63): byte offset inside chunk 0, block nb
        const int woff = ((((lt >> 6) & 3) * wrow) + nb * BN + (lt & 63)) * 16;       // + i * kc_total * wrow * 16
        const int wtap = kc_total * wrow * 16;
        int pix0, pix1, pix2, pix3, pix4, pix5;
#define WS_PIX(I, NN, Y0, X0)                                                                    \
    {                                                                                            \
        const int y_ = (Y0) - 1 + (hyx[I] >> 8), x_ = (X0) - 1 + (hyx[I] & 255);                 \
        const bool ok_ = (unsigned)y_ < (unsigned)a.h && (unsigned)x_ < (unsigned)a.w;           \
        if constexpr (MOS) pix##I = mosaic_pix(g, a.n, a.h, a.w, y_, x_);      /* (Y0, X0): mosaic coordinates, NN == 0 */ \
        else pix##I = ok_ ? ((NN) * a.h + y_) * a.w + x_ : -1;                                   \
    }
        // stage cursor of the next issue: local item k (clamped to the last one), chunk ch
#define WS_PIXELS(K)                                                                             \
    {                                                                                            \
        const int tile_ = o.slot0 + min((K), nloc - 1) * o.stride;                               \
        const int r_ = tile_ / g.tiles_x;                                                        \
        const int x0_ = (tile_ - r_ * g.tiles_x) << 4;                                           \
        const int nn_ = r_ / g.tiles_y;                                                          \
        const int y0_ = (r_ - nn_ * g.tiles_y) << 4;                                             \
        WS_PIX(0, nn_, y0_, x0_) WS_PIX(1, nn_, y0_, x0_) WS_PIX(2, nn_, y0_, x0_)               \
        WS_PIX(3, nn_, y0_, x0_) WS_PIX(4, nn_, y0_, x0_) WS_PIX(5, nn_, y0_, x0_)               \
    }
#define WS_XLD(RS, RB, OB, PIXV) \
    __builtin_amdgcn_raw_buffer_load_b128((RS), (PIXV) >= 0 ? (unsigned)((PIXV) * (RB) + (OB) + part16) : WR_OOB, 0, 0)
#define WS_WLD(I, CH) __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)(woff + (I) * wtap + (CH) * 4 * wrow * 16), 0, 0)
        u32x4 xa0, xa1, xa2, xa3, xa4, xa5, wa0, wa1, wa2, wa3, wa4, wa5, wa6, wa7, wa8;
        u32x4 xb_0, xb_1, xb_2, xb_3, xb_4, xb_5, wb0, wb1, wb2, wb3, wb4, wb5, wb6, wb7, wb8;
#define WS_ISSUE(X0_, X1_, X2_, X3_, X4_, X5_, W0_, W1_, W2_, W3_, W4_, W5_, W6_, W7_, W8_, CH)              \
    {                                                                                                        \
        const bool first_ = (CH) * P::CK < a.c1;                                                             \
        const auto rs_ = first_ ? rsx1 : rsx2;                                                               \
        const int rb_ = (first_ ? a.c1 : a.c2) * TSZ;                                                        \
        const int ob_ = (first_ ? (CH) * P::CK : (CH) * P::CK - a.c1) * TSZ;                                 \
        X0_ = WS_XLD(rs_, rb_, ob_, pix0); X1_ = WS_XLD(rs_, rb_, ob_, pix1); X2_ = WS_XLD(rs_, rb_, ob_, pix2); \
        X3_ = WS_XLD(rs_, rb_, ob_, pix3); X4_ = WS_XLD(rs_, rb_, ob_, pix4); X5_ = WS_XLD(rs_, rb_, ob_, pix5); \
        W0_ = WS_WLD(0, CH); W1_ = WS_WLD(1, CH); W2_ = WS_WLD(2, CH); W3_ = WS_WLD(3, CH); W4_ = WS_WLD(4, CH); \
        W5_ = WS_WLD(5, CH); W6_ = WS_WLD(6, CH); W7_ = WS_WLD(7, CH); W8_ = WS_WLD(8, CH);                  \
        asm volatile("" ::: "memory");                                                                       \
    }
#define WS_ISSUE_A(CH) WS_ISSUE(xa0, xa1, xa2, xa3, xa4, xa5, wa0, wa1, wa2, wa3, wa4, wa5, wa6, wa7, wa8, CH)
#define WS_ISSUE_B(CH) WS_ISSUE(xb_0, xb_1, xb_2, xb_3, xb_4, xb_5, wb0, wb1, wb2, wb3, wb4, wb5, wb6, wb7, wb8, CH)
#define WS_STORE(XT, WT_, V0, V1, V2, V3, V4, V5, U0, U1, U2, U3, U4, U5, U6, U7, U8)                        \
    *reinterpret_cast<u32x4*>((XT) + lds_slot) = V0;                                                         \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 64 * PIXB) = V1;                                             \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 128 * PIXB) = V2;                                            \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 192 * PIXB) = V3;                                            \
    *reinterpret_cast<u32x4*>((XT) + lds_slot + 256 * PIXB) = V4;                                            \
    if (lt < 16) *reinterpret_cast<u32x4*>((XT) + lds_slot + 320 * PIXB) = V5;                               \
    *reinterpret_cast<u32x4*>((WT_) + lt * 16) = U0;             *reinterpret_cast<u32x4*>((WT_) + (lt + 256) * 16) = U1;  \
    *reinterpret_cast<u32x4*>((WT_) + (lt + 512) * 16) = U2;     *reinterpret_cast<u32x4*>((WT_) + (lt + 768) * 16) = U3;  \
    *reinterpret_cast<u32x4*>((WT_) + (lt + 1024) * 16) = U4;    *reinterpret_cast<u32x4*>((WT_) + (lt + 1280) * 16) = U5; \
    *reinterpret_cast<u32x4*>((WT_) + (lt + 1536) * 16) = U6;    *reinterpret_cast<u32x4*>((WT_) + (lt + 1792) * 16) = U7; \
    *reinterpret_cast<u32x4*>((WT_) + (lt + 2048) * 16) = U8;
#define WS_STORE_A WS_STORE(xb0, wt0, xa0, xa1, xa2, xa3, xa4, xa5, wa0, wa1, wa2, wa3, wa4, wa5, wa6, wa7, wa8)
#define WS_STORE_B WS_STORE(xb1, wt1, xb_0, xb_1, xb_2, xb_3, xb_4, xb_5, wb0, wb1, wb2, wb3, wb4, wb5, wb6, wb7, wb8)
        // Stage s = (local item s / nch, chunk s % nch); even stages use set A / buffers 0, odd ones set B / buffers 1.
        // Load order: A(0) B(1) A(2) | B(3) A(4) | B(5) A(6) | ...   (kq, cq): cursor of the next even stage to issue.
        int kq = 0, cq = 0;
        WS_PIXELS(kq)
        WS_ISSUE_A(cq)
        WS_ISSUE_B(cq + 1)
        WS_STORE_A
        cq += 2; if (cq == nch) { cq = 0; ++kq; }
        WS_PIXELS(kq)                                  // pix* describe the item of stages (kq, cq), (kq, cq + 1)
        WS_ISSUE_A(cq)
        for (int sidx = 0; sidx < nloc * nch; sidx += 2) {
            lds_barrier();                             // even stage ready in buffers 0; buffers 1 are free
            WS_STORE_B
            WS_ISSUE_B(cq + 1)
            lds_barrier();                             // odd stage ready in buffers 1; buffers 0 are free
            WS_STORE_A
            cq += 2; if (cq == nch) { cq = 0; ++kq; }
            WS_PIXELS(kq)
            WS_ISSUE_A(cq)
        }
#undef WS_PIX
#undef WS_PIXELS
#undef WS_XLD
#undef WS_WLD
#undef WS_ISSUE
#undef WS_ISSUE_A
#undef WS_ISSUE_B
#undef WS_STORE
#undef WS_STORE_A
#undef WS_STORE_B
    } else {
        ws_mma_role<P, EPI, MOS>(a, xb0, xb1, wt0, wt1, gb, wave, lane, o, nch);
    }
}

// Epilogue shared by the small-map kernels below, whose four waves each hold a partial [4 m-tiles][NT n-tiles] result of
// the same 64 pixels x 16*NT channels (K split): partials go through LDS (red: 4 * 4 * NT * 1 KB), wave w adds the four
// partials of m-tile w in a fixed order, applies bias / ReLU and stores pixel gpix (lane's pixel of that m-tile) if valid.
template <typename T, int NT>
__device__ __forceinline__ void ksplit4_epilogue(float* red, const f32x4 (&acc)[4][NT], const ConvArgs& a, int wave, int lane,
                                                 int nb, bool valid, size_t gpix) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<f32x4*>(red + (((wave * 4 + mt) * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
    lds_barrier();
    if (!valid) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cg = (nb * NT + nt) * 16;                         // first channel of this n-tile
        if (cg >= a.cout_real) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(red + (((0 * 4 + wave) * NT + nt) * 64 + lane) * 4);
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) v += *reinterpret_cast<const f32x4*>(red + (((w2 * 4 + wave) * NT + nt) * 64 + lane) * 4);
        if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + cg + (lane >> 4) * 4);
        if (a.epilogue == AD_EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        T* dst = (cg < a.cy1 ? reinterpret_cast<T*>(a.y1) + gpix * a.cy1 + cg
                             : reinterpret_cast<T*>(a.y2) + gpix * (a.cout_real - a.cy1) + (cg - a.cy1)) + (lane >> 4) * 4;
        typedef typename Half16<T>::v4 h4;
        *reinterpret_cast<h4*>(dst) = h4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
    }
}

// ------------------------------------------------------------------ forward / dgrad on 4x4 feature maps (bf16 / fp16)
// The 4x4 level of a deep model (K2': 64 images x 16 pixels, 256 ... 1024 channels) is a GEMM of only 1 024 rows with
// K = 9 * Cin up to 9 216: 256-pixel tiles give 32 work items, which the generic kernel spreads over the chip by
// splitting K 16 ways through fp32 slabs in memory (33 MB written and read back for 1 MB of output, plus the finalize
// launch).  Here a workgroup owns 64 pixels (four whole images) x 16*NT output channels and the K split happens INSIDE
// it: each of the four waves takes one 32-channel chunk of every 128-channel phase (all nine taps) and the four partial
// tiles are added through LDS at the end, in a fixed order.  No slab, no second launch.
//   * activations: the four images' 6x6 zero-bordered halos of the phase's 128 channels are staged in LDS (registers ->
//     LDS, double buffered, one barrier per phase), one read of a fragment per tap and m-tile;
//   * weights: no reuse inside a workgroup (every wave multiplies different channels), so the fragments go from L2
//     straight into the registers that feed the MFMAs; the fragment of tap t is refilled with the next phase's tap t
//     right after its last use: a prefetch distance of exactly one phase (9 tap steps) with static register indices.
// Loads are issued unconditionally (the last phase re-fetches itself) so that every s_waitcnt vmcnt stays exact.
constexpr int M4_T = 256;
constexpr int M4_IMG = 4;                        // images per workgroup = m-tiles per wave
constexpr int M4_NPH = M4_IMG * 36;              // halo pixels
constexpr int M4_XCH = M4_NPH * PIXB;            // one 32-channel chunk of the halo tile (13 824 B)
constexpr int M4_XB = 4 * M4_XCH;                // one phase
constexpr int M4_XS = M4_NPH * 4 * 4 / M4_T;     // 16-byte staging slots per thread and phase (= 9)
constexpr size_t M4_LDS = 2 * (size_t)M4_XB + M4_NPH * 4;

template <typename P, int NT>
__global__ __launch_bounds__(M4_T, 1) void conv3x3_map4_kernel(ConvArgs a) {
    typedef typename P::T T;
    typedef typename P::v8 v8;
    static_assert(sizeof(T) == 2 && M4_XS == 9, "bf16 / fp16 path, nine staging slots per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TSZ = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* gtab = reinterpret_cast<int*>(smem + 2 * M4_XB);
    const int nbk = (a.cout_real + 16 * NT - 1) / (16 * NT);
    const int mb = blockIdx.x / nbk, nb = blockIdx.x - mb * nbk;     // neighbours in the grid share the input tile
    const int cin = a.c1 + a.c2;
    const int nph = cin / 128;
    const int kc_total = cin / P::KV;

    // halo table: flat pixel index, or its complement where the halo pixel is padding / past the batch
    for (int hp = tid; hp < M4_NPH; hp += M4_T) {
        const int img = hp / 36, r = hp - img * 36, hy = r / 6, hx = r - hy * 6;
        const int nn = mb * M4_IMG + img, y = hy - 1, x = hx - 1;
        const bool ok = nn < a.n && (unsigned)y < 4u && (unsigned)x < 4u;
        const int idx = (min(nn, a.n - 1) * 4 + min(max(y, 0), 3)) * 4 + min(max(x, 0), 3);
        gtab[hp] = ok ? idx : ~idx;
    }
    __syncthreads();
    // this thread's nine staging slots of a phase: slot s = tid + 256 i -> halo pixel s >> 4 (= tid / 16 + 16 i), 16-byte
    // piece s & 15 of the pixel's 256 bytes (chunk (s >> 2) & 3, part s & 3): 16 lanes read one contiguous 256-byte run.
    // The concat boundary is phase aligned (launcher), so source, row pitch and channel offset are uniform per phase.
    int ix[M4_XS];
#pragma unroll
    for (int i = 0; i < M4_XS; ++i) ix[i] = gtab[(tid >> 4) + 16 * i];
    const int ko = (tid & 15) * 16;
    const int lds_slot = ((tid >> 2) & 3) * M4_XCH + (tid >> 4) * PIXB + (tid & 3) * 16;     // + i * 16 * PIXB
    uint4 xs[M4_XS];
    auto issue_x = [&](int ph) {
        const int c0 = ph * 128;
        const bool first = c0 < a.c1;
        const char* src = (first ? a.x1 : a.x2) + (first ? c0 : c0 - a.c1) * TSZ;
        const int rb = (first ? a.c1 : a.c2) * TSZ;
#pragma unroll
        for (int i = 0; i < M4_XS; ++i) {
            const unsigned idx = ix[i] >= 0 ? ix[i] : ~ix[i];
            xs[i] = *reinterpret_cast<const uint4*>(src + (idx * (unsigned)rb + (unsigned)ko));
        }
    };
    auto store_x = [&](char* xb) {
#pragma unroll
        for (int i = 0; i < M4_XS; ++i)
            *reinterpret_cast<uint4*>(xb + lds_slot + i * 16 * PIXB) = ix[i] >= 0 ? xs[i] : make_uint4(0, 0, 0, 0);
    };
    // weights: fragment (tap, n-tile) of chunk ch = lanes (l & 15) -> output channel, (l >> 4) -> 8-channel group;
    // a uniform base per (phase, tap) plus one 32-bit lane offset
    const unsigned wlo = (unsigned)(((lane >> 4) * a.cout + nb * 16 * NT + (lane & 15)) * 16);
    v8 wq[9][NT];
    auto issue_w = [&](int ph, int t) {
        const char* wc = a.wp + ((size_t)(t * kc_total + (ph * 4 + wave) * 4) * a.cout) * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[t][nt] = *reinterpret_cast<const v8*>(wc + (wlo + nt * 256));
    };
    int abase[M4_IMG];
#pragma unroll
    for (int mt = 0; mt < M4_IMG; ++mt) {
        const int pq = lane & 15;
        abase[mt] = wave * M4_XCH + ((mt * 6 + (pq >> 2) + 1) * 6 + (pq & 3) + 1) * PIXB + P::a_lane_off(lane);
    }
    f32x4 acc[M4_IMG][NT];
#pragma unroll
    for (int i = 0; i < M4_IMG; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_x(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) issue_w(0, t);
    store_x(smem);
    lds_barrier();
    int cur = 0;
    for (int ph = 0; ph < nph; ++ph) {
        const int php = min(ph + 1, nph - 1);          // the last phase re-fetches itself: loads stay unconditional
        const char* xb = smem + cur * M4_XB;
        issue_x(php);
        __builtin_amdgcn_sched_barrier(0);             // the X loads first, then one weight refill per tap: the waits
        v8 xf[2][M4_IMG];                              // below count on this order (vmcnt is in-order)
#pragma unroll
        for (int mt = 0; mt < M4_IMG; ++mt) xf[0][mt] = *reinterpret_cast<const v8*>(xb + abase[mt] - 7 * PIXB);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) {
                const int toff = (((t + 1) / 3 - 1) * 6 + ((t + 1) % 3 - 1)) * PIXB;
#pragma unroll
                for (int mt = 0; mt < M4_IMG; ++mt) xf[(t + 1) & 1][mt] = *reinterpret_cast<const v8*>(xb + abase[mt] + toff);
            }
#pragma unroll
            for (int mt = 0; mt < M4_IMG; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Half16<T>::mfma(wq[t][nt], xf[t & 1][mt], acc[mt][nt]);
            __builtin_amdgcn_sched_barrier(0);
            issue_w(php, t);
            __builtin_amdgcn_sched_barrier(0);
        }
        store_x(smem + (cur ^ 1) * M4_XB);
        lds_barrier();
        cur ^= 1;
    }
    // the four waves' partial tiles are added through LDS (both halo buffers are free now); wave w finishes m-tile w
    ksplit4_epilogue<T, NT>(reinterpret_cast<float*>(smem), acc, a, wave, lane, nb, mb * M4_IMG + wave < a.n,
                            (size_t)(mb * M4_IMG + wave) * 16 + (lane & 15));
}

// ------------------------------------------------------------------ forward / dgrad on 1x1 feature maps (bf16 / fp16)
// With "same" padding only the centre tap of a 1x1 map sees data: the layer is a GEMM [N images] x [Cin] x [Cout] (K2':
// 64 x 1024 x 1024, 134 MFLOP) whose cost is one memory round trip.  A workgroup owns 64 images x 16*NT channels; its four
// waves split the 32-channel chunks, each issues ALL its fragment loads (activations and centre-tap weights, straight
// from global memory in MFMA operand layout) before the first MFMA, eight chunk steps at a time, and the partial tiles
// are added through LDS.  Steps past the end of K re-load the last chunk and multiply zeros: loads stay unconditional.
constexpr int M1_T = 256;
constexpr int M1_G = 8;                            // chunk steps in flight per wave

template <typename P, int NT>
__global__ __launch_bounds__(M1_T, 1) void conv3x3_map1_kernel(ConvArgs a) {
    typedef typename P::T T;
    typedef typename P::v8 v8;
    static_assert(sizeof(T) == 2, "bf16 / fp16 path");
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * NT * 64 * 4];
    constexpr int TSZ = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbk = (a.cout_real + 16 * NT - 1) / (16 * NT);
    const int mb = blockIdx.x / nbk, nb = blockIdx.x - mb * nbk;
    const int cin = a.c1 + a.c2;
    const int nch = cin / P::CK;                   // multiple of 4 (launcher): every wave has nch / 4 steps
    const int kc_total = cin / P::KV;
    const int nsteps = nch / 4;
    unsigned row[4];                               // image of this lane in each m-tile (clamped: rows past the batch are not stored)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) row[mt] = (unsigned)min(mb * 64 + mt * 16 + (lane & 15), a.n - 1);
    const unsigned klo = (unsigned)((lane >> 4) * P::KV * TSZ);
    const unsigned wlo = (unsigned)(((lane >> 4) * a.cout + nb * 16 * NT + (lane & 15)) * 16);
    const char* wc4 = a.wp + (size_t)4 * kc_total * a.cout * 16;          // centre tap
    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < nsteps; j0 += M1_G) {
        v8 xf[M1_G][4], wf[M1_G][NT];
#pragma unroll
        for (int j = 0; j < M1_G; ++j) {
            const int ch = wave + 4 * min(j0 + j, nsteps - 1);
            const int c0 = ch * P::CK;
            const bool first = c0 < a.c1;
            const char* src = (first ? a.x1 : a.x2) + (first ? c0 : c0 - a.c1) * TSZ;
            const unsigned rb = (unsigned)((first ? a.c1 : a.c2) * TSZ);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[j][mt] = *reinterpret_cast<const v8*>(src + (row[mt] * rb + klo));
            const char* wc = wc4 + (size_t)ch * 4 * a.cout * 16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[j][nt] = *reinterpret_cast<const v8*>(wc + (wlo + nt * 256));
        }
#pragma unroll
        for (int j = 0; j < M1_G; ++j) {
            if (j0 + j < nsteps) {                 // uniform
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Half16<T>::mfma(wf[j][nt], xf[j][mt], acc[mt][nt]);
            }
        }
    }
    ksplit4_epilogue<T, NT>(red, acc, a, wave, lane, nb, mb * 64 + wave * 16 + (lane & 15) < a.n,
                            (size_t)(mb * 64 + wave * 16 + (lane & 15)));
}

// ------------------------------------------------------------------ wgrad
struct WgradArgs {
    const char* x1; const char* x2; int c1, c2;
    const char* dz;
    float* ws;
    float* dw; int cin_real;   // nsplit == 1: the kernel writes dw_hwio itself, there is nothing to reduce
    int n, h, w, cout;
    int ntiles, tiles_per_split, ncib, ncob;
    Geo g;
};

template <typename P> struct WgradPol;

template <typename E> struct WgradPol<Pol16<E>> {
    typedef typename Half16<E>::v8 v8;
    static constexpr int NACC = 2;   // n-tiles per wave (one m-tile of 16 input channels)
    static constexpr int DZS = BN * 2 + 32;   // 160 B: 8 consecutive pixels x 32 B tile the 256-B bank row
    static constexpr int DSLOTS = BN * 2 / 16;   // 16-byte dz slots per thread (one pixel row each)
    static __device__ __forceinline__ v8 tr_pair(const char* p0, const char* p1) {
        typedef __attribute__((address_space(3))) short4_t* lds_p;
        short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
        short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        short8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(v8, r);
    }
    // Per-lane LDS offsets of the 8 k-steps (tile independent): k index 8*grp + j of a k-step <-> pixel
    // ks*32 + 16*(j>>2) + 4*grp + (j&3), so each transposed read covers 8 consecutive pixels per 32-lane half
    // (conflict-free); the x (A) and dz (B) operands use the same map.
    struct Lane {
        int xa[TM / 32], xb[TM / 32];   // halo-tile byte offsets of the two transposed reads of the A fragment
        int dz;                         // dz-tile byte offset of k-step 0 (k-steps are 32 * DZS apart)
    };
    static __device__ __forceinline__ Lane lane_setup(const int* hbase, int lane, int wave) {
        const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int mt = wave & 1, nt0 = (wave >> 1) * 2;
        Lane l;
#pragma unroll
        for (int ks = 0; ks < TM / 32; ++ks) {
            const int m = ks * 32 + grp * 4 + q;
            l.xa[ks] = hbase[m] * PIXB + mt * 32 + p * 8;
            l.xb[ks] = hbase[m + 16] * PIXB + mt * 32 + p * 8;
        }
        l.dz = (grp * 4 + q) * DZS + nt0 * 32 + p * 8;
        return l;
    }
    // One tile (256 pixels) of K: 8 k-steps x (9 taps x 2 MFMAs).  Software pipelined inside the wave: a ring of
    // NPRE A fragments is read NPRE taps ahead and the B fragments one k-step ahead, so the transposed LDS reads
    // overlap the MFMAs of earlier taps instead of stalling every pair of them.
    template <bool HALO>
    static __device__ __forceinline__ void tile(f32x4 (&acc)[9][NACC], const char* xt, const char* dzt, const Lane& l,
                                                int row_bytes) {
        constexpr int NT = HALO ? 9 : 1;
        constexpr int NS = (TM / 32) * NT;     // stages = (k-step, tap)
        constexpr int NPRE = 4;
        v8 afr[NPRE];
        v8 bfr[2][NACC];
#define AD_LOAD_A(S)                                                                                          \
    {                                                                                                         \
        constexpr int ks_ = (S) / NT, t_ = (S) % NT;                                                          \
        const int toff_ = HALO ? (t_ / 3 - 1) * row_bytes + (t_ % 3 - 1) * PIXB : 0;                          \
        afr[(S) % NPRE] = tr_pair(xt + l.xa[ks_] + toff_, xt + l.xb[ks_] + toff_);                            \
    }
#define AD_LOAD_B(KS)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < NACC; ++j)                                                          \
        bfr[(KS) & 1][j] = tr_pair(dzt + l.dz + (KS) * 32 * DZS + j * 32, dzt + l.dz + ((KS) * 32 + 16) * DZS + j * 32);
        AD_LOAD_B(0)
        AD_LOAD_A(0) AD_LOAD_A(1 < NS ? 1 : 0) AD_LOAD_A(2 < NS ? 2 : 0) AD_LOAD_A(3 < NS ? 3 : 0)
        ad_tile_stages<HALO, 0>(acc, xt, dzt, l, row_bytes, afr, bfr);
#undef AD_LOAD_A
#undef AD_LOAD_B
    }
    // compile-time recursion over the stages keeps every fragment index a constant (no scratch arrays)
    template <bool HALO, int S>
    static __device__ __forceinline__ void ad_tile_stages(f32x4 (&acc)[9][NACC], const char* xt, const char* dzt,
                                                          const Lane& l, int row_bytes, v8 (&afr)[4],
                                                          v8 (&bfr)[2][NACC]) {
        constexpr int NT = HALO ? 9 : 1;
        constexpr int NS = (TM / 32) * NT;
        constexpr int NPRE = 4;
        if constexpr (S < NS) {
            constexpr int ks = S / NT, t = S % NT;
            constexpr int tap = HALO ? t : 4;
            if constexpr (t == 0 && ks + 1 < TM / 32) {
#pragma unroll
                for (int j = 0; j < NACC; ++j)
                    bfr[(ks + 1) & 1][j] = tr_pair(dzt + l.dz + (ks + 1) * 32 * DZS + j * 32,
                                                   dzt + l.dz + ((ks + 1) * 32 + 16) * DZS + j * 32);
            }
            const v8 a_cur = afr[S % NPRE];
            if constexpr (S + NPRE < NS) {
                constexpr int ks2 = (S + NPRE) / NT, t2 = (S + NPRE) % NT;
                const int toff2 = HALO ? (t2 / 3 - 1) * row_bytes + (t2 % 3 - 1) * PIXB : 0;
                afr[S % NPRE] = tr_pair(xt + l.xa[ks2] + toff2, xt + l.xb[ks2] + toff2);
            }
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[tap][j] = Half16<E>::mfma(a_cur, bfr[ks & 1][j], acc[tap][j]);
            ad_tile_stages<HALO, S + 1>(acc, xt, dzt, l, row_bytes, afr, bfr);
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
    static constexpr int DSLOTS = BN * 4 / 16;
    struct Lane {
        const int* hbase;
        int kk, i, wave;
    };
    static __device__ __forceinline__ Lane lane_setup(const int* hbase, int lane, int wave) {
        return Lane{hbase, lane >> 4, lane & 15, wave};
    }
    template <bool HALO>
    static __device__ __forceinline__ void tile(f32x4 (&acc)[9][NACC], const char* xt, const char* dzt, const Lane& l,
                                                int row_bytes) {
        for (int ks = 0; ks < TM / 4; ++ks) {
            const int m = ks * 4 + l.kk;
            const int hb = l.hbase[m] * PIXB + l.i * 4;
            const float bfr = *reinterpret_cast<const float*>(dzt + m * DZS + (l.wave * 16 + l.i) * 4);
#pragma unroll
            for (int t = 0; t < (HALO ? 9 : 1); ++t) {
                const int tap = HALO ? t : 4;
                const int toff = HALO ? (t / 3 - 1) * row_bytes + (t % 3 - 1) * PIXB : 0;
                const float af = *reinterpret_cast<const float*>(xt + hb + toff);
                acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfr, acc[tap][0], 0, 0, 0);
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
// LDS: [gtab0][gtab1][hbase 256][x halo chunk][dz tile]; next tile's data waits in registers.
template <typename P, int XS, bool HALO>
__global__ __launch_bounds__(256, XS <= 6 ? 2 : 1) void conv3x3_wgrad_kernel(WgradArgs a) {
    typedef typename P::T T;
    typedef WgradPol<P> WP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gtab_bytes = g.NPHP * 4;
    int* gtab0 = reinterpret_cast<int*>(smem);
    int* gtab1 = reinterpret_cast<int*>(smem + gtab_bytes);
    int* hbase = reinterpret_cast<int*>(smem + 2 * gtab_bytes);
    char* xt = smem + 2 * gtab_bytes + TM * 4;
    char* dzt = xt + ((g.NPH * PIXB + 15) & ~15);
    constexpr int TSZ = (int)sizeof(T);

    const int split = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;
    const int c0 = cib * P::CK;
    const char* src; int row_bytes, off_bytes;
    if (c0 < a.c1) { src = a.x1; row_bytes = a.c1 * TSZ; off_bytes = c0 * TSZ; }
    else { src = a.x2; row_bytes = a.c2 * TSZ; off_bytes = (c0 - a.c1) * TSZ; }

    for (int m = tid; m < TM; m += 256) hbase[m] = halo_of(m, g);
    lds_barrier();
    const typename WP::Lane lsetup = WP::lane_setup(hbase, lane, wave);
    // dz slot s = tid + 256*i -> pixel s / DSLOTS, part s % DSLOTS; its halo index is tile independent
    int dz_hb[WP::DSLOTS];
#pragma unroll
    for (int i = 0; i < WP::DSLOTS; ++i) dz_hb[i] = halo_of((tid + 256 * i) / WP::DSLOTS, g);

    f32x4 acc[9][WP::NACC];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < WP::NACC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 xr[XS];
    uint4 dr[WP::DSLOTS];
    // 16-byte dz parts of this output block that exist in a row (8 / 16 of them unless the last block is ragged)
    const int dz_parts = min(BN, a.cout - cob * BN) * TSZ / 16;
#define WG_ISSUE(GT)                                                                                           \
    do {                                                                                                       \
        load_halo<XS, 256>(xr, (GT), src, row_bytes, off_bytes, tid);                                          \
        _Pragma("unroll") for (int i = 0; i < WP::DSLOTS; ++i) {                                               \
            const int s_ = tid + 256 * i;                                                                      \
            const int gp_ = (GT)[dz_hb[i]];                                                                    \
            const int idx_ = gp_ >= 0 ? gp_ : ~gp_;                                                            \
            const int part_ = s_ % WP::DSLOTS < dz_parts ? s_ % WP::DSLOTS : 0;   /* past the row: re-read part 0 */ \
            dr[i] = *reinterpret_cast<const uint4*>(                                                           \
                a.dz + ((size_t)idx_ * a.cout + cob * BN) * TSZ + part_ * 16);                                 \
        }                                                                                                      \
    } while (0)

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
    int cur = 0;
    if (t_begin < t_end) {
        build_gtab(gtab0, g, decode_tile(t_begin, g), a.n, a.h, a.w, tid);
        lds_barrier();
        WG_ISSUE(gtab0);
    }
    for (int tile = t_begin; tile < t_end; ++tile) {
        int* gt_cur = cur ? gtab1 : gtab0;
        int* gt_nxt = cur ? gtab0 : gtab1;
        store_halo<XS, 256>(xr, xt, gt_cur, g.NPH, tid);
#pragma unroll
        for (int i = 0; i < WP::DSLOTS; ++i) {
            const int s = tid + 256 * i;
            const bool ok = gt_cur[dz_hb[i]] >= 0 && s % WP::DSLOTS < dz_parts;
            *reinterpret_cast<uint4*>(dzt + (s / WP::DSLOTS) * WP::DZS + (s % WP::DSLOTS) * 16) =
                ok ? dr[i] : make_uint4(0, 0, 0, 0);
        }
        const bool has_next = tile + 1 < t_end;
        if (has_next) build_gtab(gt_nxt, g, decode_tile(tile + 1, g), a.n, a.h, a.w, tid);
        lds_barrier();
        if (has_next) WG_ISSUE(gt_nxt);
        WP::template tile<HALO>(acc, xt, dzt, lsetup, g.HW * PIXB);
        lds_barrier();
        cur ^= 1;
    }

#undef WG_ISSUE
    float* slab = a.ws + ((size_t)(split * a.ncib + cib) * a.ncob + cob) * (9 * P::CK * BN);
    const bool direct = gridDim.x == 1 && a.dw != nullptr;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < WP::NACC; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ci, co;
                WP::coords(wave, lane, j, r, &ci, &co);
                if (direct) {
                    if (cib * P::CK + ci < a.cin_real && cob * BN + co < a.cout)
                        a.dw[((size_t)tap * a.cin_real + cib * P::CK + ci) * a.cout + cob * BN + co] = acc[tap][j][r];
                } else {
                    slab[(tap * P::CK + ci) * BN + co] = acc[tap][j][r];
                }
            }
}

// ------------------------------------------------------------------ wgrad, wave specialised (bf16, 16x16 tiles)
// One 512-thread workgroup accumulates 9 taps x 64 input channels x 64 output channels over its share of tiles.
//   * waves 0-3 (MFMA): wave w owns input channels 16w..16w+15 and ALL 64 output channels, 9 x 4 accumulator tiles
//     (144 registers).  Per k-step that is 9 transposed A reads (one per tap) + 4 B reads for 36 MFMAs, against
//     9 + 2 for 18 in conv3x3_wgrad_kernel: the LDS traffic per MFMA, which bounds that kernel, drops by 40 %, and
//     dz is fetched once per 64 input channels instead of once per 32.
//   * waves 4-7 (loaders): buffer loads (out-of-image -> zeros) a full stage ahead, then LDS stores in the shadow of
//     the MFMA phases.
// A tile is processed as two halves (tile rows 0-7 = k-steps 0-3, rows 8-15 = k-steps 4-7), each with its own x
// halo buffer (10 halo rows x 18 columns, two 32-channel chunks) and dz buffer, so one half is refilled while the
// other is consumed; two barriers per tile.  LDS: 4 x 17,280 + 2 x 20,480 = 110,080 B.
constexpr int W2_T = 512;
constexpr int W2_XPIX = 10 * 18;                 // halo pixels per half
constexpr int W2_XB = W2_XPIX * PIXB;            // one 32-channel chunk of one half (17,280 B)
constexpr int W2_DZS = BN * 2 + 32;              // dz pixel stride (160 B), as WgradPol<PolBF16>::DZS
constexpr int W2_DZB = 128 * W2_DZS;             // dz half tile (20,480 B)
constexpr size_t W2_LDS = 4 * W2_XB + 2 * W2_DZB;
constexpr int W2_XSL = 6;                        // x slots per loader thread and half (1440 of 1536 used)
constexpr int W2_DSL = 4;                        // dz slots per loader thread and half

// One half tile (4 k-steps) of K for one MFMA wave.  xh: this wave's 32-channel chunk of the half's halo buffer,
// already offset to its 16 channels; dzh: the half's dz buffer.  Same k <-> pixel map and fragment ring as
// WgradPol<PolBF16>::tile.
template <typename E, int S>
__device__ __forceinline__ void w2_stages(f32x4 (&acc)[9][4], const char* xh, const char* dzh, const int (&xa)[4],
                                          const int (&xb)[4], int dzo, typename Half16<E>::v8 (&afr)[4], typename Half16<E>::v8 (&bfr)[2][4]) {
    typedef WgradPol<Pol16<E>> WP;
    constexpr int NS = 4 * 9, NPRE = 4, RB = 18 * PIXB;
    if constexpr (S < NS) {
        constexpr int ks = S / 9, t = S % 9;
        if constexpr (t == 0 && ks + 1 < 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[(ks + 1) & 1][j] = WP::tr_pair(dzh + dzo + (ks + 1) * 32 * W2_DZS + j * 32,
                                                   dzh + dzo + ((ks + 1) * 32 + 16) * W2_DZS + j * 32);
        }
        const typename Half16<E>::v8 a_cur = afr[S % NPRE];
        if constexpr (S + NPRE < NS) {
            constexpr int ks2 = (S + NPRE) / 9, t2 = (S + NPRE) % 9;
            constexpr int toff2 = (t2 / 3 - 1) * RB + (t2 % 3 - 1) * PIXB;
            afr[S % NPRE] = WP::tr_pair(xh + xa[ks2] + toff2, xh + xb[ks2] + toff2);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[t][j] = Half16<E>::mfma(a_cur, bfr[ks & 1][j], acc[t][j]);
        w2_stages<E, S + 1>(acc, xh, dzh, xa, xb, dzo, afr, bfr);
    }
}

template <typename E>
__device__ __forceinline__ void w2_half(f32x4 (&acc)[9][4], const char* xh, const char* dzh, const int (&xa)[4],
                                        const int (&xb)[4], int dzo) {
    typedef WgradPol<Pol16<E>> WP;
    constexpr int RB = 18 * PIXB;
    typename Half16<E>::v8 afr[4], bfr[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bfr[0][j] = WP::tr_pair(dzh + dzo + j * 32, dzh + dzo + 16 * W2_DZS + j * 32);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int toff = (t / 3 - 1) * RB + (t % 3 - 1) * PIXB;
        afr[t] = WP::tr_pair(xh + xa[0] + toff, xh + xb[0] + toff);
    }
    w2_stages<E, 0>(acc, xh, dzh, xa, xb, dzo, afr, bfr);
}

// grid: x = K split, y = 64-input-channel block, z = 64-output-channel block
// ws slab layout: [split][cib][cob][tap][64][64] fp32
template <typename E, bool MOS = false>
__global__ __launch_bounds__(W2_T, 1) void conv3x3_wgrad_ws_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo& g = a.g;      // geometry is (1, 16, 16)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int TSZ = 2;
    const int split = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;
    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
    const int npix = a.n * a.h * a.w;
    AD_CLOCK_BEGIN
    // LDS: half hf: x chunk c at smem + (2*hf + c) * W2_XB, dz at smem + 4*W2_XB + hf * W2_DZB

    if (wave >= 4) {
        // ------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        // the two 32-channel chunks of this input block (virtual concat: either may come from x2)
        const int c0 = cib * 64, c1 = c0 + 32;
        const bool s0 = c0 < a.c1, s1 = c1 < a.c1;
        const int rb0 = (s0 ? a.c1 : a.c2) * TSZ, ob0 = (s0 ? c0 : c0 - a.c1) * TSZ;
        const int rb1 = (s1 ? a.c1 : a.c2) * TSZ, ob1 = (s1 ? c1 : c1 - a.c1) * TSZ;
        const auto rs0 = wave_uniform_rsrc(s0 ? a.x1 : a.x2, npix * rb0);
        const auto rs1 = wave_uniform_rsrc(s1 ? a.x1 : a.x2, npix * rb1);
        const int rbz = a.cout * TSZ;
        const auto rsz = wave_uniform_rsrc(a.dz, npix * rbz);
        // x slot i: chunk i / 3 (compile time, so each load names one descriptor), s = lt + 256 (i % 3) < 720 ->
        // halo pixel s / 4 = (row hy, col hx), part s % 4
        int xdesc[W2_XSL], xlds[W2_XSL];
#pragma unroll
        for (int i = 0; i < W2_XSL; ++i) {
            const int sidx = lt + 256 * (i % 3);
            const int px = sidx >> 2, part = sidx & 3;
            const int hy = px / 18, hx = px - hy * 18;
            const bool used = sidx < 720;
            xdesc[i] = used ? (hy << 8) | hx : (64 << 8);                      // row 64: never inside an image
            xlds[i] = used ? (i / 3) * W2_XB + px * PIXB + part * 16 : -1;
        }
        const int xpart = (lt & 3) * 16;        // 256 % 4 == 0: the part is lt & 3 for every slot
        // dz slot i: s = lt + 256 i -> half-tile pixel s / 8 (row s / 128, col (s / 8) % 16), part s % 8
        const int dpart = (lt & 7) * 16;
        const int dcol = (lt >> 3) & 15;
        const int drow0 = lt >> 7;              // + 2 i
        const int dlds = (lt >> 3) * W2_DZS + dpart;      // + 32 i * W2_DZS
        u32x4 pa0, pa1, pa2, pa3, pa4, pa5, pb0, pb1, pb2, pb3, pb4, pb5;   // x slots of half A / half B
        u32x4 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;                       // dz slots
#define W2_XLD(I, NN, YB, X0)                                                                              \
    ({                                                                                                     \
        const int y_ = (YB) + (xdesc[I] >> 8), x_ = (X0) - 1 + (xdesc[I] & 255);                           \
        const int mp_ = MOS ? mosaic_pix(g, a.n, a.h, a.w, y_, x_) : 0;     /* (YB, X0): mosaic coordinates, NN == 0 */ \
        const bool ok_ = MOS ? mp_ >= 0 : (unsigned)y_ < (unsigned)a.h && (unsigned)x_ < (unsigned)a.w;   \
        const int pix_ = MOS ? mp_ : ((NN) * a.h + y_) * a.w + x_;                                         \
        (I) < 3 ? __builtin_amdgcn_raw_buffer_load_b128(rs0, ok_ ? (unsigned)(pix_ * rb0 + ob0 + xpart) : WR_OOB, 0, 0) \
                : __builtin_amdgcn_raw_buffer_load_b128(rs1, ok_ ? (unsigned)(pix_ * rb1 + ob1 + xpart) : WR_OOB, 0, 0); \
    })
#define W2_ZLD(I, NN, YB, X0)                                                                              \
    ({                                                                                                     \
        const int y_ = (YB) + drow0 + 2 * (I), x_ = (X0) + dcol;                                           \
        const int mp_ = MOS ? mosaic_pix(g, a.n, a.h, a.w, y_, x_) : 0;                                    \
        const bool ok_ = MOS ? mp_ >= 0 : y_ < a.h && x_ < a.w;                                            \
        const int pix_ = MOS ? mp_ : ((NN) * a.h + y_) * a.w + x_;                                         \
        __builtin_amdgcn_raw_buffer_load_b128(rsz, ok_ ? (unsigned)(pix_ * rbz + cob * BN * TSZ + dpart) : WR_OOB, 0, 0); \
    })
#define W2_TILE(TILE)                                                                                      \
    const int tile_ = min((TILE), t_end - 1);   /* past the end: re-fetch the last tile, loads stay unconditional */ \
    const int r_ = tile_ / g.tiles_x;                                                                      \
    const int x0_ = (tile_ - r_ * g.tiles_x) << 4;                                                         \
    const int nn_ = r_ / g.tiles_y;                                                                        \
    const int y0_ = (r_ - nn_ * g.tiles_y) << 4;
#define W2_ISSUE_A(TILE)                                                                                   \
    {                                                                                                      \
        W2_TILE(TILE)                                                                                      \
        const int yb_ = y0_ - 1;                                                                           \
        pa0 = W2_XLD(0, nn_, yb_, x0_); pa1 = W2_XLD(1, nn_, yb_, x0_); pa2 = W2_XLD(2, nn_, yb_, x0_);    \
        pa3 = W2_XLD(3, nn_, yb_, x0_); pa4 = W2_XLD(4, nn_, yb_, x0_); pa5 = W2_XLD(5, nn_, yb_, x0_);    \
        qa0 = W2_ZLD(0, nn_, y0_, x0_); qa1 = W2_ZLD(1, nn_, y0_, x0_);                                    \
        qa2 = W2_ZLD(2, nn_, y0_, x0_); qa3 = W2_ZLD(3, nn_, y0_, x0_);                                    \
        asm volatile("" ::: "memory");                                                                     \
    }
#define W2_ISSUE_B(TILE)                                                                                   \
    {                                                                                                      \
        W2_TILE(TILE)                                                                                      \
        const int yb_ = y0_ + 7;                                                                           \
        pb0 = W2_XLD(0, nn_, yb_, x0_); pb1 = W2_XLD(1, nn_, yb_, x0_); pb2 = W2_XLD(2, nn_, yb_, x0_);    \
        pb3 = W2_XLD(3, nn_, yb_, x0_); pb4 = W2_XLD(4, nn_, yb_, x0_); pb5 = W2_XLD(5, nn_, yb_, x0_);    \
        qb0 = W2_ZLD(0, nn_, y0_ + 8, x0_); qb1 = W2_ZLD(1, nn_, y0_ + 8, x0_);                            \
        qb2 = W2_ZLD(2, nn_, y0_ + 8, x0_); qb3 = W2_ZLD(3, nn_, y0_ + 8, x0_);                            \
        asm volatile("" ::: "memory");                                                                     \
    }
#define W2_STORE(HF, P0, P1, P2, P3, P4, P5, Q0, Q1, Q2, Q3)                                               \
    {                                                                                                      \
        char* xh_ = smem + 2 * (HF) * W2_XB;                                                               \
        char* zh_ = smem + 4 * W2_XB + (HF) * W2_DZB + dlds;                                               \
        *reinterpret_cast<u32x4*>(xh_ + xlds[0]) = P0; *reinterpret_cast<u32x4*>(xh_ + xlds[1]) = P1;     \
        *reinterpret_cast<u32x4*>(xh_ + xlds[3]) = P3; *reinterpret_cast<u32x4*>(xh_ + xlds[4]) = P4;     \
        if (xlds[2] >= 0) {                                                                                \
            *reinterpret_cast<u32x4*>(xh_ + xlds[2]) = P2; *reinterpret_cast<u32x4*>(xh_ + xlds[5]) = P5; \
        }                                                                                                  \
        *reinterpret_cast<u32x4*>(zh_) = Q0; *reinterpret_cast<u32x4*>(zh_ + 32 * W2_DZS) = Q1;            \
        *reinterpret_cast<u32x4*>(zh_ + 64 * W2_DZS) = Q2; *reinterpret_cast<u32x4*>(zh_ + 96 * W2_DZS) = Q3; \
    }
        // load order (the vmcnt arithmetic depends on it): A(0) B(0) A(1) | B(1) A(2) | ...
        int tile = t_begin;
        W2_ISSUE_A(tile)
        W2_ISSUE_B(tile)
        W2_STORE(0, pa0, pa1, pa2, pa3, pa4, pa5, qa0, qa1, qa2, qa3)
        W2_ISSUE_A(tile + 1)
        for (; tile < t_end; ++tile) {
            lds_barrier();                           // B0: half A of this tile is complete; half B is free
            W2_STORE(1, pb0, pb1, pb2, pb3, pb4, pb5, qb0, qb1, qb2, qb3)
            W2_ISSUE_B(tile + 1)
            lds_barrier();                           // B1: half B complete; half A is free
            W2_STORE(0, pa0, pa1, pa2, pa3, pa4, pa5, qa0, qa1, qa2, qa3)
            W2_ISSUE_A(tile + 2)
        }
#undef W2_XLD
#undef W2_ZLD
#undef W2_TILE
#undef W2_ISSUE_A
#undef W2_ISSUE_B
#undef W2_STORE
    } else {
        // ------------------------------------------------------------ MFMA waves
        const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        int xa[4], xb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int m = ks * 32 + grp * 4 + q;                      // pixel of the half tile (row m>>4, col m&15)
            xa[ks] = (((m >> 4) + 1) * 18 + (m & 15) + 1) * PIXB + p * 8;
            xb[ks] = (((m >> 4) + 2) * 18 + (m & 15) + 1) * PIXB + p * 8;
        }
        const int dzo = (grp * 4 + q) * W2_DZS + p * 8;
        const int xw = (wave >> 1) * W2_XB + (wave & 1) * 32;        // this wave's chunk and 16-channel half of it
        f32x4 acc[9][4];
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tile = t_begin; tile < t_end; ++tile) {
            lds_barrier();                           // B0
            w2_half<E>(acc, smem + xw, smem + 4 * W2_XB, xa, xb, dzo);
            lds_barrier();                           // B1
            w2_half<E>(acc, smem + 2 * W2_XB + xw, smem + 4 * W2_XB + W2_DZB, xa, xb, dzo);
        }
        if (a.dw) {
            // one split (>= 256 channel-block pairs: the 512 ... 1 536-channel levels): this workgroup's sums ARE the gradient
            // of its 64 x 64 block; they go to dw_hwio [tap][cin_real][cout] directly, no slab and no reduce launch
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = cib * 64 + wave * 16 + (lane >> 4) * 4 + r;
                    float* row = a.dw + ((size_t)tap * a.cin_real + ci) * a.cout + cob * BN + (lane & 15);
                    if (ci < a.cin_real) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) row[j * 16] = acc[tap][j][r];
                    }
                }
        } else {
            float* slab = a.ws + ((size_t)(split * a.ncib + cib) * a.ncob + cob) * (9 * 64 * BN);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        slab[(tap * 64 + wave * 16 + (lane >> 4) * 4 + r) * BN + j * 16 + (lane & 15)] = acc[tap][j][r];
        }
        AD_CLOCK_END(wave, lane, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))
    }
}

// ------------------------------------------------------------------ first layer: 3 input channels (bf16)
// The network's first conv has Cin = 3.  As an implicit GEMM over 9 taps x 32 zero-padded channels it wastes 90 % of
// its MFMA work and needs a 32-channel padded copy of the input.  Here K is the 27 real (tap, channel) products
// padded to ONE 32-deep MFMA step, k = tap * 3 + ch: the halo tile of the raw fp32 input (18 x 18 x 3) is staged in
// LDS as bf16 and every lane gathers the 8 K-values of its pixel; the 64 x 32 weights live in registers (read from
// the fp32 HWIO master weights, no pack).  Forward writes z and a = relu(LayerNorm(z)) (the fused epilogue of the
// wave-specialised kernels); wgrad contracts the same patches against dz with the transposed LDS read.  Both are
// pure streams: 12 B/pixel in, 128 B/pixel out (forward) or in (wgrad).
struct C3Args {
    const float* x;            // [N, H, W, 3] fp32
    const float* w;            // [27, 64] fp32 (HWIO flattened), forward
    const float* bias; const float* gamma; const float* beta; float eps;
    char* z; char* act; float* mean; float* rstd;      // forward outputs
    const char* dz; float* ws;                         // wgrad: dz [N, H, W, 64] bf16, slabs [grid][27][64] fp32
    int n, h, w_img, tiles_x, tiles_y, ntiles;
};
constexpr int C3_T = 256;
constexpr int C3_HALO = 18 * 18 * 3;          // elements of one halo tile
constexpr int C3_HB = 1984;                   // bytes of one bf16 halo buffer (972 * 2, padded)

// element e of the halo tile = (row e / 54, col (e % 54) / 3, channel e % 3); four per thread (972 of 1024 used)
struct C3Halo {
    int dy[4], dx[4], ch[4];
};
__device__ __forceinline__ C3Halo c3_halo_setup(int tid) {
    C3Halo hsl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + C3_T * i;
        const int r = e / 54, rem = e - r * 54;
        hsl.dy[i] = e < C3_HALO ? r : 1 << 20;       // far outside every image: loads zeros, never stored
        hsl.dx[i] = rem / 3;
        hsl.ch[i] = rem - (rem / 3) * 3;
    }
    return hsl;
}
#define C3_LOAD_HALO(V, RSX, NN, Y0, X0)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        const int y_ = (Y0) - 1 + hsl.dy[i], x_ = (X0) - 1 + hsl.dx[i];                                   \
        const bool ok_ = (unsigned)y_ < (unsigned)a.h && (unsigned)x_ < (unsigned)a.w_img;                \
        V[i] = __builtin_amdgcn_raw_buffer_load_b32(                                                      \
            RSX, ok_ ? (unsigned)((((NN) * a.h + y_) * a.w_img + x_) * 12 + hsl.ch[i] * 4) : WR_OOB, 0, 0); \
    }
#define C3_STORE_HALO(V, BUF)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
        if (tid + C3_T * i < C3_HALO)                                                                     \
            reinterpret_cast<E*>(BUF)[tid + C3_T * i] = (E)__builtin_bit_cast(float, V[i]);
#define C3_TILE(T)                                                                                        \
    const int t_ = (T) < a.ntiles ? (T) : a.ntiles - 1;      /* past the end: re-fetch, loads stay unconditional */ \
    const int r_ = t_ / a.tiles_x;                                                                        \
    const int x0 = (t_ - r_ * a.tiles_x) << 4;                                                            \
    const int nn = r_ / a.tiles_y;                                                                        \
    const int y0 = (r_ - nn * a.tiles_y) << 4;

// MODE 0: z, act = relu(LayerNorm(z)), mean, rstd; 1: act alone (z == NULL: inference); 2: z = conv + bias alone (the first
// Conv2D of the BatchNorm segmentation model, Segmenation/code/train_adaptive_unet.py:326: its BatchNorm needs batch statistics)
template <typename E, int MODE = 0>
__global__ __launch_bounds__(C3_T, 2) void conv3x3_c3_fwd_kernel(C3Args a) {
    constexpr bool ACT_ONLY = MODE == 1, PLAIN = MODE == 2;
    typedef typename Half16<E>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* hb0 = smem;
    char* hb1 = smem + C3_HB;
    float* gb = reinterpret_cast<float*>(smem + 2 * C3_HB);      // [gamma][beta][bias]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
    const int npix = a.n * a.h * a.w_img;
    if (tid < 64) {
        gb[tid] = PLAIN ? 0.f : a.gamma[tid]; gb[64 + tid] = PLAIN ? 0.f : a.beta[tid]; gb[128 + tid] = a.bias ? a.bias[tid] : 0.f;
    }
    // weight fragments (MFMA operand A): row = output channel nt*16 + (lane & 15), k = 8 grp .. 8 grp + 7
    v8 wf[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * grp + j;
            wf[nt][j] = (E)(k < 27 ? a.w[k * 64 + nt * 16 + (lane & 15)] : 0.f);
        }
    // patch gather: K index k of pixel (row, col) sits at halo element (row * 18 + col) * 3 + k + 45 * (k / 9)
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * grp + j;
        koff[j] = k < 27 ? (k + 45 * (k / 9)) * 2 : -1;
    }
    const C3Halo hsl = c3_halo_setup(tid);
    const auto rsx = wave_uniform_rsrc(a.x, npix * 12);
    const auto rsz = wave_uniform_rsrc(ACT_ONLY ? a.act : a.z, npix * 128);
    const auto rsa = wave_uniform_rsrc(PLAIN ? a.z : a.act, npix * 128);
    const auto rsm = wave_uniform_rsrc(ACT_ONLY ? (const void*)a.act : PLAIN ? (const void*)a.z : (const void*)a.mean, npix * 4);
    const auto rsr = wave_uniform_rsrc(ACT_ONLY ? (const void*)a.act : PLAIN ? (const void*)a.z : (const void*)a.rstd, npix * 4);
    int soff[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        soff[mt] = (((wave * 4 + mt) * a.w_img + (lane & 15)) * 64 + (grp & 1) * 16 + (grp >> 1) * 8) * 2;
    unsigned hv[4];
    {
        C3_TILE((int)blockIdx.x)
        C3_LOAD_HALO(hv, rsx, nn, y0, x0)
    }
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        char* hb = buf ? hb1 : hb0;
        C3_STORE_HALO(hv, hb)
        lds_barrier();                              // also orders gb on the first pass
        {
            C3_TILE(tile + (int)gridDim.x)
            C3_LOAD_HALO(hv, rsx, nn, y0, x0)        // next tile, lands during this tile's work
        }
        C3_TILE(tile)
        f32x4 acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b4 = *reinterpret_cast<const float4*>(gb + 128 + j * 16 + grp * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const char* pb = hb + (((wave * 4 + mt) * 18 + (lane & 15)) * 3) * 2;
            v8 xf;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xf[j] = koff[j] >= 0 ? *reinterpret_cast<const E*>(pb + koff[j]) : (E)0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = Half16<E>::mfma(wf[nt], xf, acc[mt][nt]);
        }
        u32x4 pend[MODE ? 8 : 16];
        unsigned pvo[4];
        ws_pack_tile<ACT_ONLY ? 6 : PLAIN ? 0 : 2, E>(acc, gb, a.eps, wave, lane, a.h, a.w_img, nn, y0, x0, 64, soff, rsm, rsr, pend, pvo);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (ACT_ONLY) {
                __builtin_amdgcn_raw_buffer_store_b128(pend[i], rsa, pvo[i >> 1], (i & 1) * 64, 0);
            } else if constexpr (PLAIN) {
                __builtin_amdgcn_raw_buffer_store_b128(pend[i], rsz, pvo[i >> 1], (i & 1) * 64, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(pend[i], rsz, pvo[i >> 1], (i & 1) * 64, 0);
                __builtin_amdgcn_raw_buffer_store_b128(pend[8 + i], rsa, pvo[i >> 1], (i & 1) * 64, 0);
            }
        }
        buf ^= 1;
    }
}

// wgrad of the same layer: slab[k][co] = sum over this workgroup's tiles of patch[pixel][k] * dz[pixel][co].
// Wave w owns output channels 16 w .. 16 w + 15 and both 16-row halves of k; per k-step (32 pixels) the dz fragment
// is one transposed-read pair (layout and k <-> pixel map of WgradPol<PolBF16>) and the patch fragments are gathered
// from the bf16 halo.  LDS: two halo buffers + one dz tile (256 pixels x 160 B); dz and the next halo wait in registers.
template <typename E>
__global__ __launch_bounds__(C3_T, 2) void conv3x3_c3_wgrad_kernel(C3Args a) {
    typedef typename Half16<E>::v8 v8;
    typedef WgradPol<Pol16<E>> WP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* hb0 = smem;
    char* hb1 = smem + C3_HB;
    char* dzt = smem + 2 * C3_HB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int npix = a.n * a.h * a.w_img;
    const C3Halo hsl = c3_halo_setup(tid);
    const auto rsx = wave_uniform_rsrc(a.x, npix * 12);
    const auto rsd = wave_uniform_rsrc(a.dz, npix * 128);
    // patch fragment (operand A): row m = (lane & 15) + 16 mt is the K index, element j is pixel
    // ks*32 + 16 (j >> 2) + 4 grp + (j & 3) of the tile = (row 2 ks + (j >> 2), col 4 grp + (j & 3))
    int moff[2], joff[8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = (lane & 15) + 16 * mt;
        moff[mt] = m < 27 ? (m + 45 * (m / 9)) * 2 : -1;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) joff[j] = (((j >> 2) * 18 + 4 * grp + (j & 3)) * 3) * 2;
    const int dzo = (grp * 4 + q) * WP::DZS + wave * 32 + p * 8;
    // dz slot i: s = tid + 256 i -> tile pixel s / 8, 16-byte part s % 8
    const int dpart = (tid & 7) * 16, dcol = (tid >> 3) & 15, drow0 = tid >> 7;       // row drow0 + 2 i
    const int dlds = (tid >> 3) * WP::DZS + dpart;                                      // + 32 i * DZS
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    unsigned hv[4];
    u32x4 dv[8];
#define C3_LOAD_DZ(NN, Y0, X0)                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                       \
        const int y_ = (Y0) + drow0 + 2 * i, x_ = (X0) + dcol;                                            \
        const bool ok_ = y_ < a.h && x_ < a.w_img;                                                        \
        dv[i] = __builtin_amdgcn_raw_buffer_load_b128(                                                    \
            rsd, ok_ ? (unsigned)((((NN) * a.h + y_) * a.w_img + x_) * 128 + dpart) : WR_OOB, 0, 0);      \
    }
    {
        C3_TILE((int)blockIdx.x)
        C3_LOAD_HALO(hv, rsx, nn, y0, x0)
        C3_LOAD_DZ(nn, y0, x0)
    }
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        char* hb = buf ? hb1 : hb0;
        C3_STORE_HALO(hv, hb)
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dzt + dlds + 32 * i * WP::DZS) = dv[i];
        lds_barrier();
        {
            C3_TILE(tile + (int)gridDim.x)
            C3_LOAD_HALO(hv, rsx, nn, y0, x0)
            C3_LOAD_DZ(nn, y0, x0)
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const v8 bfr = WP::tr_pair(dzt + dzo + ks * 32 * WP::DZS, dzt + dzo + (ks * 32 + 16) * WP::DZS);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                v8 afr;
                const char* pb = hb + (2 * ks * 18 * 3) * 2 + (moff[mt] >= 0 ? moff[mt] : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    afr[j] = moff[mt] >= 0 ? *reinterpret_cast<const E*>(pb + joff[j]) : (E)0.f;
                acc[mt] = Half16<E>::mfma(afr, bfr, acc[mt]);
            }
        }
        lds_barrier();                              // the dz tile is single buffered
        buf ^= 1;
    }
    // D[row = K index][col = channel]: lane holds rows (lane >> 4) * 4 + r of column lane & 15
    float* slab = a.ws + (size_t)blockIdx.x * (27 * 64);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * mt + grp * 4 + r;
            if (k < 27) slab[k * 64 + wave * 16 + (lane & 15)] = acc[mt][r];
        }
#undef C3_LOAD_DZ
}
#undef C3_LOAD_HALO
#undef C3_STORE_HALO
#undef C3_TILE

// dw_hwio[tap][ci][co] = sum over splits of the slabs, fixed order (deterministic).
// block = 32 float4 column vectors (128 consecutive (tap, ci, co) outputs) x 8 split groups; a thread walks its splits
// four at a time (four independent 16-byte loads in flight), the eight groups are folded through LDS in a fixed order.
// (cout % 4 == 0 and the slab rows are 64 floats, so a float4 never straddles a row or a channel block.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int nsplit, int ncib, int ncob, int ck, int cin_real,
                                                           int cout, int accumulate = 0) {
    __shared__ float4 sm[8][32];
    const int tid = threadIdx.x, cl = tid & 31, rg = tid >> 5;
    const int idx = (blockIdx.x * 32 + cl) * 4;
    const int total = 9 * cin_real * cout;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total) {
        const int co = idx % cout;
        const int r = idx / cout;
        const int ci = r % cin_real;
        const int tap = r / cin_real;
        const int cib = ci / ck, cil = ci % ck, cob = co / BN, col = co % BN;
        const size_t base = (((size_t)cib * ncob + cob) * 9 + tap) * (ck * BN) + cil * BN + col;
        const size_t stride = (size_t)ncib * ncob * 9 * ck * BN;
        int sp = rg;
        for (; sp + 24 < nsplit; sp += 32) {
            const float4 v0 = *reinterpret_cast<const float4*>(ws + base + (size_t)sp * stride);
            const float4 v1 = *reinterpret_cast<const float4*>(ws + base + (size_t)(sp + 8) * stride);
            const float4 v2 = *reinterpret_cast<const float4*>(ws + base + (size_t)(sp + 16) * stride);
            const float4 v3 = *reinterpret_cast<const float4*>(ws + base + (size_t)(sp + 24) * stride);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; sp < nsplit; sp += 8) {
            const float4 v = *reinterpret_cast<const float4*>(ws + base + (size_t)sp * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && idx < total) {
        float4 t = sm[0][cl];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = sm[r][cl]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float4* dst = reinterpret_cast<float4*>(dw + idx);
        if (accumulate) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }   // later image runs add
        *dst = t;
    }
}

// dbias[c] = sum over the MFMA-wave rows of the workgroups that own output block c / 64 (ws_order: workgroup b owns
// block (b >> 3) % nblk, i.e. b = ((q * nblk + blk) << 3) + x).  One block of 1024 threads per output block: 64 channels
// x 16 row groups; a group walks its rows in ascending order, eight independent loads in flight (a dependent chain of
// 128 L2 round trips took 53 us), and the sixteen partial sums are folded in a fixed order: deterministic.
__global__ __launch_bounds__(1024) void mask_dbias_reduce_kernel(const float* __restrict__ part, int nwg, int nblk, int cy1,
                                                                 float* __restrict__ dbias) {
    __shared__ float sm[16][BN];
    const int blk = blockIdx.x, col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int nrows = nwg / nblk * 4;                  // nblk divides nwg / 8 (launcher)
    float s = 0.f;
#pragma unroll 8
    for (int k = rg; k < nrows; k += 16) {
        const int wv = k & 3, x = (k >> 2) & 7, q = k >> 5;
        const int b = ((q * nblk + blk) << 3) + x;
        s += part[((size_t)b * 4 + wv) * BN + col];
    }
    sm[rg][col] = s;
    __syncthreads();
    if (rg == 0 && blk * BN + col < cy1) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += sm[r][col];
        dbias[blk * BN + col] = t;
    }
}

// dgamma / dbeta / dbias of ad_conv3x3_dgrad_ln_bwd: part[workgroup][wave][3][64]; block q sums quantity q over the
// 4 * nwg rows: 16 row groups in ascending order with eight loads in flight, folded in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void lnb_reduce_kernel(const float* __restrict__ part, int nrows, float* __restrict__ o0,
                                                          float* __restrict__ o1, float* __restrict__ o2) {
    __shared__ float sm[16][BN];
    const int q = blockIdx.x, col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    float s = 0.f;
#pragma unroll 8
    for (int r = rg; r < nrows; r += 16) s += part[(size_t)r * (3 * BN) + q * BN + col];
    sm[rg][col] = s;
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += sm[r][col];
        float* dst = q == 0 ? o0 : q == 1 ? o1 : o2;
        if (dst) dst[col] = t;
    }
}

// ------------------------------------------------------------------ weight packing
// w_fwd[tap][kc][co][kv] = W[tap][kc*KV+kv][co]; w_dgrad[tap'][kc][ci][kv] = W[8-tap'][ci][kc*KV+kv].
// The output-channel dimension of each pack (co of w_fwd, ci of w_dgrad) is padded with zeros to whole 64-channel
// blocks (ad_conv3x3_pack_elems), so a layer with 32 output channels is one block whose upper half is never stored.
static __host__ __device__ inline int pad64(int c) { return (c + BN - 1) / BN * BN; }

// One element per thread (a 16-byte-vector-per-thread variant with coalesced reads for the forward pack measured
// slower, 0.285 vs 0.179 ms for the 24 layers of K2': the dgrad half then reads one 32-byte sector per lane).
template <typename T>
__device__ __forceinline__ void pack_element(const float* __restrict__ w, int cin, int cout, int cin_pad, T* __restrict__ wf,
                                             T* __restrict__ wd, int e) {
    constexpr int KV = 16 / (int)sizeof(T);
    const int cout_p = (cout + BN - 1) / BN * BN, cin_o = (cin_pad + BN - 1) / BN * BN;
    const int total_f = 9 * cin_pad * cout_p;
    if (e < total_f) {
        int kv = e % KV, r = e / KV;
        int co = r % cout_p; r /= cout_p;
        int kc = r % (cin_pad / KV), tap = r / (cin_pad / KV);
        int ci = kc * KV + kv;
        wf[e] = (T)(ci < cin && co < cout ? w[((size_t)tap * cin + ci) * cout + co] : 0.f);
    } else {
        const int i = e - total_f;
        int kv = i % KV, r = i / KV;
        int ci = r % cin_o; r /= cin_o;
        int kc = r % (cout / KV), tap = r / (cout / KV);
        int co = kc * KV + kv;
        wd[i] = (T)(ci < cin ? w[((size_t)(8 - tap) * cin + ci) * cout + co] : 0.f);
    }
}

template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, int cin, int cout, int cin_pad, T* __restrict__ wf,
                            T* __restrict__ wd) {
    const int total = 9 * cin_pad * pad64(cout) + (wd ? 9 * cout * pad64(cin_pad) : 0);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x)
        pack_element<T>(w, cin, cout, cin_pad, wf, wd, e);
}

// All layers of a model in one launch.  A block repacks one 64 x 64 (input channel x output channel) tile of one tap
// of one layer through LDS: the fp32 master weights are read in 256-byte rows, both operand layouts are written as
// consecutive 16-byte vectors (the element-wise kernel above reads with a stride of cout floats: 1.65 TB/s of traffic for
// the 24 layers of K2').  Block b serves tile b - first_block of the job whose block range contains it.
struct PackJob {
    const float* w; void* wf; void* wd;     // wd may be NULL
    int cin, cout, cin_pad, first_block;
};
static __host__ __device__ inline int pack_job_blocks(int cin_pad, int cout) { return 9 * (pad64(cin_pad) / BN) * (pad64(cout) / BN); }

template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
    constexpr int KV = 16 / (int)sizeof(T);
    __shared__ float tile[64][65];
    int jn = 0;
    while (jn + 1 < njobs && jobs[jn + 1].first_block <= (int)blockIdx.x) ++jn;      // block-uniform scan, <= 64 jobs
    const PackJob j = jobs[jn];
    const int cout_p = pad64(j.cout), cin_o = pad64(j.cin_pad);
    const int ncot = cout_p / BN, ncit = cin_o / BN;
    const int b = (int)blockIdx.x - j.first_block;
    const int cot = b % ncot, cit = (b / ncot) % ncit, tap = b / (ncot * ncit);
    const int tid = threadIdx.x;
    for (int k = tid; k < 64 * 64; k += 256) {
        const int r = k >> 6, c = k & 63;
        const int ci = cit * 64 + r, co = cot * 64 + c;
        tile[r][c] = ci < j.cin && co < j.cout ? j.w[((size_t)tap * j.cin + ci) * j.cout + co] : 0.f;
    }
    __syncthreads();
    T* wf = (T*)j.wf;
    T* wd = (T*)j.wd;
    for (int k = tid; k < (64 / KV) * 64; k += 256) {
        const int lo6 = k & 63, kcl = k >> 6;
        // forward operand: KV input channels of output channel co (consecutive threads = consecutive co)
        const int ci0 = cit * 64 + kcl * KV;
        if (ci0 < j.cin_pad) {
            float f[KV];
#pragma unroll
            for (int kv = 0; kv < KV; ++kv) f[kv] = tile[kcl * KV + kv][lo6];
            Vec16<T> st;
            st.from_f32(f);
            st.store(wf + (((size_t)tap * (j.cin_pad / KV) + ci0 / KV) * cout_p + cot * 64 + lo6) * KV);
        }
        // dgrad operand: KV output channels of input channel ci, taps rotated by 180 degrees
        const int co0 = cot * 64 + kcl * KV;
        if (wd && co0 < j.cout) {
            float f[KV];
#pragma unroll
            for (int kv = 0; kv < KV; ++kv) f[kv] = tile[lo6][kcl * KV + kv];
            Vec16<T> st;
            st.from_f32(f);
            st.store(wd + (((size_t)(8 - tap) * (j.cout / KV) + co0 / KV) * cin_o + cit * 64 + lo6) * KV);
        }
    }
}

template <typename K>
static void allow_big_lds(K kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Which wave-specialised bf16 forward kernel takes a launch: 0 = none (generic kernel), 1 = weights resident
// (Cin = 64, >= 4 items per CU), 2 = streamed weights (even number of 32-channel chunks, >= 1 item per CU).
// Both need 16x16 tiles, 256 % (cout / 64) == 0 and tensors below 2 GiB (32-bit buffer offsets); the fused
// LayerNorm epilogue additionally needs cout == 64 (all channels of a pixel inside one wave).
static int fwd_ws_kind(int n, int h, int w, int c1, int c2, int cout, bool ln) {
    Geo g;
    pick_geo(n, h, w, &g);
    const bool geo16 = g.lti == 0 && g.lth == 4 && g.ltw == 4 && g.ph && g.pw;
    if (cout % BN) return 0;                        // a ragged last output block: generic kernel
    const int nblk = cout / BN, nch = (c1 + c2) / PolBF16::CK;
    const long long nitems = (long long)g.tiles_x * g.tiles_y * g.tiles_i * nblk;
    const long long npix = (long long)n * h * w;
    const long long widest = c1 > c2 ? (c1 > cout ? c1 : cout) : (c2 > cout ? c2 : cout);
    const bool fits = npix * widest * 2 <= WR_MAX_BYTES && 9LL * (c1 + c2) * cout * 2 <= WR_MAX_BYTES;
    if (!geo16 || nblk == 0 || (NUM_CU / 8) % nblk != 0 || !fits || (ln && nblk != 1)) return 0;
    if (nch == 2 && nitems >= 4 * NUM_CU) return 1;
    if (nch >= 2 && nch % 2 == 0 && nitems >= NUM_CU) return 2;
    return 0;
}

// Tensors of 2 GiB and more do not fit the 32-bit buffer offsets of the wave-specialised kernels; a batch that is
// only too large as a whole is cut into runs of `chunk` images (NHWC: a run is a pointer offset) when every run
// still qualifies for a specialised kernel.  chunk == n: a single launch.
static int images_per_launch(int n, int h, int w, int c1, int c2, int cout, bool ln, bool wgrad) {
    const long long widest = c1 > c2 ? (c1 > cout ? c1 : cout) : (c2 > cout ? c2 : cout);
    const long long per_img = (long long)h * w * widest * 2;
    if ((long long)n * per_img <= WR_MAX_BYTES || per_img > WR_MAX_BYTES || n < 2) return n;
    const int cap = (int)(WR_MAX_BYTES / per_img);
    const int runs = (n + cap - 1) / cap;
    const int chunk = (n + runs - 1) / runs;
    const int last = n - (runs - 1) * chunk;
    (void)wgrad;
    if (last < 1) return n;
    if (!wgrad && (!fwd_ws_kind(chunk, h, w, c1, c2, cout, ln) || !fwd_ws_kind(last, h, w, c1, c2, cout, ln))) return n;
    return chunk;
}

template <typename P>
int launch_fwd(ConvArgs a, void* ws, size_t ws_bytes, hipStream_t s);

// streamed-weights kernel, bias / bias + ReLU epilogue: the image mosaic when it needs fewer tiles and still gives every
// workgroup an item
static bool fwd_mosaic(int n, int h, int w, int nblk, Geo* g) {
    Geo mg = *g;
    if (!plan_mosaic(n, h, w, NUM_CU / nblk, &mg) || (long long)mg.tiles_x * mg.tiles_y * nblk < NUM_CU) return false;
    *g = mg;
    return true;
}

// conv3x3_map1_kernel: 1x1 maps, four chunks per step group, the concat boundary on a chunk, the output split on 16-channel tiles
static bool map1_ok(int n, int h, int w, int c1, int c2, int cout, int cy1) {
    if (ad_option(AD_OPT_NO_MAP1)) return false;                      // A/B switch (ad_set_option)
    const long long in_bytes = (long long)n * (c1 > c2 ? c1 : c2) * 2;        // 32-bit offsets inside the kernel
    // small batches of 1x1 maps only (a bottleneck level): every workgroup re-reads its 64 images' activations from
    // L2, which is the right trade for a few hundred images and the wrong one for the 1x1 GEMM over 262 144 "images"
    // that a Conv2DTranspose becomes (seg_model: those stay on the generic kernel, weights staged in LDS per tile)
    return h == 1 && w == 1 && n >= 1 && n <= 1024 && in_bytes <= WR_MAX_BYTES && (c1 + c2) % 128 == 0 && c1 % 32 == 0 &&
           c1 + c2 <= 8192 && cout % 16 == 0 && cout <= 8192 && cy1 % 16 == 0;
}

// conv3x3_map4_kernel: 4x4 maps, whole 128-channel phases, the concat boundary and the output split on 16-channel tiles
static bool map4_ok(int n, int h, int w, int c1, int c2, int cout, int cy1) {
    if (ad_option(AD_OPT_NO_MAP4)) return false;                      // A/B switch (ad_set_option)
    const long long in_bytes = (long long)n * 16 * (c1 > c2 ? c1 : c2) * 2;   // 32-bit offsets inside the kernel
    return h == 4 && w == 4 && n >= 1 && in_bytes <= WR_MAX_BYTES && (c1 + c2) % 128 == 0 && c1 % 128 == 0 && c1 + c2 <= 8192 &&
           cout % 16 == 0 && cout <= 8192 && cy1 % 16 == 0;
}

template <typename P>
int launch_fwd_runs(ConvArgs a, void* ws, size_t ws_bytes, hipStream_t s) {
    int chunk = a.n;
    if constexpr (sizeof(typename P::T) == 2)
        chunk = images_per_launch(a.n, a.h, a.w, a.c1, a.c2, a.cout_real, a.epilogue == AD_EPI_LN_RELU || a.epilogue == AD_EPI_LN_STATS || a.epilogue == AD_EPI_LN_ACT, false);
    if (chunk >= a.n) return launch_fwd<P>(a, ws, ws_bytes, s);
    constexpr size_t TSZ = sizeof(typename P::T);
    for (int i0 = 0; i0 < a.n; i0 += chunk) {
        ConvArgs b = a;
        const size_t pix0 = (size_t)i0 * a.h * a.w;
        b.n = a.n - i0 < chunk ? a.n - i0 : chunk;
        b.x1 = a.x1 + pix0 * a.c1 * TSZ;
        if (a.x2) b.x2 = a.x2 + pix0 * a.c2 * TSZ;
        b.y1 = a.y1 + pix0 * a.cy1 * TSZ;
        if (a.y2) b.y2 = a.y2 + pix0 * (a.cout_real - a.cy1) * TSZ;
        if (a.a_out) b.a_out = a.a_out + pix0 * a.cout_real * TSZ;
        if (a.ln_mean) { b.ln_mean = a.ln_mean + pix0; b.ln_rstd = a.ln_rstd + pix0; }
        pick_geo(b.n, b.h, b.w, &b.g);
        b.ntiles = b.g.tiles_x * b.g.tiles_y * b.g.tiles_i;
        const int rc = launch_fwd<P>(b, ws, ws_bytes, s);
        if (rc) return rc;
    }
    return AD_OK;
}

template <typename P>
int launch_fwd(ConvArgs a, void* ws, size_t ws_bytes, hipStream_t s) {
    Geo& g = a.g;
    const int xs = g.NPH * 4 <= 6 * FT ? 6 : 16;
    g.NPHP = xs * FT / 4;
    const bool halo = g.ph && g.pw;
    if (!halo && (g.ph || g.pw || xs != 6))
        return ad_set_error(AD_ERR_ARG, "conv3x3_fwd: feature maps with exactly one unit extent (%dx%d) are not supported", a.h, a.w);
    const int nblk = a.cout / BN;
    const int nch = (a.c1 + a.c2) / P::CK;
    const int nitems = a.ntiles * nblk;
    static std::atomic<unsigned long long> attr_set{0};
    if (ad_first_on_device(attr_set)) {
        allow_big_lds(conv3x3_fwd_kernel<P, 6, true>);
        allow_big_lds(conv3x3_fwd_kernel<P, 16, true>);
        allow_big_lds(conv3x3_fwd_kernel<P, 6, false>);
        if constexpr (sizeof(typename P::T) == 2) {
            allow_big_lds(conv3x3_fwd_wres_kernel<P, 0>);
            allow_big_lds(conv3x3_fwd_wres_kernel<P, 1>);
            allow_big_lds(conv3x3_fwd_wres_kernel<P, 2>);
            allow_big_lds(conv3x3_fwd_ws_kernel<P, 0>);
            allow_big_lds(conv3x3_fwd_ws_kernel<P, 1>);
            allow_big_lds(conv3x3_fwd_ws_kernel<P, 2>);
            allow_big_lds(conv3x3_fwd_wres_kernel<P, 3>);
            allow_big_lds(conv3x3_fwd_wres_kernel<P, 4>);
            allow_big_lds(conv3x3_fwd_wres_kernel<P, 5>);
        }
    }
    if constexpr (sizeof(typename P::T) == 2) {
        a.ksplit = 1; a.slab = nullptr;
        if (map4_ok(a.n, a.h, a.w, a.c1, a.c2, a.cout_real, a.cy1) && a.epilogue <= AD_EPI_RELU) {
            static std::atomic<unsigned long long> m4_attr{0};
            if (ad_first_on_device(m4_attr)) {
                allow_big_lds(conv3x3_map4_kernel<P, 2>);
                allow_big_lds(conv3x3_map4_kernel<P, 4>);
            }
            const int mblk = (a.n + M4_IMG - 1) / M4_IMG;
            // 64-channel blocks keep the LDS and L1 traffic per MFMA lowest; 32-channel blocks when that leaves CUs idle
            if (mblk * ((a.cout_real + 63) / 64) >= 200 || a.cout_real % 32)
                conv3x3_map4_kernel<P, 4><<<mblk * ((a.cout_real + 63) / 64), M4_T, M4_LDS, s>>>(a);
            else
                conv3x3_map4_kernel<P, 2><<<mblk * ((a.cout_real + 31) / 32), M4_T, M4_LDS, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_map4");
            return AD_OK;
        }
        if (map1_ok(a.n, a.h, a.w, a.c1, a.c2, a.cout_real, a.cy1) && a.epilogue <= AD_EPI_RELU) {
            const int mblk = (a.n + 63) / 64;
            conv3x3_map1_kernel<P, 2><<<mblk * ((a.cout_real + 31) / 32), M1_T, 0, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_map1");
            return AD_OK;
        }
        int kind = fwd_ws_kind(a.n, a.h, a.w, a.c1, a.c2, a.cout_real, a.epilogue == AD_EPI_LN_RELU || a.epilogue == AD_EPI_LN_STATS || a.epilogue == AD_EPI_LN_ACT);
        if (a.cy1 % BN && a.cy1 != a.cout_real) kind = 0;          // a split inside a 64-channel block: generic kernel
        // Two output tensors whose block counts each suit the XCD-aware work order although their sum does not (the
        // dgrad of a Concatenate of 2 nf + nf channels: 3, 6, 12, 24 blocks): one launch per output on a slice of the
        // pack's output blocks (wcout).  The input is read twice; the generic kernel it replaces runs at 0.25 - 0.35 of peak.
        if (kind == 0 && a.y2 && a.wcout == 0 && a.epilogue <= AD_EPI_RELU && a.cy1 % BN == 0 && a.cout_real % BN == 0 &&
            fwd_ws_kind(a.n, a.h, a.w, a.c1, a.c2, a.cy1, false) && fwd_ws_kind(a.n, a.h, a.w, a.c1, a.c2, a.cout_real - a.cy1, false)) {
            ConvArgs b = a;
            b.wcout = a.cout; b.y2 = nullptr;
            b.cout = b.cout_real = a.cy1;
            int rc = launch_fwd<P>(b, ws, ws_bytes, s);
            if (rc) return rc;
            b.cout = b.cout_real = b.cy1 = a.cout_real - a.cy1;
            b.y1 = a.y2;
            b.wp = a.wp + (size_t)a.cy1 * 16;                      // [tap][Cin / KV][Cout][KV]: 16 bytes per output channel
            if (a.bias) b.bias = a.bias + a.cy1;
            return launch_fwd<P>(b, ws, ws_bytes, s);
        }
        // the ReLU-grad mask epilogue exists for the weights-resident kernel only (the streamed-weights variant would
        // spill: 256 registers + scratch); ad_conv3x3_dgrad_relu_is_fused says so to the caller
        if (a.epilogue == AD_EPI_MASK) {
            if (kind != 1) return AD_ERR_UNFUSED;
            conv3x3_fwd_wres_kernel<P, 3><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_fwd_wres (relu-grad)");
            return AD_OK;
        }
        if (a.epilogue == AD_EPI_LNBWD) {
            if (kind != 1 || a.cout_real != BN) return AD_ERR_UNFUSED;
            conv3x3_fwd_wres_kernel<P, 4><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_fwd_wres (layernorm-bwd)");
            return AD_OK;
        }
        if (a.epilogue == AD_EPI_LN_ACT) {
            if (kind == 0 || a.cout_real != BN) return AD_ERR_UNFUSED;
            static std::atomic<unsigned long long> act_attr{0};
            if (ad_first_on_device(act_attr)) {
                allow_big_lds(conv3x3_fwd_wres_kernel<P, 6>);
                allow_big_lds(conv3x3_fwd_ws_kernel<P, 6>);
            }
            if (kind == 1) conv3x3_fwd_wres_kernel<P, 6><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
            else conv3x3_fwd_ws_kernel<P, 6><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_fwd (layernorm + relu, activation only)");
            return AD_OK;
        }
        if (a.epilogue == AD_EPI_LN_STATS) {
            if (kind != 1 || a.cout_real != BN) return AD_ERR_UNFUSED;
            conv3x3_fwd_wres_kernel<P, 5><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_fwd_wres (layernorm statistics)");
            return AD_OK;
        }
#define AD_WS_LAUNCH(KERN, NAME)                                                                     \
    {                                                                                                \
        if (a.epilogue == AD_EPI_LN_RELU) KERN<P, 2><<<NUM_CU, WR_T, WR_LDS, s>>>(a);                \
        else if (a.epilogue == AD_EPI_RELU) KERN<P, 1><<<NUM_CU, WR_T, WR_LDS, s>>>(a);              \
        else KERN<P, 0><<<NUM_CU, WR_T, WR_LDS, s>>>(a);                                             \
        AD_LAUNCH_CHECK(NAME);                                                                       \
        return AD_OK;                                                                                \
    }
        if (kind == 1) AD_WS_LAUNCH(conv3x3_fwd_wres_kernel, "conv3x3_fwd_wres")
        if (kind == 2 && a.epilogue <= AD_EPI_RELU) {
            // maps whose extent is not a multiple of 16: all images as one mosaic when that needs fewer tiles (Geo)
            Geo mg = a.g;
            if (fwd_mosaic(a.n, a.h, a.w, nblk, &mg)) {
                static std::atomic<unsigned long long> mos_attr{0};
                if (ad_first_on_device(mos_attr)) {
                    allow_big_lds(conv3x3_fwd_ws_kernel<P, 0, true>);
                    allow_big_lds(conv3x3_fwd_ws_kernel<P, 1, true>);
                }
                a.g = mg;
                a.ntiles = mg.tiles_x * mg.tiles_y;
                if (a.epilogue == AD_EPI_RELU) conv3x3_fwd_ws_kernel<P, 1, true><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
                else conv3x3_fwd_ws_kernel<P, 0, true><<<NUM_CU, WR_T, WR_LDS, s>>>(a);
                AD_LAUNCH_CHECK("conv3x3_fwd_ws (mosaic)");
                return AD_OK;
            }
        }
        if (kind == 2) AD_WS_LAUNCH(conv3x3_fwd_ws_kernel, "conv3x3_fwd_ws")
#undef AD_WS_LAUNCH
    }
    if (a.epilogue == AD_EPI_LN_RELU || a.epilogue == AD_EPI_MASK || a.epilogue == AD_EPI_LNBWD || a.epilogue == AD_EPI_LN_STATS ||
        a.epilogue == AD_EPI_LN_ACT)
        return AD_ERR_UNFUSED;     // the caller runs two launches
    size_t lds = 2 * (size_t)g.NPHP * 4 + 2 * BN * 4 + (((size_t)g.NPH * PIXB + 15) & ~15) + (size_t)FWS * FT * 16;
    if (lds > 160 * 1024) return ad_set_error(AD_ERR_ARG, "conv3x3_fwd: LDS %zu too large", lds);
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t npix = (int64_t)a.n * a.h * a.w;
    a.ksplit = pick_ksplit(nitems, nch);
    if (a.ksplit > 1 && (ws == nullptr || ws_bytes < (size_t)a.ksplit * npix * a.cout * sizeof(float))) a.ksplit = 1;
    a.slab = (float*)ws;
    int grid = NUM_CU * per_cu;
    if (grid > nitems * a.ksplit) grid = nitems * a.ksplit;
    if (!halo)
        conv3x3_fwd_kernel<P, 6, false><<<grid, FT, lds, s>>>(a);
    else if (xs == 6)
        conv3x3_fwd_kernel<P, 6, true><<<grid, FT, lds, s>>>(a);
    else
        conv3x3_fwd_kernel<P, 16, true><<<grid, FT, lds, s>>>(a);
    AD_LAUNCH_CHECK("conv3x3_fwd");
    if (a.ksplit > 1) {
        typedef typename P::T T;
        const int64_t total = npix * (a.cout / 4);
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        splitk_finalize_kernel<T><<<blocks, 256, 0, s>>>(a.slab, a.bias, (T*)a.y1, (T*)a.y2, a.cy1, npix, a.cout, a.ksplit,
                                                         a.epilogue == AD_EPI_RELU, a.cout_real);
        AD_LAUNCH_CHECK("splitk_finalize");
    }
    return AD_OK;
}

struct WgradPlan {
    Geo g;
    int ntiles, nsplit, tiles_per_split, ncib, ncob, ck;
    bool specialised;          // conv3x3_wgrad_ws_kernel (bf16, 16x16 tiles, 64-channel input blocks)
    size_t ws_bytes;
};

static void plan_wgrad(int n, int h, int w, int c1, int c2, int cout, int dtype, WgradPlan* p) {
    const int cin = c1 + c2;
    pick_geo(n, h, w, &p->g, ad_is_half(dtype) ? 1 << 30 : 576);
    p->ntiles = p->g.tiles_x * p->g.tiles_y * p->g.tiles_i;
    p->ncob = (cout + BN - 1) / BN;
    const long long widest = (long long)n * h * w * (c1 > c2 ? (c1 > cout ? c1 : cout) : (c2 > cout ? c2 : cout)) * 2;
    p->specialised = ad_is_half(dtype) && p->g.lti == 0 && p->g.lth == 4 && p->g.ltw == 4 && cin % 64 == 0 && cout % BN == 0 &&
                     c1 % 32 == 0 && widest <= WR_MAX_BYTES && p->ntiles >= 4 * NUM_CU / ((cin / 64) * p->ncob);
    if (p->specialised) {      // the image mosaic where it needs fewer tiles and still fills the chip (Geo)
        Geo mg = p->g;
        const int wgs = NUM_CU / ((cin / 64) * p->ncob);          // workgroups that share the tiles of one channel-block pair
        if (plan_mosaic(n, h, w, wgs < 1 ? 1 : wgs, &mg) && mg.tiles_x * mg.tiles_y >= 2 * wgs) {
            p->g = mg;
            p->ntiles = mg.tiles_x * mg.tiles_y;
        }
    }
    p->ck = p->specialised ? 64 : ad_is_half(dtype) ? PolBF16::CK : PolF32::CK;
    p->ncib = cin / p->ck;
    int want = ((p->specialised ? 1 : 2) * NUM_CU) / (p->ncib * p->ncob);   // workgroups per CU in total
    if (want < 1) want = 1;
    if (want > p->ntiles) want = p->ntiles;
    // tiny maps (<= 4 tiles): the channel blocks alone fill the chip when there are >= 128 of them; one split lets the
    // kernel write dw itself instead of round-tripping 4 x |dw| through the slab reduce
    if (!p->specialised && p->ntiles <= 4 && p->ncib * p->ncob >= NUM_CU / 2) want = 1;
    p->tiles_per_split = (p->ntiles + want - 1) / want;
    p->nsplit = (p->ntiles + p->tiles_per_split - 1) / p->tiles_per_split;
    p->ws_bytes = (size_t)p->nsplit * p->ncib * p->ncob * 9 * p->ck * BN * sizeof(float);
}

template <typename P>
int launch_wgrad(WgradArgs a, const WgradPlan& p, hipStream_t s) {
    typedef WgradPol<P> WP;
    Geo& g = a.g;
    const int xs = g.NPH * 4 <= 6 * 256 ? 6 : 16;
    g.NPHP = xs * 256 / 4;
    size_t lds = 2 * (size_t)g.NPHP * 4 + TM * 4 + (((size_t)g.NPH * PIXB + 15) & ~15) + (size_t)TM * WP::DZS;
    if (lds > 160 * 1024) return ad_set_error(AD_ERR_ARG, "conv3x3_wgrad: LDS %zu too large", lds);
    static std::atomic<unsigned long long> attr_set{0};
    if (ad_first_on_device(attr_set)) {
        allow_big_lds(conv3x3_wgrad_kernel<P, 6, true>);
        allow_big_lds(conv3x3_wgrad_kernel<P, 16, true>);
        allow_big_lds(conv3x3_wgrad_kernel<P, 6, false>);
    }
    dim3 grid(p.nsplit, p.ncib, p.ncob);
    if constexpr (sizeof(typename P::T) == 2) {
        if (p.specialised) {
            static std::atomic<unsigned long long> attr2{0};
            if (ad_first_on_device(attr2)) {
                allow_big_lds(conv3x3_wgrad_ws_kernel<typename P::T>);
                allow_big_lds(conv3x3_wgrad_ws_kernel<typename P::T, true>);
            }
            if (g.mos) conv3x3_wgrad_ws_kernel<typename P::T, true><<<grid, W2_T, W2_LDS, s>>>(a);
            else conv3x3_wgrad_ws_kernel<typename P::T><<<grid, W2_T, W2_LDS, s>>>(a);
            AD_LAUNCH_CHECK("conv3x3_wgrad_ws");
            return AD_OK;
        }
    }
    const bool halo = g.ph && g.pw;
    if (!halo && (g.ph || g.pw || xs != 6))
        return ad_set_error(AD_ERR_ARG, "conv3x3_wgrad: feature maps with exactly one unit extent (%dx%d) are not supported", a.h, a.w);
    if (!halo)
        conv3x3_wgrad_kernel<P, 6, false><<<grid, 256, lds, s>>>(a);
    else if (xs == 6)
        conv3x3_wgrad_kernel<P, 6, true><<<grid, 256, lds, s>>>(a);
    else
        conv3x3_wgrad_kernel<P, 16, true><<<grid, 256, lds, s>>>(a);
    AD_LAUNCH_CHECK("conv3x3_wgrad");
    return AD_OK;
}

static int launch_fwd_dtype(int dtype, const ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t s) {
    if (dtype == AD_BF16) return launch_fwd_runs<PolBF16>(a, ws, ws_bytes, s);
    if (dtype == AD_F16) return launch_fwd_runs<PolF16>(a, ws, ws_bytes, s);
    return launch_fwd_runs<PolF32>(a, ws, ws_bytes, s);
}

}  // namespace

static unsigned long long* g_dbg = nullptr;
#ifdef AD_STAMP
extern "C" void ad_dbg_set_stamp_buffer(void* p) { g_dbg = (unsigned long long*)p; }
#endif
#ifdef AD_CLOCK
// host copy of the (cycles, 10-ns ticks) pairs of the last stamped launches; `reset` zeroes them
extern "C" int ad_dbg_clock_read(unsigned long long* host, int pairs, int reset) {
    if (pairs > 4096) pairs = 4096;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ad_clock), (size_t)pairs * 16) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zeros[2 * 4096];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ad_clock), zeros, sizeof(zeros)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int ad_conv3x3_pack(const float* w_hwio, int cin, int cout, int cin_pad, void* w_fwd, void* w_dgrad,
                               int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_conv3x3_pack: bad dtype %d", dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(cin > 0 && cout > 0 && cin_pad >= cin && cin_pad % gran == 0,
               "ad_conv3x3_pack: cin=%d cin_pad=%d must be a multiple of %d", cin, cin_pad, gran);
    AD_REQUIRE(w_fwd != nullptr, "ad_conv3x3_pack: w_fwd is NULL");
    if (w_dgrad) AD_REQUIRE(cout % gran == 0, "ad_conv3x3_pack: dgrad layout needs cout %% %d == 0 (got %d)", gran, cout);
    hipStream_t s = (hipStream_t)stream;
    int total = 9 * cin_pad * pad64(cout) + (w_dgrad ? 9 * cout * pad64(cin_pad) : 0);
    int blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    AD_DISPATCH_DTYPE(dtype, T_, pack_kernel<T_><<<blocks, 256, 0, s>>>(w_hwio, cin, cout, cin_pad, (T_*)w_fwd, (T_*)w_dgrad);)
    AD_LAUNCH_CHECK("ad_conv3x3_pack");
    return AD_OK;
}

extern "C" size_t ad_conv3x3_pack_job_bytes(void) { return sizeof(PackJob); }

extern "C" size_t ad_conv3x3_pack_elems(int cin_pad, int cout, int dgrad) {
    return dgrad ? (size_t)9 * cout * pad64(cin_pad) : (size_t)9 * cin_pad * pad64(cout);
}

extern "C" int ad_conv3x3_pack_job_blocks(int cin_pad, int cout) { return pack_job_blocks(cin_pad, cout); }

extern "C" int ad_conv3x3_pack_batch(const void* jobs_dev, int njobs, int nblocks, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_conv3x3_pack_batch: bad dtype %d", dtype);
    AD_REQUIRE(jobs_dev != nullptr && njobs > 0 && njobs <= 64 && nblocks > 0,
               "ad_conv3x3_pack_batch: bad job table (%d jobs, %d blocks)", njobs, nblocks);
    AD_DISPATCH_DTYPE(dtype, T_, pack_batch_kernel<T_><<<nblocks, 256, 0, (hipStream_t)stream>>>((const PackJob*)jobs_dev, njobs);)
    AD_LAUNCH_CHECK("ad_conv3x3_pack_batch");
    return AD_OK;
}

extern "C" int ad_conv3x3_fwd(const void* x1, int c1, const void* x2, int c2, const void* w_packed, const float* bias,
                              void* y1, int cy1, void* y2, int n, int h, int w, int cout, int epilogue, void* ws,
                              size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_conv3x3_fwd: bad dtype %d", dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0, "ad_conv3x3_fwd: bad shape n=%d h=%d w=%d", n, h, w);
    AD_REQUIRE((long)n * h * w < (1L << 31), "ad_conv3x3_fwd: more than 2^31 pixels");
    AD_REQUIRE(x1 && c1 > 0 && c1 % gran == 0, "ad_conv3x3_fwd: c1=%d must be a positive multiple of %d", c1, gran);
    AD_REQUIRE((x2 == nullptr) == (c2 == 0) && c2 % gran == 0, "ad_conv3x3_fwd: c2=%d / x2 mismatch", c2);
    AD_REQUIRE(cout > 0 && cout % 16 == 0, "ad_conv3x3_fwd: cout=%d must be a multiple of 16", cout);
    AD_REQUIRE(cy1 > 0 && cy1 <= cout && cy1 % 16 == 0 && ((cy1 == cout) == (y2 == nullptr)),
               "ad_conv3x3_fwd: bad output split cy1=%d cout=%d", cy1, cout);
    AD_REQUIRE(epilogue == AD_EPI_NONE || epilogue == AD_EPI_RELU, "ad_conv3x3_fwd: bad epilogue %d", epilogue);
    ConvArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.c1 = c1; a.c2 = c2;
    a.wp = (const char*)w_packed; a.bias = bias;
    a.y1 = (char*)y1; a.y2 = (char*)y2; a.cy1 = cy1;
    a.n = n; a.h = h; a.w = w; a.cout = pad64(cout); a.cout_real = cout; a.epilogue = epilogue;
    a.dbg = g_dbg;
    a.ksplit = 1; a.slab = nullptr;
    a.ln_gamma = a.ln_beta = nullptr; a.ln_eps = 0.f; a.a_out = nullptr; a.ln_mean = a.ln_rstd = nullptr;
    a.mask1 = nullptr; a.dbias_part = nullptr; a.lnb_z = nullptr;
    pick_geo(n, h, w, &a.g);
    a.ntiles = a.g.tiles_x * a.g.tiles_y * a.g.tiles_i;
    hipStream_t s = (hipStream_t)stream;
    return launch_fwd_dtype(dtype, a, ws, ws_bytes, s);
}

extern "C" int ad_layernorm_relu_fwd(const void* z, const float* gamma, const float* beta, void* y, float* mean,
                                     float* rstd, int64_t npix, int c, float eps, int relu, int dtype, void* stream);

extern "C" int ad_conv3x3_ln_relu_is_fused(int n, int h, int w, int c1, int c2, int cout, int dtype) {
    if (!ad_is_half(dtype) || !pixels_ok(n, h, w) || c1 <= 0 || c2 < 0 || cout <= 0 || cout % BN) return 0;
    const int chunk = images_per_launch(n, h, w, c1, c2, cout, true, false);
    return fwd_ws_kind(chunk, h, w, c1, c2, cout, true) != 0;
}

// act == NULL in ad_conv3x3_ln_relu_fwd: the weights-resident kernel (Cin = 64 -> Cout = 64) on every image run
extern "C" int ad_conv3x3_ln_stats_is_fused(int n, int h, int w, int c1, int c2, int cout, int dtype) {
    if (!ad_is_half(dtype) || !pixels_ok(n, h, w) || c1 <= 0 || c2 < 0 || cout != BN) return 0;
    const int chunk = images_per_launch(n, h, w, c1, c2, cout, true, false);
    const int last = n - (n - 1) / chunk * chunk;
    return fwd_ws_kind(chunk, h, w, c1, c2, cout, true) == 1 && fwd_ws_kind(last, h, w, c1, c2, cout, true) == 1;
}

extern "C" int ad_conv3x3_ln_relu_fwd(const void* x1, int c1, const void* x2, int c2, const void* w_packed,
                                      const float* bias, const float* gamma, const float* beta, float eps, void* z,
                                      void* act, float* mean, float* rstd, int n, int h, int w, int cout, void* ws,
                                      size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_conv3x3_ln_relu_fwd: bad dtype %d", dtype);
    AD_REQUIRE(gamma && beta, "ad_conv3x3_ln_relu_fwd: NULL LayerNorm parameter");
    AD_REQUIRE(z || (act && ad_conv3x3_ln_relu_is_fused(n, h, w, c1, c2, cout, dtype)),
               "ad_conv3x3_ln_relu_fwd: z == NULL (activation only) has no kernel for n=%d %dx%d c1=%d c2=%d cout=%d dtype=%d "
               "(ask ad_conv3x3_ln_relu_is_fused first)", n, h, w, c1, c2, cout, dtype);
    AD_REQUIRE(!z || (mean && rstd), "ad_conv3x3_ln_relu_fwd: NULL statistics operand");
    AD_REQUIRE(act || ad_conv3x3_ln_stats_is_fused(n, h, w, c1, c2, cout, dtype),
               "ad_conv3x3_ln_relu_fwd: act == NULL (statistics only) has no kernel for n=%d %dx%d c1=%d c2=%d cout=%d dtype=%d "
               "(ask ad_conv3x3_ln_stats_is_fused first)", n, h, w, c1, c2, cout, dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && (long)n * h * w < (1L << 31), "ad_conv3x3_ln_relu_fwd: bad shape n=%d h=%d w=%d", n, h, w);
    AD_REQUIRE(x1 && c1 > 0 && c1 % gran == 0 && (x2 == nullptr) == (c2 == 0) && c2 % gran == 0,
               "ad_conv3x3_ln_relu_fwd: bad input channels c1=%d c2=%d", c1, c2);
    AD_REQUIRE(cout > 0 && cout % 16 == 0, "ad_conv3x3_ln_relu_fwd: cout=%d must be a multiple of 16", cout);
    ConvArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.c1 = c1; a.c2 = c2;
    a.wp = (const char*)w_packed; a.bias = bias;
    a.y1 = (char*)(z ? z : act); a.y2 = nullptr; a.cy1 = cout;       // (z == NULL: the activation is the launch's one output)
    a.n = n; a.h = h; a.w = w; a.cout = pad64(cout); a.cout_real = cout;
    a.epilogue = !z ? AD_EPI_LN_ACT : act ? AD_EPI_LN_RELU : AD_EPI_LN_STATS;
    a.dbg = g_dbg;
    a.ksplit = 1; a.slab = nullptr;
    a.ln_gamma = gamma; a.ln_beta = beta; a.ln_eps = eps;
    a.a_out = (char*)(z ? act : nullptr); a.ln_mean = mean; a.ln_rstd = rstd;
    a.mask1 = nullptr; a.dbias_part = nullptr; a.lnb_z = nullptr;
    pick_geo(n, h, w, &a.g);
    a.ntiles = a.g.tiles_x * a.g.tiles_y * a.g.tiles_i;
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_fwd_dtype(dtype, a, ws, ws_bytes, s);
    if (rc != AD_ERR_UNFUSED) return rc;
    AD_REQUIRE(act && z, "ad_conv3x3_ln_relu_fwd: act == NULL / z == NULL need the fused kernel");
    a.epilogue = AD_EPI_NONE;
    rc = launch_fwd_dtype(dtype, a, ws, ws_bytes, s);
    if (rc) return rc;
    return ad_layernorm_relu_fwd(z, gamma, beta, act, mean, rstd, (int64_t)n * h * w, cout, eps, 1, dtype, stream);
}

extern "C" int ad_conv3x3_mosaic(int n, int h, int w, int c1, int c2, int cout, int dtype, int wgrad) {
    if (!ad_is_half(dtype) || !pixels_ok(n, h, w) || c1 <= 0 || c2 < 0 || cout <= 0 || (c1 + c2) % ad_cin_granule(dtype)) return 0;
    // follows the launch path: a batch of 2 GiB and more runs as image chunks, each planned on its own (launch_fwd_runs, the run
    // loop of ad_conv3x3_wgrad); the answer describes the full-size runs of `chunk` images
    const int chunk = images_per_launch(n, h, w, c1, c2, cout, false, wgrad != 0);
    if (wgrad) {
        WgradPlan p;
        plan_wgrad(chunk, h, w, c1, c2, cout, dtype, &p);
        return p.specialised && p.g.mos ? p.g.mix : 0;
    }
    if (map4_ok(chunk, h, w, c1, c2, cout, cout) || map1_ok(chunk, h, w, c1, c2, cout, cout)) return 0;
    if (fwd_ws_kind(chunk, h, w, c1, c2, cout, false) != 2) return 0;
    Geo g;
    pick_geo(chunk, h, w, &g);
    return fwd_mosaic(chunk, h, w, pad64(cout) / BN, &g) ? g.mix : 0;        // the block count launch_fwd uses
}

extern "C" size_t ad_conv3x3_fwd_ws_bytes(int n, int h, int w, int cin, int cout, int dtype) {
    if (!pixels_ok(n, h, w) || cin <= 0 || cout <= 0 || !ad_dtype_ok(dtype)) return 0;
    cout = pad64(cout);
    Geo g;
    pick_geo(n, h, w, &g);
    const long long nitems = (long long)g.tiles_x * g.tiles_y * g.tiles_i * (cout / BN);      // (64-bit: a query may name any shape)
    if (nitems >= NUM_CU) return 0;                                                            // split-K only below one item per CU
    const int ks = pick_ksplit((int)nitems, cin / (ad_is_half(dtype) ? PolBF16::CK : PolF32::CK));
    return ks > 1 ? (size_t)ks * n * h * w * cout * sizeof(float) : 0;
}

// ---- dgrad with the producer's ReLU-grad fused (decoder: dgrad of conv_block's first conv -> gradient of the up-conv's
// ReLU output, train_adaptive_unet.py:259-262)
extern "C" int ad_conv3x3_dgrad_relu_is_fused(int n, int h, int w, int c1, int cout, int cy1, int dtype) {
    if (!ad_is_half(dtype) || !pixels_ok(n, h, w) || c1 <= 0 || cout <= 0 || cout % BN || cy1 <= 0 || cy1 % BN) return 0;
    if (images_per_launch(n, h, w, c1, 0, cout, false, false) < n) return 0;        // image runs: plain path
    return fwd_ws_kind(n, h, w, c1, 0, cout, false) == 1;      // weights-resident kernel (contraction over 64 channels)
}

extern "C" size_t ad_conv3x3_dgrad_relu_ws_bytes(void) { return (size_t)NUM_CU * 4 * BN * sizeof(float); }

extern "C" int ad_conv3x3_dgrad_relu(const void* dz, int c1, const void* w_dgrad, const void* relu_out, void* y1, int cy1,
                                     void* y2, float* dbias, int n, int h, int w, int cout, void* ws, size_t ws_bytes,
                                     int dtype, void* stream) {
    AD_REQUIRE(ad_conv3x3_dgrad_relu_is_fused(n, h, w, c1, cout, cy1, dtype),
               "ad_conv3x3_dgrad_relu: no fused kernel for n=%d %dx%d c1=%d cout=%d cy1=%d dtype=%d (ask _is_fused first)", n, h, w,
               c1, cout, cy1, dtype);
    AD_REQUIRE(dz && w_dgrad && relu_out && y1 && dbias && ((cy1 == cout) == (y2 == nullptr)), "ad_conv3x3_dgrad_relu: bad operands");
    if (!ws || ws_bytes < ad_conv3x3_dgrad_relu_ws_bytes())
        return ad_set_error(AD_ERR_WS, "ad_conv3x3_dgrad_relu: workspace %zu < %zu bytes", ws_bytes, ad_conv3x3_dgrad_relu_ws_bytes());
    ConvArgs a;
    a.x1 = (const char*)dz; a.x2 = nullptr; a.c1 = c1; a.c2 = 0;
    a.wp = (const char*)w_dgrad; a.bias = nullptr;
    a.y1 = (char*)y1; a.y2 = (char*)y2; a.cy1 = cy1;
    a.n = n; a.h = h; a.w = w; a.cout = cout; a.cout_real = cout; a.epilogue = AD_EPI_MASK;
    a.dbg = g_dbg;
    a.ksplit = 1; a.slab = nullptr;
    a.ln_gamma = a.ln_beta = nullptr; a.ln_eps = 0.f; a.a_out = nullptr; a.ln_mean = a.ln_rstd = nullptr;
    a.mask1 = (const char*)relu_out; a.dbias_part = (float*)ws; a.lnb_z = nullptr;
    pick_geo(n, h, w, &a.g);
    a.ntiles = a.g.tiles_x * a.g.tiles_y * a.g.tiles_i;
    hipStream_t s = (hipStream_t)stream;
    const int rc = launch_fwd_dtype(dtype, a, nullptr, 0, s);
    if (rc) return rc == AD_ERR_UNFUSED ? ad_set_error(AD_ERR_ARG, "ad_conv3x3_dgrad_relu: launch not specialised") : rc;
    mask_dbias_reduce_kernel<<<(cy1 + BN - 1) / BN, 1024, 0, s>>>((const float*)ws, NUM_CU, cout / BN, cy1, dbias);
    AD_LAUNCH_CHECK("mask_dbias_reduce");
    return AD_OK;
}

// ---- dgrad with the producer's LayerNorm + ReLU backward fused (conv_block's second conv, Super_resolution/code/
// train_adaptive_unet.py:200-210: its input is the first LayerNorm's activation)
extern "C" int ad_conv3x3_dgrad_ln_bwd_is_fused(int n, int h, int w, int c1, int cout, int dtype) {
    const bool off = ad_option(AD_OPT_NO_DGRAD_LN) != 0;                  // A/B switch (ad_set_option)
    if (off || !ad_is_half(dtype) || !pixels_ok(n, h, w) || c1 <= 0 || cout != BN) return 0;
    if (images_per_launch(n, h, w, c1, 0, cout, false, false) < n) return 0;        // image runs: plain path
    return fwd_ws_kind(n, h, w, c1, 0, cout, false) == 1;      // weights-resident kernel (contraction over 64 channels)
}

extern "C" size_t ad_conv3x3_dgrad_ln_bwd_ws_bytes(void) { return (size_t)NUM_CU * 4 * 3 * BN * sizeof(float); }

extern "C" int ad_conv3x3_dgrad_ln_bwd(const void* dz, int c1, const void* w_dgrad, const void* z_prev, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta, void* dz_prev,
                                       float* dgamma, float* dbeta, float* dbias, int n, int h, int w, int cout, void* ws,
                                       size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_conv3x3_dgrad_ln_bwd_is_fused(n, h, w, c1, cout, dtype),
               "ad_conv3x3_dgrad_ln_bwd: no fused kernel for n=%d %dx%d c1=%d cout=%d dtype=%d (ask _is_fused first)", n, h, w, c1,
               cout, dtype);
    AD_REQUIRE(dz && w_dgrad && z_prev && mean && rstd && gamma && beta && dz_prev, "ad_conv3x3_dgrad_ln_bwd: bad operands");
    if (!ws || ws_bytes < ad_conv3x3_dgrad_ln_bwd_ws_bytes())
        return ad_set_error(AD_ERR_WS, "ad_conv3x3_dgrad_ln_bwd: workspace %zu < %zu bytes", ws_bytes,
                            ad_conv3x3_dgrad_ln_bwd_ws_bytes());
    ConvArgs a;
    a.x1 = (const char*)dz; a.x2 = nullptr; a.c1 = c1; a.c2 = 0;
    a.wp = (const char*)w_dgrad; a.bias = nullptr;
    a.y1 = (char*)dz_prev; a.y2 = nullptr; a.cy1 = cout;
    a.n = n; a.h = h; a.w = w; a.cout = cout; a.cout_real = cout; a.epilogue = AD_EPI_LNBWD;
    a.dbg = g_dbg;
    a.ksplit = 1; a.slab = nullptr;
    a.ln_gamma = gamma; a.ln_beta = beta; a.ln_eps = 0.f; a.a_out = nullptr;
    a.ln_mean = const_cast<float*>(mean); a.ln_rstd = const_cast<float*>(rstd);
    a.mask1 = nullptr; a.dbias_part = (float*)ws; a.lnb_z = (const char*)z_prev;
    pick_geo(n, h, w, &a.g);
    a.ntiles = a.g.tiles_x * a.g.tiles_y * a.g.tiles_i;
    hipStream_t s = (hipStream_t)stream;
    const int rc = launch_fwd_dtype(dtype, a, nullptr, 0, s);
    if (rc) return rc == AD_ERR_UNFUSED ? ad_set_error(AD_ERR_ARG, "ad_conv3x3_dgrad_ln_bwd: launch not specialised") : rc;
    lnb_reduce_kernel<<<3, 1024, 0, s>>>((const float*)ws, NUM_CU * 4, dgamma, dbeta, dbias);
    AD_LAUNCH_CHECK("lnb_reduce");
    return AD_OK;
}

// ---- first layer (3 input channels, 64 output channels, bf16)
static bool c3_plan(int n, int h, int w, C3Args* a, int* grid) {
    if (n <= 0 || h <= 0 || w <= 0 || (long long)n * h * w * 128 > WR_MAX_BYTES) return false;
    a->n = n; a->h = h; a->w_img = w;
    a->tiles_x = (w + 15) / 16; a->tiles_y = (h + 15) / 16;
    a->ntiles = a->tiles_x * a->tiles_y * n;
    *grid = a->ntiles < 2 * NUM_CU ? a->ntiles : 2 * NUM_CU;
    return true;
}

extern "C" int ad_conv3x3_c3_supported(int n, int h, int w, int cout, int dtype) {
    C3Args a; int grid;
    return ad_is_half(dtype) && cout == 64 && c3_plan(n, h, w, &a, &grid);
}

extern "C" int ad_conv3x3_c3_ln_relu_fwd(const float* x, const float* w_hwio, const float* bias, const float* gamma,
                                         const float* beta, float eps, void* z, void* act, float* mean, float* rstd,
                                         int n, int h, int w, int dtype, void* stream) {
    AD_REQUIRE(ad_is_half(dtype), "ad_conv3x3_c3_ln_relu_fwd: 16-bit storage types only (dtype %d)", dtype);
    AD_REQUIRE(x && w_hwio && gamma && beta && act && (!z || (mean && rstd)), "ad_conv3x3_c3_ln_relu_fwd: NULL operand");
    C3Args a; int grid;
    AD_REQUIRE(c3_plan(n, h, w, &a, &grid), "ad_conv3x3_c3_ln_relu_fwd: unsupported shape n=%d h=%d w=%d", n, h, w);
    a.x = x; a.w = w_hwio; a.bias = bias; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.z = (char*)z; a.act = (char*)act; a.mean = mean; a.rstd = rstd; a.dz = nullptr; a.ws = nullptr;
    if (!z) {        // inference: the activation only (as ad_conv3x3_ln_relu_fwd with z == NULL)
        if (dtype == AD_BF16) conv3x3_c3_fwd_kernel<bf16_t, 1><<<grid, C3_T, 2 * C3_HB + 3 * BN * 4, (hipStream_t)stream>>>(a);
        else conv3x3_c3_fwd_kernel<f16_t, 1><<<grid, C3_T, 2 * C3_HB + 3 * BN * 4, (hipStream_t)stream>>>(a);
    } else if (dtype == AD_BF16) conv3x3_c3_fwd_kernel<bf16_t><<<grid, C3_T, 2 * C3_HB + 3 * BN * 4, (hipStream_t)stream>>>(a);
    else conv3x3_c3_fwd_kernel<f16_t><<<grid, C3_T, 2 * C3_HB + 3 * BN * 4, (hipStream_t)stream>>>(a);
    AD_LAUNCH_CHECK("ad_conv3x3_c3_ln_relu_fwd");
    return AD_OK;
}

extern "C" int ad_conv3x3_c3_fwd(const float* x, const float* w_hwio, const float* bias, void* z, int n, int h, int w, int dtype,
                                 void* stream) {
    AD_REQUIRE(ad_is_half(dtype), "ad_conv3x3_c3_fwd: 16-bit storage types only (dtype %d)", dtype);
    AD_REQUIRE(x && w_hwio && z, "ad_conv3x3_c3_fwd: NULL operand");
    C3Args a; int grid;
    AD_REQUIRE(c3_plan(n, h, w, &a, &grid), "ad_conv3x3_c3_fwd: unsupported shape n=%d h=%d w=%d", n, h, w);
    a.x = x; a.w = w_hwio; a.bias = bias; a.gamma = a.beta = nullptr; a.eps = 0.f;
    a.z = (char*)z; a.act = nullptr; a.mean = a.rstd = nullptr; a.dz = nullptr; a.ws = nullptr;
    if (dtype == AD_BF16) conv3x3_c3_fwd_kernel<bf16_t, 2><<<grid, C3_T, 2 * C3_HB + 3 * BN * 4, (hipStream_t)stream>>>(a);
    else conv3x3_c3_fwd_kernel<f16_t, 2><<<grid, C3_T, 2 * C3_HB + 3 * BN * 4, (hipStream_t)stream>>>(a);
    AD_LAUNCH_CHECK("ad_conv3x3_c3_fwd");
    return AD_OK;
}

extern "C" size_t ad_conv3x3_c3_wgrad_ws_bytes(int n, int h, int w) {
    C3Args a; int grid;
    if (!c3_plan(n, h, w, &a, &grid)) return 0;
    return (size_t)grid * 27 * 64 * sizeof(float);
}

extern "C" int ad_conv3x3_c3_wgrad(const float* x, const void* dz, float* dw_hwio, int n, int h, int w, void* ws,
                                   size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_is_half(dtype), "ad_conv3x3_c3_wgrad: 16-bit storage types only (dtype %d)", dtype);
    AD_REQUIRE(x && dz && dw_hwio && (uintptr_t)dw_hwio % 16 == 0, "ad_conv3x3_c3_wgrad: NULL or unaligned operand");
    C3Args a; int grid;
    AD_REQUIRE(c3_plan(n, h, w, &a, &grid), "ad_conv3x3_c3_wgrad: unsupported shape n=%d h=%d w=%d", n, h, w);
    const size_t need = (size_t)grid * 27 * 64 * sizeof(float);
    if (ws == nullptr || ws_bytes < need)
        return ad_set_error(AD_ERR_WS, "ad_conv3x3_c3_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    a.x = x; a.w = nullptr; a.bias = a.gamma = a.beta = nullptr; a.eps = 0.f;
    a.z = a.act = nullptr; a.mean = a.rstd = nullptr; a.dz = (const char*)dz; a.ws = (float*)ws;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == AD_BF16) conv3x3_c3_wgrad_kernel<bf16_t><<<grid, C3_T, 2 * C3_HB + TM * WgradPol<PolBF16>::DZS, s>>>(a);
    else conv3x3_c3_wgrad_kernel<f16_t><<<grid, C3_T, 2 * C3_HB + TM * WgradPol<PolF16>::DZS, s>>>(a);
    AD_LAUNCH_CHECK("ad_conv3x3_c3_wgrad");
    // slabs [grid][27][64] -> dw_hwio [3][3][3][64]: the generic slab reduce with 3-channel input blocks
    wgrad_reduce_kernel<<<(27 * 64 + 127) / 128, 256, 0, s>>>((const float*)ws, dw_hwio, grid, 1, 1, 3, 3, 64);
    AD_LAUNCH_CHECK("wgrad_reduce");
    return AD_OK;
}

extern "C" size_t ad_conv3x3_wgrad_ws_bytes(int n, int h, int w, int cin, int cout, int dtype) {
    WgradPlan p;
    // the same argument rules as ad_conv3x3_wgrad (cin a multiple of the channel granule): the plan below divides by the number
    // of input-channel blocks (r04: a query with cin = 3 was a division by zero -- found by the host-side sanitizer build,
    // tests/test_host_sanitizers.py)
    if (!pixels_ok(n, h, w) || cin <= 0 || cout <= 0 || !ad_dtype_ok(dtype) || cin % ad_cin_granule(dtype)) return 0;
    plan_wgrad(n, h, w, cin, 0, cout, dtype, &p);     // the split of cin does not change the slab size
    size_t need = p.ws_bytes;
    const int chunk = ad_is_half(dtype) ? images_per_launch(n, h, w, cin, 0, cout, false, true) : n;
    if (chunk < n) {                                  // image runs of a >= 2 GiB batch plan their own slabs
        plan_wgrad(chunk, h, w, cin, 0, cout, dtype, &p);
        if (p.ws_bytes > need) need = p.ws_bytes;
        plan_wgrad(n - (n - 1) / chunk * chunk, h, w, cin, 0, cout, dtype, &p);
        if (p.ws_bytes > need) need = p.ws_bytes;
    }
    return need;
}

extern "C" int ad_conv3x3_wgrad(const void* x1, int c1, const void* x2, int c2, const void* dz, float* dw_hwio,
                                int cin_real, int n, int h, int w, int cout, void* ws, size_t ws_bytes, int dtype,
                                void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_conv3x3_wgrad: bad dtype %d", dtype);
    const int gran = ad_cin_granule(dtype);
    AD_REQUIRE(n > 0 && h > 0 && w > 0, "ad_conv3x3_wgrad: bad shape");
    AD_REQUIRE((long)n * h * w < (1L << 31), "ad_conv3x3_wgrad: more than 2^31 pixels");
    AD_REQUIRE(x1 && c1 > 0 && c1 % gran == 0, "ad_conv3x3_wgrad: c1=%d must be a positive multiple of %d", c1, gran);
    AD_REQUIRE((x2 == nullptr) == (c2 == 0) && c2 % gran == 0, "ad_conv3x3_wgrad: c2=%d / x2 mismatch", c2);
    AD_REQUIRE(cout > 0 && cout % 16 == 0, "ad_conv3x3_wgrad: cout=%d must be a multiple of 16", cout);
    const int cin = c1 + c2;
    AD_REQUIRE(cin_real > 0 && cin_real <= cin, "ad_conv3x3_wgrad: cin_real=%d", cin_real);
    AD_REQUIRE((uintptr_t)dw_hwio % 16 == 0, "ad_conv3x3_wgrad: dw_hwio must be 16-byte aligned");
    // batches whose tensors reach 2 GiB: runs of `chunk` images, the first run writes dw, the others add to it
    const int chunk = ad_is_half(dtype) ? images_per_launch(n, h, w, c1, c2, cout, false, true) : n;
    hipStream_t s = (hipStream_t)stream;
    const size_t tsz = ad_is_half(dtype) ? 2 : 4;
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int nr = n - i0 < chunk ? n - i0 : chunk;
        const size_t pix0 = (size_t)i0 * h * w;
        WgradPlan p;
        plan_wgrad(nr, h, w, c1, c2, cout, dtype, &p);
        if (ws == nullptr || ws_bytes < p.ws_bytes)
            return ad_set_error(AD_ERR_WS, "ad_conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
        WgradArgs a;
        a.x1 = (const char*)x1 + pix0 * c1 * tsz; a.x2 = x2 ? (const char*)x2 + pix0 * c2 * tsz : nullptr; a.c1 = c1; a.c2 = c2;
        a.dz = (const char*)dz + pix0 * cout * tsz; a.ws = (float*)ws;
        a.n = nr; a.h = h; a.w = w; a.cout = cout;
        a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split; a.ncib = p.ncib; a.ncob = p.ncob;
        a.g = p.g;
        const bool direct = p.nsplit == 1 && chunk >= n;       // (the specialised kernel too, r05: cout % 64 == 0 there)
        a.dw = direct ? dw_hwio : nullptr; a.cin_real = cin_real;
        int rc = dtype == AD_BF16 ? launch_wgrad<PolBF16>(a, p, s)
                 : dtype == AD_F16 ? launch_wgrad<PolF16>(a, p, s) : launch_wgrad<PolF32>(a, p, s);
        if (rc || direct) return rc;
        int total = 9 * cin_real * cout;
        wgrad_reduce_kernel<<<(total + 127) / 128, 256, 0, s>>>((const float*)ws, dw_hwio, p.nsplit, p.ncib, p.ncob, p.ck,
                                                             cin_real, cout, i0 > 0);
        AD_LAUNCH_CHECK("wgrad_reduce");
    }
    return AD_OK;
}
