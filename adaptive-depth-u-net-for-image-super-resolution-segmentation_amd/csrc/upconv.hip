// The decoder step "dec_up -> Conv2D(nf, 3, padding='same', activation='relu')" without the up-resized tensor.
//
//   Super_resolution/code/train_adaptive_unet.py:258-259   x = dec_up([x, skip]); x = L.Conv2D(nf, 3, ...)(x)
//   shared/custom_layers.py:121-125                        ResizeToMatch (tf.image.resize, bilinear, antialias)
//
// The resize mixes pixels, the convolution's contraction mixes channels, so they commute:
//     conv3x3(U x)[p] = b + sum_tap (U x)[p + tap] W_tap = b + sum_tap (U (x W_tap))[p + tap]     (zero outside the image)
// The step therefore runs as
//   1. a bank of nine 1x1 convolutions on the LOW-resolution map, Y[q][tap][co] = sum_ci x[q][ci] W[tap][ci][co]:
//      one GEMM [pixels] x [Cin] x [9 Cout] on the matrix cores with 1 / ratio^2 of the 3x3 convolution's FLOPs
//      (pw_gemm_kernel);
//   2. a gather at the high resolution that interpolates and shifts, out[p] = relu(b + sum_tap (U Y_tap)[p + tap])
//      (upconv_gather_fwd_kernel: fp32 arithmetic, HBM-bound: reads Y through the L2, writes the activation once);
// and backwards as the transposes: dY = gather^T(dz) (upconv_gather_bwd_kernel), dx = dY Bank^T (pw_gemm_kernel again)
// and dBank = x^T dY (pw_wgrad_kernel; on the fp32 path the 1x1 case of conv3x3_wgrad + bank_grad_kernel), written
// straight into the Keras kernel gradient's layout.
// Neither the up-resized activation (K2' level 0: 1.07 GB per step in bf16) nor its gradient exists in memory: five
// passes over a full-resolution 2 nf-channel tensor and 15/16 of the up-conv's FLOPs (ratio 4) are gone.
// Restated tap by tap in oracle/ops.py (upconv_bank_* / upconv_gather_*), whose equality with
// relu(conv2d(resize(x))) and its gradients is tests/test_oracle_factored_upconv.py.
#include "common.h"
#include <atomic>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned PW_OOB = 0x80000000u;
constexpr long long PW_MAX_BYTES = 0x7fffffffLL;

__device__ __forceinline__ auto pw_rsrc(const void* p, long long bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// ------------------------------------------------------------------ pointwise GEMM on the matrix cores
// Y[m][n] = sum_k X[m][k] B[k][n], X / Y row-major activations ([pixels][channels], NHWC with the pixels flattened),
// B pre-packed to [K / KV][N][KV] (KV = 16 bytes of k), so that an MFMA weight fragment is one 16-byte load.
// Orientation as in conv.hip: D[n][m] = B^T X^T, a lane ends up with 4 consecutive output channels of one pixel.
// A wave owns 64 pixels x 64 output channels (4 x 4 accumulator tiles) and walks K in chunks of 64; fragments come
// straight from global memory (the bank is L2 resident, the pixel rows are re-read per 64-channel block from L1 / L2):
// these GEMMs are 1 / ratio^2 of the step's up-conv FLOPs, so the kernel is kept simple rather than LDS-tiled.
// Every access is a buffer load / store with out-of-range offsets past the last pixel (zeros in, store dropped).
template <typename T> struct PwPol;
template <typename E> struct PwPol16 {
    typedef typename Half16<E>::v8 frag;
    static constexpr int KV = 8;          // k per 16 bytes
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return Half16<E>::mfma(a, b, c); }
    static __device__ __forceinline__ u32x2 pack4(f32x4 v) {
        union { typename Half16<E>::v4 h; u32x2 u; } p;
        p.h[0] = (E)v.x; p.h[1] = (E)v.y; p.h[2] = (E)v.z; p.h[3] = (E)v.w;
        return p.u;
    }
};
template <> struct PwPol<bf16_t> : PwPol16<bf16_t> {};
template <> struct PwPol<f16_t> : PwPol16<f16_t> {};
template <> struct PwPol<float> {
    typedef f32x4 frag;
    static constexpr int KV = 4;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    }
};

struct PwArgs {
    const char* x; const char* bp; char* y;
    int m, k, n;               // pixels, contraction, output channels (n % 64 == 0, k % 64 == 0)
    int nb_per_wg;             // 64-channel output blocks per workgroup
    int whole_groups;          // LDS-tiled kernel: PlOrder's A/B switch
    unsigned long long* dbg;   // diagnostic builds only (-DAD_STAMP, tools/stamps_pw.py): cycle sums by phase of wave 0
};
#ifdef AD_STAMP
static unsigned long long* g_pw_dbg = nullptr;
extern "C" void ad_dbg_set_pw_stamp_buffer(void* p) { g_pw_dbg = (unsigned long long*)p; }
#endif

constexpr int PW_T = 256;      // 4 waves, 64 pixels each

template <typename T>
__global__ __launch_bounds__(PW_T, 2) void pw_gemm_kernel(PwArgs a) {
    typedef PwPol<T> P;
    typedef typename P::frag frag;
    constexpr int TSZ = (int)sizeof(T);
    constexpr int KV = P::KV;
    constexpr int KSTEP = 4 * KV;              // k covered by one fragment pair (4 lane groups x KV)
    constexpr int NKS = 64 / KSTEP;            // fragment steps per 64-k chunk: 2 (16-bit), 4 (fp32)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, l15 = lane & 15;
    const int nblk = a.n / 64;
    const int wg_per_mt = (nblk + a.nb_per_wg - 1) / a.nb_per_wg;
    const int mtile = blockIdx.x / wg_per_mt;
    const int nb0 = (blockIdx.x - mtile * wg_per_mt) * a.nb_per_wg;
    const int nb1 = min(nb0 + a.nb_per_wg, nblk);
    const int m0 = mtile * 256 + wave * 64;
    const auto rsx = pw_rsrc(a.x, (long long)a.m * a.k * TSZ);
    const auto rsb = pw_rsrc(a.bp, (long long)a.k * a.n * TSZ);
    const auto rsy = pw_rsrc(a.y, (long long)a.m * a.n * TSZ);
    // pixel-row fragment of m-tile mt, k-step ks of chunk kc: X[m0 + 16 mt + l15][kc*64 + ks*KSTEP + grp*KV ..]
    unsigned xo[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int mrow = m0 + mt * 16 + l15;
        xo[mt] = mrow < a.m ? (unsigned)((mrow * a.k + grp * KV) * TSZ) : PW_OOB;
    }
    // Fragment stream: step (nb, ks) = the 4 pixel-row and 4 bank fragments of k-step ks of output block nb.  Two register
    // sets alternate; the loads of step t + 2 are issued right after the MFMAs of step t, BEFORE the stores that end an
    // output block, so that a wait for fragments never has the block's stores in front of it in the vmcnt queue (vmcnt
    // retires in issue order: with the stores older than the next loads every block boundary cost a store round trip).
    // Every step issues the same 8 loads and every block the same 8 / 16 stores, unconditionally (past the end the cursor
    // re-reads the last step), which keeps hipcc's s_waitcnt vmcnt counts exact.
    const int kpn = a.k / KSTEP;                         // k-steps per output block (even: k % 64 == 0)
    frag xf0[4], bf0[4], xf1[4], bf1[4];
    int inb = nb0, iks = 0;                              // cursor of the next step to issue
#define PW_ISSUE(XF, BF)                                                                                         \
    {                                                                                                            \
        const unsigned koff_ = (unsigned)(iks * KSTEP * TSZ);                                                    \
        const unsigned brow_ = (unsigned)(((iks * KSTEP / KV) * a.n + inb * 64) * 16);                           \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                          \
            XF[t] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsx, xo[t], koff_, 0));       \
            BF[t] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsb, bo0 + t * 256, brow_, 0)); \
        }                                                                                                        \
        if (++iks == kpn) { if (inb + 1 < nb1) { iks = 0; ++inb; } else { iks = kpn - 1; } }                     \
    }
#define PW_MMA(XF, BF)                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                             \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = P::mma(BF[nt], XF[mt], acc[mt][nt]);
    // bank fragment of n-tile nt: Bp[ks*KSTEP / KV + grp][nb*64 + 16 nt + l15][KV]
    const unsigned bo0 = (unsigned)(((grp * a.n) + l15) * 16);
    PW_ISSUE(xf0, bf0)
    PW_ISSUE(xf1, bf1)
    for (int nb = nb0; nb < nb1; ++nb) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < kpn; ks += 2) {
            PW_MMA(xf0, bf0)
            PW_ISSUE(xf0, bf0)
            PW_MMA(xf1, bf1)
            PW_ISSUE(xf1, bf1)
        }
        // lane holds channels nb*64 + 16 nt + 4 grp .. +3 of pixel m0 + 16 mt + l15
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int mrow = m0 + mt * 16 + l15;
            if constexpr (TSZ == 2) {
                // 16-byte stores: v_permlane16_swap trades the odd 16-lane rows of one n-tile with the even rows of the
                // next, after which lane group g holds channels (g & 1) * 16 + (g >> 1) * 8 .. + 7 of the n-tile pair
                const unsigned yo = mrow < a.m ? (unsigned)((mrow * a.n + nb * 64 + (grp & 1) * 16 + (grp >> 1) * 8) * TSZ) : PW_OOB;
#pragma unroll
                for (int np = 0; np < 2; ++np) {
                    const u32x2 pa = P::pack4(acc[mt][2 * np]), pb = P::pack4(acc[mt][2 * np + 1]);
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rsy, yo, np * 32 * TSZ, 0);
                }
            } else {
                const unsigned yo = mrow < a.m ? (unsigned)((mrow * a.n + nb * 64 + grp * 4) * TSZ) : PW_OOB;
                // The n-tile offset goes into the VECTOR offset (hipcc folds it into the instruction's immediate), never into
                // the scalar soffset operand: a 16-byte buffer store whose soffset is an SGPR (128 and 192 are not inline
                // constants) reads its data registers late, hipcc assumes that form has no store-data hazard and lets the
                // next VALU write reuse them at once -- seen as address words in place of accumulator values (gfx950, ROCm 7.2)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][nt]), rsy, yo + nt * 16 * TSZ, 0, 0);
            }
        }
    }
#undef PW_ISSUE
#undef PW_MMA
}

// ---- LDS-tiled form of the same product for the 16-bit types (the bank GEMMs of ratio ~0.5-0.7 pyramids carry a third to a
// half of the up-conv's FLOPs: fragments straight from L2 cap the kernel above at ~0.2 of the MFMA peak).
// A 512-thread workgroup owns a 256-pixel x NT-channel tile (NT = 32 NW: 192 for the bank's 9 Cout columns, 128 for the
// Cin columns of dx), 8 waves as 4 (pixels) x 2 (channels), a wave 64 pixels x 16 NW channels.  K advances in stages of 64:
// pixel rows and bank columns of a stage go global -> registers -> LDS one stage ahead of their use (double-buffered
// stage: [2 chunks][256 rows][96 B] + [2 chunks][4 k-slots][NT][16 B], the row stride and k-slot layout of conv.hip, both
// conflict-free for the ds_read_b128 fragment reads), one barrier per stage.  Workgroups are persistent and walk the tiles
// n-fastest (the pixel rows of a tile row stay in L2); the fragment stream runs across tile boundaries, the loads of the
// next tile's first stage are in flight while a tile's results are packed and stored.
constexpr int PL_T = 512;
constexpr int PL_XB = 256 * 96;                 // one 32-k chunk of 256 pixel rows (24,576 B)

// XCD-aware tile order of pw_gemm_lds_kernel (r05), shared by the kernel and the host (ad_pw_gemm_tile_order: the coverage
// test and the launcher's round count).  Workgroup b runs on XCD b % 8 and every XCD has its own 4 MiB L2.  Until r04
// workgroup b took tiles b, b + grid, ... of the n-fastest list: the tiles_n tiles that share a block of pixel rows sat on all
// eight XCDs, every XCD pulled every pixel row AND the whole bank through its L2 -- the launch read its operands at ~7 TB/s, the
// rate the Infinity Cache serves (MI355X_MICROARCH.md: 33.5 GB/s per CU from the Infinity Cache against 66-73 from the XCD's own
// L2), with 20 % of a wave's time in MFMAs (tools/stamps_pw.py).  Now a round of an XCD's gridDim.x / 8 workgroups is a BLOCK of
// GM row tiles x GN column tiles: a staged slice of pixel rows is read by GN workgroups behind ONE L2, a bank slice by GM.
// Blocks are ordered row group by row group (a row group = GM row tiles; its n_ng column groups in sequence: its pixel rows
// stay hot).  Whole sets of eight row groups go one group per XCD.  The mg_total % 8 row groups left over are dealt to the XCDs
// BLOCK by block (second half of r05: until then each was one XCD's, and e.g. 10 row groups of 9 column groups cost 18 rounds
// on XCDs 0 and 1 against 9 on the others -- 0.57 of the launch's workgroup-rounds did work).
// GN: the largest of 8, 4, 2, 1 (dividing the workgroups of an XCD) that leaves the fewest slots of the block without a tile,
// e.g. 8 for the 24 column tiles of 1 024 -> 4 608, 1 for the 3 of 128 -> 576.  Only the order changes: every tile is the same
// arithmetic as before.
struct PlOrder {
    int tiles_m, tiles_n, grid, bid;
    bool xo;
    int xcd, rr, GN, GM, n_ng, mg_total, S_full, S_total;
    __host__ __device__ void init(int tiles_m_, int tiles_n_, int grid_, int bid_, bool whole_groups = false) {
        tiles_m = tiles_m_; tiles_n = tiles_n_; grid = grid_; bid = bid_;
        const int ntiles = tiles_m * tiles_n;
        xo = (grid & 7) == 0 && ntiles >= grid;            // (smaller launches keep the linear order)
        xcd = bid & 7; rr = bid >> 3;
        const int per_xcd = grid >> 3;
        GN = 1;
        if (xo) {
            int best_idle = 1 << 30;
            for (int c = 8; c >= 1; c >>= 1) {
                if (c > per_xcd || per_xcd % c) continue;
                const int idle = (tiles_n + c - 1) / c * c - tiles_n;
                if (idle < best_idle) { best_idle = idle; GN = c; }
            }
        }
        GM = xo ? per_xcd / GN : 1;
        n_ng = (tiles_n + GN - 1) / GN;
        mg_total = (tiles_m + GM - 1) / GM;
        if (xo) {
            // whole_groups (A/B switch, option "no_pw_wide"): the order until the first half of r05, every row group one XCD's
            const int tail_blocks = whole_groups ? 0 : (mg_total & 7) * n_ng;   // blocks of the row groups beyond the whole sets of eight
            S_full = (whole_groups ? (mg_total > xcd ? (mg_total - xcd + 7) / 8 : 0) : mg_total >> 3) * n_ng;
            S_total = S_full + (tail_blocks > xcd ? (tail_blocks - xcd + 7) / 8 : 0);
        } else {
            S_full = 0;
            S_total = bid < ntiles ? (ntiles - bid + grid - 1) / grid : 0;
        }
    }
    // the tile of this workgroup's round sq; false where the block has no tile there (ragged last row / column group)
    __host__ __device__ bool tile_at(int sq, int& tm, int& tn) const {
        if (!xo) {
            const int t = bid + sq * grid;
            tm = t / tiles_n; tn = t - tm * tiles_n;
            return true;
        }
        int g, ng;
        if (sq < S_full) {
            const int gl = sq / n_ng;
            ng = sq - gl * n_ng;
            g = gl * 8 + xcd;
        } else {
            const int t = (sq - S_full) * 8 + xcd;         // tail block t of the (mg_total % 8) * n_ng left over
            const int gt = t / n_ng;
            ng = t - gt * n_ng;
            g = (mg_total & ~7) + gt;
        }
        tm = g * GM + rr / GN; tn = ng * GN + rr % GN;
        return tm < tiles_m && tn < tiles_n;
    }
    __host__ __device__ int next_round(int sq) const {
        int tm, tn;
        while (sq < S_total && !tile_at(sq, tm, tn)) ++sq;
        return sq;
    }
};
template <int NW> struct PlGeo {
    static constexpr int NT = 32 * NW;                      // channels per tile
    static constexpr int WB = 4 * NT * 16;                  // one 32-k chunk of the bank tile
    static constexpr int STAGE = 2 * PL_XB + 2 * WB;
    static constexpr int WSL = 8 * NT / PL_T;               // bank slots per thread and stage (3 or 2)
};

template <typename E, int NW, int DEPTH>
__global__ __launch_bounds__(PL_T, 1) void pw_gemm_lds_kernel(PwArgs a) {
    typedef PwPol<E> P;
    typedef typename P::frag frag;
    typedef PlGeo<NW> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int grp = lane >> 4, l15 = lane & 15;
    const int tiles_n = a.n / G::NT, tiles_m = (a.m + 255) / 256;
    const int ntiles = tiles_n * tiles_m;
    const int ks_per_tile = a.k / 64;
    const auto rsx = pw_rsrc(a.x, (long long)a.m * a.k * 2);
    const auto rsb = pw_rsrc(a.bp, (long long)a.k * a.n * 2);
    const auto rsy = pw_rsrc(a.y, (long long)a.m * a.n * 2);
    // staging slots.  Pixel rows: slot s = tid + 512 i -> row s >> 3, 16-byte part s & 7 (8 lanes = one 128-byte row piece).
    // (r05, measured: the 8 lanes the LDS serves per clock write one row's two chunk pieces PL_XB apart, i.e. into the same banks --
    // SQ_LDS_BANK_CONFLICT = 1.0 cycle per LDS instruction of the launch, all of it these stores.  A slot map without the
    // conflict, two rows of one chunk per 8 lanes, changed nothing (half-line global loads cost what the stores gained): the
    // kernel is not waiting for its LDS stores.)
    // Bank: slot s = tid + 512 i -> k-slot s / NT, column s % NT (a k-slot's NT columns are contiguous in the pack).
    int xrow[4], xlds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = tid + PL_T * i;
        xrow[i] = sl >> 3;
        xlds[i] = ((sl & 7) >> 2) * PL_XB + (sl >> 3) * 96 + (sl & 3) * 16;
    }
    const int xpart = (tid & 7) * 16;
    int wsrc[G::WSL], wlds[G::WSL];
#pragma unroll
    for (int i = 0; i < G::WSL; ++i) {
        const int sl = tid + PL_T * i;
        const int j = sl / G::NT, n = sl - j * G::NT;
        wsrc[i] = (j * a.n + n) * 16;
        wlds[i] = 2 * PL_XB + (j >> 2) * G::WB + ((j & 3) * G::NT + n) * 16;
    }
    // DEPTH stages of loads are in flight in registers (DEPTH = 2 where a tile has an even number of k-stages: slots by stage
    // parity, static): a stage is ~0.9 us of MFMAs per SIMD, less than the latency of its loads when K is long (one stage
    // ahead: 763 TFLOP/s at K = 1 024)
    u32x4 xr[DEPTH][4], wr[DEPTH][G::WSL];
    PlOrder ord;                                     // XCD-aware tile order (PlOrder above)
    ord.init(tiles_m, tiles_n, (int)gridDim.x, (int)blockIdx.x, a.whole_groups != 0);
    const int S_total = ord.S_total;
    auto tile_at = [&](int sq, int& tm, int& tn) -> bool { return ord.tile_at(sq, tm, tn); };
    auto next_round = [&](int sq) -> int { return ord.next_round(sq); };
    int qs = next_round(0), iks = 0;                 // cursor of the next stage to issue (round, k-stage)
    int qtm = 0, qtn = 0;                            // its tile; past the end: the last one again (re-read, never stored)
    if (qs < S_total) tile_at(qs, qtm, qtn);
#define PL_ISSUE(SLOT)                                                                                           \
    {                                                                                                            \
        const int tm_ = qtm, tn_ = qtn;                                                                          \
        const int kk_ = iks;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
            const int mrow_ = tm_ * 256 + xrow[i];                                                               \
            xr[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, mrow_ < a.m ? (unsigned)(mrow_ * a.k * 2 + xpart) : PW_OOB, \
                                                                (unsigned)(kk_ * 128), 0);                       \
        }                                                                                                        \
        const unsigned wb_ = (unsigned)((kk_ * 8 * a.n + tn_ * G::NT) * 16);                                     \
        _Pragma("unroll") for (int i = 0; i < G::WSL; ++i)                                                       \
            wr[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (unsigned)wsrc[i], wb_, 0);                 \
        if (++iks == ks_per_tile) {                                                                              \
            iks = 0;                                                                                             \
            qs = next_round(qs + 1);                                                                             \
            if (qs < S_total) tile_at(qs, qtm, qtn);                                                             \
        }                                                                                                        \
    }
#ifdef AD_STAMP
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long pt_last = clock64();
    const unsigned long long pt_begin = pt_last;
#define PSTAMP(slot) do { unsigned long long now_ = clock64(); pst[slot] += now_ - pt_last; pt_last = now_; } while (0)
#else
#define PSTAMP(slot) do {} while (0)
#endif
#define PL_STAGE(SLOT)                                                                                           \
    {                                                                                                            \
        char* sb = smem + buf * G::STAGE;                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(sb + xlds[i]) = xr[SLOT][i];     \
        _Pragma("unroll") for (int i = 0; i < G::WSL; ++i) *reinterpret_cast<u32x4*>(sb + wlds[i]) = wr[SLOT][i]; \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        PSTAMP(0);                          /* wait for the staged loads + LDS stores */                         \
        asm volatile("s_barrier" ::: "memory");                                                                  \
        PSTAMP(1);                          /* barrier */                                                        \
        PL_ISSUE(SLOT)                                                                                           \
        PSTAMP(2);                          /* issue of the next loads */                                        \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                          \
            frag xf[4], wf[NW];                                                                                  \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                     \
                xf[mt] = *reinterpret_cast<const frag*>(sb + c * PL_XB + xfo + mt * 16 * 96);                    \
            _Pragma("unroll") for (int nt = 0; nt < NW; ++nt)                                                    \
                wf[nt] = *reinterpret_cast<const frag*>(sb + c * G::WB + wfo + nt * 256);                        \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                     \
                _Pragma("unroll") for (int nt = 0; nt < NW; ++nt) acc[mt][nt] = P::mma(wf[nt], xf[mt], acc[mt][nt]); \
        }                                                                                                        \
        PSTAMP(3);                          /* fragment reads + MFMAs of both chunks */                          \
        buf ^= 1;                                                                                                \
    }
    const int xfo = (wm * 64 + l15) * 96 + grp * 16;                           // + mt * 16 * 96 (+ chunk * PL_XB)
    const int wfo = 2 * PL_XB + (grp * G::NT + wn * NW * 16 + l15) * 16;       // + nt * 256 (+ chunk * WB)
    PL_ISSUE(0)
    if (DEPTH == 2) PL_ISSUE(DEPTH - 1)
    int buf = 0;
    for (int sc = next_round(0); sc < S_total; sc = next_round(sc + 1)) {
        f32x4 acc[4][NW];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ks_per_tile; ks += DEPTH) {
            PL_STAGE(0)
            if (DEPTH == 2) PL_STAGE(DEPTH - 1)
        }
        int tm, tn;
        tile_at(sc, tm, tn);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int mrow = tm * 256 + wm * 64 + mt * 16 + l15;
            const unsigned yo = mrow < a.m ? (unsigned)((mrow * a.n + tn * G::NT + wn * NW * 16 + (grp & 1) * 16 + (grp >> 1) * 8) * 2) : PW_OOB;
#pragma unroll
            for (int np = 0; np < NW / 2; ++np) {
                const u32x2 pa = P::pack4(acc[mt][2 * np]), pb = P::pack4(acc[mt][2 * np + 1]);
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rsy, yo + np * 64, 0, 0);
            }
        }
        PSTAMP(4);                          // tile epilogue (pack + stores)
    }
#ifdef AD_STAMP
    if (wave == 0 && lane == 0 && a.dbg) {
        for (int i = 0; i < 5; ++i) a.dbg[blockIdx.x * 8 + i] = pst[i];
        a.dbg[blockIdx.x * 8 + 7] = clock64() - pt_begin;
    }
#endif
#undef PSTAMP
#undef PL_ISSUE
#undef PL_STAGE
}

// ---- The same product with the two waves of every SIMD a phase apart (r05): 256 pixels x 256 channels per workgroup, 8 waves
// as 2 (pixels) x 4 (channels), a wave 128 pixels x 64 channels (128 accumulator registers).  K advances in 32-deep chunks; a
// wave alternates
//     R(q): [chunk q opens a 64-deep stage: write the NEXT stage's staged registers to the other LDS buffer, issue the loads of
//            the stage after it]  read the 8 pixel + 4 bank fragments of chunk q into registers, wait for them;
//     M(q): 32 MFMAs on those registers;
// with one workgroup barrier after each.  Waves 4-7 (the second pixel half) pass ONE extra barrier before their first phase, so
// on every SIMD one wave is in M while the other is in R: the matrix pipe always has an issuing wave, and the LDS traffic, the
// global loads and the tile epilogue of one half sit under the MFMAs of the other (in pw_gemm_lds_kernel all eight waves stage,
// wait, read and multiply in step: tools/stamps_pw.py had 63 % of a wave's time waiting).  Why this is race free: with global
// phases P = 1, 2, ..., waves 0-3 run R(q) in P = 2q + 1, waves 4-7 in P = 2q + 2.  Stage S + 1 goes to the buffer stage S - 1
// lived in, whose last reads (chunk 2S - 1) are issued in P = 4S - 1 / 4S and complete before the barrier that ends that phase
// (lgkmcnt(0) in front of it); the writes come in P = 4S + 1 / 4S + 2, complete before THEIR barrier, and the first read of
// stage S + 1 is in P = 4S + 5.  Loads stay in flight for four phases (one stage of MFMAs on both waves, ~1.2 us).
// The bank is read in the packed layout [K / 8][N][8] as before; the tile order is PlOrder.  N % 256 == 0, K % 64 == 0.
constexpr int PP_NT = 256;
constexpr int PP_WB = 4 * PP_NT * 16;              // one 32-k chunk of the bank tile (16 384 B)
constexpr int PP_STAGE = 2 * PL_XB + 2 * PP_WB;    // 81 920 B; two stages = the CU's 160 KiB

template <typename E>
__global__ __launch_bounds__(PL_T, 1) void pw_gemm_pp_kernel(PwArgs a) {
    typedef PwPol<E> P;
    typedef typename P::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;        // pixel half (= phase group: waves w and w + 4 share a SIMD), channel quarter
    const int grp = lane >> 4, l15 = lane & 15;
    const int tiles_n = a.n / PP_NT, tiles_m = (a.m + 255) / 256;
    const int ks_per_tile = a.k / 64;
    const auto rsx = pw_rsrc(a.x, (long long)a.m * a.k * 2);
    const auto rsb = pw_rsrc(a.bp, (long long)a.k * a.n * 2);
    const auto rsy = pw_rsrc(a.y, (long long)a.m * a.n * 2);
    // staging slots of a stage: pixel rows 256 x 8 sixteen-byte parts, bank 8 k-slots x 256 columns: 4 + 4 per thread
    int xrow[4], xlds[4], wsrc[4], wlds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = tid + PL_T * i;
        xrow[i] = sl >> 3;
        xlds[i] = ((sl & 7) >> 2) * PL_XB + (sl >> 3) * 96 + (sl & 3) * 16;
        const int j = sl >> 8, n = sl & 255;
        wsrc[i] = (j * a.n + n) * 16;
        wlds[i] = 2 * PL_XB + (j >> 2) * PP_WB + ((j & 3) * PP_NT + n) * 16;
    }
    const int xpart = (tid & 7) * 16;
    PlOrder ord;
    ord.init(tiles_m, tiles_n, (int)gridDim.x, (int)blockIdx.x, a.whole_groups != 0);
    const int S_total = ord.S_total;
    // cursor of the next stage to LOAD (round of the tile order, k-stage); past the end it stays on the last stage (re-read)
    int qs = ord.next_round(0), iks = 0, qtm = 0, qtn = 0;
    if (qs < S_total) ord.tile_at(qs, qtm, qtn);
    u32x4 xr[4], wr[4];
    // (measured and withdrawn, r05: one-dword loads that warm the L2 two or three stages ahead -- 979 -> 818 TFLOP/s on the
    // 1 024 -> 4 608 dx product: the launch is not waiting for misses to HBM either)
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mrow = qtm * 256 + xrow[i];
            xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, mrow < a.m ? (unsigned)(mrow * a.k * 2 + xpart) : PW_OOB,
                                                          (unsigned)(iks * 128), 0);
        }
        const unsigned wb = (unsigned)((iks * 8 * a.n + qtn * PP_NT) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (unsigned)wsrc[i], wb, 0);
        if (++iks == ks_per_tile) {
            const int nq = ord.next_round(qs + 1);
            if (nq < S_total) { qs = nq; iks = 0; ord.tile_at(qs, qtm, qtn); }
            else iks = ks_per_tile - 1;            // no further tile: keep re-reading the last stage (never used)
        }
    };
    auto stage_to_lds = [&](char* sb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(sb + xlds[i]) = xr[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(sb + wlds[i]) = wr[i];
    };
    const int xfo = (wm * 128 + l15) * 96 + grp * 16;                              // + mt * 16 * 96 (+ chunk * PL_XB)
    const int wfo = 2 * PL_XB + (grp * PP_NT + wn * 64 + l15) * 16;                // + nt * 256 (+ chunk * PP_WB)
    if (S_total == 0) return;                       // (uniform per workgroup: no barrier is left waiting)
    // prologue: stage 0 into buffer 0, stage 1 into the registers
    issue();
    stage_to_lds(smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue();
    if (wm == 1) __builtin_amdgcn_s_barrier();      // the second half runs one phase behind
    int buf = 0;
    for (int sc = ord.next_round(0); sc < S_total; sc = ord.next_round(sc + 1)) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ks_per_tile; ++ks) {
            char* sb = smem + buf * PP_STAGE;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // ---- R
                if (c == 0) {
                    stage_to_lds(smem + (buf ^ 1) * PP_STAGE);
                    issue();
                }
                frag xf[8], wf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const frag*>(sb + c * PP_WB + wfo + nt * 256);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) xf[mt] = *reinterpret_cast<const frag*>(sb + c * PL_XB + xfo + mt * 16 * 96);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---- M
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = P::mma(wf[nt], xf[mt], acc[mt][nt]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            buf ^= 1;
        }
        // tile epilogue: issued at the head of this wave's next R phase, under the other half's MFMAs
        int tm, tn;
        ord.tile_at(sc, tm, tn);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int mrow = tm * 256 + wm * 128 + mt * 16 + l15;
            const unsigned yo = mrow < a.m ? (unsigned)((mrow * a.n + tn * PP_NT + wn * 64 + (grp & 1) * 16 + (grp >> 1) * 8) * 2) : PW_OOB;
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                const u32x2 pa = P::pack4(acc[mt][2 * np]), pb = P::pack4(acc[mt][2 * np + 1]);
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rsy, yo + np * 64, 0, 0);
            }
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();      // the first half passes the barrier the second half took at the start
}

// Bank operands from the fp32 Keras kernel W[3][3][Cin][Cout] (HWIO):
//   forward  bank  B[k = ci][n = tap * Cout + co]  -> bf[ci / KV][tap * Cout + co][KV]
//   backward bank  B^T[k = tap * Cout + co][n = ci] -> bd[(tap * Cout + co) / KV][ci][KV]
template <typename T>
__global__ __launch_bounds__(256) void pw_bank_pack_kernel(const float* __restrict__ w, int cin, int cout, T* __restrict__ bf,
                                                           T* __restrict__ bd) {
    constexpr int KV = 16 / (int)sizeof(T);
    const int n9 = 9 * cout;
    const int total = cin * n9;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * total; e += gridDim.x * 256) {
        if (e < total) {
            const int kv = e % KV, r = e / KV;
            const int nn = r % n9, kc = r / n9;
            const int ci = kc * KV + kv, tap = nn / cout, co = nn - tap * cout;
            bf[e] = (T)w[((size_t)tap * cin + ci) * cout + co];
        } else {
            const int i = e - total;
            const int kv = i % KV, r = i / KV;
            const int ci = r % cin, kc = r / cin;
            const int kk = kc * KV + kv, tap = kk / cout, co = kk - tap * cout;
            bd[i] = (T)w[((size_t)tap * cin + ci) * cout + co];
        }
    }
}

// dW[tap][ci][co] (Keras HWIO gradient) = dBank[ci][tap * Cout + co], the centre tap of the [3][3][Cin][9 Cout] tensor
// that the 1x1 case of ad_conv3x3_wgrad writes.
__global__ __launch_bounds__(256) void bank_grad_kernel(const float* __restrict__ dw9, int cin, int cout, float* __restrict__ dw) {
    const int n9 = 9 * cout;
    const float* centre = dw9 + (size_t)4 * cin * n9;
    const int total = 9 * cin * cout;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int co = e % cout, r = e / cout;
        const int ci = r % cin, tap = r / cin;
        dw[e] = centre[(size_t)ci * n9 + tap * cout + co];
    }
}

// ------------------------------------------------------------------ bank weight gradient (16-bit types)
// dBank[k][n] = sum_m X[m][k] dY[m][n]: the contraction runs over PIXELS, which both operands have as their slow axis, so
// the MFMA fragments (8 consecutive contraction indices per lane) are transposed reads from LDS (ds_read_b64_tr_b16, as
// conv.hip's wgrad).  A 512-thread workgroup owns 128 input channels x 576 bank columns (nine taps x 64 output channels)
// for its share of the pixels: 8 waves as 2 (k) x 4 (n), a wave holds 4 x 9 accumulator tiles (144 registers), so per
// 32-pixel step it reads 8 + 18 transposed fragments for 36 MFMAs and X / dY are fetched from HBM ONCE per 128 / 576
// channels -- the generic 1x1 wgrad re-reads X nine times and dY four times and writes eight zero taps.
// Stages of 32 pixels are double buffered in LDS ([32][128] + [32][576] elements, row strides 288 / 1184 B: an odd number
// of 32-byte units, so the eight rows a 32-lane half reads land on distinct banks), global loads one stage ahead in
// registers, one barrier per stage.  Each workgroup writes its partial sums to a slab [split][K][N]; pw_wgrad_reduce sums
// the splits in a fixed order (bitwise deterministic) straight into the Keras layout dW[tap][ci][co].
constexpr int PWG_T = 512;
constexpr int PWG_KT = 128, PWG_NT = 576;
constexpr int PWG_XS = PWG_KT * 2 + 32;           // 288
constexpr int PWG_DS = PWG_NT * 2 + 32;           // 1184
constexpr int PWG_STAGE = 32 * PWG_XS + 32 * PWG_DS;      // 47,104 B
constexpr int PWG_LDS = 2 * PWG_STAGE;

struct PwWgradArgs {
    const char* x; const char* dy; float* slab;
    int m, k, n;
    int mps;                 // pixels per split (multiple of 32)
};

template <typename E>
__device__ __forceinline__ typename Half16<E>::v8 pw_tr_pair(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) short4_t* lds_p;
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
    return __builtin_bit_cast(typename Half16<E>::v8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <typename E>
__global__ __launch_bounds__(PWG_T, 1) void pw_wgrad_kernel(PwWgradArgs a) {
    typedef typename Half16<E>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    const int split = blockIdx.x, kg = blockIdx.y, ng = blockIdx.z;
    const int m_begin = split * a.mps, m_end = min(a.m, m_begin + a.mps);
    const int nst = (m_end - m_begin + 31) / 32;
    const auto rsx = pw_rsrc(a.x, (long long)a.m * a.k * 2);
    const auto rsd = pw_rsrc(a.dy, (long long)a.m * a.n * 2);
    // staging slots: X 32 rows x 16 parts (one per thread), dY 32 rows x 72 parts (slots tid + 512 i, i < 5)
    const int xrow = tid >> 4, xpart = tid & 15;
    int drow[5], dpart[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int sl = tid + PWG_T * i;
        drow[i] = sl / 72; dpart[i] = sl - drow[i] * 72;       // rows >= 32: beyond the stage, never stored
    }
    u32x4 xr, dr[5];
#define PWG_ISSUE(S)                                                                                             \
    {                                                                                                            \
        const int mb_ = m_begin + (S) * 32;                                                                      \
        const int mx_ = mb_ + xrow;                                                                              \
        xr = __builtin_amdgcn_raw_buffer_load_b128(rsx, mx_ < m_end ? (unsigned)((mx_ * a.k + kg * PWG_KT + xpart * 8) * 2) : PW_OOB, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                          \
            const int md_ = mb_ + drow[i];                                                                       \
            dr[i] = __builtin_amdgcn_raw_buffer_load_b128(                                                       \
                rsd, (drow[i] < 32 && md_ < m_end) ? (unsigned)((md_ * a.n + ng * PWG_NT + dpart[i] * 8) * 2) : PW_OOB, 0, 0); \
        }                                                                                                        \
    }
#define PWG_STORE(BUF)                                                                                           \
    {                                                                                                            \
        char* xs_ = smem + (BUF) * PWG_STAGE;                                                                    \
        char* ds_ = xs_ + 32 * PWG_XS;                                                                           \
        *reinterpret_cast<u32x4*>(xs_ + xrow * PWG_XS + xpart * 16) = xr;                                        \
        _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                            \
            if (drow[i] < 32) *reinterpret_cast<u32x4*>(ds_ + drow[i] * PWG_DS + dpart[i] * 16) = dr[i];         \
    }
    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposed-read addresses: lane (G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3) supplies row 4 G + q (+ 16 for
    // the second half of the fragment), 8 bytes at column 4 p of the 16-channel tile; contraction index 8 G + j of the
    // MFMA <-> pixel 16 (j >> 2) + 4 G + (j & 3) of the stage, the same map for both operands
    const int G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int xa = (4 * G + q) * PWG_XS + wk * 128 + pp * 8;              // + kt * 32 (+ 16 rows for the second read)
    const int da = (4 * G + q) * PWG_DS + wn * 288 + pp * 8;              // + nt * 32
    if (nst > 0) PWG_ISSUE(0)
    for (int s = 0; s < nst; ++s) {
        PWG_STORE(s & 1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        PWG_ISSUE(s + 1)                   // (past the split: every offset out of range, zeros that are never stored)
        const char* xs = smem + (s & 1) * PWG_STAGE;
        const char* ds = xs + 32 * PWG_XS;
        v8 xf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) xf[kt] = pw_tr_pair<E>(xs + xa + kt * 32, xs + xa + kt * 32 + 16 * PWG_XS);
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const v8 df = pw_tr_pair<E>(ds + da + nt * 32, ds + da + nt * 32 + 16 * PWG_DS);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[kt][nt] = Half16<E>::mfma(xf[kt], df, acc[kt][nt]);
        }
    }
#undef PWG_ISSUE
#undef PWG_STORE
    // lane holds dBank[k = kg*128 + wk*64 + kt*16 + 4 G + r][n = ng*576 + wn*144 + nt*16 + (lane & 15)]
    float* slab = a.slab + (size_t)split * a.k * a.n;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* row = slab + (size_t)(kg * PWG_KT + wk * 64 + kt * 16 + 4 * G + r) * a.n + ng * PWG_NT + wn * 144 + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 9; ++nt) row[nt * 16] = acc[kt][nt][r];
        }
}

// dW[tap][ci][co] = sum over the splits, in ascending order, of slab[split][ci][tap * cout + co]
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int cin, int cout,
                                                              float* __restrict__ dw) {
    const int n9 = 9 * cout;
    const int total4 = 9 * cin * cout / 4;
    const size_t stride = (size_t)cin * n9;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total4; e += gridDim.x * 256) {
        const int co = (e * 4) % cout, r = (e * 4) / cout;
        const int ci = r % cin, tap = r / cin;
        const float* src = slab + (size_t)ci * n9 + tap * cout + co;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (; sp + 3 < nsplit; sp += 4) {               // four independent loads in flight, added in a fixed order
            const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)sp * stride);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(sp + 1) * stride);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)(sp + 2) * stride);
            const float4 v3 = *reinterpret_cast<const float4*>(src + (size_t)(sp + 3) * stride);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; sp < nsplit; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sp * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dw + (size_t)e * 4) = s;
    }
}

// ------------------------------------------------------------------ gather (forward)
// out[n][oy][ox][c] = act(bias[c] + sum_{dy,dx} [row oy+dy, column ox+dx inside the image]
//                         sum_{a,b < 2} wy[oy+dy][a] wx[ox+dx][b] Y[n][sy[oy+dy]+a][sx[ox+dx]+b][tap(dy,dx)][c])
// Tables: two taps per output index (an up-resize never has more), the second index clamped and weighted 0 where the
// source has no such row / column; sy is non-decreasing and sy[r+1] - sy[r-1] <= window - 2 (checked by the host;
// window = 3 for every resize of ratio >= 2).
// A thread owns one (output column, 16-byte channel vector) and MARCHES down a strip of output rows.  It keeps, for each
// of the three vertical taps dy, the horizontally interpolated and shifted sums
//     H_dy[q] = sum_dx sum_b wx[ox+dx][b] Y[q][sx[ox+dx]+b][tap(dy,dx)]          (3 dx x 2 taps = 6 loads each)
// of a sliding window of three or four low-resolution rows q in registers (fp32): every low-resolution row is interpolated ONCE per
// strip and column (the strip-per-8-rows version of this kernel redid it for every strip: 5x the arithmetic at ratio 4),
// and an output row is two fused multiply-adds per dy on window entries picked by workgroup-uniform table values.
struct GatherArgs {
    const char* y; const float* bias; char* out;
    const int* sy; const float* wy; const int* sx; const float* wx;
    int h, w, oh, ow, c, relu, rows;      // rows: output rows per strip
    int slab;                             // LDS bytes of one staged bank-row piece (multiple of 16)
};

// Workgroups are dealt to the 8 XCDs round-robin in launch order and every XCD has its own L2: XCD k takes the k-th
// contiguous eighth of the (x, y, z) grid, so that the strips of the forward gather that share bank rows and columns run
// behind ONE L2 (3 - 15 % per launch on the Experiment-2 levels, 2 % on the x4 levels).  The transpose does NOT use it: it
// is bound by its arithmetic, not by the rows adjacent workgroups re-read (2.0x its operand bytes from HBM,
// profiles/r03_pmc_summary.txt) -- the mapping gave 4 % at batch 64 and cost 25 % at batch 8, where it puts one image
// on each XCD and all eight walk the same offsets of their images in step.
__device__ __forceinline__ void xcd_block(int& bx, int& by, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
    unsigned l = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = nb >> 3;
    if (l < per * 8) l = (l & 7) * per + (l >> 3);       // (the last nb % 8 workgroups keep their place)
    bx = (int)(l % gx);
    by = (int)((l / gx) % gy);
    bz = (int)(l / (gx * gy));
}

constexpr int GF_MAXROWS = 64;
constexpr int GB_MAXKY = 30;   // most transposed vertical taps the backward gather stages
constexpr int GB_BATCH = 5;    // window columns loaded at once by upconv_gather_bwd_kernel

// 8-byte vector of T, unpacked to / packed from fp32 (the gathers keep per-thread state small: occupancy hides their latency)
template <typename T> struct VecH;
template <> struct VecH<float> {
    static constexpr int N = 2;
    float2 v;
    __device__ __forceinline__ void load(const void* p) { v = *reinterpret_cast<const float2*>(p); }
    __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<float2*>(p) = v; }
    __device__ __forceinline__ void to_f32(float* f) const { f[0] = v.x; f[1] = v.y; }
    __device__ __forceinline__ void from_f32(const float* f) { v = make_float2(f[0], f[1]); }
};
template <typename E> struct VecH16 {
    static constexpr int N = 4;
    typename Half16<E>::v4 v;
    __device__ __forceinline__ void load(const void* p) { v = *reinterpret_cast<const typename Half16<E>::v4*>(p); }
    __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<typename Half16<E>::v4*>(p) = v; }
    __device__ __forceinline__ void to_f32(float* f) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
    }
    __device__ __forceinline__ void from_f32(const float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (E)f[i];
    }
};
template <> struct VecH<bf16_t> : VecH16<bf16_t> {};
template <> struct VecH<f16_t> : VecH16<f16_t> {};

// Four waves per SIMD for the two-slot instantiation (64 channels: the full-resolution level of every pyramid).  hipcc takes 156
// registers for it when left alone (three waves); held to 128 it does not spill and the launch goes from 2.9 to 3.7 TB/s
// (K2' level 0: 0.289 -> 0.224 ms) -- the kernel hides its ~60 vector instructions per stored row behind other waves' memory
// time, so a fourth wave is a third more to hide behind.  The four- and eight-slot instantiations spill under the same bound
// (0.20 -> 0.43 ms at eight slots; three registers at four slots, which the no-spill guard of tests/test_isa_guards.py
// rules out for 12 us), as does the four-row window; they keep the default.
template <typename T, int GW, int NSL>
__global__ __launch_bounds__(256, GW == 3 && NSL <= 2 ? 4 : 1) void upconv_gather_fwd_kernel(GatherArgs a) {
    constexpr int EPT = VecH<T>::N;
    constexpr int TSZ = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int vecs = a.c / EPT;                        // divides 256 (launcher)
    const int nox = 256 / vecs;                        // output columns of this workgroup
    int bx, by, nn;
    xcd_block(bx, by, nn);
    const int oy0 = by * a.rows, oy1 = min(oy0 + a.rows, a.oh);
    const int ox0 = bx * nox;
    const int oxl = threadIdx.x / vecs, v = threadIdx.x - oxl * vecs;
    const int ox = min(ox0 + oxl, a.ow - 1);
    const bool live = ox0 + oxl < a.ow;
    // The low-resolution columns this workgroup reads: ONE contiguous piece of every bank row (columns xlo .. xhi, nine
    // taps, all channels), staged in LDS by all threads with 16-byte loads -- the 18 vectors a thread needs per
    // low-resolution row are then LDS reads, and the global loads of the NEXT row are few enough (NSL per thread) to stay
    // in flight in registers across the output rows in between.
    const int xlo = a.sx[max(ox0 - 1, 0)];
    const int xhi = min(a.sx[min(ox0 + nox, a.ow - 1)] + 1, a.w - 1);
    const int slab_bytes = (xhi - xlo + 1) * 9 * a.c * TSZ;
    char* sbuf = gsm;                                  // [2][a.slab] bytes
    float (*s_w)[12] = reinterpret_cast<float (*)[12]>(gsm + 2 * a.slab);
    int* s_adv = reinterpret_cast<int*>(gsm + 2 * a.slab + GF_MAXROWS * 48);
    // horizontal taps of the three shifted columns: LDS byte offsets inside the slab
    float fx[3][2];
    int xoff[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int cc = ox + d - 1;
        const bool ok = cc >= 0 && cc < a.ow;
        const int ccl = min(max(cc, 0), a.ow - 1);
        const int x0 = a.sx[ccl];
        fx[d][0] = ok ? a.wx[2 * ccl] : 0.f;
        fx[d][1] = ok ? a.wx[2 * ccl + 1] : 0.f;
        xoff[d][0] = (((x0 - xlo) * 9 + d) * a.c + v * EPT) * TSZ;
        xoff[d][1] = (((min(x0 + 1, a.w - 1) - xlo) * 9 + d) * a.c + v * EPT) * TSZ;
    }
    float bv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) bv[e] = a.bias ? a.bias[v * EPT + e] : 0.f;
    const long long ybytes = (long long)a.h * a.w * 9 * a.c * TSZ;                  // one image of the bank (< 2 GiB: launcher)
    const auto rsy = pw_rsrc(a.y + (size_t)nn * ybytes, ybytes);
    float hs[3][GW][EPT];
    u32x4 pend[NSL];
#define GF_ISSUE(Q)                                                                                        \
    {                                                                                                      \
        const unsigned rb_ = (unsigned)((min((Q), a.h - 1) * a.w + xlo) * 9 * a.c * TSZ);                  \
        _Pragma("unroll") for (int i_ = 0; i_ < NSL; ++i_) {                                               \
            const int o_ = (threadIdx.x + 256 * i_) * 16;                                                  \
            pend[i_] = __builtin_amdgcn_raw_buffer_load_b128(rsy, o_ < slab_bytes ? rb_ + o_ : PW_OOB, 0, 0); \
        }                                                                                                  \
    }
#define GF_STAGE(BUF)                                                                                      \
    {                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < NSL; ++i_) {                                               \
            const int o_ = (threadIdx.x + 256 * i_) * 16;                                                  \
            if (o_ < a.slab) *reinterpret_cast<u32x4*>(sbuf + (BUF) * a.slab + o_) = pend[i_];             \
        }                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                    \
    }
#define GF_TAKE(BUF, SLOT)                                                                                 \
    {                                                                                                      \
        const char* sl_ = sbuf + (BUF) * a.slab;                                                           \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                                 \
            VecH<T> ld_[3][2];                                                                             \
            _Pragma("unroll") for (int d_ = 0; d_ < 3; ++d_)                                               \
                _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) ld_[d_][b_].load(sl_ + xoff[d_][b_] + s_ * 3 * a.c * TSZ); \
            _Pragma("unroll") for (int e_ = 0; e_ < EPT; ++e_) hs[s_][SLOT][e_] = 0.f;                     \
            _Pragma("unroll") for (int d_ = 0; d_ < 3; ++d_)                                               \
                _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) {                                         \
                    float t_[EPT];                                                                         \
                    ld_[d_][b_].to_f32(t_);                                                                \
                    _Pragma("unroll") for (int e_ = 0; e_ < EPT; ++e_) hs[s_][SLOT][e_] += fx[d_][b_] * t_[e_]; \
                }                                                                                          \
        }                                                                                                  \
    }
    // Per-row table values of the strip, staged in LDS once: the window advance of each output row and, per vertical tap
    // dy, the weights of the window rows (two of them non-zero; all zero where row oy + dy is outside the image).
    // Read back as LDS broadcasts: table loads from global memory inside the row loop would sit in the same vmcnt queue as
    // the row stores and turn every row into three dependent memory round trips.
    for (int r = threadIdx.x; r < oy1 - oy0; r += 256) {
        const int oy = oy0 + r;
        const int qb = a.sy[max(oy - 1, 0)];
        s_adv[r] = r == 0 ? 0 : qb - a.sy[max(oy - 2, 0)];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            const int rr = oy + s_ - 1;
            const bool ok = rr >= 0 && rr < a.oh;
            const int k0 = ok ? a.sy[rr] - qb : 0;
            const float f0 = ok ? a.wy[2 * rr] : 0.f, f1 = ok ? a.wy[2 * rr + 1] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s_w[r][s_ * 4 + k] = k == k0 ? f0 : (k == k0 + 1 ? f1 : 0.f);
        }
    }
    int qbase = a.sy[max(oy0 - 1, 0)];                 // the window holds rows qbase .. qbase + GW - 1 (workgroup-uniform)
    int cur = 0;                                       // LDS buffer the next staged row goes to
    GF_ISSUE(qbase)
#pragma unroll
    for (int k = 0; k < GW; ++k) {                     // fill the window; row qbase + GW is left in flight
        GF_STAGE(cur)
        GF_ISSUE(qbase + k + 1)
        GF_TAKE(cur, k)
        cur ^= 1;
    }
    T* on = reinterpret_cast<T*>(a.out) + ((size_t)nn * a.oh * a.ow + ox) * a.c + v * EPT;
    for (int r = 0; r < oy1 - oy0; ++r) {
        for (int adv = __builtin_amdgcn_readfirstlane(s_adv[r]); adv > 0; --adv) {    // slide: drop row qbase, add row qbase + GW
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int k = 0; k + 1 < GW; ++k)
#pragma unroll
                    for (int e = 0; e < EPT; ++e) hs[s_][k][e] = hs[s_][k + 1][e];
            GF_STAGE(cur)                               // row qbase + GW: its loads were issued one slide ago
            ++qbase;
            GF_ISSUE(qbase + GW)
            GF_TAKE(cur, GW - 1)
            cur ^= 1;
        }
        float o[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) o[e] = bv[e];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            const float4 wv = *reinterpret_cast<const float4*>(&s_w[r][s_ * 4]);
            const float wk[4] = {wv.x, wv.y, wv.z, wv.w};          // (wk[3] is 0 when the window has three rows)
#pragma unroll
            for (int k = 0; k < GW; ++k)
#pragma unroll
                for (int e = 0; e < EPT; ++e) o[e] += wk[k] * hs[s_][k][e];
        }
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        if (live) {
            VecH<T> st;
            st.from_f32(o);
            st.store(on + (size_t)(oy0 + r) * a.ow * a.c);
        }
    }
#undef GF_ISSUE
#undef GF_STAGE
#undef GF_TAKE
}

// ------------------------------------------------------------------ gather (backward): the transpose
// dY[n][qy][qx][tap(dy,dx)][c] = sum_{r, cc} wyt[qy][r] wxt[qx][cc] g[n][r - dy][cc - dx][c]     (r - dy, cc - dx inside)
// with the TRANSPOSED tables of the resize (first hi-res row / column that reads low-res index q, and the weights of the
// KYT / KXT rows / columns from there on).  A thread owns one (low-resolution column, channel vector) of one
// low-resolution row: it walks the KYT + 2 gradient rows that reach that row, forms per row the three horizontally
// contracted sums (one per dx, from KXT + 2 loads) and adds them to the 3 x 3 tap accumulators with the rows' vertical
// weights (workgroup-uniform).  KX2 >= kxt + 2 bounds the column window in registers.  (Measured and withdrawn: 8-byte
// channel vectors as in the forward gather -- 164 instead of 224 registers, a third wave per SIMD, twice the load
// instructions: 0.274 -> 0.314 ms on the 64 -> 256 level.  Also withdrawn: the gradient rows staged in LDS as in the forward
// gather, with 16- and with 8-byte vectors at two to four waves per SIMD -- 0.264 / 0.296 against 0.262 ms.  Per wave and row
// the kernel issues ~250 vector instructions (156 v_pk_fma_f32, the bf16 unpacking, addresses): ~0.13 ms of issue time per
// launch at two waves per SIMD, which the loads only partly hide; neither fewer load instructions nor more waves moved it.)
struct GatherBwdArgs {
    const char* g; char* dy;
    const int* ryt; const float* wyt; int kyt;      // [h], [h][kyt]
    const int* cxt; const float* wxt; int kxt;      // [w], [w][kxt]
    int h, w, oh, ow, c;
};

// r05 (second half): the kernel held 224 registers (two waves per SIMD): 30 per-lane tap weights (three shifted views of the SAME
// kxt weights), 10 clamped column offsets and 10 validity flags beside the 72 accumulators.  Now the kxt weights are kept once
// and indexed statically (window column j, shift d -> tap j + d - 2), columns outside the image are out-of-range buffer loads
// (zeros) instead of clamped loads with a zero weight, the column offsets are one register plus a uniform stride, and the ten-column
// window is loaded as two batches of five: 162 / 168 registers for the six- / ten-column instantiations, three waves per SIMD, no
// spill (the explicit bound is given to the six-column one only: under it hipcc spills the ten-column one, which fits by itself).
// Same products and the same order of additions as before.
template <typename T, int KX2>
__global__ __launch_bounds__(256, KX2 <= 6 ? 3 : 1) void upconv_gather_bwd_kernel(GatherBwdArgs a) {
    constexpr int EPT = ElemTraits<T>::EPT;
    constexpr int TSZ = (int)sizeof(T);
    constexpr int KT = KX2 - 2;                        // most transposed horizontal taps of this instantiation
    const int vecs = a.c / EPT;
    const int qy = blockIdx.y, nn = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.w * vecs;
    const int qx = live ? i / vecs : 0, v = live ? i - qx * vecs : 0;
    const int c0 = a.cxt[qx] - 1;                      // first column of the window
    float wk[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) wk[k] = k < a.kxt ? a.wxt[qx * a.kxt + k] : 0.f;
    float acc[3][3][EPT];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < EPT; ++e) acc[s][d][e] = 0.f;
    const long long gbytes = (long long)a.oh * a.ow * a.c * TSZ;                    // one image (< 2 GiB: launcher)
    const auto rsg = pw_rsrc(a.g + (size_t)nn * gbytes, gbytes);
    const int r0 = a.ryt[qy];
    // vertical weights of the window rows, per dy, staged in LDS (workgroup-uniform; read back as broadcasts instead of
    // global loads that would serialise with the gradient loads on vmcnt): row j <-> gradient row r0 - 1 + j, which is
    // hi-res row r0 - 1 + j + dy of the resized tensor, transposed tap k = j + s - 2
    __shared__ __attribute__((aligned(16))) float s_fy[GB_MAXKY + 2][4];
    for (int t = threadIdx.x; t < (a.kyt + 2) * 3; t += 256) {
        const int j = t / 3, s_ = t - j * 3, k = j + s_ - 2;
        s_fy[j][s_] = (k >= 0 && k < a.kyt) ? a.wyt[qy * a.kyt + k] : 0.f;
    }
    __syncthreads();
    if (!live) return;
    const int cstride = a.c * TSZ;
    const int voff0 = (c0 * a.c + v * EPT) * TSZ;      // byte offset of window column 0 inside a row (negative left of the image)
    for (int p = max(r0 - 1, 0); p <= min(r0 + a.kyt, a.oh - 1); ++p) {
        const unsigned rowb = (unsigned)(p * a.ow * a.c * TSZ);
        float hs[3][EPT];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < EPT; ++e) hs[d][e] = 0.f;
        // the window's loads in batches of at most GB_BATCH columns (all of a batch in flight): ten at once cost the ten-column
        // instantiation two registers more than the step to three waves per SIMD
        constexpr int NB = KX2 > 6 ? (KX2 + GB_BATCH - 1) / GB_BATCH : 1, BW = (KX2 + NB - 1) / NB;     // (six columns: one batch)
#pragma unroll
        for (int jb = 0; jb < KX2; jb += BW) {
            Vec16<T> ld[BW];
#pragma unroll
            for (int jj = 0; jj < BW; ++jj) {
                const int j = jb + jj;
                const bool ok = j < KX2 && (unsigned)(c0 + j) < (unsigned)a.ow && j < a.kxt + 2;
                ld[jj].v = __builtin_bit_cast(decltype(ld[jj].v), __builtin_amdgcn_raw_buffer_load_b128(rsg, ok ? (unsigned)(voff0 + j * cstride) : PW_OOB, rowb, 0));
            }
#pragma unroll
            for (int jj = 0; jj < BW; ++jj) {
                const int j = jb + jj;
                float t[EPT];
                ld[jj].to_f32(t);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int k = j + d - 2;               // compile-time after unrolling
                    if (j < KX2 && k >= 0 && k < KT) {
#pragma unroll
                        for (int e = 0; e < EPT; ++e) hs[d][e] += wk[k] * t[e];
                    }
                }
            }
            if (NB > 1) __builtin_amdgcn_sched_barrier(0);      // (hipcc would hoist the next batch's loads above this one's arithmetic)
        }
        const float4 fyv = *reinterpret_cast<const float4*>(&s_fy[p - (r0 - 1)][0]);
        const float fy[3] = {fyv.x, fyv.y, fyv.z};
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[s_][d][e] += fy[s_] * hs[d][e];
    }
    T* dst = reinterpret_cast<T*>(a.dy) + ((((size_t)nn * a.h + qy) * a.w + qx) * 9) * a.c + v * EPT;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            Vec16<T> st;
            st.from_f32(acc[s][d]);
            st.store(dst + (s * 3 + d) * a.c);
        }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- C ABI
extern "C" int ad_pw_supported(int64_t m, int k, int n, int dtype) {
    if (!ad_dtype_ok(dtype) || m <= 0 || k <= 0 || n <= 0 || k % 64 || n % 64) return 0;
    const long long tsz = ad_is_half(dtype) ? 2 : 4;
    const long long widest = k > n ? k : n;
    return m * widest * tsz <= PW_MAX_BYTES && (long long)k * n * tsz <= PW_MAX_BYTES;
}

extern "C" size_t ad_pw_bank_elems(int cin, int cout) { return (size_t)9 * cin * cout; }

extern "C" int ad_pw_bank_pack(const float* w_hwio, int cin, int cout, void* bank_fwd, void* bank_bwd, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_pw_bank_pack: bad dtype %d", dtype);
    AD_REQUIRE(w_hwio && bank_fwd && bank_bwd && cin > 0 && cout > 0 && cin % 64 == 0 && cout % 64 == 0,
               "ad_pw_bank_pack: cin=%d cout=%d must be positive multiples of 64", cin, cout);
    const long long total = 2LL * 9 * cin * cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    AD_DISPATCH_DTYPE(dtype, T_, pw_bank_pack_kernel<T_><<<blocks, 256, 0, (hipStream_t)stream>>>(w_hwio, cin, cout, (T_*)bank_fwd,
                                                                                            (T_*)bank_bwd);)
    AD_LAUNCH_CHECK("ad_pw_bank_pack");
    return AD_OK;
}

// which kernel ad_pw_gemm launches for a shape: 0 = fragments straight from L2 (fp32, few pixels, odd widths), 1 = LDS-tiled
// persistent kernel with one k-stage of loads in flight, 2 = the same with two (K >= 256 in stage pairs).  Tests ask it to
// make sure a parity case reaches the variant it is meant for.
static int pw_gemm_variant(int64_t m, int k, int n, int dtype) {
    if (!(ad_is_half(dtype) && m >= 2048 && (n % 192 == 0 || n % 128 == 0))) return 0;
    return (k / 64) % 2 == 0 && k >= 256 ? 2 : 1;
}
// rounds of the slowest workgroup of an LDS-tiled launch with NT-channel tiles (PlOrder; workgroup 0 sits on XCD 0, which takes
// the first block of every ragged set)
static int pw_rounds(int64_t m, int n, int nt, int* grid_out) {
    const int tiles_m = (int)((m + 255) / 256), tiles_n = n / nt;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < ad_num_cu() ? ntiles : ad_num_cu();
    PlOrder o;
    o.init(tiles_m, tiles_n, grid, 0);
    if (grid_out) *grid_out = grid;
    return o.S_total;
}
// channels per tile / 32.  256-channel tiles (a wave: 64 pixels x 128 channels, fewer LDS bytes per MFMA than the 192- and
// 128-channel tiles: +5 ... +19 % per tile-column on the dx products of the Experiment-2 levels) where the width allows, K is
// long enough for the arithmetic to dominate and the launch does not lose more to whole rounds than the tile gains
// (rounds x tile width / measured relative rate).  Option "no_pw_wide": off.
static bool pw_new_order() { return !ad_option(AD_OPT_NO_PW_WIDE); }     // A/B switch (ad_set_option "no_pw_wide")
static int pw_gemm_nw(int64_t m, int k, int n) {
    const bool nw8 = pw_new_order();
    const int base = n % 192 == 0 ? 6 : 4;
    if (!(nw8 && n % 256 == 0 && k >= 256)) return base;
    // measured per-column rates relative to the 192-channel tile at equal round quantisation (E2s07's levels): 256: 1.05-1.08,
    // 128: 0.88
    const double c8 = (double)pw_rounds(m, n, 256, nullptr) * 256 / 1.08;
    const double cb = (double)pw_rounds(m, n, 32 * base, nullptr) * 32 * base / (base == 6 ? 1.0 : 0.88);
    return c8 <= cb ? 8 : base;
}
extern "C" int ad_pw_gemm_variant(int64_t m, int k, int n, int dtype) {
    return ad_pw_supported(m, k, n, dtype) ? pw_gemm_variant(m, k, n, dtype) : -1;
}

extern "C" int ad_pw_gemm_tile_channels(int64_t m, int k, int n, int dtype) {
    if (!ad_pw_supported(m, k, n, dtype) || pw_gemm_variant(m, k, n, dtype) == 0) return 0;
    return 32 * pw_gemm_nw(m, k, n);
}

extern "C" int ad_pw_gemm_tile_order(int tiles_m, int tiles_n, int grid, int* tiles, int max_rounds) {
    if (tiles_m <= 0 || tiles_n <= 0 || grid <= 0 || !tiles || max_rounds <= 0) return -1;
    int rounds = 0;
    for (int b = 0; b < grid; ++b) {
        PlOrder o;
        o.init(tiles_m, tiles_n, grid, b);
        if (o.S_total > max_rounds) return -1;
        if (o.S_total > rounds) rounds = o.S_total;
        for (int sq = 0; sq < max_rounds; ++sq) {
            int tm = 0, tn = 0;
            tiles[(size_t)b * max_rounds + sq] = (sq < o.S_total && o.tile_at(sq, tm, tn)) ? tm * tiles_n + tn : -1;
        }
    }
    return rounds;
}

extern "C" int ad_pw_gemm(const void* x, const void* bank, void* y, int64_t m, int k, int n, int dtype, void* stream) {
    AD_REQUIRE(ad_pw_supported(m, k, n, dtype), "ad_pw_gemm: unsupported m=%lld k=%d n=%d dtype=%d (ask ad_pw_supported)",
               (long long)m, k, n, dtype);
    AD_REQUIRE(x && bank && y, "ad_pw_gemm: NULL operand");
    PwArgs a;
    a.x = (const char*)x; a.bp = (const char*)bank; a.y = (char*)y;
    a.m = (int)m; a.k = k; a.n = n;
    a.nb_per_wg = 0;
    a.whole_groups = pw_new_order() ? 0 : 1;
    a.dbg = nullptr;
#ifdef AD_STAMP
    a.dbg = g_pw_dbg;
#endif
    const int variant = pw_gemm_variant(m, k, n, dtype);
    if (variant > 0) {                                                              // LDS-tiled persistent kernel
        hipStream_t s = (hipStream_t)stream;
        const int nw = pw_gemm_nw(m, k, n);
        const int ntiles = (int)((m + 255) / 256) * (n / (32 * nw));
        const int grid = ntiles < ad_num_cu() ? ntiles : ad_num_cu();
        // two stages of loads in flight where a tile has >= 4 k-stages that pair up (K = 128, two stages a tile: 10 % slower)
        const int depth = variant;
#define PL_LAUNCH(E_, NW_, D_)                                                                                       \
    {                                                                                                                \
        static std::atomic<unsigned long long> attr_{0};                                                             \
        if (ad_first_on_device(attr_)) {                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_gemm_lds_kernel<E_, NW_, D_>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PlGeo<NW_>::STAGE);            \
        }                                                                                                            \
        pw_gemm_lds_kernel<E_, NW_, D_><<<grid, PL_T, 2 * PlGeo<NW_>::STAGE, s>>>(a);                                \
    }
        if (nw == 8) {
            static std::atomic<unsigned long long> pp_attr{0};
            if (ad_first_on_device(pp_attr)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_gemm_pp_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_STAGE);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_gemm_pp_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_STAGE);
            }
            if (dtype == AD_BF16) pw_gemm_pp_kernel<bf16_t><<<grid, PL_T, 2 * PP_STAGE, s>>>(a);
            else pw_gemm_pp_kernel<f16_t><<<grid, PL_T, 2 * PP_STAGE, s>>>(a);
        } else if (nw == 6) {
            if (dtype == AD_BF16) { if (depth == 2) PL_LAUNCH(bf16_t, 6, 2) else PL_LAUNCH(bf16_t, 6, 1) }
            else { if (depth == 2) PL_LAUNCH(f16_t, 6, 2) else PL_LAUNCH(f16_t, 6, 1) }
        } else {
            if (dtype == AD_BF16) { if (depth == 2) PL_LAUNCH(bf16_t, 4, 2) else PL_LAUNCH(bf16_t, 4, 1) }
            else { if (depth == 2) PL_LAUNCH(f16_t, 4, 2) else PL_LAUNCH(f16_t, 4, 1) }
        }
#undef PL_LAUNCH
        AD_LAUNCH_CHECK("ad_pw_gemm (lds)");
        return AD_OK;
    }
    const int mtiles = (int)((m + 255) / 256), nblk = n / 64;
    // all output blocks of a pixel tile in one workgroup (its pixel rows stay in L1) unless that leaves CUs idle
    int per = nblk;
    while (per > 1 && (long long)mtiles * ((nblk + per - 1) / per) < 1024) per = (per + 1) / 2;
    a.nb_per_wg = per;
    const int grid = mtiles * ((nblk + per - 1) / per);
    AD_DISPATCH_DTYPE(dtype, T_, pw_gemm_kernel<T_><<<grid, PW_T, 0, (hipStream_t)stream>>>(a);)
    AD_LAUNCH_CHECK("ad_pw_gemm");
    return AD_OK;
}

extern "C" int ad_pw_bank_grad(const float* dw9, int cin, int cout, float* dw_hwio, void* stream) {
    AD_REQUIRE(dw9 && dw_hwio && cin > 0 && cout > 0, "ad_pw_bank_grad: bad operands");
    const int total = 9 * cin * cout;
    const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
    bank_grad_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dw9, cin, cout, dw_hwio);
    AD_LAUNCH_CHECK("ad_pw_bank_grad");
    return AD_OK;
}

// pixels per split and number of splits of the bank weight gradient
static void pw_wgrad_plan(int64_t m, int k, int n, int* mps, int* nsplit) {
    const int groups = (k / PWG_KT) * (n / PWG_NT);
    int want = 256 / groups;
    if (want < 1) want = 1;
    int per = (int)((m + want - 1) / want);
    per = (per + 31) / 32 * 32;
    *mps = per;
    *nsplit = (int)((m + per - 1) / per);
}

extern "C" int ad_pw_wgrad_supported(int64_t m, int cin, int cout, int dtype) {
    if (!ad_is_half(dtype) || m <= 0 || cin <= 0 || cout <= 0 || cin % PWG_KT || cout % 64) return 0;
    return m * (cin > 9LL * cout ? cin : 9LL * cout) * 2 <= PW_MAX_BYTES;
}

extern "C" size_t ad_pw_wgrad_ws_bytes(int64_t m, int cin, int cout) {
    // (pixel counts beyond what ad_pw_wgrad_supported accepts -- every operand below 2 GiB -- would overflow the 32-bit plan)
    if (m <= 0 || cin <= 0 || cout <= 0 || cin % PWG_KT || cout % 64 || m * (cin > 9LL * cout ? cin : 9LL * cout) * 2 > PW_MAX_BYTES)
        return 0;
    int mps, nsplit;
    pw_wgrad_plan(m, cin, 9 * cout, &mps, &nsplit);
    return (size_t)nsplit * cin * 9 * cout * sizeof(float);
}

extern "C" int ad_pw_wgrad(const void* x, const void* dybank, float* dw_hwio, int64_t m, int cin, int cout, void* ws,
                           size_t ws_bytes, int dtype, void* stream) {
    AD_REQUIRE(ad_pw_wgrad_supported(m, cin, cout, dtype), "ad_pw_wgrad: unsupported m=%lld cin=%d cout=%d dtype=%d", (long long)m,
               cin, cout, dtype);
    AD_REQUIRE(x && dybank && dw_hwio && (uintptr_t)dw_hwio % 16 == 0, "ad_pw_wgrad: NULL or unaligned operand");
    const size_t need = ad_pw_wgrad_ws_bytes(m, cin, cout);
    if (!ws || ws_bytes < need) return ad_set_error(AD_ERR_WS, "ad_pw_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    PwWgradArgs a;
    a.x = (const char*)x; a.dy = (const char*)dybank; a.slab = (float*)ws;
    a.m = (int)m; a.k = cin; a.n = 9 * cout;
    int nsplit;
    pw_wgrad_plan(m, a.k, a.n, &a.mps, &nsplit);
    hipStream_t s = (hipStream_t)stream;
    static std::atomic<unsigned long long> attr{0};
    if (ad_first_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, PWG_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, PWG_LDS);
    }
    dim3 grid(nsplit, a.k / PWG_KT, a.n / PWG_NT);
    if (dtype == AD_BF16) pw_wgrad_kernel<bf16_t><<<grid, PWG_T, PWG_LDS, s>>>(a);
    else pw_wgrad_kernel<f16_t><<<grid, PWG_T, PWG_LDS, s>>>(a);
    AD_LAUNCH_CHECK("ad_pw_wgrad");
    const int total4 = 9 * cin * cout / 4;
    pw_wgrad_reduce_kernel<<<(total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048, 256, 0, s>>>((const float*)ws, nsplit, cin, cout, dw_hwio);
    AD_LAUNCH_CHECK("pw_wgrad_reduce");
    return AD_OK;
}

extern "C" int ad_upconv_gather_fwd_supported(int c, int slab_cols, int dtype) {
    if (!ad_dtype_ok(dtype) || c <= 0 || slab_cols <= 0) return 0;
    const int ept = ad_is_half(dtype) ? 4 : 2, tsz = ad_is_half(dtype) ? 2 : 4;
    if (c % ept || 256 % (c / ept)) return 0;                               // 8-byte channel vectors, a divisor of 256 per column
    return (long long)slab_cols * 9 * c * tsz <= 12 * 256 * 16;            // <= 12 staging slots per thread (48 KB per row piece)
}

// Host-side: the slab_cols argument of ad_upconv_gather_fwd for a horizontal table `sx_host` (a HOST copy of the sx the launch
// will pass), so that callers do not restate the kernel's staging rule; -1 when the table is not a non-decreasing map into [0, w).
extern "C" int ad_upconv_slab_cols(const int* sx_host, int w, int ow, int c, int dtype) {
    if (!sx_host || w <= 0 || ow <= 0 || c <= 0 || !ad_dtype_ok(dtype)) return -1;
    const int ept = ad_is_half(dtype) ? 4 : 2;
    if (c % ept || 256 % (c / ept)) return -1;
    for (int i = 0; i < ow; ++i)
        if (sx_host[i] < 0 || sx_host[i] >= w || (i && sx_host[i] < sx_host[i - 1])) return -1;
    const int nox = 256 / (c / ept);
    int most = 0;
    for (int ox0 = 0; ox0 < ow; ox0 += nox) {                 // the same xlo / xhi the kernel derives per workgroup
        const int lo = sx_host[ox0 > 0 ? ox0 - 1 : 0];
        int hi = sx_host[ox0 + nox < ow ? ox0 + nox : ow - 1] + 1;
        if (hi > w - 1) hi = w - 1;
        if (hi - lo + 1 > most) most = hi - lo + 1;
    }
    return most;
}

extern "C" int ad_upconv_gather_fwd(const void* ybank, const float* bias, void* out, const int* sy, const float* wy,
                                    const int* sx, const float* wx, int window, int slab_cols, int n, int h, int w, int oh,
                                    int ow, int c, int relu, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_upconv_gather_fwd: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 4 : 2, tsz = ad_is_half(dtype) ? 2 : 4;
    AD_REQUIRE(ybank && out && sy && wy && sx && wx, "ad_upconv_gather_fwd: NULL operand");
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && oh >= h && ow >= w && c > 0,
               "ad_upconv_gather_fwd: bad shape n=%d %dx%d -> %dx%d c=%d", n, h, w, oh, ow, c);
    AD_REQUIRE(ad_upconv_gather_fwd_supported(c, slab_cols, dtype),
               "ad_upconv_gather_fwd: c=%d / slab_cols=%d not supported (ask ad_upconv_gather_fwd_supported)", c, slab_cols);
    AD_REQUIRE((long long)h * w * 9 * c * tsz < (1LL << 31) && (long long)oh * ow * c < (1LL << 31),
               "ad_upconv_gather_fwd: an image of 2 GiB or more");
    GatherArgs a;
    a.y = (const char*)ybank; a.bias = bias; a.out = (char*)out;
    a.sy = sy; a.wy = wy; a.sx = sx; a.wx = wx;
    a.h = h; a.w = w; a.oh = oh; a.ow = ow; a.c = c; a.relu = relu;
    AD_REQUIRE(window == 3 || window == 4, "ad_upconv_gather_fwd: window=%d (3: sy advances by at most 1 over two rows; 4: by 2)", window);
    a.slab = (slab_cols * 9 * c * tsz + 15) / 16 * 16;
    const int nsl = (a.slab / 16 + 255) / 256;                                // staging slots per thread: 1 .. 8
    // strips: long enough to amortise the window rows a strip starts with, short enough to fill the chip
    const int nox = 256 / (c / ept);
    const int bx = (ow + nox - 1) / nox;
    int rows = 64;
    while (rows > 16 && (long long)bx * ((oh + rows - 1) / rows) * n < 4096) rows /= 2;
    a.rows = rows;
    dim3 grid(bx, (oh + rows - 1) / rows, n);
    const size_t lds = 2 * (size_t)a.slab + GF_MAXROWS * 48 + GF_MAXROWS * 4;
    hipStream_t s = (hipStream_t)stream;
    // every instantiation can need more than the 64 KB default: 2 x slab + tables exceeds it from slab > 31 104 bytes on, which the
    // 8-slot variants reach (scale 0.8 / depth 5, level 2: 132 -> 164 at 256 channels stages 2 x 32 256 bytes), so the limit is
    // raised on the instantiation that is launched, once each
#define GF_LAUNCH(GW_, NSL_)                                                                                              \
    {                                                                                                                     \
        AD_DISPATCH_DTYPE(dtype, T_,                                                                                      \
            static std::atomic<unsigned long long> big_{0};                       /* one bit per device ordinal */        \
            if (lds > 48 * 1024 && ad_first_on_device(big_)) {                                                            \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(upconv_gather_fwd_kernel<T_, GW_, NSL_>),           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess)            \
                    return ad_set_error(AD_ERR_LAUNCH, "ad_upconv_gather_fwd: cannot raise the dynamic LDS limit to %zu bytes", lds); \
            }                                                                                                             \
            upconv_gather_fwd_kernel<T_, GW_, NSL_><<<grid, 256, lds, s>>>(a);)                                           \
    }
    if (window == 3) { if (nsl <= 2) GF_LAUNCH(3, 2) else if (nsl <= 4) GF_LAUNCH(3, 4) else if (nsl <= 8) GF_LAUNCH(3, 8) else GF_LAUNCH(3, 12) }
    else { if (nsl <= 2) GF_LAUNCH(4, 2) else if (nsl <= 4) GF_LAUNCH(4, 4) else if (nsl <= 8) GF_LAUNCH(4, 8) else GF_LAUNCH(4, 12) }
#undef GF_LAUNCH
    AD_LAUNCH_CHECK("ad_upconv_gather_fwd");
    return AD_OK;
}

extern "C" int ad_upconv_gather_bwd_supported(int kxt) { return kxt >= 1 && kxt + 2 <= 16; }     // (kyt: GB_MAXKY, checked at the call)

extern "C" int ad_upconv_gather_bwd(const void* g, void* dybank, const int* ryt, const float* wyt, int kyt, const int* cxt,
                                    const float* wxt, int kxt, int n, int h, int w, int oh, int ow, int c, int dtype,
                                    void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_upconv_gather_bwd: bad dtype %d", dtype);
    const int ept = ad_is_half(dtype) ? 8 : 4;
    AD_REQUIRE(g && dybank && ryt && wyt && cxt && wxt, "ad_upconv_gather_bwd: NULL operand");
    AD_REQUIRE(n > 0 && h > 0 && w > 0 && oh >= h && ow >= w && c > 0 && c % ept == 0 && kyt >= 1,
               "ad_upconv_gather_bwd: bad shape n=%d %dx%d <- %dx%d c=%d", n, h, w, oh, ow, c);
    AD_REQUIRE(kyt <= GB_MAXKY, "ad_upconv_gather_bwd: %d vertical taps (at most %d)", kyt, GB_MAXKY);
    AD_REQUIRE(ad_upconv_gather_bwd_supported(kxt), "ad_upconv_gather_bwd: %d horizontal taps (at most 14)", kxt);
    AD_REQUIRE((long long)h * w * 9 * c < (1LL << 31) && (long long)oh * ow * c < (1LL << 31),
               "ad_upconv_gather_bwd: an image of more than 2^31 elements");
    GatherBwdArgs a;
    a.g = (const char*)g; a.dy = (char*)dybank;
    a.ryt = ryt; a.wyt = wyt; a.kyt = kyt; a.cxt = cxt; a.wxt = wxt; a.kxt = kxt;
    a.h = h; a.w = w; a.oh = oh; a.ow = ow; a.c = c;
    dim3 grid((w * (c / ept) + 255) / 256, h, n);
    hipStream_t s = (hipStream_t)stream;
    if (kxt + 2 <= 6) { AD_DISPATCH_DTYPE(dtype, T_, upconv_gather_bwd_kernel<T_, 6><<<grid, 256, 0, s>>>(a);) }
    else if (kxt + 2 <= 10) { AD_DISPATCH_DTYPE(dtype, T_, upconv_gather_bwd_kernel<T_, 10><<<grid, 256, 0, s>>>(a);) }
    else { AD_DISPATCH_DTYPE(dtype, T_, upconv_gather_bwd_kernel<T_, 16><<<grid, 256, 0, s>>>(a);) }
    AD_LAUNCH_CHECK("ad_upconv_gather_bwd");
    return AD_OK;
}
