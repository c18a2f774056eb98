// Premise check for VERDICT r04 item 2 (overlap the LayerNorm-backward epilogue with MFMA instead of adding it), run BEFORE
// touching conv3x3_fwd_wres_kernel<., 4>.  Built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/roles tools/mfma_power/role_alternation.hip && /tmp/roles
//
// The conv kernels' item (256 pixels x 64 channels, Cin = 64) is 18 tap steps of 16 v_mfma_f32_16x16x32_bf16 per MFMA wave,
// followed by an epilogue of E vector instructions in the SAME wave (E ~ 200 plain, ~ 440 LayerNorm forward, ~ 880 LayerNorm
// backward: tools/stamps_lnb.py).  One workgroup per CU, 8 waves = 2 per SIMD, as the kernels.  Three organisations:
//   serial      (today)  waves 0-3: [18 x 16 MFMA][E VALU] per item; waves 4-7 ("loaders"): L VALU per item; 2 barriers per item
//   alternating          all 8 waves run both roles, the two waves of a SIMD half an item out of phase:
//                          wave A: [MFMA item 2k  ] | barrier | [E + L VALU       ] | barrier
//                          wave B: [E + L VALU    ] | barrier | [MFMA item 2k + 1 ] | barrier
//                        -- the wave tile (64 pixels x 64 channels), the LDS traffic per MFMA and the two barriers per item
//                        are those of today's kernel; what changes is that a SIMD's matrix pipe has work during every epilogue
//   alternating, no barrier  the same without the barriers (upper bound of what the organisation can give)
// VALU work = v_fma_f32 / v_pk_fma_f32 (1 : 1) on 8 independent chains.  Random bf16 MFMA operands (the clock is data dependent).
// Prints microseconds per item.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define VALU_BLOCK(N)                                                    /* N: a multiple of 16 */         \
    _Pragma("unroll 1") for (int o_ = 0; o_ < (N) / 16; ++o_) {                                              \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                   \
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q_]) : "v"(c1), "v"(c2));                       \
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(g[q_]) : "v"(c3), "v"(c4));                    \
        }                                                                                                    \
    }

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define MFMA_ITEM                                                                                            \
    _Pragma("unroll 1") for (int st = 0; st < 18; ++st) {                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);         \
    }

// MODE 0: serial, 1: alternating, 2: alternating without barriers
template <int MODE, int E, int L>
__global__ __launch_bounds__(512, 1) void roles(const bf16x8* __restrict__ src, float* __restrict__ out, int items) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float f[8];
    f32x2 g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = (float)(tid + i) * 1e-6f; g[i] = f32x2{f[i], -f[i]}; }
    float c1 = 1.0000001f, c2 = 1e-9f;
    f32x2 c3 = {1.0000001f, 0.9999999f}, c4 = {1e-9f, -1e-9f};
    asm volatile("" : "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4));
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(tid * 8 + i) & 4095];
        b[i] = src[(tid * 8 + 4 + i) & 4095];
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        if (wave < 4) {
            for (int k = 0; k < items; ++k) {
                lds_barrier();
                MFMA_ITEM
                lds_barrier();
                VALU_BLOCK(E)
            }
        } else {
            for (int k = 0; k < items; ++k) {
                lds_barrier();
                VALU_BLOCK(L / 2)
                lds_barrier();
                VALU_BLOCK(L / 2)
            }
        }
    } else {
        if (wave < 4) {
            for (int k = 0; k < items; k += 2) {
                MFMA_ITEM
                if (MODE == 1) lds_barrier();
                VALU_BLOCK(E + L)
                if (MODE == 1) lds_barrier();
            }
        } else {
            for (int k = 0; k < items; k += 2) {
                VALU_BLOCK(E + L)
                if (MODE == 1) lds_barrier();
                MFMA_ITEM
                if (MODE == 1) lds_barrier();
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += f[i] + g[i].x + g[i].y;
    if (sum == 12345.678f) out[blockIdx.x * 512 + tid] = sum;      // keeps everything alive
}

template <int MODE, int E, int L>
double run(const bf16x8* src, float* out, int items, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) roles<MODE, E, L><<<256, 512>>>(src, out, items);      // warm-up: lets the clock settle
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) roles<MODE, E, L><<<256, 512>>>(src, out, items);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps / items;        // us per item (each workgroup runs `items` items: MODE 0 on 4 waves, MODE 1/2 on 8)
}

#define ROW(E)                                                                                                \
    printf("%5d  %10.2f  %12.2f  %12.2f\n", E, run<0, E, 96>(src, out, items, reps), run<1, E, 96>(src, out, items, reps), \
           run<2, E, 96>(src, out, items, reps));

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1234);
    for (auto& v : h) {
        const unsigned s = rand() & 1, e = 124 + (rand() & 3), m = rand() & 127;
        v = (unsigned short)((s << 15) | (e << 7) | m);
    }
    bf16x8* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int items = 256, reps = 40;
    printf("us per item (item = 18 x 16 MFMA 16x16x32 bf16 per SIMD + E epilogue VALU + 96 loader VALU), one workgroup per CU\n");
    printf("    E      serial   alternating   alt, no barrier\n");
    ROW(0) ROW(208) ROW(448) ROW(880)
    return 0;
}
