// How many vector (VALU) instructions hide beside the MFMAs of a conv-shaped main loop, by MFMA shape and by WHERE they run
// (diagnostic for DESIGN 4.7 / the r05 plan; built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/mfma_power/mfma_valu_coissue.hip && /tmp/coissue
// One workgroup per CU.  Waves 0-3 (one per SIMD) own a 64 x 64 fp32 accumulator tile each, as the conv kernels' MFMA waves do,
// and issue per step either 16 v_mfma_f32_16x16x32_bf16 or 8 v_mfma_f32_32x32x16_bf16 (the same FLOPs) on register operands.
//   same wave : V independent v_fma_f32 are interleaved with the step's MFMAs in the MFMA wave itself
//   partner   : waves 4-7 (the second wave of each SIMD, as the conv kernels' loader waves) issue the V v_fma_f32 per step,
//               free-running (no barrier), the MFMA waves issue MFMAs only
// Prints microseconds per 1000 steps; the MFMA-only time is the V = 0 row.  Random bf16 operands (the clock is data dependent).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V fused multiply-adds on NV independent chains, spread evenly between the NM MFMAs of a step by the caller
// (inline asm: hipcc would SLP-pack adjacent scalar fused multiply-adds into v_pk_fma_f32, which issues differently)
#define FMA_BLOCK(N)                                                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < (N); ++q_) {                                                     \
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q_ & 7]) : "v"(c1), "v"(c2));                       \
    }

template <int SHAPE, int V, bool PARTNER>
__global__ __launch_bounds__(PARTNER ? 512 : 256, 1) void burn(const bf16x8* __restrict__ src, float* __restrict__ out, int steps) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)(tid + i) * 1e-6f;
    float c1 = 1.0000001f, c2 = 1e-9f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    float sum = 0.f;
    if (PARTNER && wave >= 4) {
        for (int s = 0; s < steps; ++s) {
            FMA_BLOCK(V)
        }
    } else {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = src[(tid * 8 + i) & 4095];
            b[i] = src[(tid * 8 + 4 + i) & 4095];
        }
        constexpr int VS = PARTNER ? 0 : V;              // VALU work of the MFMA wave itself
        if (SHAPE == 16) {
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < steps; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        FMA_BLOCK(VS / 16)
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
        } else {
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
            for (int s = 0; s < steps; ++s) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * kh + i], b[2 * kh + j], acc[i][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            FMA_BLOCK(VS / 8)
                            __builtin_amdgcn_sched_barrier(0);
                        }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += f[i];
    if (sum == 12345.678f) out[blockIdx.x * 512 + tid] = sum;      // keeps everything alive
}

template <int SHAPE, int V, bool PARTNER>
double run(const bf16x8* src, float* out, int steps, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int thr = PARTNER ? 512 : 256;
    for (int r = 0; r < reps; ++r) burn<SHAPE, V, PARTNER><<<256, thr>>>(src, out, steps);      // warm-up: lets the clock settle
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) burn<SHAPE, V, PARTNER><<<256, thr>>>(src, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps / steps * 1000.0;        // us per 1000 steps
}

#define ROW(V)                                                                                                    \
    printf("%4d  %10.1f %10.1f   %10.1f %10.1f\n", V, run<16, V, false>(src, out, steps, reps), run<32, V, false>(src, out, steps, reps), \
           run<16, V, true>(src, out, steps, reps), run<32, V, true>(src, out, steps, reps));

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
    const int steps = 2000, reps = 60;
    printf("us per 1000 steps (a step = 64 x 64 x 32 MACs per wave = 16 MFMA 16x16x32 or 8 MFMA 32x32x16); V = v_fma_f32 per step\n");
    printf("   V   same wave: 16x16x32   32x32x16    partner wave: 16x16x32   32x32x16\n");
    ROW(0) ROW(16) ROW(32) ROW(48) ROW(64) ROW(96) ROW(128)
    return 0;
}
