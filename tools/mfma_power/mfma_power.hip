// Power-limited throughput of the matrix pipes by MFMA shape and operand source (diagnostic; built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/mfma_power/mfma_power.hip && /tmp/mfma_power
// A workgroup of 4 waves (one per SIMD) per CU x 2; every wave owns a 64 x 64 fp32 accumulator tile, as the conv kernels'
// MFMA waves do, and multiplies random bf16 operands into it:
//   v16   16 v_mfma_f32_16x16x32_bf16 per step, operands constant in registers
//   v32    8 v_mfma_f32_32x32x16_bf16 per step (the same FLOPs), operands in registers
//   v16l  as v16, the four A and four B fragments of a step re-read from LDS every step (8 ds_read_b128: the conv kernels' ratio)
//   v32l  as v32, its fragments re-read from LDS every step (8 ds_read_b128)
// Prints TFLOP/s of each (HIP events over a few hundred ms, after a warm-up long enough for the clock to settle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void burn(const bf16x8* __restrict__ src, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) bf16x8 lds[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(tid * 8 + i) & 4095];
        b[i] = src[(tid * 8 + 4 + i) & 4095];
        lds[tid * 8 + i] = a[i];
        lds[tid * 8 + 4 + i] = b[i];
    }
    __syncthreads();
    float sum = 0.f;
    if (MODE == 0 || MODE == 2) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < steps; ++s) {
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = lds[((tid + s) & 255) * 8 + i];
                    b[i] = lds[((tid + s) & 255) * 8 + 4 + i];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = lds[((tid + s) & 255) * 8 + i];
                    b[i] = lds[((tid + s) & 255) * 8 + 4 + i];
                }
            }
            // 64 x 64 x 32 per step: two k-halves (a[2 kh + i], b[2 kh + j]) x 2 x 2 tiles of 32 x 32 x 16
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * kh + i], b[2 * kh + j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
    }
    if (sum == 12345.678f) out[blockIdx.x * 256 + tid] = sum;      // keeps the accumulators alive
    (void)lane;
}

template <int MODE>
double run(const bf16x8* src, float* out, int steps, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) burn<MODE><<<512, 256>>>(src, out, steps);      // warm-up: lets the clock settle
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) burn<MODE><<<512, 256>>>(src, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 64 * 64 * 32 * (double)steps * 4 * 512 * reps;     // per wave-step 64 x 64 x 32 MACs
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1234);
    for (auto& v : h) {               // random bf16 in about [-2, 2): sign, exponent 124 .. 127, random mantissa
        const unsigned s = rand() & 1, e = 124 + (rand() & 3), m = rand() & 127;
        v = (unsigned short)((s << 15) | (e << 7) | m);
    }
    bf16x8* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int steps = 4000, reps = 300;
    printf("v16  (16x16x32, registers)     %7.0f TFLOP/s\n", run<0>(src, out, steps, reps));
    printf("v32  (32x32x16, registers)     %7.0f TFLOP/s\n", run<1>(src, out, steps, reps));
    printf("v16l (16x16x32, LDS fragments) %7.0f TFLOP/s\n", run<2>(src, out, steps, reps));
    printf("v32l (32x32x16, LDS fragments) %7.0f TFLOP/s\n", run<3>(src, out, steps, reps));
    printf("v16  again                     %7.0f TFLOP/s\n", run<0>(src, out, steps, reps));
    return 0;
}
