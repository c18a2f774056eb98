// Keras-form Adam over flat fp32 buffers (tf.keras.optimizers.Adam as compiled at
// Super_resolution/code/train_adaptive_unet.py:489-494).  HBM-bound: 28 bytes per parameter.
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                   float alpha_host, const float* __restrict__ alpha_dev, float omb1,
                                                   float omb2, float eps, float gscale) {
    const float alpha = alpha_dev ? alpha_dev[0] : alpha_host;
    const int64_t nvec = count / 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pa = reinterpret_cast<float*>(&pp);
        float* ga = reinterpret_cast<float*>(&gg);
        float* ma = reinterpret_cast<float*>(&mm);
        float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gr = ga[e] * gscale;
            ma[e] += (gr - ma[e]) * omb1;
            va[e] += (gr * gr - va[e]) * omb2;
            pa[e] -= ma[e] * alpha / (sqrtf(va[e]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        float gr = g[i] * gscale;
        float mv = m[i] + (gr - m[i]) * omb1;
        float vv = v[i] + (gr * gr - v[i]) * omb2;
        m[i] = mv;
        v[i] = vv;
        p[i] -= mv * alpha / (sqrtf(vv) + eps);
    }
}

}  // namespace

extern "C" int ad_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float b1, float b2,
                            float eps, int step, float gscale, void* stream) {
    AD_REQUIRE(count >= 0 && step >= 1, "ad_adam_step: count=%ld step=%d", (long)count, step);
    AD_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "ad_adam_step: buffers must be 16-byte aligned");
    if (count == 0) return AD_OK;
    // alpha = lr * sqrt(1 - b2^t) / (1 - b1^t), evaluated in double then rounded once
    double alpha = (double)lr * sqrt(1.0 - pow((double)b2, step)) / (1.0 - pow((double)b1, step));
    int64_t nvec = (count + 3) / 4;
    int blocks = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
    adam_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, count, (float)alpha, nullptr, 1.0f - b1, 1.0f - b2, eps, gscale);
    AD_LAUNCH_CHECK("ad_adam_step");
    return AD_OK;
}

extern "C" float ad_adam_alpha(float lr, float b1, float b2, int step) {
    return (float)((double)lr * sqrt(1.0 - pow((double)b2, step)) / (1.0 - pow((double)b1, step)));
}

extern "C" int ad_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t count, const float* alpha_dev,
                                float b1, float b2, float eps, float gscale, void* stream) {
    AD_REQUIRE(count >= 0 && alpha_dev != nullptr, "ad_adam_step_dev: count=%ld", (long)count);
    AD_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "ad_adam_step_dev: buffers must be 16-byte aligned");
    if (count == 0) return AD_OK;
    int64_t nvec = (count + 3) / 4;
    int blocks = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
    adam_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, count, 0.f, alpha_dev, 1.0f - b1, 1.0f - b2, eps, gscale);
    AD_LAUNCH_CHECK("ad_adam_step_dev");
    return AD_OK;
}
