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

// ---- dynamic loss scaling (Keras LossScaleOptimizer, the reference's mixed_float16 policy:
// Super_resolution/code/train_adaptive_unet.py:471-477).  The scaler lives in device memory so that a train step
// captured in a hipGraph carries it:  state[0] = scale, [1] = 1 / scale, [2] = finite steps since the last change,
// [3] = overflow flag of the current step, [4] = optimizer applications so far, [5] = skipped steps so far.
__global__ __launch_bounds__(256) void nonfinite_scan_kernel(const float* __restrict__ g, int64_t count, float* __restrict__ state) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.4028234663852886e38f);       // NaN or +-inf
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) state[3] = 1.f;    // every writer stores the same value: deterministic
}

__global__ void loss_scale_update_kernel(float* __restrict__ state, int growth_steps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[3] != 0.f) {                 // non-finite gradients: the step was skipped, halve the scale
        state[0] = fmaxf(state[0] * 0.5f, 1.f);
        state[2] = 0.f;
        state[5] += 1.f;
    } else {
        state[4] += 1.f;
        state[2] += 1.f;
        if (state[2] >= (float)growth_steps) { state[0] *= 2.f; state[2] = 0.f; }
    }
    state[1] = 1.f / state[0];
    state[3] = 0.f;
}

// Adam under the scaler: skipped when the overflow flag is set; otherwise the gradients are unscaled (state[1]) and
// the step index of the bias correction is the number of APPLIED updates (state[4] + 1), as Keras' inner optimizer
// counts them.  lr comes from device memory (graph replay).
__global__ __launch_bounds__(256) void adam_scaled_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                          const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                          float gscale_host, const float* __restrict__ state) {
    if (state[3] != 0.f) return;
    const double t = (double)state[4] + 1.0;
    const float alpha = (float)((double)lr_dev[0] * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    const float gscale = gscale_host * state[1];
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const float gr = g[i] * gscale;
        const float mv = m[i] + (gr - m[i]) * omb1;
        const float vv = v[i] + (gr * gr - v[i]) * omb2;
        m[i] = mv;
        v[i] = vv;
        p[i] -= mv * alpha / (sqrtf(vv) + eps);
    }
}

}  // namespace

extern "C" int ad_loss_scale_check(const float* grads, int64_t count, float* state, void* stream) {
    AD_REQUIRE(grads && state && count >= 0, "ad_loss_scale_check: NULL operand");
    if (count == 0) return AD_OK;
    int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    nonfinite_scan_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(grads, count, state);
    AD_LAUNCH_CHECK("ad_loss_scale_check");
    return AD_OK;
}

extern "C" int ad_loss_scale_update(float* state, int growth_steps, void* stream) {
    AD_REQUIRE(state && growth_steps > 0, "ad_loss_scale_update: bad arguments");
    loss_scale_update_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, growth_steps);
    AD_LAUNCH_CHECK("ad_loss_scale_update");
    return AD_OK;
}

extern "C" int ad_adam_step_scaled(float* p, const float* g, float* m, float* v, int64_t count, const float* lr_dev, float b1,
                                   float b2, float eps, float gscale, const float* state, void* stream) {
    AD_REQUIRE(count >= 0 && lr_dev && state, "ad_adam_step_scaled: NULL operand");
    if (count == 0) return AD_OK;
    int blocks = (int)((count + 255) / 256 < 8192 ? (count + 255) / 256 : 8192);
    adam_scaled_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, count, lr_dev, b1, b2, eps, gscale, state);
    AD_LAUNCH_CHECK("ad_adam_step_scaled");
    return AD_OK;
}

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
