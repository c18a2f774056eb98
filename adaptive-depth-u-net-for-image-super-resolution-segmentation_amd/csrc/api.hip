// Error reporting + small utility entry points of the C ABI.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

static thread_local char g_err[512] = "";

int ad_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int ad_version(void) { return 1; }

// ---- the only process-wide state of the library besides the last error string: the CU count of the device (read once)
// and the options a caller sets explicitly
int ad_num_cu() {
    // per device (a process may drive several, one thread each; the one-time dynamic-LDS raises are per device too:
    // ad_first_on_device): the count sizes every persistent grid, workspace and the XCD
    // tile order.  Relaxed atomics: racing first calls compute the same value.
    static std::atomic<int> cache[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const int slot = dev < 16 ? dev : 15;
    int ncu = cache[slot].load(std::memory_order_relaxed);
    if (ncu == 0 || dev >= 15) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 64)
            ncu = v / 64 * 64;          // whole groups of 8 XCDs x 8 (the XCD-aware work order deals tiles in such groups)
        else
            ncu = 256;                  // no device visible (the CPU-only build check) or fewer than 64 CUs: MI355X / MI350X geometry
        cache[slot].store(ncu, std::memory_order_relaxed);
    }
    return ncu;
}

static int g_options[AD_OPT_COUNT] = {0, 0, 0, 0, 0};
static const char* const g_option_names[AD_OPT_COUNT] = {"no_map1", "no_map4", "no_dgrad_ln", "no_mosaic", "no_pw_wide"};
int ad_option(int which) { return which >= 0 && which < AD_OPT_COUNT ? g_options[which] : 0; }

extern "C" int ad_set_option(const char* name, int value) {
    for (int i = 0; i < AD_OPT_COUNT; ++i)
        if (name && strcmp(name, g_option_names[i]) == 0) { g_options[i] = value; return AD_OK; }
    return ad_set_error(AD_ERR_ARG, "ad_set_option: unknown option '%s' (no_map1, no_map4, no_dgrad_ln, no_mosaic, no_pw_wide)", name ? name : "(null)");
}

extern "C" int ad_get_option(const char* name) {
    for (int i = 0; i < AD_OPT_COUNT; ++i)
        if (name && strcmp(name, g_option_names[i]) == 0) return g_options[i];
    return -1;
}

extern "C" int ad_device_cus(void) { return ad_num_cu(); }
extern "C" const char* ad_last_error(void) { return g_err; }
extern "C" int ad_cin_granule(int dtype) { return ad_is_half(dtype) ? 32 : 16; }

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = (TO)(float)x[i];
}

extern "C" int ad_cast(const void* x, int dtype_in, void* y, int dtype_out, int64_t count, void* stream) {
    if (count <= 0) return AD_OK;
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (!ad_dtype_ok(dtype_in) || !ad_dtype_ok(dtype_out))
        return ad_set_error(AD_ERR_ARG, "ad_cast: bad dtype %d -> %d", dtype_in, dtype_out);
    AD_DISPATCH_DTYPE(dtype_in, TI_,
        AD_DISPATCH_DTYPE(dtype_out, TO_, cast_kernel<TI_, TO_><<<blocks, 256, 0, s>>>((const TI_*)x, (TO_*)y, count);))
    AD_LAUNCH_CHECK("ad_cast");
    return AD_OK;
}

// one thread per 16-byte output vector (EPT channels): the 3 real channels sit in the first vector of a pixel
template <typename T>
__global__ void pad_channels_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t npix, int c, int cpad) {
    constexpr int EPT = ElemTraits<T>::EPT;
    const int vecs = cpad / EPT;
    const int64_t total = npix * vecs;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t p = i / vecs;
        const int v = (int)(i - p * vecs);
        float f[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int ch = v * EPT + e;
            f[e] = ch < c ? x[p * c + ch] : 0.0f;
        }
        Vec16<T> st;
        st.from_f32(f);
        st.store(y + i * EPT);
    }
}

extern "C" int ad_pad_channels(const float* x, void* y, int64_t npix, int c, int cpad, int dtype, void* stream) {
    AD_REQUIRE(ad_dtype_ok(dtype), "ad_pad_channels: bad dtype %d", dtype);
    AD_REQUIRE(c > 0 && cpad >= c && cpad % (ad_is_half(dtype) ? 8 : 4) == 0, "ad_pad_channels: c=%d cpad=%d", c, cpad);
    if (npix <= 0) return AD_OK;
    hipStream_t s = (hipStream_t)stream;
    int64_t total = npix * (cpad / (ad_is_half(dtype) ? 8 : 4));
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    AD_DISPATCH_DTYPE(dtype, T_, pad_channels_kernel<T_><<<blocks, 256, 0, s>>>(x, (T_*)y, npix, c, cpad);)
    AD_LAUNCH_CHECK("ad_pad_channels");
    return AD_OK;
}


// ---- feed path helpers (pipeline.DeviceDegrader): all fp32, one thread per pixel
// hr3[p][0..2] = u8[p][0..2] / 255 ; hr4[p] = (r, g, b, 0): the float HR batch and its 16-byte-vector copy for ad_resample
__global__ void u8_feed_kernel(const unsigned char* __restrict__ x, float* __restrict__ hr3, float* __restrict__ hr4, int64_t npix) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const float r = x[p * 3] * (1.0f / 255.0f), g = x[p * 3 + 1] * (1.0f / 255.0f), b = x[p * 3 + 2] * (1.0f / 255.0f);
        hr3[p * 3] = r; hr3[p * 3 + 1] = g; hr3[p * 3 + 2] = b;
        reinterpret_cast<float4*>(hr4)[p] = make_float4(r, g, b, 0.f);
    }
}

// y[p][0..cpad) = clip(x[p][0..c), 0, 1) then zeros (degrade_image clips the HR patch first, shared/pipeline.py:84)
__global__ void pad_clip_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix, int c, int cpad) {
    const int64_t total = npix * cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / cpad;
        const int ch = (int)(i - p * cpad);
        y[i] = ch < c ? fminf(fmaxf(x[p * c + ch], 0.f), 1.f) : 0.f;
    }
}

__global__ void take_channels_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix, int cin, int cout) {
    const int64_t total = npix * cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / cout;
        y[i] = x[p * cin + (i - p * cout)];
    }
}

static int feed_blocks(int64_t work) { return (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192); }

extern "C" int ad_u8_to_float_pad(const void* x_u8, float* hr3, float* hr4, int64_t npix, void* stream) {
    AD_REQUIRE(x_u8 && hr3 && hr4 && npix > 0, "ad_u8_to_float_pad: bad arguments");
    u8_feed_kernel<<<feed_blocks(npix), 256, 0, (hipStream_t)stream>>>((const unsigned char*)x_u8, hr3, hr4, npix);
    AD_LAUNCH_CHECK("ad_u8_to_float_pad");
    return AD_OK;
}

extern "C" int ad_pad_clip_f32(const float* x, float* y, int64_t npix, int c, int cpad, void* stream) {
    AD_REQUIRE(x && y && npix > 0 && c > 0 && cpad >= c, "ad_pad_clip_f32: bad arguments");
    pad_clip_kernel<<<feed_blocks(npix * cpad), 256, 0, (hipStream_t)stream>>>(x, y, npix, c, cpad);
    AD_LAUNCH_CHECK("ad_pad_clip_f32");
    return AD_OK;
}

extern "C" int ad_take_channels(const float* x, float* y, int64_t npix, int cin, int cout, void* stream) {
    AD_REQUIRE(x && y && npix > 0 && cout > 0 && cin >= cout, "ad_take_channels: bad arguments");
    take_channels_kernel<<<feed_blocks(npix * cout), 256, 0, (hipStream_t)stream>>>(x, y, npix, cin, cout);
    AD_LAUNCH_CHECK("ad_take_channels");
    return AD_OK;
}
