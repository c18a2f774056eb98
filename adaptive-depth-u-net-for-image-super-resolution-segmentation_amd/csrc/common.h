// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/adunet.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

int ad_set_error(int code, const char* fmt, ...);
// Compute units of the current device, queried once (api.hip): the persistent kernels launch one workgroup per CU.
int ad_num_cu();
// One-time per-DEVICE set-up (hipFuncSetAttribute applies to the current device's function object: a process that drives several
// devices must raise the dynamic-LDS limit on each of them).  `done` holds one bit per device ordinal; true = this caller sets up.
// Racing first calls on one device both run the set-up, which is idempotent.
#include <atomic>
static inline bool ad_first_on_device(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;      // beyond 63 ordinals: set up on every call
    const unsigned long long bit = 1ULL << dev;
    if (dev != 63 && (done.load(std::memory_order_acquire) & bit)) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}
// Explicit library options (ad_set_option, include/adunet.h): the library itself never reads the environment.
enum { AD_OPT_NO_MAP1 = 0, AD_OPT_NO_MAP4 = 1, AD_OPT_NO_DGRAD_LN = 2, AD_OPT_NO_MOSAIC = 3, AD_OPT_NO_PW_WIDE = 4, AD_OPT_COUNT = 5 };
int ad_option(int which);

static inline bool ad_is_half(int dtype) { return dtype == AD_BF16 || dtype == AD_F16; }   // 16-bit storage types
static inline bool ad_dtype_ok(int dtype) { return dtype == AD_F32 || dtype == AD_BF16 || dtype == AD_F16; }

// The two 16-bit element types share every kernel; they differ in the vector types and in the MFMA instruction
// (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x32_f16), fp32 accumulation in both.
template <typename E> struct Half16;
template <> struct Half16<bf16_t> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half16<f16_t> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// Runs `...` with T_ bound to the element type of `dtype` (validated by the caller).
#define AD_DISPATCH_DTYPE(dtype, T_, ...)                           \
    switch (dtype) {                                                \
        case AD_BF16: { typedef bf16_t T_; __VA_ARGS__ } break;     \
        case AD_F16: { typedef f16_t T_; __VA_ARGS__ } break;       \
        default: { typedef float T_; __VA_ARGS__ } break;           \
    }

#define AD_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return ad_set_error(AD_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define AD_LAUNCH_CHECK(name)                                                            \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                            \
            return ad_set_error(AD_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_));   \
    } while (0)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int EPT = 4;  // elements per 16 bytes
    static constexpr int DT = AD_F32;
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int EPT = 8;
    static constexpr int DT = AD_BF16;
};
template <> struct ElemTraits<f16_t> {
    static constexpr int EPT = 8;
    static constexpr int DT = AD_F16;
};

// 16-byte vector of T, unpacked to / packed from fp32.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    float4 v;
    __device__ __forceinline__ void load(const void* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void to_f32(float* f) const { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    __device__ __forceinline__ void from_f32(const float* f) { v = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Vec16<bf16_t> {
    bf16x8 v;
    __device__ __forceinline__ void load(const void* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.0f;
    }
    __device__ __forceinline__ void to_f32(float* f) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    __device__ __forceinline__ void from_f32(const float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
    }
};

template <> struct Vec16<f16_t> {
    f16x8 v;
    __device__ __forceinline__ void load(const void* p) { v = *reinterpret_cast<const f16x8*>(p); }
    __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<f16x8*>(p) = v; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (f16_t)0.0f;
    }
    __device__ __forceinline__ void to_f32(float* f) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    __device__ __forceinline__ void from_f32(const float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (f16_t)f[i];
    }
};

#ifdef __HIPCC__
// Sum over the G lanes (a power of two up to 64, aligned) that own one pixel / row, returned in each of them.  Inside a row of
// 16 lanes the partners arrive as DPP operands of the adds (quad_perm for the lanes 1 and 2 away; once the four lanes of a quad
// agree, row_half_mirror / row_mirror deliver the other quad's / the other half row's sum) instead of ds_bpermute round trips
// through LDS (what __shfl_xor compiles to): head_ln_bwd_kernel, bound by its instruction count, had 15 of them per pixel.
template <int CTRL>
__device__ __forceinline__ float ad_dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int G>
__device__ __forceinline__ float ad_group_sum(float v) {
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "lane groups are powers of two");
    if (G >= 2) v += ad_dpp_f32<0xB1>(v);            // quad_perm [1, 0, 3, 2]
    if (G >= 4) v += ad_dpp_f32<0x4E>(v);            // quad_perm [2, 3, 0, 1]
    if (G >= 8) v += ad_dpp_f32<0x141>(v);           // row_half_mirror
    if (G >= 16) v += ad_dpp_f32<0x140>(v);          // row_mirror
#pragma unroll
    for (int o = 16; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif

template <typename A, typename B> struct ad_same_type { static constexpr bool value = false; };
template <typename A> struct ad_same_type<A, A> { static constexpr bool value = true; };

static inline int ad_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline size_t ad_align(size_t v, size_t a) { return (v + a - 1) / a * a; }
