// Internal helpers shared by the gfx950 kernels of libr3d_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define R3D_OK 0
#define R3D_EINVAL (-1)     // bad argument (null pointer, negative size, unsupported shape)
#define R3D_EALIGN (-2)     // pointer / leading dimension violates an alignment the entry point documents
#define R3D_ENORCCL (-3)
#define R3D_ERCCL_BASE (-100)

#define R3D_EXPORT extern "C" __attribute__((visibility("default")))

// Entry points never throw and never synchronise: they validate, enqueue on `stream`, and return
// either R3D_OK, a negative argument error, or the positive hipError_t of the failed launch.
#define R3D_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

#define R3D_REQUIRE(cond) do { if (!(cond)) return R3D_EINVAL; } while (0)

static inline bool r3d_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int r3d_cdiv(int a, int b) { return (a + b - 1) / b; }

namespace r3d {

constexpr int kWave = 64;

// Wave-wide reductions without the LDS crossbar: a row_ror butterfly inside each 16-lane DPP row (4 x v_add_f32_dpp,
// every lane of a row ends with the row's value), then the four row values are read with v_readlane and combined in
// a fixed order -- ~12 instructions against six dependent ds_bpermute round trips for the __shfl_xor form (the row
// kernels of the step are latency chains of such reductions).  All 64 lanes must be active at the call (they are:
// every call site reduces over a whole wave with out-of-range lanes contributing the identity).
template <int CTRL> __device__ __forceinline__ float dpp_mov_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov_f<0x128>(v);          // row_ror:8
    v += dpp_mov_f<0x124>(v);          // row_ror:4
    v += dpp_mov_f<0x122>(v);          // row_ror:2
    v += dpp_mov_f<0x121>(v);          // row_ror:1
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov_f<0x128>(v));
    v = fmaxf(v, dpp_mov_f<0x124>(v));
    v = fmaxf(v, dpp_mov_f<0x122>(v));
    v = fmaxf(v, dpp_mov_f<0x121>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ double dpp_mov_d(double v, int) { return v; }
template <int CTRL> __device__ __forceinline__ double dpp_mov_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_mov_d<0x128>(v);
    v += dpp_mov_d<0x124>(v);
    v += dpp_mov_d<0x122>(v);
    v += dpp_mov_d<0x121>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// exact-erf GELU (nn.GELU(approximate='none'), model/extras/transformerblock.py:80,86)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

}  // namespace r3d
