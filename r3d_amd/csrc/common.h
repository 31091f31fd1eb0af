// Internal helpers shared by the gfx950 kernels of libr3d_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define R3D_OK 0
#define R3D_EINVAL (-1)     // bad argument (null pointer, negative size, unsupported shape)
#define R3D_EALIGN (-2)     // pointer / leading dimension violates an alignment the entry point documents

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// exact-erf GELU (nn.GELU(approximate='none'), model/extras/transformerblock.py:80,86)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

}  // namespace r3d
