// Exactness probe of the two three-way bf16 splits (gemm_bf3.hip's pair-wise split3_pair, chain_bf3.h's bf3_split1):
// for random fp32 values x (both signs, 40 binades), h + m + l must reconstruct x EXACTLY in fp32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I r3d_amd/csrc -o tools/_build/split_probe tools/split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "chain_bf3.h"

__device__ __forceinline__ void pair_split(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u);
    const float r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u);
    const float q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// the two subtractions of a level as one v_pk_add_f32, written on a 2-vector
__device__ __forceinline__ void pair_split_vec(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2_t x = {x0, x1};
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const f32x2_t hf = {__builtin_bit_cast(float, u0 & 0xffff0000u), __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const f32x2_t r = x - hf;
    const float r0 = r[0], r1 = r[1];      // (__builtin_bit_cast straight on a vector ELEMENT reads element 0 for both)
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const f32x2_t mf = {__builtin_bit_cast(float, v0 & 0xffff0000u), __builtin_bit_cast(float, v1 & 0xffff0000u)};
    const f32x2_t q = r - mf;
    const float q0 = q[0], q1 = q[1];
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
}
// ... and with the packed subtraction as inline assembly
__device__ __forceinline__ f32x2_t pk_sub(f32x2_t a, f32x2_t b) {
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void pair_split_asm(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2_t x = {x0, x1};
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const f32x2_t hf = {__builtin_bit_cast(float, u0 & 0xffff0000u), __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const f32x2_t r = pk_sub(x, hf);
    const float r0 = r[0], r1 = r[1];      // (__builtin_bit_cast straight on a vector ELEMENT reads element 0 for both)
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const f32x2_t mf = {__builtin_bit_cast(float, v0 & 0xffff0000u), __builtin_bit_cast(float, v1 & 0xffff0000u)};
    const f32x2_t q = pk_sub(r, mf);
    const float q0 = q[0], q1 = q[1];
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
}
__device__ inline float bf(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

template <int V>
__global__ void probe(const float* x, int n, float* err_pair, float* err_one) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned h, m, l;
    if (V == 0) pair_split(x0, x1, h, m, l);
    if (V == 1) pair_split_vec(x0, x1, h, m, l);
    if (V == 2) pair_split_asm(x0, x1, h, m, l);
    const float y0 = (bf(h & 0xffff) + bf(m & 0xffff)) + bf(l & 0xffff);
    const float y1 = (bf(h >> 16) + bf(m >> 16)) + bf(l >> 16);
    err_pair[2 * i] = fabsf(y0 - x0) / fmaxf(fabsf(x0), 1e-30f);
    err_pair[2 * i + 1] = fabsf(y1 - x1) / fmaxf(fabsf(x1), 1e-30f);
    unsigned short a, b, c;
    r3d::bf3_split1(x0, a, b, c);
    err_one[2 * i] = fabsf(((bf(a) + bf(b)) + bf(c)) - x0) / fmaxf(fabsf(x0), 1e-30f);
    r3d::bf3_split1(x1, a, b, c);
    err_one[2 * i + 1] = fabsf(((bf(a) + bf(b)) + bf(c)) - x1) / fmaxf(fabsf(x1), 1e-30f);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const float mant = 1.f + (float)rand() / (float)RAND_MAX;
        hx[i] = ldexpf(mant, rand() % 40 - 20) * ((rand() & 1) ? 1.f : -1.f);
    }
    float *x, *e1, *e2;
    hipMalloc(&x, n * 4); hipMalloc(&e1, n * 4); hipMalloc(&e2, n * 4);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> h1(n), h2(n);
    const char* names[3] = {"pair-wise, scalar subtractions", "pair-wise, 2-vector subtraction, elements laundered",
                            "pair-wise, v_pk_add_f32 as inline assembly"};
    for (int v = 0; v < 3; ++v) {
        if (v == 0) probe<0><<<n / 2 / 256, 256>>>(x, n, e1, e2);
        if (v == 1) probe<1><<<n / 2 / 256, 256>>>(x, n, e1, e2);
        if (v == 2) probe<2><<<n / 2 / 256, 256>>>(x, n, e1, e2);
        hipMemcpy(h1.data(), e1, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), e2, n * 4, hipMemcpyDeviceToHost);
        float m1 = 0, m2 = 0;
        for (int i = 0; i < n; ++i) { m1 = fmaxf(m1, h1[i]); m2 = fmaxf(m2, h2[i]); }
        printf("max |h+m+l - x| / |x| over %d values: %-60s %.3e   (one value, shifts: %.3e)\n", n, names[v], m1, m2);
    }
    return 0;
}
