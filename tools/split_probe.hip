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
__device__ inline float bf(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

__global__ void probe(const float* x, int n, float* err_pair, float* err_one) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned h, m, l;
    pair_split(x0, x1, h, m, l);
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
    probe<<<n / 2 / 256, 256>>>(x, n, e1, e2);
    std::vector<float> h1(n), h2(n);
    hipMemcpy(h1.data(), e1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), e2, n * 4, hipMemcpyDeviceToHost);
    float m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m1 = fmaxf(m1, h1[i]); m2 = fmaxf(m2, h2[i]); }
    printf("max |h+m+l - x| / |x| over %d values: pair-wise (v_perm) %.3e   one value (shifts) %.3e\n", n, m1, m2);
    return 0;
}
