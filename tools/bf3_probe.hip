// Probe: accuracy of the bf16x3 product on v_mfma_f32_16x16x32_bf16 against fp64 for one 16 x 16 x 128 tile.
// hipcc --offload-arch=gfx950 -O3 -o tools/_build/bf3_probe tools/bf3_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split1(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    h = (unsigned short)(u >> 16);
    const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);
    const unsigned ur = __builtin_bit_cast(unsigned, r);
    m = (unsigned short)(ur >> 16);
    const float q = r - __builtin_bit_cast(float, ur & 0xffff0000u);
    l = (unsigned short)(__builtin_bit_cast(unsigned, q) >> 16);
}
__device__ inline void split8(const float* x, uint4& h, uint4& m, uint4& l) {
    unsigned hh[4], mm[4], ll[4];
    for (int p = 0; p < 4; ++p) {
        unsigned short h0, m0, l0, h1, m1, l1;
        split1(x[2 * p], h0, m0, l0); split1(x[2 * p + 1], h1, m1, l1);
        hh[p] = h0 | ((unsigned)h1 << 16); mm[p] = m0 | ((unsigned)m1 << 16); ll[p] = l0 | ((unsigned)l1 << 16);
    }
    h = make_uint4(hh[0], hh[1], hh[2], hh[3]); m = make_uint4(mm[0], mm[1], mm[2], mm[3]); l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}
#define B8(x) __builtin_bit_cast(bf16x8_t, x)
// mode 0: six products small-first; 1: six products, each class into its own accumulator, summed at the end; 2: nine products
__global__ void probe(const float* A, const float* B, float* C, int mode) {
    const int lane = threadIdx.x, li = lane & 15, q = lane >> 4;
    f32x4 acc = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    for (int s = 0; s < 4; ++s) {
        uint4 ah, am, al, bh, bm, bl;
        split8(A + li * 128 + 32 * s + 8 * q, ah, am, al);
        split8(B + li * 128 + 32 * s + 8 * q, bh, bm, bl);
        if (mode == 1) {
            s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(ah), B8(bl), s2, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(al), B8(bh), s2, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(am), B8(bm), s2, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(ah), B8(bm), s1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(am), B8(bh), s1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(ah), B8(bh), acc, 0, 0, 0);
        } else {
            if (mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(al), B8(bl), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(am), B8(bl), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(al), B8(bm), acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(ah), B8(bl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(al), B8(bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(am), B8(bm), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(ah), B8(bm), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(am), B8(bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(ah), B8(bh), acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + li] = mode == 1 ? (acc[i] + (s1[i] + s2[i])) : acc[i];
}
int main() {
    std::vector<float> A(16 * 128), B(16 * 128), C(256);
    srand(7);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0, w32 = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double r = 0; float f = 0;
            for (int k = 0; k < 128; ++k) { r += (double)A[i * 128 + k] * (double)B[j * 128 + k]; f += A[i * 128 + k] * B[j * 128 + k]; }
            worst = fmax(worst, fabs(C[i * 16 + j] - r)); scale = fmax(scale, fabs(r)); w32 = fmax(w32, fabs((double)f - r));
        }
        printf("mode %d: max err / scale %.3e   (plain fp32 loop: %.3e)\n", mode, worst / scale, w32 / scale);
    }
    return 0;
}
