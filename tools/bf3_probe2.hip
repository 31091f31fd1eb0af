// Probe of the chain_bf3.h primitives: A staged with bf3_store4 (mode 0) or bf3_store1 (mode 1), B as operand-order planes,
// one 16 x 16 x 128 tile through bf3_bload + bf3_chunk, against fp64.
#include "../r3d_amd/csrc/chain_bf3.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace r3d;
__global__ void planes(const float* B, unsigned short* dst) {      // N = 16, K = 128: tile 0, k-steps 0..3
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int n = lane & 15, k0 = 32 * s + 8 * (lane >> 4);
    unsigned hh[4], mm[4], ll[4];
    for (int p = 0; p < 4; ++p) {
        unsigned short h0, m0, l0, h1, m1, l1;
        bf3_split1(B[n * 128 + k0 + 2 * p], h0, m0, l0); bf3_split1(B[n * 128 + k0 + 2 * p + 1], h1, m1, l1);
        hh[p] = h0 | ((unsigned)h1 << 16); mm[p] = m0 | ((unsigned)m1 << 16); ll[p] = l0 | ((unsigned)l1 << 16);
    }
    uint4* d = reinterpret_cast<uint4*>(dst) + (size_t)s * (3 * 64) + lane;
    d[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]); d[64] = make_uint4(mm[0], mm[1], mm[2], mm[3]); d[128] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}
__global__ void probe(const float* A, const unsigned short* pl, float* C, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short img[3 * 16 * 136];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
    if (mode == 0) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + (tid >> 5) * 128 + 4 * (tid & 31));
        bf3_stage_tile(img, 136, 0, v, tid);
    } else {
        for (int e = tid; e < 16 * 128; e += 512) bf3_store1(img, 136, e / 128, e % 128, A[e]);
    }
    __syncthreads();
    if (tid >= 64) return;
    Bf3B b;
    bf3_bload<4>(b, pl, 4, 0, 0, lane);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    bf3_chunk<4>(img, 136, li, q, 0, b, acc0, acc1);
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + li] = acc0[i] + acc1[i];
}
int main() {
    std::vector<float> A(16 * 128), B(16 * 128), C(256);
    srand(7);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
    float *dA, *dB, *dC; unsigned short* dP;
    hipMalloc(&dA, 8192); hipMalloc(&dB, 8192); hipMalloc(&dC, 1024); hipMalloc(&dP, 4 * 3 * 64 * 16);
    hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(planes, dim3(1), dim3(256), 0, 0, dB, dP);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, dA, dP, dC, mode);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double r = 0;
            for (int k = 0; k < 128; ++k) r += (double)A[i * 128 + k] * (double)B[j * 128 + k];
            worst = fmax(worst, fabs(C[i * 16 + j] - r)); scale = fmax(scale, fabs(r));
        }
        printf("primitives, A staged with %s: max err / scale %.3e\n", mode == 0 ? "bf3_store4" : "bf3_store1", worst / scale);
    }
    return 0;
}
