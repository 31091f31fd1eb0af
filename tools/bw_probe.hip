// Micro-benchmark: achievable read bandwidth for the depth-projection operand access pattern on MI355X.
//   mode 0: linear float4 stream (the AdamW pattern)
//   mode 1: GEMM pattern: each workgroup reads a [ROWS x SEG floats] panel per step, stepping along k (row pitch 50176)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void linear_read(const float4* __restrict__ p, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// grid = (row_tiles, splits); each WG: rows [rt*ROWS, +ROWS), k range [s*kps, (s+1)*kps) in steps of SEG floats
template <int ROWS, int SEG>
__global__ void panel_read(const float* __restrict__ p, int ld, int kps, float* out) {
    const int rt = blockIdx.x, s = blockIdx.y;
    constexpr int Q4 = SEG / 4;
    constexpr int NLD = ROWS * Q4 / 256;
    float acc = 0.f;
    for (int k0 = s * kps; k0 < (s + 1) * kps; k0 += SEG) {
        float4 v[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int f = threadIdx.x + q * 256;
            const int row = rt * ROWS + f / Q4, k = k0 + (f % Q4) * 4;
            v[q] = *reinterpret_cast<const float4*>(p + (size_t)row * ld + k);
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const int R = 128, K = 50176;
    const size_t n = (size_t)R * K;
    float *a, *b, *out;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn, const char* name, double bytes) {
        for (int i = 0; i < 3; ++i) fn();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.2f us  %7.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    timeit([&] { hipLaunchKernelGGL(linear_read, dim3(2048), dim3(256), 0, 0, (const float4*)a, n / 4, out);
                 hipLaunchKernelGGL(linear_read, dim3(2048), dim3(256), 0, 0, (const float4*)b, n / 4, out); }, "linear float4, 2 x 25.7 MB", 2.0 * n * 4);
    timeit([&] { hipLaunchKernelGGL((panel_read<64, 64>), dim3(2, 64), dim3(256), 0, 0, a, K, 784, out);
                 hipLaunchKernelGGL((panel_read<64, 64>), dim3(2, 64), dim3(256), 0, 0, b, K, 784, out); }, "panel 64 rows x 64 floats, 128 WGs each", 2.0 * n * 4);
    timeit([&] { hipLaunchKernelGGL((panel_read<64, 64>), dim3(2, 392), dim3(256), 0, 0, a, K, 128, out);
                 hipLaunchKernelGGL((panel_read<64, 64>), dim3(2, 392), dim3(256), 0, 0, b, K, 128, out); }, "panel 64x64, 784 WGs each", 2.0 * n * 4);
    timeit([&] { hipLaunchKernelGGL((panel_read<64, 256>), dim3(2, 196), dim3(256), 0, 0, a, K, 256, out);
                 hipLaunchKernelGGL((panel_read<64, 256>), dim3(2, 196), dim3(256), 0, 0, b, K, 256, out); }, "panel 64 rows x 256 floats, 392 WGs each", 2.0 * n * 4);
    timeit([&] { hipLaunchKernelGGL((panel_read<32, 256>), dim3(4, 196), dim3(256), 0, 0, a, K, 256, out);
                 hipLaunchKernelGGL((panel_read<32, 256>), dim3(4, 196), dim3(256), 0, 0, b, K, 256, out); }, "panel 32 rows x 256 floats, 784 WGs each", 2.0 * n * 4);
    timeit([&] { hipLaunchKernelGGL((panel_read<16, 1024>), dim3(8, 49), dim3(256), 0, 0, a, K, 1024, out);
                 hipLaunchKernelGGL((panel_read<16, 1024>), dim3(8, 49), dim3(256), 0, 0, b, K, 1024, out); }, "panel 16 rows x 1024 floats, 392 WGs each", 2.0 * n * 4);
    // one launch reading both operands like the GEMM does (2x2 tiles of 64x64 per k-slab: each panel read by 2 WGs)
    return 0;
}
