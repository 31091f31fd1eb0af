// Micro-benchmark: what does a dependent phase boundary cost on MI355X?
//   A) P tiny dependent kernels captured in a hipGraph (what the training step's 33 small launches pay today)
//   B) ONE persistent kernel with P phases separated by a grid-wide barrier (monotonic atomic counter in device memory,
//      agent scope): the megakernel alternative.  Every workgroup does the same trivial read-modify-write on a small
//      buffer per phase, so the numbers are pure boundary cost.
// The barrier has a bounded spin (exits with an error flag instead of hanging if a workgroup never arrives).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void work(float* buf, int n, int phase) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = buf[i] * 0.999f + (float)phase;
}

__global__ void tiny(float* buf, int n, int phase) { work(buf, n, phase); }

__global__ void persistent(float* buf, int n, int phases, unsigned* counter, unsigned* err) {
    for (int p = 0; p < phases; ++p) {
        work(buf, n, p);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(p + 1) * gridDim.x;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 20000000u) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

int main() {
    const int n = 64 * 1024, P = 32;
    float* buf; unsigned *counter, *err;
    CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMemset(buf, 0, n * sizeof(float)));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {8, 32, 64, 128, 256}) {
        // A: graph of P dependent tiny kernels
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int p = 0; p < P; ++p) hipLaunchKernelGGL(tiny, dim3(G), dim3(256), 0, s, buf, n, p);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float msA; CK(hipEventElapsedTime(&msA, e0, e1));
        // B: one persistent kernel, P phases (G <= 256 workgroups of 256 threads are co-resident on 256 CUs)
        float msB = 0.f;
        for (int i = 0; i < 23; ++i) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            if (i == 3) CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, s, buf, n, P, counter, err);
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&msB, e0, e1));
        unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("G=%3d workgroups: graph of %d tiny kernels %.2f us/phase | persistent kernel %.2f us/phase (err=%u)\n", G, P,
               msA * 1e3 / 20 / P, msB * 1e3 / 20 / P, herr);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
