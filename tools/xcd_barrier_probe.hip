// Micro-benchmark: a phase boundary between workgroups that all live on ONE XCD (one shared L2).
// Workgroups are dealt round-robin over the 8 XCDs, so of a grid of 8*G workgroups the ones with blockIdx.x % 8 == 0 sit on
// the same XCD; the others leave at once.  The barrier is a monotonic counter bumped with an L2-local atomic (no sc1: the
// L2 of the XCD is the point of coherence for its own CUs), polled with a returning atomic (never served by the vL1D);
// data hand-over = vmcnt(0) on the writer side, a vL1D invalidate on the reader side.  Every phase each workgroup
// rewrites its 1 KB slot and checks the slot its neighbour wrote in the previous phase (visibility check).
// Slots are double-buffered by phase parity (a neighbour may run one phase ahead).
// Variants: 3 = L2-local arrive counter + a release flag written by the last arriver and polled with sc0 loads,
//           0 = L2-local (buffer_inv sc0 only), 1 = agent-scope fences (buffer_wbl2 sc1 / buffer_inv sc1), same XCD,
//           2 = agent-scope fences, workgroups spread over all XCDs (the round-1 measurement).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VAR>
__global__ __launch_bounds__(256) void persistent(float* buf, int G, int phases, unsigned* counter, unsigned* err, unsigned* xcc) {
    int w;
    if (VAR == 2) { w = blockIdx.x; if (w >= G) return; }
    else { if (blockIdx.x & 7) return; w = blockIdx.x >> 3; }
    if (threadIdx.x == 0) xcc[w] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf;
    const int t = threadIdx.x;
    for (int p = 0; p < phases; ++p) {
        if (p > 0) {
            const float v = buf[(((p - 1) & 1) * 64 + (w + 1) % G) * 256 + t];
            if (v != (float)(p - 1)) atomicAdd(err + 1, 1u);
        }
        __syncthreads();            // (everyone has read before the slot is rewritten... by its owner only: no hazard)
        buf[((p & 1) * 64 + w) * 256 + t] = (float)p;
        if (VAR == 0 || VAR == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (t == 0) {
            const unsigned target = (unsigned)(p + 1) * G;
            if (VAR == 3) {
                unsigned one = 1u, got, flag;
                asm volatile("global_atomic_add %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(got) : "v"(counter), "v"(one) : "memory");
                if (got == target - 1) {
                    unsigned ph = (unsigned)(p + 1);
                    asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(counter + 16), "v"(ph) : "memory");
                } else {
                    unsigned spins = 0;
                    do {
                        asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(flag) : "v"(counter + 16) : "memory");
                        if (++spins > 200000u) { *err = 1; break; }
                    } while (flag < (unsigned)(p + 1));
                }
            } else if (VAR == 0) {
                unsigned one = 1u, zero = 0u, got;
                asm volatile("global_atomic_add %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(counter), "v"(one) : "memory");
                unsigned spins = 0;
                do {
                    asm volatile("global_atomic_add %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(got) : "v"(counter), "v"(zero) : "memory");
                    if (++spins > 20000u) { *err = 1; break; }
                } while (got < target && *((volatile unsigned*)err) == 0);
            } else {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 20000u || *((volatile unsigned*)err) != 0) { *err = 1; break; }
                }
            }
        }
        __syncthreads();
        if (VAR == 0 || VAR == 3) asm volatile("buffer_inv sc0" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

template <int VAR>
static int run(int G, float* buf, unsigned* counter, unsigned* err, unsigned* xcc, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int P = 32;
    float ms = 0.f;
    CK(hipMemsetAsync(err, 0, 8, s));
    for (int i = 0; i < 23; ++i) {
        CK(hipMemsetAsync(counter, 0, 128, s));
        if (i == 3) CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(persistent<VAR>, dim3(VAR == 2 ? G : 8 * G), dim3(256), 0, s, buf, G, P, counter, err, xcc);
    }
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr[2], hx[64];
    CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx, xcc, sizeof(unsigned) * (G < 64 ? G : 64), hipMemcpyDeviceToHost));
    unsigned mask = 0;
    for (int i = 0; i < (G < 64 ? G : 64); ++i) mask |= 1u << hx[i];
    printf("variant %d G=%2d: %.2f us/phase (incl. ~%.2f us launch+memset amortised over %d phases) timeout=%u stale=%u xcc mask=0x%x\n",
           VAR, G, ms * 1e3 / 20 / P, 0.0, P, herr[0], herr[1], mask);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    float* buf; unsigned *counter, *err, *xcc;
    CK(hipMalloc(&buf, 2 * 64 * 256 * sizeof(float))); CK(hipMemset(buf, 0, 2 * 64 * 256 * sizeof(float)));
    CK(hipMalloc(&counter, 128)); CK(hipMalloc(&err, 8)); CK(hipMalloc(&xcc, 64 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {8, 16, 32}) {
        if ((only < 0 || only == 2) && run<2>(G, buf, counter, err, xcc, s, e0, e1)) return 1;
        if ((only < 0 || only == 1) && run<1>(G, buf, counter, err, xcc, s, e0, e1)) return 1;
        if ((only < 0 || only == 3) && run<3>(G, buf, counter, err, xcc, s, e0, e1)) return 1;
        if ((only < 0 || only == 0) && run<0>(G, buf, counter, err, xcc, s, e0, e1)) return 1;
    }
    return 0;
}
