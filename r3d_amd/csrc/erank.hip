// Effective rank of the fused token matrix via a batched one-sided (Hestenes) Jacobi SVD.
//
// Build-side addition: the reference has NO SVD / effective-rank code; README.md:8-14 only describes
// the quantity.  Definition used everywhere in this repo (Roy & Vetterli): sigma = svd(X[R,C]),
// p = sigma / sum(sigma), erank = exp(-sum p log p).  The checker is torch.linalg.svdvals on the oracle's fused
// features (oracle/futr_oracle.py: effective_rank).
//
// Kernel: one workgroup (16 waves) per matrix.  The C column vectors (length R) are held TRANSPOSED in LDS
// ([C][R|1] floats, odd row stride -> conflict-free transposing fill, coalesced HBM column sweeps on load/store);
// each round of a round-robin tournament gives every wave disjoint column pairs: three wave-reduced dot products
// (alpha, beta, gamma), one Givens rotation applied in LDS.  Sweeps repeat until no pair exceeds
// |gamma| > tol * sqrt(alpha * beta), tol = sqrt(R) * eps_f32 (the sgesvj criterion), or 30 sweeps.
// Outputs: singular values (unsorted), entropy, erank, sweep count and optionally the rotated columns
// Af^T = (X V)^T [C][R], from which the backward  dX = Af diag(g / sigma^3) (Af^T X)  is two MFMA GEMMs
// (U diag(g) V^T with V^T = Sigma^-2 Af^T X; no accumulation of V in the sweep).
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

constexpr int kJacWaves = 16;

struct ErankArgs {
    const float* x; int ld; long long batch_stride;      // [batch][R][ld]
    int R, C;
    float* sigma;          // [batch][C]
    float* af_t;           // [batch][C][R] or NULL
    float* stats;          // [batch][4] = {erank, entropy, sum sigma, sweeps}
    int sqrt_out;          // input is a Gram matrix: report sqrt of its singular values
    int max_sweeps;
};

__global__ __launch_bounds__(64 * kJacWaves) void erank_jacobi_kernel(const ErankArgs a) {
    extern __shared__ __attribute__((aligned(16))) float A[];     // [Cp][Rp]
    __shared__ int rotated;
    __shared__ float wred[kJacWaves];
    const int R = a.R, C = a.C;
    const int Rp = R | 1;
    const int Cp = (C + 1) & ~1;                                   // even number of players (last may be a dummy)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = a.x + (size_t)blockIdx.x * a.batch_stride;

    for (int e = tid; e < R * C; e += 64 * kJacWaves) {
        const int r = e / C, c = e % C;
        A[c * Rp + r] = X[(size_t)r * a.ld + c];
    }
    if (Cp != C)
        for (int r = tid; r < R; r += 64 * kJacWaves) A[C * Rp + r] = 0.f;
    __syncthreads();

    // ||X||_F^2: columns whose squared norm falls below (1e-6 ||X||_F)^2 are numerically zero (rank-deficient
    // input, e.g. R < C) and are not rotated against each other -- their mutual "angles" are rounding noise.
    float fpart = 0.f;
    for (int c = wave; c < C; c += kJacWaves)
        for (int r = lane; r < R; r += 64) { const float u = A[c * Rp + r]; fpart += u * u; }
    fpart = wave_sum(fpart);
    if (lane == 0) wred[wave] = fpart;
    __syncthreads();
    float fro2 = 0.f;
    for (int w = 0; w < kJacWaves; ++w) fro2 += wred[w];
    __syncthreads();
    const float negl = fro2 * 1e-12f;
    const float tol = sqrtf((float)R) * 1.1920929e-7f;
    const int npairs = Cp / 2, nrounds = Cp - 1;
    int sweeps = 0;
    for (; sweeps < a.max_sweeps; ++sweeps) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        int my_rot = 0;
        for (int rd = 0; rd < nrounds; ++rd) {
            for (int k = wave; k < npairs; k += kJacWaves) {
                int i, j;
                if (k == 0) { i = Cp - 1; j = rd; }
                else { i = (rd + k) % (Cp - 1); j = (rd - k + (Cp - 1)) % (Cp - 1); }
                if (i > j) { const int t = i; i = j; j = t; }
                float* ai = A + i * Rp;
                float* aj = A + j * Rp;
                float al = 0.f, be = 0.f, ga = 0.f;
                for (int r = lane; r < R; r += 64) {
                    const float u = ai[r], v = aj[r];
                    al += u * u; be += v * v; ga += u * v;
                }
                al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
                if (fabsf(ga) > tol * sqrtf(al * be) && al > negl && be > negl) {
                    const float zeta = (be - al) / (2.f * ga);
                    const float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                    const float c = 1.f / sqrtf(1.f + t * t), s = c * t;
                    for (int r = lane; r < R; r += 64) {
                        const float u = ai[r], v = aj[r];
                        ai[r] = c * u - s * v;
                        aj[r] = s * u + c * v;
                    }
                    my_rot = 1;
                }
            }
            __syncthreads();
        }
        if (my_rot && lane == 0) rotated = 1;
        __syncthreads();
        const int any = rotated;
        __syncthreads();
        if (!any) { ++sweeps; break; }
    }

    // singular values, entropy, erank
    float* sig = a.sigma + (size_t)blockIdx.x * C;
    for (int c = wave; c < C; c += kJacWaves) {
        float s2 = 0.f;
        for (int r = lane; r < R; r += 64) { const float u = A[c * Rp + r]; s2 += u * u; }
        s2 = wave_sum(s2);
        float s = sqrtf(s2);
        if (a.sqrt_out) s = sqrtf(s);
        if (lane == 0) sig[c] = s;
    }
    __syncthreads();                         // sig[] written by this workgroup: visible after the barrier (same CU)
    __threadfence_block();
    float part = 0.f;
    for (int c = tid; c < C; c += 64 * kJacWaves) part += sig[c];
    part = wave_sum(part);
    if (lane == 0) wred[wave] = part;
    __syncthreads();
    float total = 0.f;
    for (int w = 0; w < kJacWaves; ++w) total += wred[w];
    __syncthreads();
    float ent = 0.f;
    for (int c = tid; c < C; c += 64 * kJacWaves) {
        const float p = sig[c] / total;
        if (p > 0.f) ent -= p * logf(p);
    }
    ent = wave_sum(ent);
    if (lane == 0) wred[wave] = ent;
    __syncthreads();
    if (tid == 0) {
        float H = 0.f;
        for (int w = 0; w < kJacWaves; ++w) H += wred[w];
        float* st = a.stats + (size_t)blockIdx.x * 4;
        st[0] = expf(H); st[1] = H; st[2] = total; st[3] = (float)sweeps;
    }
    if (a.af_t) {
        float* out = a.af_t + (size_t)blockIdx.x * C * R;
        for (int e = tid; e < R * C; e += 64 * kJacWaves) {
            const int c = e / R, r = e % R;
            out[e] = A[c * Rp + r];
        }
    }
}

// coef[k] = gout * d erank / d sigma_k / sigma_k^3, zero where sigma_k is negligible (rank-deficient directions
// carry no defined singular vectors; p log p -> 0 there: SURVEY.md Appendix A.11)
__global__ __launch_bounds__(256) void erank_coef_kernel(const float* sigma, const float* stats, const float* gout,
                                                         float* coef, int C) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C) return;
    const float er = stats[0], H = stats[1], total = stats[2];
    const float s = sigma[k];
    float smax = 0.f;
    for (int j = 0; j < C; ++j) smax = fmaxf(smax, sigma[j]);
    float c = 0.f;
    if (s > 1e-6f * smax && s > 0.f) {
        const float p = s / total;
        const float g = -er * (logf(p) + H) / total;
        c = (gout ? *gout : 1.f) * g / (s * s * s);
    }
    coef[k] = c;
}

__global__ __launch_bounds__(256) void scale_rows_kernel(float* x, int ld, int rows, int cols, const float* coef) {
    const size_t total = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        x[(size_t)r * ld + c] *= coef[r];
    }
}

}  // namespace r3d

using namespace r3d;

R3D_EXPORT int64_t r3d_erank_lds_bytes(int R, int C) { return (int64_t)((C + 1) & ~1) * (R | 1) * 4; }

/* Batched effective rank.  x: [batch][R][ld] (row-major matrices X[R,C]); sigma [batch][C] (unsorted);
 * af_t (optional) [batch][C][R] rotated columns for the backward; stats [batch][4] = {erank, entropy, sum sigma,
 * sweeps}.  gram != 0: x holds Gram matrices X^T X (R == C) and sigma reports sqrt of their singular values.
 * Returns R3D_EINVAL when the matrix does not fit the 160 KB LDS of one CU (see r3d_erank_lds_bytes). */
R3D_EXPORT int r3d_erank_jacobi(const float* x, int ld, int64_t batch_stride, int batch, int R, int C, int gram,
                                float* sigma, float* af_t, float* stats, int max_sweeps, void* stream) {
    R3D_REQUIRE(x && sigma && stats && batch > 0 && R > 0 && C > 0 && ld >= C);
    R3D_REQUIRE(!gram || R == C);
    const int64_t lds = r3d_erank_lds_bytes(R, C);
    R3D_REQUIRE(lds <= 160 * 1024 - 256);
    ErankArgs a{x, ld, (long long)batch_stride, R, C, sigma, af_t, stats, gram, max_sweeps > 0 ? max_sweeps : 30};
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)erank_jacobi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(erank_jacobi_kernel, dim3(batch), dim3(64 * kJacWaves), (size_t)lds, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* coef[k] = (*gout) * (d erank / d sigma_k) / sigma_k^3  for the backward  dX = Af diag(coef) (Af^T X). */
R3D_EXPORT int r3d_erank_bwd_coef(const float* sigma, const float* stats, const float* gout, float* coef, int C,
                                  void* stream) {
    R3D_REQUIRE(sigma && stats && coef && C > 0);
    hipLaunchKernelGGL(erank_coef_kernel, dim3(r3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sigma, stats, gout,
                       coef, C);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* x[r, :] *= coef[r] */
R3D_EXPORT int r3d_scale_rows(float* x, int ld, int rows, int cols, const float* coef, void* stream) {
    R3D_REQUIRE(x && coef && rows > 0 && cols > 0 && ld >= cols);
    const size_t total = (size_t)rows * cols;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, coef);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
