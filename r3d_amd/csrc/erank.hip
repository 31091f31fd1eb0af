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

// ---------------------------------------------------------------------------------------------------------------
// Matrices that do not fit one CU's LDS: two-level (block) one-sided Jacobi.  The columns live transposed in HBM
// (At [Cpad][R], a few MB: L2 / Infinity-Cache resident), grouped in nblk blocks of b columns.  One sweep =
//   1 launch  "within": workgroup p orthogonalises the b columns of block p against each other,
//   nblk-1 launches "cross": round-robin over block pairs (p,q); a workgroup holds both blocks in LDS (2b columns)
//                    and rotates every (i in p, j in q) pair: b inner rounds of b disjoint pairs, one wave per pair.
// so every column pair is visited exactly once per sweep (a cyclic ordering -> the usual quadratic convergence).
// The launches of all sweeps are enqueued up front; a device-side flag turns the remainder into no-ops once a sweep
// made no rotation.  ctrl (ints): [0] done, [1] sweeps run, [2] float bits of ||X||_F^2, [4+s] rotations in sweep s.
struct ErankBlk {
    float* at; int R, C, b, nblk;      // nblk even (the last block may be a dummy one: index >= nreal)
    int nreal;
    int* ctrl;
};

__global__ __launch_bounds__(256) void erank_blk_init_kernel(const float* __restrict__ x, int ld, ErankBlk g) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    float part = 0.f;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        const float v = (r < g.R && c < g.C) ? x[(size_t)r * ld + c] : 0.f;
        tile[k][tx] = v;
        part += v * v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < g.nreal * g.b && r < g.R) g.at[(size_t)c * g.R + r] = tile[tx][k];
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<float*>(g.ctrl + 2), red[0] + red[1] + red[2] + red[3]);
}

// mode 0: within-block (grid = nreal), mode 1: cross, round rd of the block tournament (grid = nblk / 2)
__global__ __launch_bounds__(64 * kJacWaves) void erank_blk_round_kernel(ErankBlk g, int mode, int rd, int sweep) {
    extern __shared__ __attribute__((aligned(16))) float A[];     // [2b][Rp]
    if (g.ctrl[0]) return;
    const int R = g.R, b = g.b, Rp = R | 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int p, q;
    if (mode == 0) { p = blockIdx.x; q = -1; }
    else {
        const int k = blockIdx.x, n1 = g.nblk - 1;
        if (k == 0) { p = n1; q = rd; }
        else { p = (rd + k) % n1; q = (rd - k + n1) % n1; }
        if (p > q) { const int t = p; p = q; q = t; }
        if (q >= g.nreal) return;                                  // paired with the dummy block
    }
    const int ncol = (mode == 0) ? b : 2 * b;
    for (int e = tid; e < ncol * R; e += 64 * kJacWaves) {
        const int c = e / R, r = e - c * R;
        const int blk = c < b ? p : q;
        A[c * Rp + r] = g.at[((size_t)blk * b + (c % b)) * R + r];
    }
    __syncthreads();
    const float negl = __int_as_float(g.ctrl[2]) * 1e-12f;
    const float tol = sqrtf((float)R) * 1.1920929e-7f;
    int my_rot = 0;
    const int npairs = (mode == 0) ? b / 2 : b;
    const int nrounds = (mode == 0) ? b - 1 : b;
    for (int t = 0; t < nrounds; ++t) {
        for (int k = wave; k < npairs; k += kJacWaves) {
            int i, j;
            if (mode == 0) {
                if (k == 0) { i = b - 1; j = t; }
                else { i = (t + k) % (b - 1); j = (t - k + (b - 1)) % (b - 1); }
            } else { i = k; j = b + (k + t) % b; }
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
                const float tt = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                const float c = 1.f / sqrtf(1.f + tt * tt), s = c * tt;
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
    for (int e = tid; e < ncol * R; e += 64 * kJacWaves) {
        const int c = e / R, r = e - c * R;
        const int blk = c < b ? p : q;
        g.at[((size_t)blk * b + (c % b)) * R + r] = A[c * Rp + r];
    }
    if (my_rot && lane == 0) atomicAdd(g.ctrl + 4 + sweep, 1);
}

__global__ void erank_blk_sweep_end_kernel(ErankBlk g, int sweep) {
    if (g.ctrl[0]) return;
    g.ctrl[1] = sweep + 1;
    if (g.ctrl[4 + sweep] == 0) g.ctrl[0] = 1;
}

__global__ __launch_bounds__(64 * kJacWaves) void erank_blk_finish_kernel(ErankBlk g, float* sig, float* stats) {
    __shared__ float wred[kJacWaves];
    const int R = g.R, C = g.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < C; c += kJacWaves) {
        float s2 = 0.f;
        for (int r = lane; r < R; r += 64) { const float u = g.at[(size_t)c * R + r]; s2 += u * u; }
        s2 = wave_sum(s2);
        if (lane == 0) sig[c] = sqrtf(s2);
    }
    __syncthreads();
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
        stats[0] = expf(H); stats[1] = H; stats[2] = total; stats[3] = (float)g.ctrl[1];
    }
}

// coef[k] = gout * d erank / d sigma_k / sigma_k^3, zero where sigma_k is negligible (rank-deficient directions
// carry no defined singular vectors; p log p -> 0 there: SURVEY.md Appendix A.11)
__global__ __launch_bounds__(256) void erank_coef_kernel(const float* sigma, const float* stats, const float* gout,
                                                         float* coef, int C, int max_rank) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C) return;
    const float er = stats[0], H = stats[1], total = stats[2];
    const float s = sigma[k];
    float smax = 0.f;
    int larger = 0;                      // singular values above this one: only the min(R, C) largest are real
    for (int j = 0; j < C; ++j) {
        smax = fmaxf(smax, sigma[j]);
        larger += (sigma[j] > s || (sigma[j] == s && j < k)) ? 1 : 0;
    }
    float c = 0.f;
    if (s > 1e-6f * smax && s > 0.f && (max_rank <= 0 || larger < max_rank)) {
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


/* Block size (columns per block) of the two-level Jacobi for column length R: two blocks must fit one CU's LDS. */
static int erank_blk_b(int R) {
    int b = 16;
    while (b > 2 && (int64_t)2 * b * (R | 1) * 4 > 150 * 1024) b >>= 1;
    return b;
}

/* Sizes for r3d_erank_blocked: out[0] = floats of af_t ([Cpad][R], Cpad = C rounded up to the block size),
 * out[1] = ints of ctrl.  Returns R3D_EINVAL when even two 2-column blocks exceed the LDS (R > ~9500). */
R3D_EXPORT int r3d_erank_blocked_sizes(int R, int C, int max_sweeps, int64_t* out) {
    R3D_REQUIRE(out && R > 0 && C > 0);
    const int b = erank_blk_b(R);
    R3D_REQUIRE((int64_t)2 * b * (R | 1) * 4 <= 150 * 1024);
    const int nreal = r3d_cdiv(C, b);
    out[0] = (int64_t)nreal * b * R;
    out[1] = 4 + (max_sweeps > 0 ? max_sweeps : 30);
    return R3D_OK;
}

/* Effective rank of ONE matrix X[R, C] (row-major, leading dimension ld) of any size: two-level one-sided Jacobi
 * with the columns in HBM (see above).  af_t [Cpad][R] receives the rotated columns (X V)^T (rows >= C are zero
 * padding); ctrl is integer scratch; sizes from r3d_erank_blocked_sizes.  sigma [C], stats [4] as r3d_erank_jacobi.
 * Enqueues 2 + max_sweeps * (nblk + 1) launches on the stream, never synchronises. */
R3D_EXPORT int r3d_erank_blocked(const float* x, int ld, int R, int C, float* sigma, float* af_t, int* ctrl, float* stats,
                                 int max_sweeps, void* stream) {
    R3D_REQUIRE(x && sigma && af_t && ctrl && stats && R > 0 && C > 0 && ld >= C);
    const int b = erank_blk_b(R);
    const int64_t lds = (int64_t)2 * b * (R | 1) * 4;
    R3D_REQUIRE(lds <= 150 * 1024);
    const int ms = max_sweeps > 0 ? max_sweeps : 30;
    const int nreal = r3d_cdiv(C, b);
    const int nblk = (nreal + 1) & ~1;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ctrl, 0, sizeof(int) * (4 + ms), st);
    if (e != hipSuccess) return (int)e;
    ErankBlk g{af_t, R, C, b, nblk, nreal, ctrl};
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute((const void*)erank_blk_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(erank_blk_init_kernel, dim3(r3d_cdiv(nreal * b, 32), r3d_cdiv(R, 32)), dim3(256), 0, st, x, ld, g);
    for (int s = 0; s < ms; ++s) {
        hipLaunchKernelGGL(erank_blk_round_kernel, dim3(nreal), dim3(64 * kJacWaves), (size_t)lds / 2 + 16, st, g, 0, 0, s);
        if (nblk > 1)
            for (int rd = 0; rd < nblk - 1; ++rd)
                hipLaunchKernelGGL(erank_blk_round_kernel, dim3(nblk / 2), dim3(64 * kJacWaves), (size_t)lds, st, g, 1, rd, s);
        hipLaunchKernelGGL(erank_blk_sweep_end_kernel, dim3(1), dim3(1), 0, st, g, s);
    }
    hipLaunchKernelGGL(erank_blk_finish_kernel, dim3(1), dim3(64 * kJacWaves), 0, st, g, sigma, stats);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* coef[k] = (*gout) * (d erank / d sigma_k) / sigma_k^3  for the backward  dX = Af diag(coef) (Af^T X). */
R3D_EXPORT int r3d_erank_bwd_coef(const float* sigma, const float* stats, const float* gout, float* coef, int C,
                                  int max_rank, void* stream) {
    R3D_REQUIRE(sigma && stats && coef && C > 0);
    hipLaunchKernelGGL(erank_coef_kernel, dim3(r3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sigma, stats, gout,
                       coef, C, max_rank);
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
