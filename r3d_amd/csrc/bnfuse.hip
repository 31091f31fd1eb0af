// The BN-blend token fuser of model/futr_safuser_batchnormalization.py, the seam between the two
// embeddings and the fuser block:
//   forward : BatchNorm1d over the (B,T) rows per channel on both embeddings (:45-46; batch statistics + running-stat
//             update when training, running statistics otherwise), score = |BN gamma| (:48-49, selection by
//             r3d_token_select with k = int(0.1 C)), exchanged = alpha*own + (1-alpha)*other on the selected channels
//             (:65-74), embd_drop (:95), fuser.blocks.0.norm1 (transformerblock.py:122).
//   backward: norm1 backward (+ the residual gradient) -> dropout -> blend backward -> per-element terms whose column sums
//             are the BatchNorm / alpha parameter gradients (summed by r3d_rowmod_sum_batched) -> BatchNorm input
//             gradient (+ the ReLU of the RGB embedding :194).
// Row kernels: one workgroup (2 waves = the two modality tokens) per frame; column statistics: one workgroup per 64
// channels.  All loads unconditional from clamped columns, issued up front; fixed-order reductions.
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

constexpr float kEpsBN = 1e-5f;
constexpr float kEpsLN = 1e-5f;

// ---- statistics: x [2][N][C] given as two pointers.  out: mean [2][C], rstd [2][C], absgamma [2][C].
struct BnStatsArgs {
    const float* x[2]; float* run_mean[2]; float* run_var[2]; int64_t* nbt[2]; const float* gamma[2];
    float* mean; float* rstd; float* absg; int N, C, training; float momentum;
};

__global__ __launch_bounds__(256) void bn_stats_kernel(const BnStatsArgs a) {
    __shared__ float4 red[16][16];
    const int t = blockIdx.y, c0 = blockIdx.x * 64, C = a.C, N = a.N;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = c0 + cq * 4;
    const bool cok = c < C;                                     // C % 4 == 0
    const float* x = a.x[t];
    float4 mean4 = make_float4(0.f, 0.f, 0.f, 0.f), var4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.training) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok)
            for (int r = rl; r < N; r += 16) {
                const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[rl][cq] = s;
        __syncthreads();
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) { tot.x += red[k][cq].x; tot.y += red[k][cq].y; tot.z += red[k][cq].z; tot.w += red[k][cq].w; }
        mean4 = make_float4(tot.x / N, tot.y / N, tot.z / N, tot.w / N);
        __syncthreads();
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok)
            for (int r = rl; r < N; r += 16) {
                const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
                const float dx = v.x - mean4.x, dy = v.y - mean4.y, dz = v.z - mean4.z, dw = v.w - mean4.w;
                q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
            }
        red[rl][cq] = q;
        __syncthreads();
        tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) { tot.x += red[k][cq].x; tot.y += red[k][cq].y; tot.z += red[k][cq].z; tot.w += red[k][cq].w; }
        var4 = make_float4(tot.x / N, tot.y / N, tot.z / N, tot.w / N);          // biased: used for normalisation
    } else if (cok) {
        mean4 = *reinterpret_cast<const float4*>(a.run_mean[t] + c);
        var4 = *reinterpret_cast<const float4*>(a.run_var[t] + c);
    }
    if (rl != 0 || !cok) return;
    const float m4[4] = {mean4.x, mean4.y, mean4.z, mean4.w}, v4[4] = {var4.x, var4.y, var4.z, var4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        a.mean[(size_t)t * C + c + u] = m4[u];
        a.rstd[(size_t)t * C + c + u] = 1.0f / sqrtf(v4[u] + kEpsBN);
        a.absg[(size_t)t * C + c + u] = fabsf(a.gamma[t][c + u]);
        if (a.training) {                                       // nn.BatchNorm1d running statistics (unbiased variance)
            const float unb = N > 1 ? v4[u] * (float)N / (float)(N - 1) : v4[u];
            a.run_mean[t][c + u] = (1.f - a.momentum) * a.run_mean[t][c + u] + a.momentum * m4[u];
            a.run_var[t][c + u] = (1.f - a.momentum) * a.run_var[t][c + u] + a.momentum * unb;
        }
    }
    if (a.training && blockIdx.x == 0 && threadIdx.x == 0 && a.nbt[t]) *a.nbt[t] += 1;
}

struct BnBlendArgs {
    const float* rgb; const float* dep;                         // [N][C] embeddings (post ReLU)
    const float* mean; const float* rstd;                       // [2][C]
    const float* g_r; const float* b_r; const float* g_d; const float* b_d; const float* alpha;
    const float* m_rgb; const float* m_dep;                     // [C] 1 = selected channel
    const uint8_t* drop; float drop_scale; const float* ln1_g; const float* ln1_b;
    float* x0; float* h1; float* m1; float* r1;
    // backward only
    const float* d_h1; const float* add1; int training;
    float* t_drb; float* t_drbx; float* t_ddb; float* t_ddbx; float* t_dal; float* ws_n1;
    int N, C;
};

template <int EPL>
__global__ __launch_bounds__(128) void bn_blend_fwd_kernel(const BnBlendArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x & 63, t = threadIdx.x >> 6, C = a.C;
    const size_t rowo = (size_t)n * C, row = (size_t)2 * n + t;
    float x[EPL], g1[EPL], b1[EPL];
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e, cc = c < C ? c : C - 1;
        const float r = a.rgb[rowo + cc], d = a.dep[rowo + cc];
        const float rb = (r - a.mean[cc]) * a.rstd[cc] * a.g_r[cc] + a.b_r[cc];
        const float db = (d - a.mean[C + cc]) * a.rstd[C + cc] * a.g_d[cc] + a.b_d[cc];
        const float al = a.alpha[cc];
        const float own = t == 0 ? rb : db, oth = t == 0 ? db : rb;
        const float sel = (t == 0 ? a.m_rgb : a.m_dep)[cc];
        const float keep = a.drop ? a.drop_scale * (float)a.drop[row * C + cc] : 1.f;
        float v = (sel != 0.f ? al * own + (1.f - al) * oth : own) * keep;
        if (c >= C) v = 0.f;
        x[e] = v; s1 += v;
        g1[e] = a.ln1_g[cc]; b1[e] = a.ln1_b[cc];
        if (c < C) a.x0[row * C + c] = v;
    }
    const float mean1 = wave_sum(s1) / (float)C;
    float q1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float dl = (lane + 64 * e < C) ? x[e] - mean1 : 0.f;
        q1 += dl * dl;
    }
    const float rstd1 = 1.0f / sqrtf(wave_sum(q1) / (float)C + kEpsLN);
    if (lane == 0) { a.m1[row] = mean1; a.r1[row] = rstd1; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < C) a.h1[row * C + c] = (x[e] - mean1) * rstd1 * g1[e] + b1[e];
    }
}

template <int EPL>
__global__ __launch_bounds__(128) void bn_blend_bwd_kernel(const BnBlendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // G[2][C] | P[2][2][C]
    const int n = blockIdx.x, lane = threadIdx.x & 63, t = threadIdx.x >> 6, C = a.C;
    float* G = lds;
    float* P = lds + 2 * C;
    const size_t rowo = (size_t)n * C, row = (size_t)2 * n + t;
    const float mean1 = a.m1[row], rstd1 = a.r1[row];
    float xh[EPL], gg[EPL], a1[EPL], keep[EPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e, cc = c < C ? c : C - 1;
        const float dh = a.d_h1[row * C + cc], xv = a.x0[row * C + cc];
        a1[e] = a.add1 ? a.add1[row * C + cc] : 0.f;
        keep[e] = a.drop ? a.drop_scale * (float)a.drop[row * C + cc] : 1.f;
        float xhat = 0.f, g = 0.f;
        if (c < C) {
            xhat = (xv - mean1) * rstd1;
            g = dh * a.ln1_g[cc];
            P[(t * 2 + 0) * C + c] = dh * xhat;
            P[(t * 2 + 1) * C + c] = dh;
        }
        xh[e] = xhat; gg[e] = g; s1 += g; s2 += g * xhat;
    }
    s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < C) G[t * C + c] = (rstd1 * (gg[e] - s1 - xh[e] * s2) + a1[e]) * keep[e];
    }
    __syncthreads();
    if (t == 1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < C) {
                a.ws_n1[((size_t)n * 2 + 0) * C + c] = P[0 * C + c] + P[2 * C + c];
                a.ws_n1[((size_t)n * 2 + 1) * C + c] = P[1 * C + c] + P[3 * C + c];
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c >= C) continue;
        const float g0 = G[c], g1v = G[C + c];
        const float xr = (a.rgb[rowo + c] - a.mean[c]) * a.rstd[c];                 // x-hat of the two BatchNorms
        const float xd = (a.dep[rowo + c] - a.mean[C + c]) * a.rstd[C + c];
        const float rb = xr * a.g_r[c] + a.b_r[c], db = xd * a.g_d[c] + a.b_d[c];
        const float al = a.alpha[c];
        const bool sr = a.m_rgb[c] != 0.f, sd = a.m_dep[c] != 0.f;
        const float drb = g0 * (sr ? al : 1.f) + g1v * (sd ? (1.f - al) : 0.f);
        const float ddb = g0 * (sr ? (1.f - al) : 0.f) + g1v * (sd ? al : 1.f);
        a.t_drb[rowo + c] = drb; a.t_drbx[rowo + c] = drb * xr;
        a.t_ddb[rowo + c] = ddb; a.t_ddbx[rowo + c] = ddb * xd;
        a.t_dal[rowo + c] = (sr ? g0 * (rb - db) : 0.f) + (sd ? g1v * (db - rb) : 0.f);
    }
}

// d(input of BatchNorm) for both embeddings from the summed parameter gradients (dgamma = sum dy*xhat, dbeta = sum dy):
// training: gamma*rstd*(dy - dbeta/N - xhat*dgamma/N); eval (running statistics): gamma*rstd*dy.  RGB: times [rgb > 0].
struct BnApplyArgs {
    const float* rgb; const float* dep; const float* mean; const float* rstd; const float* g_r; const float* g_d;
    const float* t_drb; const float* t_ddb; const float* dg_r; const float* db_r; const float* dg_d; const float* db_d;
    float* d_rgb_pre; float* d_dep; int N, C, training;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnApplyArgs a) {
    const size_t total = (size_t)a.N * a.C;
    const float invn = a.training ? 1.0f / (float)a.N : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % a.C);
        const float r = a.rgb[i], d = a.dep[i];
        const float xr = (r - a.mean[c]) * a.rstd[c], xd = (d - a.mean[a.C + c]) * a.rstd[a.C + c];
        const float gr = a.g_r[c] * a.rstd[c] * (a.t_drb[i] - invn * a.db_r[c] - xr * invn * a.dg_r[c]);
        const float gd = a.g_d[c] * a.rstd[a.C + c] * (a.t_ddb[i] - invn * a.db_d[c] - xd * invn * a.dg_d[c]);
        a.d_rgb_pre[i] = r > 0.f ? gr : 0.f;
        a.d_dep[i] = gd;
    }
}

// ---- data-parallel BatchNorm (parallel.SyncBatchNorm): moments of the global batch from per-rank moments -------------
// pack    : out [4C + 1] = (local mean [2][C], local M2 = N * var [2][C], N)   (var recovered from rstd: 1/rstd^2 - eps)
// finalize: all [W][4C + 1] (rank order) -> Chan's combination  mean = sum n_r mean_r / n,  M2 = sum M2_r + sum n_r
//           (mean_r - mean)^2  in rank order (deterministic); writes mean / rstd [2][C], the nn.BatchNorm1d running-statistics
//           update from the global moments (unbiased with the global count), num_batches_tracked += 1, and
//           nfrac = n_local / n (the factor that turns bn_bwd_apply's 1 / n_local into 1 / n).
__global__ __launch_bounds__(256) void bn_sync_pack_kernel(const float* mean, const float* rstd, int N, int C, float* out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * C; i += gridDim.x * 256) {
        const float r = rstd[i];
        out[i] = mean[i];
        out[2 * C + i] = fmaxf(1.0f / (r * r) - kEpsBN, 0.f) * (float)N;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[4 * C] = (float)N;
}

struct BnSyncArgs {
    const float* all; int W, n_local, C; float* mean; float* rstd; float* run_mean[2]; float* run_var[2]; int64_t* nbt[2];
    float* nfrac; float momentum;
};

__global__ __launch_bounds__(256) void bn_sync_finalize_kernel(const BnSyncArgs a) {
    const int C = a.C, L = 4 * C + 1;
    float n = 0.f;
    for (int r = 0; r < a.W; ++r) n += a.all[(size_t)r * L + 4 * C];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * C; i += gridDim.x * 256) {
        float sm = 0.f;
        for (int r = 0; r < a.W; ++r) sm += a.all[(size_t)r * L + 4 * C] * a.all[(size_t)r * L + i];
        const float mean = sm / n;
        float m2 = 0.f, between = 0.f;
        for (int r = 0; r < a.W; ++r) {
            const float d = a.all[(size_t)r * L + i] - mean;
            m2 += a.all[(size_t)r * L + 2 * C + i];
            between += a.all[(size_t)r * L + 4 * C] * d * d;
        }
        const float var = (m2 + between) / n;
        a.mean[i] = mean;
        a.rstd[i] = 1.0f / sqrtf(var + kEpsBN);
        const int t = i / C, c = i - t * C;
        const float unb = n > 1.f ? var * (n / (n - 1.f)) : var;
        a.run_mean[t][c] = (1.f - a.momentum) * a.run_mean[t][c] + a.momentum * mean;
        a.run_var[t][c] = (1.f - a.momentum) * a.run_var[t][c] + a.momentum * unb;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.nfrac[0] = (float)a.n_local / n;
        if (a.nbt[0]) *a.nbt[0] += 1;
        if (a.nbt[1]) *a.nbt[1] += 1;
    }
}

}  // namespace r3d

using namespace r3d;

/* BatchNorm statistics of the two [N, C] embeddings (C % 4 == 0, contiguous): mean / rstd [2][C] (batch statistics when
 * training, else from the running buffers), absgamma [2][C] = |BN weight| (the selection score), and -- training -- the
 * nn.BatchNorm1d running-statistics update (momentum, unbiased variance, num_batches_tracked += 1). */
R3D_EXPORT int r3d_bn_stats(const float* x_rgb, const float* x_dep, float* run_mean_rgb, float* run_var_rgb,
                            int64_t* nbt_rgb, float* run_mean_dep, float* run_var_dep, int64_t* nbt_dep,
                            const float* gamma_rgb, const float* gamma_dep, float* mean, float* rstd, float* absgamma,
                            int N, int C, int training, float momentum, void* stream) {
    R3D_REQUIRE(x_rgb && x_dep && run_mean_rgb && run_var_rgb && run_mean_dep && run_var_dep && gamma_rgb && gamma_dep);
    R3D_REQUIRE(mean && rstd && absgamma && N > 0 && C > 0 && (C % 4) == 0);
    if (!(r3d_aligned16(x_rgb) && r3d_aligned16(x_dep) && r3d_aligned16(run_mean_rgb) && r3d_aligned16(run_var_rgb) &&
          r3d_aligned16(run_mean_dep) && r3d_aligned16(run_var_dep))) return R3D_EALIGN;
    BnStatsArgs a{{x_rgb, x_dep}, {run_mean_rgb, run_mean_dep}, {run_var_rgb, run_var_dep}, {nbt_rgb, nbt_dep},
                  {gamma_rgb, gamma_dep}, mean, rstd, absgamma, N, C, training, momentum};
    hipLaunchKernelGGL(bn_stats_kernel, dim3(r3d_cdiv(C, 64), 2), dim3(256), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

template <typename K>
static void bn_launch(K k2, K k8, K k16, int C, dim3 grid, size_t shmem, hipStream_t s, const BnBlendArgs& a) {
    if (C <= 128) hipLaunchKernelGGL(k2, grid, dim3(128), shmem, s, a);
    else if (C <= 512) hipLaunchKernelGGL(k8, grid, dim3(128), shmem, s, a);
    else hipLaunchKernelGGL(k16, grid, dim3(128), shmem, s, a);
}

/* Forward seam of the BN-blend fuser: x0 [2N, C] = dropout(blend(BN(rgb), BN(dep))), h1 = norm1(x0), m1 / r1 [2N]. */
R3D_EXPORT int r3d_bn_blend_fwd(const float* rgb, const float* dep, const float* mean, const float* rstd,
                                const float* gamma_rgb, const float* beta_rgb, const float* gamma_dep, const float* beta_dep,
                                const float* alpha, const float* mask_rgb, const float* mask_dep, const uint8_t* drop_mask,
                                float drop_scale, const float* ln1_gamma, const float* ln1_beta, float* x0, float* h1,
                                float* m1, float* r1, int N, int C, void* stream) {
    R3D_REQUIRE(rgb && dep && mean && rstd && gamma_rgb && beta_rgb && gamma_dep && beta_dep && alpha && mask_rgb && mask_dep);
    R3D_REQUIRE(ln1_gamma && ln1_beta && x0 && h1 && m1 && r1 && N > 0 && C > 0 && C <= 1024);
    BnBlendArgs a{};
    a.rgb = rgb; a.dep = dep; a.mean = mean; a.rstd = rstd; a.g_r = gamma_rgb; a.b_r = beta_rgb; a.g_d = gamma_dep;
    a.b_d = beta_dep; a.alpha = alpha; a.m_rgb = mask_rgb; a.m_dep = mask_dep; a.drop = drop_mask; a.drop_scale = drop_scale;
    a.ln1_g = ln1_gamma; a.ln1_b = ln1_beta; a.x0 = x0; a.h1 = h1; a.m1 = m1; a.r1 = r1; a.N = N; a.C = C;
    bn_launch(bn_blend_fwd_kernel<2>, bn_blend_fwd_kernel<8>, bn_blend_fwd_kernel<16>, C, dim3(N), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Backward seam, part 1 (per frame): norm1 backward (+ add1), dropout, blend backward.  Writes five [N, C] term matrices
 * whose COLUMN SUMS are: t_drb -> d beta_rgb, t_drbx -> d gamma_rgb, t_ddb -> d beta_dep, t_ddbx -> d gamma_dep,
 * t_dal -> d alpha (sum them with r3d_rowmod_sum_batched, mod = 1), and the norm1 parameter partials ws_n1 [N][2][C]. */
R3D_EXPORT int r3d_bn_blend_bwd(const float* d_h1, const float* x0, const float* m1, const float* r1, const float* ln1_gamma,
                                const float* add1, const uint8_t* drop_mask, float drop_scale, const float* rgb,
                                const float* dep, const float* mean, const float* rstd, const float* gamma_rgb,
                                const float* beta_rgb, const float* gamma_dep, const float* beta_dep, const float* alpha,
                                const float* mask_rgb, const float* mask_dep, float* t_drb, float* t_drbx, float* t_ddb,
                                float* t_ddbx, float* t_dal, float* ws_n1, int N, int C, void* stream) {
    R3D_REQUIRE(d_h1 && x0 && m1 && r1 && ln1_gamma && rgb && dep && mean && rstd && gamma_rgb && beta_rgb && gamma_dep);
    R3D_REQUIRE(beta_dep && alpha && mask_rgb && mask_dep && t_drb && t_drbx && t_ddb && t_ddbx && t_dal && ws_n1);
    R3D_REQUIRE(N > 0 && C > 0 && C <= 1024);
    BnBlendArgs a{};
    a.rgb = rgb; a.dep = dep; a.mean = mean; a.rstd = rstd; a.g_r = gamma_rgb; a.b_r = beta_rgb; a.g_d = gamma_dep;
    a.b_d = beta_dep; a.alpha = alpha; a.m_rgb = mask_rgb; a.m_dep = mask_dep; a.drop = drop_mask; a.drop_scale = drop_scale;
    a.ln1_g = ln1_gamma; a.x0 = const_cast<float*>(x0); a.m1 = const_cast<float*>(m1); a.r1 = const_cast<float*>(r1);
    a.d_h1 = d_h1; a.add1 = add1; a.t_drb = t_drb; a.t_drbx = t_drbx; a.t_ddb = t_ddb; a.t_ddbx = t_ddbx; a.t_dal = t_dal;
    a.ws_n1 = ws_n1; a.N = N; a.C = C;
    bn_launch(bn_blend_bwd_kernel<2>, bn_blend_bwd_kernel<8>, bn_blend_bwd_kernel<16>, C, dim3(N), (size_t)6 * C * sizeof(float),
              (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Backward seam, part 2 (after the column sums): BatchNorm input gradients.  d_rgb_pre = gradient before input_embed's
 * ReLU; d_dep = gradient w.r.t. the depth embedding (output of depth_layernorm + ReLU). */
R3D_EXPORT int r3d_bn_bwd_apply(const float* rgb, const float* dep, const float* mean, const float* rstd,
                                const float* gamma_rgb, const float* gamma_dep, const float* t_drb, const float* t_ddb,
                                const float* dgamma_rgb, const float* dbeta_rgb, const float* dgamma_dep,
                                const float* dbeta_dep, float* d_rgb_pre, float* d_dep, int N, int C, int training,
                                void* stream) {
    R3D_REQUIRE(rgb && dep && mean && rstd && gamma_rgb && gamma_dep && t_drb && t_ddb && dgamma_rgb && dbeta_rgb);
    R3D_REQUIRE(dgamma_dep && dbeta_dep && d_rgb_pre && d_dep && N > 0 && C > 0);
    BnApplyArgs a{rgb, dep, mean, rstd, gamma_rgb, gamma_dep, t_drb, t_ddb, dgamma_rgb, dbeta_rgb, dgamma_dep, dbeta_dep,
                  d_rgb_pre, d_dep, N, C, training};
    const size_t total = (size_t)N * C;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Data-parallel BatchNorm, step 1: the local moments r3d_bn_stats left in mean / rstd [2][C] (N rows) packed for the
 * exchange: out [4C + 1] = (mean, M2 = N * var, N). */
R3D_EXPORT int r3d_bn_sync_pack(const float* mean, const float* rstd, int N, int C, float* out, void* stream) {
    R3D_REQUIRE(mean && rstd && out && N > 0 && C > 0);
    hipLaunchKernelGGL(bn_sync_pack_kernel, dim3(r3d_cdiv(2 * C, 256)), dim3(256), 0, (hipStream_t)stream, mean, rstd, N, C, out);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Data-parallel BatchNorm, step 2: all [world][4C + 1] = every rank's pack in rank order -> global mean / rstd [2][C],
 * running-statistics update from the global moments, nfrac[0] = n_local / n_global. */
R3D_EXPORT int r3d_bn_sync_finalize(const float* all, int world, int n_local, int C, float* mean, float* rstd,
                                    float* run_mean_rgb, float* run_var_rgb, int64_t* nbt_rgb, float* run_mean_dep,
                                    float* run_var_dep, int64_t* nbt_dep, float* nfrac, float momentum, void* stream) {
    R3D_REQUIRE(all && mean && rstd && run_mean_rgb && run_var_rgb && run_mean_dep && run_var_dep && nfrac);
    R3D_REQUIRE(world > 0 && n_local > 0 && C > 0);
    BnSyncArgs a{all, world, n_local, C, mean, rstd, {run_mean_rgb, run_mean_dep}, {run_var_rgb, run_var_dep},
                 {nbt_rgb, nbt_dep}, nfrac, momentum};
    hipLaunchKernelGGL(bn_sync_finalize_kernel, dim3(r3d_cdiv(2 * C, 256)), dim3(256), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
