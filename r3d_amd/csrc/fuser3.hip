// Build-defined THREE-modality token fuser (BASELINE.json configs[4]: "Synthetic 3-modality (RGB+Depth+Gaze) fusion").
//
// The reference's CMFuser is structurally two-token (model/futr_safuser_tokenfusion.py:74-81 hard-codes the keys 'rgb',
// 'depth' and a 2 x 2 mask); SURVEY.md 8(d) allows an M = 3 extension checked against the build's own CPU restatement
// (oracle/futr_oracle.py: cm_fuser_m -- "build-defined, parity unpinned").  The extension keeps every line of the original
// that generalises and fixes the two that do not:
//   token_fusion  (:33-66)  per-modality scores as in the reference, k = C / 4 lowest-score channels of modality m are
//                           replaced by the same channels of the NEXT modality, cyclically (m -> (m + 1) mod 3);
//   attention mask (:68-72) eye(3) with -inf on the diagonal: every token attends to the two OTHER modality tokens -- a real
//                           softmax over two logits (for M = 2 it degenerates to the swap the two-modality path uses);
//   mean over the 3 tokens  (:94).
// The GEMMs and LayerNorms are the library's (r3d_gemm_f32, r3d_layernorm_*); this file holds the three pieces that have
// no two-modality counterpart: the 3-way exchange, the 3-token attention core and the mean over token triples, each with
// its adjoint.  Rows are frame-major: tokens of frame n are rows 3n, 3n + 1, 3n + 2.
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

// x0[3n + m][c] = (mask[m][c] != 0 ? x[(m + 1) % 3][n][c] : x[m][n][c]) * keep
__global__ __launch_bounds__(256) void exchange3_fwd_kernel(const float* x0_, const float* x1_, const float* x2_,
                                                            const float* mask, const uint8_t* drop, float drop_scale,
                                                            float* out, int N, int C) {
    const size_t total = (size_t)3 * N * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t row = e / C;
        const int m = (int)(row % 3);
        const size_t n = row / 3;
        const float* own = m == 0 ? x0_ : (m == 1 ? x1_ : x2_);
        const float* nxt = m == 0 ? x1_ : (m == 1 ? x2_ : x0_);
        const float a = own[n * C + c], b = nxt[n * C + c];
        float v = mask[(size_t)m * C + c] != 0.f ? b : a;
        if (drop) v *= drop_scale * (float)drop[e];
        out[e] = v;
    }
}

// d x[m][n][c] = (1 - mask[m][c]) g[3n + m][c] + mask[m - 1][c] g[3n + (m - 1)][c]   (g = d out * keep)
__global__ __launch_bounds__(256) void exchange3_bwd_kernel(const float* g, const float* mask, const uint8_t* drop,
                                                            float drop_scale, float* d0, float* d1, float* d2, int N, int C) {
    const size_t total = (size_t)N * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t n = e / C;
        float gm[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const size_t o = (3 * n + m) * C + c;
            gm[m] = g[o] * (drop ? drop_scale * (float)drop[o] : 1.f);
        }
        const float k0 = mask[c] != 0.f ? 1.f : 0.f, k1 = mask[(size_t)C + c] != 0.f ? 1.f : 0.f,
                    k2 = mask[(size_t)2 * C + c] != 0.f ? 1.f : 0.f;
        // modality m keeps its own slot where unmasked and feeds the PREVIOUS modality's slot where that one is masked
        d0[e] = (1.f - k0) * gm[0] + k2 * gm[2];
        d1[e] = (1.f - k1) * gm[1] + k0 * gm[0];
        d2[e] = (1.f - k2) * gm[2] + k1 * gm[1];
    }
}

// Attention over the three modality tokens of a frame with the -inf diagonal: token i attends to the two others.
// One wave per (frame, head); lanes over the head's dh channels (dh <= 128: two per lane).
// qkv: [3N][3C] rows = tokens, columns [q | k | v], head h in columns h*dh .. of each third.  probs: [N][heads][3][2]
// (the two partners of token i in increasing index order).
struct Attn3Args {
    const float* qkv; float* probs; float* out; const float* d_out; float* d_qkv; int N, C, heads; float scale;
};

__device__ __forceinline__ int partner(int i, int t) { return t == 0 ? (i == 0 ? 1 : 0) : (i == 2 ? 1 : 2); }

template <bool BWD>
__global__ __launch_bounds__(256) void attn3_kernel(const Attn3Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= a.N * a.heads) return;
    const int n = unit / a.heads, h = unit % a.heads;
    const int C = a.C, dh = C / a.heads;
    const size_t ld = (size_t)3 * C;
    float q[3][2], k[3][2], v[3][2], go[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = lane + 64 * e;
            const bool in = d < dh;
            const size_t base = (size_t)(3 * n + i) * ld + (size_t)h * dh + (in ? d : 0);
            q[i][e] = in ? a.qkv[base] : 0.f;
            k[i][e] = in ? a.qkv[base + C] : 0.f;
            v[i][e] = in ? a.qkv[base + 2 * C] : 0.f;
            go[i][e] = (BWD && in) ? a.d_out[(size_t)(3 * n + i) * C + (size_t)h * dh + d] : 0.f;
        }
    }
    float p[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = partner(i, t);
            s[t] = wave_sum(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * a.scale;
        }
        const float mx = fmaxf(s[0], s[1]);
        const float e0 = expf(s[0] - mx), e1 = expf(s[1] - mx);
        const float inv = 1.0f / (e0 + e1);
        p[i][0] = e0 * inv; p[i][1] = e1 * inv;
    }
    if (!BWD) {
        if (lane < 6) a.probs[(size_t)unit * 6 + lane] = p[lane >> 1][lane & 1];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j0 = partner(i, 0), j1 = partner(i, 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                if (d < dh) a.out[(size_t)(3 * n + i) * C + (size_t)h * dh + d] = p[i][0] * v[j0][e] + p[i][1] * v[j1][e];
            }
        }
        return;
    }
    // backward: d p_ij = d_out_i . v_j ; d s = softmax' ; d q_i = scale sum_j d s_ij k_j ; d k_j = scale sum_i d s_ij q_i ;
    // d v_j = sum_i p_ij d_out_i
    float ds[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float dp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = partner(i, t);
            dp[t] = wave_sum(go[i][0] * v[j][0] + go[i][1] * v[j][1]);
        }
        const float dot = p[i][0] * dp[0] + p[i][1] * dp[1];
        ds[i][0] = p[i][0] * (dp[0] - dot) * a.scale;
        ds[i][1] = p[i][1] * (dp[1] - dot) * a.scale;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = lane + 64 * e;
        if (d >= dh) continue;
        float dq[3] = {0.f, 0.f, 0.f}, dk[3] = {0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = partner(i, t);
                dq[i] += ds[i][t] * k[j][e];
                dk[j] += ds[i][t] * q[i][e];
                dv[j] += p[i][t] * go[i][e];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const size_t base = (size_t)(3 * n + i) * ld + (size_t)h * dh + d;
            a.d_qkv[base] = dq[i];
            a.d_qkv[base + C] = dk[i];
            a.d_qkv[base + 2 * C] = dv[i];
        }
    }
}

// out[n][c] = (y[3n][c] + y[3n+1][c] + y[3n+2][c]) / 3 ; adjoint: d y[3n + m][c] = d out[n][c] / 3
__global__ __launch_bounds__(256) void triple_mean_fwd_kernel(const float* y, float* out, int N, int C) {
    const size_t total = (size_t)N * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t n = e / C;
        const float* r = y + (3 * n) * C + c;
        out[e] = (r[0] + r[C] + r[(size_t)2 * C]) * (1.0f / 3.0f);
    }
}
__global__ __launch_bounds__(256) void triple_mean_bwd_kernel(const float* d_out, float* dy, int N, int C) {
    const size_t total = (size_t)3 * N * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t n = (e / C) / 3;
        dy[e] = d_out[n * C + c] * (1.0f / 3.0f);
    }
}

static inline int blocks_for(size_t total) { return (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096); }

}  // namespace r3d

using namespace r3d;

/* Three-modality token exchange (build-defined M = 3 extension of model/futr_safuser_tokenfusion.py:56-62 + embd_drop :83):
 * x0[3n + m] = (mask[m] ? x_{(m+1) % 3}[n] : x_m[n]) * keep.  xa / xb / xc: [N, C] dense; mask: [3][C] 1.0 / 0.0;
 * drop_mask (optional) [3N, C]. */
R3D_EXPORT int r3d_token_exchange3_fwd(const float* xa, const float* xb, const float* xc, const float* mask,
                                       const uint8_t* drop_mask, float drop_scale, float* x0, int N, int C, void* stream) {
    R3D_REQUIRE(xa && xb && xc && mask && x0 && N > 0 && C > 0);
    hipLaunchKernelGGL(exchange3_fwd_kernel, dim3(blocks_for((size_t)3 * N * C)), dim3(256), 0, (hipStream_t)stream, xa, xb, xc,
                       mask, drop_mask, drop_scale, x0, N, C);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_token_exchange3_bwd(const float* dx0, const float* mask, const uint8_t* drop_mask, float drop_scale, float* da,
                                       float* db, float* dc, int N, int C, void* stream) {
    R3D_REQUIRE(dx0 && mask && da && db && dc && N > 0 && C > 0);
    hipLaunchKernelGGL(exchange3_bwd_kernel, dim3(blocks_for((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, dx0, mask,
                       drop_mask, drop_scale, da, db, dc, N, C);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Attention of the three modality tokens of each frame with the -inf diagonal (model/extras/transformerblock.py:19-36 with
 * generate_cross_attention_mask(3)): qkv [3N, 3C] = [q | k | v] per token row, heads of C / heads <= 128 channels; probs
 * [N][heads][3][2] (token i's two partners in increasing index order); out [3N, C]. */
R3D_EXPORT int r3d_attn3_fwd(const float* qkv, float* probs, float* out, int N, int C, int heads, void* stream) {
    R3D_REQUIRE(qkv && probs && out && N > 0 && C > 0 && heads > 0 && C % heads == 0 && C / heads <= 128);
    Attn3Args a{qkv, probs, out, nullptr, nullptr, N, C, heads, 1.0f / sqrtf((float)(C / heads))};
    hipLaunchKernelGGL(attn3_kernel<false>, dim3(r3d_cdiv(N * heads, 4)), dim3(256), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_attn3_bwd(const float* qkv, const float* d_out, float* d_qkv, int N, int C, int heads, void* stream) {
    R3D_REQUIRE(qkv && d_out && d_qkv && N > 0 && C > 0 && heads > 0 && C % heads == 0 && C / heads <= 128);
    Attn3Args a{qkv, nullptr, nullptr, d_out, d_qkv, N, C, heads, 1.0f / sqrtf((float)(C / heads))};
    hipLaunchKernelGGL(attn3_kernel<true>, dim3(r3d_cdiv(N * heads, 4)), dim3(256), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Mean over the three tokens of a frame (torch.mean(x, dim=1), :94) and its adjoint. */
R3D_EXPORT int r3d_triple_mean_fwd(const float* y, float* out, int N, int C, void* stream) {
    R3D_REQUIRE(y && out && N > 0 && C > 0);
    hipLaunchKernelGGL(triple_mean_fwd_kernel, dim3(blocks_for((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, y, out, N, C);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_triple_mean_bwd(const float* d_out, float* dy, int N, int C, void* stream) {
    R3D_REQUIRE(d_out && dy && N > 0 && C > 0);
    hipLaunchKernelGGL(triple_mean_bwd_kernel, dim3(blocks_for((size_t)3 * N * C)), dim3(256), 0, (hipStream_t)stream, d_out, dy, N, C);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
