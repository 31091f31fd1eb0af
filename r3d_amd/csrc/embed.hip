// The seam between the two input projections and the SA-Fuser block as ONE launch per direction (train mode, where the
// token selection is data independent -- SURVEY F5a):
//   forward : split-K slab sums of both projections (+bias; ReLU on RGB futr_safuser_tokenfusion.py:183; LayerNorm +
//             ReLU on depth :196-197) -> token exchange + embd_drop (:56-62,83) -> fuser.blocks.0.norm1
//             (transformerblock.py:122).   Replaces 4 dependent launches (reduce, LN, exchange, LN) by one.
//   backward: norm1 backward (+ the two residual gradients) -> exchange backward (index_put / clone, ReLU of :183)
//             -> depth LayerNorm + ReLU backward.   Replaces 3 dependent launches by one.
// One workgroup (4 waves) per frame row n.  Everything is latency: all loads are unconditional (clamped columns) and
// issued up front.  Reductions are wave shuffles / fixed-order LDS sums -> bitwise reproducible.
#include "common.h"
#include "../../include/r3d_hip.h"
#include "chain_bf3.h"

namespace r3d {

constexpr float kLnEpsE = 1e-5f;

struct EmbedFwdArgs {
    const float* rgb_src; int ns_r; const float* bias_r;      // ns_r > 0: [ns_r][N][H] slabs, bias + ReLU applied here
    const float* dep_src; int ns_d; const float* bias_d;      // ns_d > 0: [ns_d][N][H] slabs (+bias); LN + ReLU here
    const float* lnd_g; const float* lnd_b; const float* m_rgb; const float* m_dep;
    const uint8_t* drop; float drop_scale; const float* ln1_g; const float* ln1_b;
    float* rgb_out; float* dep_pre_out; float* mean_d; float* rstd_d; float* dep_out;
    float* x0; float* h1; float* m1; float* r1;
    int N, H;
    // riders (r3d_embed_fuse_fwd_planes): workgroups N .. N + ceil(pl_total / 4) - 1 re-split the chain weights (chain_bf3.h)
    const r3d_plane_job* pl_jobs; int pl_njobs, pl_total;
};

template <int EPL>
__global__ __launch_bounds__(256) void embed_fuse_fwd_kernel(const EmbedFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];       // [8 partials = 4 waves x 2 halves][2 projections][H]
    if ((int)blockIdx.x >= a.N) {                                     // rider: four (tile, k-step) blocks of the weight planes
        weight_planes_block(a.pl_jobs, a.pl_njobs, a.pl_total, ((int)blockIdx.x - a.N) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, H = a.H;
    const size_t stride = (size_t)a.N * H, rowo = (size_t)n * H;
    int cc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { const int c = lane + 64 * e; cc[e] = c < H ? c : H - 1; }
    // ---- operands of the row tail, prefetched by the two waves that run it (wave t finishes token t of the frame)
    const int t = wave & 1;
    float g1[EPL], b1[EPL], gd[EPL], bd[EPL], br[EPL], bdp[EPL], msk[EPL], keep[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        g1[e] = a.ln1_g[cc[e]]; b1[e] = a.ln1_b[cc[e]];
        gd[e] = a.lnd_g[cc[e]]; bd[e] = a.lnd_b[cc[e]];
        br[e] = (a.ns_r > 0 && a.bias_r) ? a.bias_r[cc[e]] : 0.f;
        bdp[e] = a.bias_d ? a.bias_d[cc[e]] : 0.f;
        msk[e] = (t == 0 ? a.m_rgb : a.m_dep)[cc[e]];
        keep[e] = a.drop ? a.drop_scale * (float)a.drop[((size_t)2 * n + t) * H + cc[e]] : 1.f;
    }
    // ---- slab sums: wave w takes slabs w, w+4, ... of both projections; `red` holds 8 partial rows per projection
    const bool vec4 = EPL == 2 && (H & 3) == 0 &&
                      (((uintptr_t)a.rgb_src | (uintptr_t)a.dep_src) & 15) == 0;
    if (vec4) {
        // H <= 128: a row is <= 32 float4, so the two half-waves take alternate slabs of the wave's set and every lane has
        // ALL its loads (8 per projection and 64 slabs) in flight at once -- one memory round trip for the ~61 depth and
        // the RGB slabs together instead of one per batch of eight scalar loads (all clamped, none under a branch)
        const int j = lane >> 5, c4 = lane & 31, H4 = H >> 2;
        const size_t st4 = stride >> 2;
        const float4* pr = reinterpret_cast<const float4*>(a.rgb_src + rowo) + (c4 < H4 ? c4 : 0);
        const float4* pd = reinterpret_cast<const float4*>(a.dep_src + rowo) + (c4 < H4 ? c4 : 0);
        auto slab_sum = [&](const float4* p, int ns) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int base = 0; base < ns; base += 64) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int sl = base + wave + 4 * (2 * i + j);
                    v[i] = p[(size_t)(sl < ns ? sl : ns - 1) * st4];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (base + wave + 4 * (2 * i + j) >= ns) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#define R3D_A4(P_, Q_) make_float4(P_.x + Q_.x, P_.y + Q_.y, P_.z + Q_.z, P_.w + Q_.w)
                const float4 t01 = R3D_A4(v[0], v[1]), t23 = R3D_A4(v[2], v[3]), t45 = R3D_A4(v[4], v[5]), t67 = R3D_A4(v[6], v[7]);
                const float4 t03 = R3D_A4(t01, t23), t47 = R3D_A4(t45, t67), tt = R3D_A4(t03, t47);
                acc = R3D_A4(acc, tt);
#undef R3D_A4
            }
            return acc;
        };
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.ns_r > 0) ra = slab_sum(pr, a.ns_r);
        else if (wave == 0 && j == 0) ra = pr[0];
        const float4 da = slab_sum(pd, a.ns_d);
        if (c4 < H4) {
            *reinterpret_cast<float4*>(red + ((size_t)(wave * 2 + j) * 2 + 0) * H + 4 * c4) = ra;
            *reinterpret_cast<float4*>(red + ((size_t)(wave * 2 + j) * 2 + 1) * H + 4 * c4) = da;
        }
    } else if (EPL >= 4 && (H & 3) == 0 && (((uintptr_t)a.rgb_src | (uintptr_t)a.dep_src) & 15) == 0) {
        // wide rows (hidden 256 .. 1024): a lane takes EPL / 4 float4 columns of the row; the wave's slabs (w, w + 4, ...) go
        // four at a time, so EPL loads per projection are in flight per lane instead of dependent batches of scalar loads
        // (cfg4's per-GPU shape: 24.2 -> see DESIGN 4)
        constexpr int NV = EPL >= 4 ? EPL / 4 : 1;              // (EPL 2 never takes this branch)
        const int H4 = H >> 2;
        const size_t st4 = stride >> 2;
        int c4[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) c4[v] = lane + 64 * v < H4 ? lane + 64 * v : 0;
        const float4* pr = reinterpret_cast<const float4*>(a.rgb_src + rowo);
        const float4* pd = reinterpret_cast<const float4*>(a.dep_src + rowo);
        auto slab_sum = [&](const float4* p, int ns, float4 (&acc)[NV]) {
            for (int base = 0; base < ns; base += 16) {
                float4 x[4][NV];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sl = base + wave + 4 * i;
#pragma unroll
                    for (int v = 0; v < NV; ++v) x[i][v] = p[(size_t)(sl < ns ? sl : ns - 1) * st4 + c4[v]];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = base + wave + 4 * i < ns;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        acc[v].x += in ? x[i][v].x : 0.f; acc[v].y += in ? x[i][v].y : 0.f;
                        acc[v].z += in ? x[i][v].z : 0.f; acc[v].w += in ? x[i][v].w : 0.f;
                    }
                }
            }
        };
        float4 ra[NV], da[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) { ra[v] = make_float4(0.f, 0.f, 0.f, 0.f); da[v] = ra[v]; }
        if (a.ns_r > 0) slab_sum(pr, a.ns_r, ra);
        else if (wave == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) ra[v] = pr[c4[v]];
        }
        slab_sum(pd, a.ns_d, da);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            if (c < H4) {
                *reinterpret_cast<float4*>(red + ((size_t)(wave * 2) * 2 + 0) * H + 4 * c) = ra[v];
                *reinterpret_cast<float4*>(red + ((size_t)(wave * 2) * 2 + 1) * H + 4 * c) = da[v];
                *reinterpret_cast<float4*>(red + ((size_t)(wave * 2 + 1) * 2 + 0) * H + 4 * c) = z4;
                *reinterpret_cast<float4*>(red + ((size_t)(wave * 2 + 1) * 2 + 1) * H + 4 * c) = z4;
            }
        }
    } else {
        float ar[EPL], ad[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.ns_r > 0) {
                const float* p = a.rgb_src + rowo + cc[e];
                int s = wave;
                for (; s + 12 < a.ns_r; s += 16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s4[q] += p[(size_t)(s + 4 * q) * stride];
                }
                for (; s < a.ns_r; s += 4) s4[0] += p[(size_t)s * stride];
            } else if (wave == 0) {
                s4[0] = a.rgb_src[rowo + cc[e]];
            }
            ar[e] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            float d8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // depth: ~61 slabs -> 8 loads in flight per column
            {
                const float* p = a.dep_src + rowo + cc[e];
                int s = wave;
                for (; s + 28 < a.ns_d; s += 32) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) d8[q] += p[(size_t)(s + 4 * q) * stride];
                }
                for (int q = 0; s < a.ns_d; s += 4, ++q) d8[q & 7] += p[(size_t)s * stride];
            }
            ad[e] = ((d8[0] + d8[1]) + (d8[2] + d8[3])) + ((d8[4] + d8[5]) + (d8[6] + d8[7]));
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < H) {
                red[((size_t)(wave * 2) * 2 + 0) * H + c] = ar[e]; red[((size_t)(wave * 2) * 2 + 1) * H + c] = ad[e];
                red[((size_t)(wave * 2 + 1) * 2 + 0) * H + c] = 0.f; red[((size_t)(wave * 2 + 1) * 2 + 1) * H + c] = 0.f;
            }
        }
    }
    __syncthreads();
    if (wave >= 2) return;
    // ---- both tail waves rebuild the two embedding rows (identical arithmetic -> identical values)
    float r[EPL], dpre[EPL];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        float vr = 0.f, vd = 0.f;
        if (c < H) {
            // partial k of projection q sits at red[(2 k + q) H + c]; k = 2 wave + half
            vr = ((red[0 * H + c] + red[2 * H + c]) + (red[4 * H + c] + red[6 * H + c])) +
                 ((red[8 * H + c] + red[10 * H + c]) + (red[12 * H + c] + red[14 * H + c]));
            vd = ((red[1 * H + c] + red[3 * H + c]) + (red[5 * H + c] + red[7 * H + c])) +
                 ((red[9 * H + c] + red[11 * H + c]) + (red[13 * H + c] + red[15 * H + c])) + bdp[e];
            if (a.ns_r > 0) vr = fmaxf(vr + br[e], 0.f);
        }
        r[e] = vr; dpre[e] = vd;
        s += vd;
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const float dl = c < H ? dpre[e] - mean : 0.f;
        q += dl * dl;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + kLnEpsE);
    float d[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) d[e] = fmaxf((dpre[e] - mean) * rstd * gd[e] + bd[e], 0.f);
    if (wave == 0) {
        if (lane == 0) { a.mean_d[n] = mean; a.rstd_d[n] = rstd; }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < H) {
                if (a.rgb_out != a.rgb_src) a.rgb_out[rowo + c] = r[e];
                a.dep_pre_out[rowo + c] = dpre[e];
                a.dep_out[rowo + c] = d[e];
            }
        }
    }
    // ---- token t of the frame: exchange, embd_drop, norm1
    const size_t row = (size_t)2 * n + t;
    float x[EPL];
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const float own = t == 0 ? r[e] : d[e], other = t == 0 ? d[e] : r[e];
        float v = (msk[e] != 0.f ? other : own) * keep[e];
        if (c >= H) v = 0.f;
        x[e] = v;
        s1 += v;
        if (c < H) a.x0[row * H + c] = v;
    }
    const float mean1 = wave_sum(s1) / (float)H;
    float q1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const float dl = c < H ? x[e] - mean1 : 0.f;
        q1 += dl * dl;
    }
    const float rstd1 = 1.0f / sqrtf(wave_sum(q1) / (float)H + kLnEpsE);
    if (lane == 0) { a.m1[row] = mean1; a.r1[row] = rstd1; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < H) a.h1[row * H + c] = (x[e] - mean1) * rstd1 * g1[e] + b1[e];
    }
}

struct EmbedBwdArgs {
    const float* d_h1; const float* x0; const float* m1; const float* r1; const float* ln1_g;
    const float* add1; const float* add2; const uint8_t* drop; float drop_scale;
    const float* m_rgb; const float* m_dep; const float* rgb; const float* dep_pre; const float* mean_d;
    const float* rstd_d; const float* lnd_g; const float* lnd_b;
    float* d_rgb_pre; float* d_dep_pre; float* ws_n1; float* ws_dep;     // ws_*: [N][2][H] partial (dgamma, dbeta)
    int N, H;
};

template <int EPL>
__global__ __launch_bounds__(128) void embed_fuse_bwd_kernel(const EmbedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // G[2][H] | P[2][2][H]
    const int n = blockIdx.x, lane = threadIdx.x & 63, t = threadIdx.x >> 6, H = a.H;
    float* G = lds;
    float* P = lds + 2 * H;
    const size_t row = (size_t)2 * n + t, rowo = (size_t)n * H;
    int cc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { const int c = lane + 64 * e; cc[e] = c < H ? c : H - 1; }
    // ---- every load of the workgroup up front
    const float mean1 = a.m1[row], rstd1 = a.r1[row];
    float dh[EPL], xv[EPL], g1[EPL], a1[EPL], a2[EPL], keep[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        dh[e] = a.d_h1[row * H + cc[e]];
        xv[e] = a.x0[row * H + cc[e]];
        g1[e] = a.ln1_g[cc[e]];
        a1[e] = a.add1 ? a.add1[row * H + cc[e]] : 0.f;
        a2[e] = a.add2 ? a.add2[row * H + cc[e]] : 0.f;
        keep[e] = a.drop ? a.drop_scale * (float)a.drop[row * H + cc[e]] : 1.f;
    }
    float mr[EPL], md[EPL], rg[EPL], dp[EPL], gd[EPL], bd[EPL];
    float mean_d = 0.f, rstd_d = 0.f;
    if (t == 0) {
        mean_d = a.mean_d[n]; rstd_d = a.rstd_d[n];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            mr[e] = a.m_rgb[cc[e]]; md[e] = a.m_dep[cc[e]];
            rg[e] = a.rgb[rowo + cc[e]]; dp[e] = a.dep_pre[rowo + cc[e]];
            gd[e] = a.lnd_g[cc[e]]; bd[e] = a.lnd_b[cc[e]];
        }
    }
    // ---- norm1 backward of token t (+ the two residual gradients), then back through embd_drop
    float xh[EPL], gg[EPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        float xhat = 0.f, g = 0.f;
        if (c < H) {
            xhat = (xv[e] - mean1) * rstd1;
            g = dh[e] * g1[e];
            P[(t * 2 + 0) * H + c] = dh[e] * xhat;
            P[(t * 2 + 1) * H + c] = dh[e];
        }
        xh[e] = xhat; gg[e] = g;
        s1 += g; s2 += g * xhat;
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < H) G[t * H + c] = (rstd1 * (gg[e] - s1 - xh[e] * s2) + a1[e] + a2[e]) * keep[e];
    }
    __syncthreads();
    if (t == 1) {                       // norm1 parameter-gradient partial of this frame (its two token rows)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < H) {
                a.ws_n1[((size_t)n * 2 + 0) * H + c] = P[0 * H + c] + P[2 * H + c];
                a.ws_n1[((size_t)n * 2 + 1) * H + c] = P[1 * H + c] + P[3 * H + c];
            }
        }
        return;
    }
    // ---- exchange backward (index_put / clone), ReLU of the RGB embedding, depth LayerNorm + ReLU backward
    float xd[EPL], gq[EPL];
    float u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        float xhat = 0.f, g = 0.f;
        if (c < H) {
            const float g0 = G[c], g1v = G[H + c];
            const float drgb = ((mr[e] != 0.f ? 0.f : g0) + (md[e] != 0.f ? g1v : 0.f)) * (rg[e] > 0.f ? 1.f : 0.f);
            a.d_rgb_pre[rowo + c] = drgb;
            float dd = (mr[e] != 0.f ? g0 : 0.f) + (md[e] != 0.f ? 0.f : g1v);
            xhat = (dp[e] - mean_d) * rstd_d;
            if (!(xhat * gd[e] + bd[e] > 0.f)) dd = 0.f;
            a.ws_dep[((size_t)n * 2 + 0) * H + c] = dd * xhat;
            a.ws_dep[((size_t)n * 2 + 1) * H + c] = dd;
            g = dd * gd[e];
        }
        xd[e] = xhat; gq[e] = g;
        u1 += g; u2 += g * xhat;
    }
    u1 = wave_sum(u1) / (float)H;
    u2 = wave_sum(u2) / (float)H;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < H) a.d_dep_pre[rowo + c] = rstd_d * (gq[e] - u1 - xd[e] * u2);
    }
}

}  // namespace r3d

using namespace r3d;

/* Backward seam (see the file header): d_h1 [2N,H] = gradient w.r.t. norm1's output; add1 / add2 (optional) are added
 * to norm1's input gradient (the residual paths).  Outputs: d_rgb_pre [N,H] (gradient before input_embed's ReLU),
 * d_dep_pre [N,H] (gradient before depth_layernorm), and the LayerNorm parameter-gradient partials ws_n1 / ws_dep, each
 * [N][2][H] floats: one (dgamma, dbeta) pair per frame, summed by r3d_layernorm_bwd_finalize_batched with rows = -N. */
R3D_EXPORT int r3d_embed_fuse_bwd(const float* d_h1, const float* x0, const float* m1, const float* r1, const float* ln1_gamma,
                                  const float* add1, const float* add2, const uint8_t* drop_mask, float drop_scale,
                                  const float* mask_rgb, const float* mask_dep, const float* rgb, const float* dep_pre,
                                  const float* mean_d, const float* rstd_d, const float* lnd_gamma, const float* lnd_beta,
                                  float* d_rgb_pre, float* d_dep_pre, float* ws_n1, float* ws_dep, int N, int H,
                                  void* stream) {
    R3D_REQUIRE(d_h1 && x0 && m1 && r1 && ln1_gamma && mask_rgb && mask_dep && rgb && dep_pre && mean_d && rstd_d);
    R3D_REQUIRE(lnd_gamma && lnd_beta && d_rgb_pre && d_dep_pre && ws_n1 && ws_dep && N > 0 && H > 0 && H <= 1024);
    EmbedBwdArgs a{d_h1, x0, m1, r1, ln1_gamma, add1, add2, drop_mask, drop_scale, mask_rgb, mask_dep, rgb, dep_pre,
                   mean_d, rstd_d, lnd_gamma, lnd_beta, d_rgb_pre, d_dep_pre, ws_n1, ws_dep, N, H};
    const size_t shmem = (size_t)6 * H * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (H <= 128) hipLaunchKernelGGL(embed_fuse_bwd_kernel<2>, dim3(N), dim3(128), shmem, s, a);
    else if (H <= 512) hipLaunchKernelGGL(embed_fuse_bwd_kernel<8>, dim3(N), dim3(128), shmem, s, a);
    else hipLaunchKernelGGL(embed_fuse_bwd_kernel<16>, dim3(N), dim3(128), shmem, s, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Forward seam (see the file header).  rgb_src: ns_r > 0 -> split-K slabs [ns_r][N][H] of input_embed (bias_r and ReLU
 * are applied here), ns_r == 0 -> the finished [N,H] embedding (rgb_out may alias it).  dep_src: [ns_d][N][H] slabs
 * (ns_d >= 1; a finished pre-LayerNorm matrix is one slab), bias_d optional.  Outputs: rgb_out, dep_pre_out (pre-LN),
 * mean_d / rstd_d [N], dep_out (post ReLU), x0 [2N,H] (exchanged + dropped tokens), h1 = norm1(x0), m1 / r1 [2N].
 * All matrices contiguous with leading dimension H; H <= 1024. */
static int embed_fuse_fwd_launch(EmbedFwdArgs a, hipStream_t s) {
    R3D_REQUIRE(a.rgb_src && a.dep_src && a.lnd_g && a.lnd_b && a.m_rgb && a.m_dep && a.ln1_g && a.ln1_b);
    R3D_REQUIRE(a.rgb_out && a.dep_pre_out && a.mean_d && a.rstd_d && a.dep_out && a.x0 && a.h1 && a.m1 && a.r1);
    R3D_REQUIRE(a.N > 0 && a.H > 0 && a.H <= 1024 && a.ns_r >= 0 && a.ns_d >= 1);
    const size_t shmem = (size_t)16 * a.H * sizeof(float);
    const int grid = a.N + (a.pl_jobs ? r3d_cdiv(a.pl_total, 4) : 0);
    if (a.H <= 128) hipLaunchKernelGGL(embed_fuse_fwd_kernel<2>, dim3(grid), dim3(256), shmem, s, a);
    else if (a.H <= 512) hipLaunchKernelGGL(embed_fuse_fwd_kernel<8>, dim3(grid), dim3(256), shmem, s, a);
    else hipLaunchKernelGGL(embed_fuse_fwd_kernel<16>, dim3(grid), dim3(256), shmem, s, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_embed_fuse_fwd(const float* rgb_src, int ns_r, const float* bias_r, const float* dep_src, int ns_d,
                                  const float* bias_d, const float* lnd_gamma, const float* lnd_beta, const float* mask_rgb,
                                  const float* mask_dep, const uint8_t* drop_mask, float drop_scale, const float* ln1_gamma,
                                  const float* ln1_beta, float* rgb_out, float* dep_pre_out, float* mean_d, float* rstd_d,
                                  float* dep_out, float* x0, float* h1, float* m1, float* r1, int N, int H, void* stream) {
    EmbedFwdArgs a{rgb_src, ns_r, bias_r, dep_src, ns_d, bias_d, lnd_gamma, lnd_beta, mask_rgb, mask_dep, drop_mask,
                   drop_scale, ln1_gamma, ln1_beta, rgb_out, dep_pre_out, mean_d, rstd_d, dep_out, x0, h1, m1, r1, N, H,
                   nullptr, 0, 0};
    return embed_fuse_fwd_launch(a, (hipStream_t)stream);
}

/* r3d_embed_fuse_fwd + r3d_weight_planes(jobs_device, njobs, total_blocks) in ONE launch: the re-split of the chain weights
 * (which depends on the parameters only) rides as extra workgroups of the seam, whose own 128 workgroups are latency-bound
 * and leave half the chip idle. */
R3D_EXPORT int r3d_embed_fuse_fwd_planes(const float* rgb_src, int ns_r, const float* bias_r, const float* dep_src, int ns_d,
                                         const float* bias_d, const float* lnd_gamma, const float* lnd_beta,
                                         const float* mask_rgb, const float* mask_dep, const uint8_t* drop_mask,
                                         float drop_scale, const float* ln1_gamma, const float* ln1_beta, float* rgb_out,
                                         float* dep_pre_out, float* mean_d, float* rstd_d, float* dep_out, float* x0, float* h1,
                                         float* m1, float* r1, int N, int H, const r3d_plane_job* jobs_device, int njobs,
                                         int total_blocks, void* stream) {
    R3D_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0);
    EmbedFwdArgs a{rgb_src, ns_r, bias_r, dep_src, ns_d, bias_d, lnd_gamma, lnd_beta, mask_rgb, mask_dep, drop_mask,
                   drop_scale, ln1_gamma, ln1_beta, rgb_out, dep_pre_out, mean_d, rstd_d, dep_out, x0, h1, m1, r1, N, H,
                   jobs_device, njobs, total_blocks};
    return embed_fuse_fwd_launch(a, (hipStream_t)stream);
}
