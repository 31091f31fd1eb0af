// The tail of the decoder as ONE launch per direction (row-local work on the B*Q query rows, one wave per row):
//   forward : last layer's norm3 (transformer.py:329) -> decoder.norm (:182-183) -> the anticipation heads fc | fc_len
//             (futr_safuser_tokenfusion.py:219-226) as one [K+1, H] product.        Replaces 3 dependent launches.
//   backward: heads' input gradient -> decoder.norm backward -> norm3 backward (+ dropout3 :328 for the FFN branch).
//             Replaces 3 dependent launches; LayerNorm parameter-gradient partials in layernorm_bwd's layout.
// Everything is latency: all loads unconditional from clamped columns and issued up front; reductions are wave shuffles
// and a fixed-order LDS fold -> bitwise reproducible.
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

constexpr float kLnEpsT = 1e-5f;

struct TailFwdArgs {
    const float* x; const float* g3; const float* b3; const float* gF; const float* bF;
    const float* w_head; const float* b_head; int n_head;
    float* t3; float* m3; float* r3; float* tgtF; float* mF; float* rF; float* out; int ld_out;
    int rows, H;
};

template <int EPL>
__device__ __forceinline__ void ln_apply(const float (&x)[EPL], const float (&g)[EPL], const float (&b)[EPL], int H, int lane,
                                         float (&y)[EPL], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += (lane + 64 * e < H) ? x[e] : 0.f;
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float dl = (lane + 64 * e < H) ? x[e] - mean : 0.f;
        q += dl * dl;
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + kLnEpsT);
#pragma unroll
    for (int e = 0; e < EPL; ++e) y[e] = (lane + 64 * e < H) ? (x[e] - mean) * rstd * g[e] + b[e] : 0.f;
}

template <int EPL>
__global__ __launch_bounds__(256) void decoder_tail_fwd_kernel(const TailFwdArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), H = a.H;
    if (row >= a.rows) return;
    int cc[EPL];
    float x[EPL], g3[EPL], b3[EPL], gF[EPL], bF[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        cc[e] = c < H ? c : H - 1;
        x[e] = a.x[(size_t)row * H + cc[e]];
        g3[e] = a.g3[cc[e]]; b3[e] = a.b3[cc[e]]; gF[e] = a.gF[cc[e]]; bF[e] = a.bF[cc[e]];
    }
    // head weights of this lane's columns, up front (K + 1 = 18 rows for DARai): the loads travel under the two norms
    constexpr int NHP = (EPL <= 2) ? 24 : 1;
    float wh[NHP][EPL];
    const bool pre = a.n_head <= NHP;
    if (pre) {
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
            const int kc = k < a.n_head ? k : a.n_head - 1;
#pragma unroll
            for (int e = 0; e < EPL; ++e) wh[k][e] = a.w_head[(size_t)kc * H + cc[e]];
        }
    }
    float y3[EPL], yF[EPL], m, r;
    ln_apply<EPL>(x, g3, b3, H, lane, y3, m, r);
    if (lane == 0) { a.m3[row] = m; a.r3[row] = r; }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (lane + 64 * e < H) a.t3[(size_t)row * H + lane + 64 * e] = y3[e];
    ln_apply<EPL>(y3, gF, bF, H, lane, yF, m, r);
    if (lane == 0) { a.mF[row] = m; a.rF[row] = r; }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (lane + 64 * e < H) a.tgtF[(size_t)row * H + lane + 64 * e] = yF[e];
    if (pre) {
        float p[NHP];
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) t += yF[e] * wh[k][e];
            p[k] = t;
        }
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
            if (k < a.n_head) {                       // wave-uniform
                const float t = wave_sum(p[k]);
                if (lane == 0) a.out[(size_t)row * a.ld_out + k] = t + a.b_head[k];
            }
        }
        return;
    }
    // heads: 6 outputs per pass (their weight rows are loaded together, then 6 wave reductions)
    for (int k0 = 0; k0 < a.n_head; k0 += 6) {
        float p[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = k0 + j < a.n_head ? k0 + j : a.n_head - 1;
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) t += yF[e] * a.w_head[(size_t)k * H + cc[e]];
            p[j] = t;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float t = wave_sum(p[j]);
            if (lane == 0 && k0 + j < a.n_head) a.out[(size_t)row * a.ld_out + k0 + j] = t + a.b_head[k0 + j];
        }
    }
}

struct TailBwdArgs {
    const float* d_out; int ld_dout; const float* w_head; int n_head;
    const float* t3; const float* mF; const float* rF; const float* gF;
    const float* x; const float* m3; const float* r3; const float* g3;
    const uint8_t* drop; float drop_scale;
    float* dx; float* dx2;
    float* dgF; float* dbF; float* dg3; float* db3; float* wsF; float* ws3;      // partials [nblocks][2][H] when nblocks > 1
    int rows, H, rows_per_block, nblocks;
};

template <int EPL>
__global__ __launch_bounds__(256) void decoder_tail_bwd_kernel(const TailBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves][4 sums][H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, H = a.H;
    const int r_begin = blockIdx.x * a.rows_per_block, r_end = min(a.rows, r_begin + a.rows_per_block);
    int cc[EPL];
    float gF[EPL], g3[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        cc[e] = c < H ? c : H - 1;
        gF[e] = a.gF[cc[e]]; g3[e] = a.g3[cc[e]];
    }
    float agF[EPL], abF[EPL], ag3[EPL], ab3[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { agF[e] = 0.f; abF[e] = 0.f; ag3[e] = 0.f; ab3[e] = 0.f; }
    for (int row = r_begin + wave; row < r_end; row += 4) {
        const float mF = a.mF[row], rF = a.rF[row], m3 = a.m3[row], r3 = a.r3[row];
        float t3[EPL], x[EPL], keep[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            t3[e] = a.t3[(size_t)row * H + cc[e]];
            x[e] = a.x[(size_t)row * H + cc[e]];
            keep[e] = a.drop ? a.drop_scale * (float)a.drop[(size_t)row * H + cc[e]] : 1.f;
        }
        // d tgtF = d_out . W_head  (each lane: its columns; d_out[row][k] is a wave-uniform scalar)
        float d[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) d[e] = 0.f;
        for (int k = 0; k < a.n_head; ++k) {
            const float dk = a.d_out[(size_t)row * a.ld_dout + k];
#pragma unroll
            for (int e = 0; e < EPL; ++e) d[e] += dk * a.w_head[(size_t)k * H + cc[e]];
        }
        // decoder.norm backward
        float xh[EPL], gg[EPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const bool in = lane + 64 * e < H;
            xh[e] = in ? (t3[e] - mF) * rF : 0.f;
            const float dd = in ? d[e] : 0.f;
            agF[e] += dd * xh[e]; abF[e] += dd;
            gg[e] = dd * gF[e];
            s1 += gg[e]; s2 += gg[e] * xh[e];
        }
        s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
        float dt[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) dt[e] = rF * (gg[e] - s1 - xh[e] * s2);
        // norm3 backward
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const bool in = lane + 64 * e < H;
            xh[e] = in ? (x[e] - m3) * r3 : 0.f;
            const float dd = in ? dt[e] : 0.f;
            ag3[e] += dd * xh[e]; ab3[e] += dd;
            gg[e] = dd * g3[e];
            u1 += gg[e]; u2 += gg[e] * xh[e];
        }
        u1 = wave_sum(u1) / (float)H; u2 = wave_sum(u2) / (float)H;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < H) {
                const float o = r3 * (gg[e] - u1 - xh[e] * u2);
                a.dx[(size_t)row * H + c] = o;
                a.dx2[(size_t)row * H + c] = o * keep[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < H) {
            red[(wave * 4 + 0) * H + c] = agF[e]; red[(wave * 4 + 1) * H + c] = abF[e];
            red[(wave * 4 + 2) * H + c] = ag3[e]; red[(wave * 4 + 3) * H + c] = ab3[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * H; i += 256) {
        const int which = i / H, c = i - which * H;
        const float s = (red[(0 * 4 + which) * H + c] + red[(1 * 4 + which) * H + c]) +
                        (red[(2 * 4 + which) * H + c] + red[(3 * 4 + which) * H + c]);
        float* fin = which == 0 ? a.dgF : which == 1 ? a.dbF : which == 2 ? a.dg3 : a.db3;
        float* ws = which < 2 ? a.wsF : a.ws3;
        if (a.nblocks == 1) fin[c] = s;
        else ws[((size_t)blockIdx.x * 2 + (which & 1)) * H + c] = s;
    }
}

}  // namespace r3d

using namespace r3d;

/* Forward tail: x = the last decoder layer's pre-norm3 rows [rows, H] (contiguous).  Outputs: t3 = norm3(x) with m3/r3,
 * tgtF = decoder.norm(t3) with mF/rF, out[rows, ld_out] (first n_head columns) = tgtF . w_head^T + b_head. */
R3D_EXPORT int r3d_decoder_tail_fwd(const float* x, const float* g3, const float* b3, const float* gF, const float* bF,
                                    const float* w_head, const float* b_head, int n_head, float* t3, float* m3, float* r3,
                                    float* tgtF, float* mF, float* rF, float* out, int ld_out, int rows, int H, void* stream) {
    R3D_REQUIRE(x && g3 && b3 && gF && bF && w_head && b_head && t3 && m3 && r3 && tgtF && mF && rF && out);
    R3D_REQUIRE(rows > 0 && H > 0 && H <= 2048 && n_head > 0 && ld_out >= n_head);
    TailFwdArgs a{x, g3, b3, gF, bF, w_head, b_head, n_head, t3, m3, r3, tgtF, mF, rF, out, ld_out, rows, H};
    const dim3 grid(r3d_cdiv(rows, 4));
    hipStream_t s = (hipStream_t)stream;
    if (H <= 128) hipLaunchKernelGGL(decoder_tail_fwd_kernel<2>, grid, dim3(256), 0, s, a);
    else if (H <= 512) hipLaunchKernelGGL(decoder_tail_fwd_kernel<8>, grid, dim3(256), 0, s, a);
    else if (H <= 1024) hipLaunchKernelGGL(decoder_tail_fwd_kernel<16>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(decoder_tail_fwd_kernel<32>, grid, dim3(256), 0, s, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Backward tail: d_out [rows, ld_dout] (first n_head columns) = gradient w.r.t. the heads' output.  Writes dx = gradient
 * w.r.t. x (pre-norm3) and dx2 = dx * dropout mask (the FFN branch, :328).  LayerNorm parameter gradients: partials
 * wsF / ws3 in r3d_layernorm_bwd's layout for `rows` rows (finalize with r3d_layernorm_bwd_finalize*), or the final values
 * in dgF/dbF/dg3/db3 when one block covers all rows. */
R3D_EXPORT int r3d_decoder_tail_bwd(const float* d_out, int ld_dout, const float* w_head, int n_head, const float* t3,
                                    const float* mF, const float* rF, const float* gF, const float* x, const float* m3,
                                    const float* r3, const float* g3, const uint8_t* drop_mask, float drop_scale, float* dx,
                                    float* dx2, float* dgF, float* dbF, float* dg3, float* db3, float* wsF, float* ws3,
                                    int rows, int H, void* stream) {
    R3D_REQUIRE(d_out && w_head && t3 && mF && rF && gF && x && m3 && r3 && g3 && dx && dx2 && dgF && dbF && dg3 && db3);
    R3D_REQUIRE(rows > 0 && H > 0 && H <= 2048 && n_head > 0 && ld_dout >= n_head);
    int rpb = (r3d_cdiv(rows, 256) + 3) / 4 * 4;          // = r3d_layernorm_bwd's blocking
    if (rpb < 4) rpb = 4;
    const int blocks = r3d_cdiv(rows, rpb);
    R3D_REQUIRE(blocks == 1 || (wsF && ws3));
    TailBwdArgs a{d_out, ld_dout, w_head, n_head, t3, mF, rF, gF, x, m3, r3, g3, drop_mask, drop_scale, dx, dx2,
                  dgF, dbF, dg3, db3, wsF, ws3, rows, H, rpb, blocks};
    const size_t shmem = (size_t)16 * H * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (H <= 128) hipLaunchKernelGGL(decoder_tail_bwd_kernel<2>, dim3(blocks), dim3(256), shmem, s, a);
    else if (H <= 512) hipLaunchKernelGGL(decoder_tail_bwd_kernel<8>, dim3(blocks), dim3(256), shmem, s, a);
    else if (H <= 1024) hipLaunchKernelGGL(decoder_tail_bwd_kernel<16>, dim3(blocks), dim3(256), shmem, s, a);
    else {
        hipError_t e = hipFuncSetAttribute((const void*)decoder_tail_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(decoder_tail_bwd_kernel<32>, dim3(blocks), dim3(256), shmem, s, a);
    }
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
