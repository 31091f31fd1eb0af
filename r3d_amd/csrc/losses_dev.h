// Device code shared by the loss launches (losses.hip) and the one-launch decoder (decoder_chain.hip): the three losses of
// train/train_proposed_depth.py:171-213 per row / clip, and the decoder tail + losses + tail backward of one clip.
#pragma once
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

struct LossArgs {
    const float* seg; int ld_seg; const float* act; int ld_act; const float* dur; int ld_dur;
    const int64_t* past_label; const int64_t* target; const float* target_dur;
    int B, S, Q, K, pad_idx, exclude_idx; int val_mode;
    const float* dur_den; float grad_scale;
    float* d_seg; int ld_dseg; float* d_act; int ld_dact; float* d_dur; int ld_ddur;
    float* loss_out; int64_t* counts;
    int64_t* tick_a; int64_t* tick_b;      // optional: ++*tick_a, ++*tick_b once per call (step counter, dropout offset)
};

// CE of one row held across a wave.  Returns loss contribution; writes gradient (softmax - onehot) * gscale if dl.
__device__ __forceinline__ float ce_row(const float* logit, int K, int64_t label, bool valid, int pad_idx,
                                        float gscale, float* dl, int lane, int* argmax_out) {
    float m = -INFINITY;
    int am = 0x7fffffff;
    if (K <= 64) {
        // one class per lane: the maximum by DPP, its first occurrence (smallest index, as torch.max on CPU) by ballot
        const float x = lane < K ? logit[lane] : -INFINITY;
        m = wave_max(x);
        const unsigned long long hit = __ballot(lane < K && x == m);
        am = hit ? __builtin_ctzll(hit) : 0x7fffffff;
    } else {
        for (int c = lane; c < K; c += 64) {
            const float x = logit[c];
            if (x > m || (x == m && c < am)) { m = x; am = c; }
        }
        // wave arg-max: largest value, smallest index among equals
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(m, off, 64);
            const int oa = __shfl_xor(am, off, 64);
            if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        }
    }
    *argmax_out = am;
    float se = 0.f;
    for (int c = lane; c < K; c += 64) se += expf(logit[c] - m);
    se = wave_sum(se);
    const float lse = m + logf(se);
    float loss = 0.f;
    if (valid) loss = lse - logit[label];
    if (dl) {
        for (int c = lane; c < K; c += 64) {
            float g = 0.f;
            if (valid) g = (expf(logit[c] - lse) - ((int64_t)c == label ? 1.f : 0.f)) * gscale;
            dl[c] = g;
        }
    }
    return loss;
}

// Unit u of the row grid: [0,N) segmentation rows, [N,N+BQ) anticipation rows, [N+BQ, N+BQ+B) duration clips.
// One wave per unit (the work is latency-bound, so units must not queue inside a wave); each writes
// part[u] = {loss contribution, correct, valid, 0}; losses_finalize_kernel adds them in a fixed order.
__device__ __forceinline__ void losses_unit(const LossArgs& a, float* part, int u, int lane) {
    const int N = a.B * a.S, BQ = a.B * a.Q;
    float out_l = 0.f, out_c = 0.f, out_v = 0.f;
    if (u < N) {
        if (a.seg) {
            const int64_t lab = a.past_label[u];
            // labels outside [0,K) that are neither pad nor excluded would raise in PyTorch; here they are ignored
            const bool valid = (lab != (int64_t)a.pad_idx) && (lab != (int64_t)a.exclude_idx) && lab >= 0 && lab < a.K;
            int am;
            float l = ce_row(a.seg + (size_t)u * a.ld_seg, a.K, lab, valid, a.pad_idx, a.grad_scale / (float)N,
                             a.d_seg ? a.d_seg + (size_t)u * a.ld_dseg : nullptr, lane, &am);
            if (valid && am == a.pad_idx) l += 2.0f;          // penalty term of cal_loss (utils.py:481-486)
            out_l = l; out_v = valid ? 1.f : 0.f; out_c = (valid && (int64_t)am == lab) ? 1.f : 0.f;
        }
    } else if (u < N + BQ) {
        const int r = u - N, b = r / a.Q;
        // per-clip weight: last observed (non-pad) label vs first future label (train_proposed_depth.py:28-50, utils.py:439)
        int last = -1;
        for (int s = lane; s < a.S; s += 64)
            if (a.past_label[(size_t)b * a.S + s] != (int64_t)a.pad_idx) last = s;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
        const int64_t ref = (last >= 0) ? a.past_label[(size_t)b * a.S + last] : (int64_t)a.pad_idx;
        const float w = (ref == a.target[(size_t)b * a.Q]) ? 1.0f : 10.0f;
        const int64_t lab = a.target[r];
        const bool valid = (lab != (int64_t)a.pad_idx) && (lab != (int64_t)a.exclude_idx) && lab >= 0 && lab < a.K;
        int am;
        const float l = ce_row(a.act + (size_t)r * a.ld_act, a.K, lab, valid, a.pad_idx, w * a.grad_scale / (float)BQ,
                               a.d_act ? a.d_act + (size_t)r * a.ld_dact : nullptr, lane, &am);
        out_l = l * w; out_v = valid ? 1.f : 0.f; out_c = (valid && (int64_t)am == lab) ? 1.f : 0.f;
    } else {
        const int b = u - N - BQ;
        // global duration-mask count (needed for the gradient scale); every duration wave recomputes it (B*Q is tiny)
        float mc = 0.f;
        for (int e = lane; e < BQ; e += 64) mc += (a.target_dur[e] != (float)a.pad_idx) ? 1.f : 0.f;
        mc = wave_sum(mc);
        const float dur_den = a.dur_den ? *a.dur_den : mc;
        float ssum = 0.f;
        for (int q = lane; q < a.Q; q += 64) {
            const float td = a.target_dur[(size_t)b * a.Q + q];
            const float mk = (td != (float)a.pad_idx) ? 1.f : 0.f;
            ssum += fabsf(expf(a.dur[((size_t)b * a.Q + q) * a.ld_dur]) * mk);
        }
        ssum = wave_sum(ssum);
        const float den = fmaxf(ssum, 1e-12f);
        float sq = 0.f, gp = 0.f;
        for (int q = lane; q < a.Q; q += 64) {
            const float td = a.target_dur[(size_t)b * a.Q + q];
            const float mk = (td != (float)a.pad_idx) ? 1.f : 0.f;
            const float p = expf(a.dur[((size_t)b * a.Q + q) * a.ld_dur]) * mk / den;
            const float t = a.val_mode ? td : td * mk * mk;
            const float diff = p - t;
            sq += diff * diff;
            gp += (2.f * diff / dur_den) * p;
        }
        sq = wave_sum(sq);
        gp = wave_sum(gp);
        if (a.d_dur) {
            for (int q = lane; q < a.Q; q += 64) {
                const float td = a.target_dur[(size_t)b * a.Q + q];
                const float mk = (td != (float)a.pad_idx) ? 1.f : 0.f;
                const float p = expf(a.dur[((size_t)b * a.Q + q) * a.ld_dur]) * mk / den;
                const float t = a.val_mode ? td : td * mk * mk;
                const float g = 2.f * (p - t) / dur_den;
                const float dd = (ssum >= 1e-12f) ? p * (g - gp) : 0.f;
                a.d_dur[((size_t)b * a.Q + q) * a.ld_ddur] = dd * a.grad_scale;
            }
        }
        out_l = sq; out_v = mc;             // every duration unit carries the same mask count
    }
    if (lane == 0) {
        // agent-scope (write-through) stores: they leave this XCD's L2, so the finishing workgroup on another XCD sees
        // them without anybody paying a whole-L2 write-back fence
        __hip_atomic_store(part + 4 * (size_t)u + 0, out_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 4 * (size_t)u + 1, out_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 4 * (size_t)u + 2, out_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void losses_finalize(const LossArgs& a, const float* part) {
    __shared__ double red[4][3][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.B * a.S, BQ = a.B * a.Q;
    // wave w sums units w, w+256... in a fixed order; 3 groups x {loss, correct, valid}
    double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int u = threadIdx.x; u < N + BQ + a.B; u += 256) {
        float4 v;
        v.x = __hip_atomic_load(part + 4 * (size_t)u + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load(part + 4 * (size_t)u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load(part + 4 * (size_t)u + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int g = u < N ? 0 : (u < N + BQ ? 1 : 2);
        acc[g][0] += v.x; acc[g][1] += v.y; acc[g][2] += v.z;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double t = wave_sum_d(acc[g][k]);
            if (lane == 0) red[wave][g][k] = t;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3][3];
        for (int g = 0; g < 3; ++g)
            for (int k = 0; k < 3; ++k) t[g][k] = (red[0][g][k] + red[1][g][k]) + (red[2][g][k] + red[3][g][k]);
        const double msum = t[2][2] / (double)a.B;                 // every clip reported the same global count
        const double dur_den = a.dur_den ? (double)*a.dur_den : msum;
        const float ls = a.seg ? (float)(t[0][0] / (double)N) : 0.f;
        const float la = (float)(t[1][0] / (double)BQ);
        const float ld = (float)(t[2][0] / dur_den);
        a.loss_out[0] = ls; a.loss_out[1] = la; a.loss_out[2] = ld; a.loss_out[3] = ls + la + ld;
        a.counts[0] = (int64_t)(t[0][1] + 0.5); a.counts[1] = (int64_t)(t[0][2] + 0.5);
        a.counts[2] = (int64_t)(t[1][1] + 0.5); a.counts[3] = (int64_t)(t[1][2] + 0.5);
        if (a.tick_a) *a.tick_a += 1;
        if (a.tick_b) *a.tick_b += 1;
    }
}

constexpr float kLnEpsTL = 1e-5f;

__device__ __forceinline__ void ln2_apply(const float (&x)[2], const float (&g)[2], const float (&b)[2], int H, int lane,
                                          float (&y)[2], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) s += (lane + 64 * e < H) ? x[e] : 0.f;
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dl = (lane + 64 * e < H) ? x[e] - mean : 0.f;
        q += dl * dl;
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + kLnEpsTL);
#pragma unroll
    for (int e = 0; e < 2; ++e) y[e] = (lane + 64 * e < H) ? (x[e] - mean) * rstd * g[e] + b[e] : 0.f;
}

// same partial sums over 8 waves instead of 4 (fixed order)
__device__ __forceinline__ void losses_finalize8(const LossArgs& a, const float* part) {
    __shared__ double red8[8][3][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.B * a.S, BQ = a.B * a.Q;
    double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int u = threadIdx.x; u < N + BQ + a.B; u += 512) {
        const float vx = __hip_atomic_load(part + 4 * (size_t)u + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float vy = __hip_atomic_load(part + 4 * (size_t)u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float vz = __hip_atomic_load(part + 4 * (size_t)u + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int g = u < N ? 0 : (u < N + BQ ? 1 : 2);
        acc[g][0] += vx; acc[g][1] += vy; acc[g][2] += vz;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double t = wave_sum_d(acc[g][k]);
            if (lane == 0) red8[wave][g][k] = t;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3][3];
        for (int g = 0; g < 3; ++g)
            for (int k = 0; k < 3; ++k)
                t[g][k] = ((red8[0][g][k] + red8[1][g][k]) + (red8[2][g][k] + red8[3][g][k])) +
                          ((red8[4][g][k] + red8[5][g][k]) + (red8[6][g][k] + red8[7][g][k]));
        const double msum = t[2][2] / (double)a.B;
        const double dur_den = a.dur_den ? (double)*a.dur_den : msum;
        const float ls = a.seg ? (float)(t[0][0] / (double)N) : 0.f;
        const float la = (float)(t[1][0] / (double)BQ);
        const float ld = (float)(t[2][0] / dur_den);
        a.loss_out[0] = ls; a.loss_out[1] = la; a.loss_out[2] = ld; a.loss_out[3] = ls + la + ld;
        a.counts[0] = (int64_t)(t[0][1] + 0.5); a.counts[1] = (int64_t)(t[0][2] + 0.5);
        a.counts[2] = (int64_t)(t[1][1] + 0.5); a.counts[3] = (int64_t)(t[1][2] + 0.5);
        if (a.tick_a) *a.tick_a += 1;
        if (a.tick_b) *a.tick_b += 1;
    }
}

constexpr int kTLHeads = 24;          // head outputs kept in registers / LDS per row (K + 1 <= 24)

// The clip role of tail_losses_kernel (see losses.hip): workgroup of 8 waves, wave q = query row b*Q + q.  lg / dl: [8][kTLHeads + 8]
// logits and their gradients, red: [8][4][128] LayerNorm parameter partials (workgroup-shared scratch).  Two workgroup barriers.
__device__ __forceinline__ void tail_clip_body(const r3d_tail_losses_args& t, const LossArgs& a, float* part, const int b,
                                               float (*lg)[kTLHeads + 8], float (*dl)[kTLHeads + 8], float (*red)[4][128]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = t.H, K = a.K, Q = a.Q, NH = t.n_head, N = a.B * a.S, BQ = a.B * a.Q;
    const int row = b * Q + wave;            // Q == 8 == waves (validated by the host)
    int cc[2];
    float x[2], g3[2], b3[2], gF[2], bF[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = lane + 64 * e;
        cc[e] = c < H ? c : H - 1;
        x[e] = t.x[(size_t)row * H + cc[e]];
        g3[e] = t.g3[cc[e]]; b3[e] = t.b3[cc[e]]; gF[e] = t.gF[cc[e]]; bF[e] = t.bF[cc[e]];
    }
    float wh[kTLHeads][2];
#pragma unroll
    for (int k = 0; k < kTLHeads; ++k) {
        const int kc = k < NH ? k : NH - 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) wh[k][e] = t.w_head[(size_t)kc * H + cc[e]];
    }
    // backward operands that do not depend on anything computed here: issued now, consumed after the barrier
    float keep[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) keep[e] = t.drop ? t.drop_scale * (float)t.drop[(size_t)row * H + cc[e]] : 1.f;
    // ... and the clip's labels, targets and durations (S <= 64, B*Q <= 64: one per lane -- validated by the host):
    // one round trip under the forward tail instead of four dependent ones after the barrier
    const bool small = a.S <= 64 && BQ <= 64;                    // (beyond that the label / mask scans loop, below)
    const int64_t pl_pre = a.past_label[(size_t)b * a.S + (lane < a.S ? lane : 0)];
    const int64_t tgt_first = a.target[(size_t)b * Q];
    const int64_t tgt_row = a.target[row];
    const float td_all = a.target_dur[lane < BQ ? lane : 0];
    const float td_clip = a.target_dur[(size_t)b * Q + (lane < Q ? lane : 0)];
    const float dden_pre = a.dur_den ? *a.dur_den : 0.f;
    // ---- forward tail: norm3 -> decoder.norm -> heads
    float y3[2], yF[2], m3, r3, mF, rF;
    ln2_apply(x, g3, b3, H, lane, y3, m3, r3);
    ln2_apply(y3, gF, bF, H, lane, yF, mF, rF);
    if (lane == 0) { t.m3[row] = m3; t.r3[row] = r3; t.mF[row] = mF; t.rF[row] = rF; }
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (lane + 64 * e < H) {
            t.t3[(size_t)row * H + lane + 64 * e] = y3[e];
            t.tgtF[(size_t)row * H + lane + 64 * e] = yF[e];
        }
    {
        float p[kTLHeads];
#pragma unroll
        for (int k = 0; k < kTLHeads; ++k) p[k] = yF[0] * wh[k][0] + yF[1] * wh[k][1];
#pragma unroll
        for (int k = 0; k < kTLHeads; ++k) {
            if (k < NH) {                                     // wave-uniform
                const float v = wave_sum(p[k]) + t.b_head[k];
                if (lane == 0) { t.out[(size_t)row * t.ld_out + k] = v; lg[wave][k] = v; }
            }
        }
    }
    __syncthreads();
    // ---- losses of this row: anticipation CE (cal_weighted_loss) ...
    float out_l, out_c, out_v;
    {
        // last observed (non-pad) label of the clip: highest lane whose label is not the pad index
        int last = -1;
        int64_t ref = (int64_t)a.pad_idx;
        if (small) {
            const unsigned long long obs = __ballot(lane < a.S && pl_pre != (int64_t)a.pad_idx);
            last = obs ? 63 - __builtin_clzll(obs) : -1;
            const int lsel = __builtin_amdgcn_readfirstlane(last < 0 ? 0 : last);
            const unsigned rlo = __builtin_amdgcn_readlane((unsigned)(unsigned long long)pl_pre, lsel);
            const unsigned rhi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)pl_pre >> 32), lsel);
            if (last >= 0) ref = (int64_t)(((unsigned long long)rhi << 32) | rlo);
        } else {
            for (int s = lane; s < a.S; s += 64)
                if (a.past_label[(size_t)b * a.S + s] != (int64_t)a.pad_idx) last = s;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
            if (last >= 0) ref = a.past_label[(size_t)b * a.S + last];
        }
        const float w = (ref == tgt_first) ? 1.0f : 10.0f;
        const int64_t lab = tgt_row;
        const bool valid = (lab != (int64_t)a.pad_idx) && (lab != (int64_t)a.exclude_idx) && lab >= 0 && lab < K;
        int am;
        const float l = ce_row(&lg[wave][0], K, lab, valid, a.pad_idx, w * a.grad_scale / (float)BQ, &dl[wave][0], lane, &am);
        out_l = l * w; out_v = valid ? 1.f : 0.f; out_c = (valid && (int64_t)am == lab) ? 1.f : 0.f;
        if (lane == 0) {
            __hip_atomic_store(part + 4 * (size_t)(N + row) + 0, out_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + 4 * (size_t)(N + row) + 1, out_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + 4 * (size_t)(N + row) + 2, out_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ... and the duration MSE (normalize_duration over the clip's Q queries; every wave redoes the clip's sums,
    // wave 0 reports the clip's loss term)
    {
        float mc = 0.f;
        if (small) mc = (lane < BQ && td_all != (float)a.pad_idx) ? 1.f : 0.f;
        else
            for (int e = lane; e < BQ; e += 64) mc += (a.target_dur[e] != (float)a.pad_idx) ? 1.f : 0.f;
        mc = wave_sum(mc);
        const float dur_den = a.dur_den ? dden_pre : mc;
        const float mkq = (lane < Q && td_clip != (float)a.pad_idx) ? 1.f : 0.f;      // lane q < Q holds query q
        const float eq = lane < Q ? expf(lg[lane < Q ? lane : 0][K]) * mkq : 0.f;
        const float ssum = wave_sum(fabsf(eq));
        const float den = fmaxf(ssum, 1e-12f);
        float sq = 0.f, gp = 0.f;
        if (lane < Q) {
            const float p = eq / den;
            const float tt = td_clip * mkq * mkq;
            const float diff = p - tt;
            sq = diff * diff;
            gp = (2.f * diff / dur_den) * p;
        }
        sq = wave_sum(sq);
        gp = wave_sum(gp);
        const float td_w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(unsigned, td_clip),
                                                                               __builtin_amdgcn_readfirstlane(wave)));
        if (lane == 0) {
            const float td = td_w;
            const float mk = (td != (float)a.pad_idx) ? 1.f : 0.f;
            const float p = expf(lg[wave][K]) * mk / den;
            const float g = 2.f * (p - td * mk * mk) / dur_den;
            dl[wave][K] = ((ssum >= 1e-12f) ? p * (g - gp) : 0.f) * a.grad_scale;
            if (wave == 0) {
                const size_t u = (size_t)N + BQ + b;
                __hip_atomic_store(part + 4 * u + 0, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + 4 * u + 1, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + 4 * u + 2, mc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the row's gradient leaves for the heads' weight gradient (a later grouped GEMM reads it from memory)
    for (int k = lane; k < NH; k += 64) t.d_out[(size_t)row * t.ld_dout + k] = dl[wave][k];
    // ---- backward tail: heads' input gradient -> decoder.norm backward -> norm3 backward (+ dropout3)
    float d[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kTLHeads; ++k) {
        if (k < NH) {
            const float dk = dl[wave][k];
            d[0] += dk * wh[k][0]; d[1] += dk * wh[k][1];
        }
    }
    float agF[2], abF[2], ag3[2], ab3[2], xh[2], gg[2], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const bool in = lane + 64 * e < H;
        xh[e] = in ? (y3[e] - mF) * rF : 0.f;
        const float dd = in ? d[e] : 0.f;
        agF[e] = dd * xh[e]; abF[e] = dd;
        gg[e] = dd * gF[e];
        s1 += gg[e]; s2 += gg[e] * xh[e];
    }
    s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
    float dt[2], u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) dt[e] = rF * (gg[e] - s1 - xh[e] * s2);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const bool in = lane + 64 * e < H;
        xh[e] = in ? (x[e] - m3) * r3 : 0.f;
        const float dd = in ? dt[e] : 0.f;
        ag3[e] = dd * xh[e]; ab3[e] = dd;
        gg[e] = dd * g3[e];
        u1 += gg[e]; u2 += gg[e] * xh[e];
    }
    u1 = wave_sum(u1) / (float)H; u2 = wave_sum(u2) / (float)H;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = lane + 64 * e;
        if (c < H) {
            const float o = r3 * (gg[e] - u1 - xh[e] * u2);
            t.dx[(size_t)row * H + c] = o;
            t.dx2[(size_t)row * H + c] = o * keep[e];
            red[wave][0][c] = agF[e]; red[wave][1][c] = abF[e]; red[wave][2][c] = ag3[e]; red[wave][3][c] = ab3[e];
        }
    }
    __syncthreads();
    // partials per 4 rows (waves 0-3 -> partial block 2b, waves 4-7 -> 2b+1): r3d_layernorm_bwd's layout for B*Q rows
    for (int i = threadIdx.x; i < 2 * 4 * H; i += 512) {
        const int half = i / (4 * H), j = i - half * 4 * H, which = j / H, c = j - which * H;
        const int w0 = 4 * half;
        const float s = (red[w0][which][c] + red[w0 + 1][which][c]) + (red[w0 + 2][which][c] + red[w0 + 3][which][c]);
        float* ws = which < 2 ? t.wsF : t.ws3;
        ws[((size_t)(2 * b + half) * 2 + (which & 1)) * H + c] = s;
    }
}

template <int EPL>
__device__ __forceinline__ void lnw_apply(const float (&x)[EPL], const float (&g)[EPL], const float (&b)[EPL], int H, int lane,
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
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + kLnEpsTL);
#pragma unroll
    for (int e = 0; e < EPL; ++e) y[e] = (lane + 64 * e < H) ? (x[e] - mean) * rstd * g[e] + b[e] : 0.f;
}

// The same role for hidden sizes up to 64 EPL (tail_losses_wide_kernel: EPL = 8, hidden <= 512): a lane holds EPL columns of its
// row; the head weights (n_head x H) and the LayerNorm parameter partials ([8][4][H]) live in dynamic LDS instead of registers /
// a fixed [8][4][128] array (whs is filled by the caller, behind a barrier).  Same arithmetic and order per row.
template <int EPL>
__device__ __forceinline__ void tail_clip_body_wide(const r3d_tail_losses_args& t, const LossArgs& a, float* part, const int b,
                                                    float (*lg)[kTLHeads + 8], float (*dl)[kTLHeads + 8], const float* whs,
                                                    float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = t.H, K = a.K, Q = a.Q, NH = t.n_head, N = a.B * a.S, BQ = a.B * a.Q;
    const int row = b * Q + wave;            // Q == 8 == waves (validated by the host)
    int cc[EPL];
    float x[EPL], g3[EPL], b3[EPL], gF[EPL], bF[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        cc[e] = c < H ? c : H - 1;
        x[e] = t.x[(size_t)row * H + cc[e]];
        g3[e] = t.g3[cc[e]]; b3[e] = t.b3[cc[e]]; gF[e] = t.gF[cc[e]]; bF[e] = t.bF[cc[e]];
    }
    // backward operands that do not depend on anything computed here: issued now, consumed after the barrier
    float keep[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) keep[e] = t.drop ? t.drop_scale * (float)t.drop[(size_t)row * H + cc[e]] : 1.f;
    // ... and the clip's labels, targets and durations (S <= 64, B*Q <= 64: one per lane -- validated by the host):
    // one round trip under the forward tail instead of four dependent ones after the barrier
    const bool small = a.S <= 64 && BQ <= 64;                    // (beyond that the label / mask scans loop, below)
    const int64_t pl_pre = a.past_label[(size_t)b * a.S + (lane < a.S ? lane : 0)];
    const int64_t tgt_first = a.target[(size_t)b * Q];
    const int64_t tgt_row = a.target[row];
    const float td_all = a.target_dur[lane < BQ ? lane : 0];
    const float td_clip = a.target_dur[(size_t)b * Q + (lane < Q ? lane : 0)];
    const float dden_pre = a.dur_den ? *a.dur_den : 0.f;
    // ---- forward tail: norm3 -> decoder.norm -> heads
    float y3[EPL], yF[EPL], m3, r3, mF, rF;
    lnw_apply<EPL>(x, g3, b3, H, lane, y3, m3, r3);
    lnw_apply<EPL>(y3, gF, bF, H, lane, yF, mF, rF);
    if (lane == 0) { t.m3[row] = m3; t.r3[row] = r3; t.mF[row] = mF; t.rF[row] = rF; }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (lane + 64 * e < H) {
            t.t3[(size_t)row * H + lane + 64 * e] = y3[e];
            t.tgtF[(size_t)row * H + lane + 64 * e] = yF[e];
        }
    {
        float p[kTLHeads];
#pragma unroll
        for (int k = 0; k < kTLHeads; ++k) {
            const int kc = k < NH ? k : NH - 1;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s += yF[e] * whs[(size_t)kc * H + cc[e]];      // (yF is 0 beyond H)
            p[k] = s;
        }
#pragma unroll
        for (int k = 0; k < kTLHeads; ++k) {
            if (k < NH) {                                     // wave-uniform
                const float v = wave_sum(p[k]) + t.b_head[k];
                if (lane == 0) { t.out[(size_t)row * t.ld_out + k] = v; lg[wave][k] = v; }
            }
        }
    }
    __syncthreads();
    // ---- losses of this row: anticipation CE (cal_weighted_loss) ...
    float out_l, out_c, out_v;
    {
        // last observed (non-pad) label of the clip: highest lane whose label is not the pad index
        int last = -1;
        int64_t ref = (int64_t)a.pad_idx;
        if (small) {
            const unsigned long long obs = __ballot(lane < a.S && pl_pre != (int64_t)a.pad_idx);
            last = obs ? 63 - __builtin_clzll(obs) : -1;
            const int lsel = __builtin_amdgcn_readfirstlane(last < 0 ? 0 : last);
            const unsigned rlo = __builtin_amdgcn_readlane((unsigned)(unsigned long long)pl_pre, lsel);
            const unsigned rhi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)pl_pre >> 32), lsel);
            if (last >= 0) ref = (int64_t)(((unsigned long long)rhi << 32) | rlo);
        } else {
            for (int s = lane; s < a.S; s += 64)
                if (a.past_label[(size_t)b * a.S + s] != (int64_t)a.pad_idx) last = s;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
            if (last >= 0) ref = a.past_label[(size_t)b * a.S + last];
        }
        const float w = (ref == tgt_first) ? 1.0f : 10.0f;
        const int64_t lab = tgt_row;
        const bool valid = (lab != (int64_t)a.pad_idx) && (lab != (int64_t)a.exclude_idx) && lab >= 0 && lab < K;
        int am;
        const float l = ce_row(&lg[wave][0], K, lab, valid, a.pad_idx, w * a.grad_scale / (float)BQ, &dl[wave][0], lane, &am);
        out_l = l * w; out_v = valid ? 1.f : 0.f; out_c = (valid && (int64_t)am == lab) ? 1.f : 0.f;
        if (lane == 0) {
            __hip_atomic_store(part + 4 * (size_t)(N + row) + 0, out_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + 4 * (size_t)(N + row) + 1, out_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + 4 * (size_t)(N + row) + 2, out_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ... and the duration MSE (normalize_duration over the clip's Q queries; every wave redoes the clip's sums,
    // wave 0 reports the clip's loss term)
    {
        float mc = 0.f;
        if (small) mc = (lane < BQ && td_all != (float)a.pad_idx) ? 1.f : 0.f;
        else
            for (int e = lane; e < BQ; e += 64) mc += (a.target_dur[e] != (float)a.pad_idx) ? 1.f : 0.f;
        mc = wave_sum(mc);
        const float dur_den = a.dur_den ? dden_pre : mc;
        const float mkq = (lane < Q && td_clip != (float)a.pad_idx) ? 1.f : 0.f;      // lane q < Q holds query q
        const float eq = lane < Q ? expf(lg[lane < Q ? lane : 0][K]) * mkq : 0.f;
        const float ssum = wave_sum(fabsf(eq));
        const float den = fmaxf(ssum, 1e-12f);
        float sq = 0.f, gp = 0.f;
        if (lane < Q) {
            const float p = eq / den;
            const float tt = td_clip * mkq * mkq;
            const float diff = p - tt;
            sq = diff * diff;
            gp = (2.f * diff / dur_den) * p;
        }
        sq = wave_sum(sq);
        gp = wave_sum(gp);
        const float td_w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(unsigned, td_clip),
                                                                               __builtin_amdgcn_readfirstlane(wave)));
        if (lane == 0) {
            const float td = td_w;
            const float mk = (td != (float)a.pad_idx) ? 1.f : 0.f;
            const float p = expf(lg[wave][K]) * mk / den;
            const float g = 2.f * (p - td * mk * mk) / dur_den;
            dl[wave][K] = ((ssum >= 1e-12f) ? p * (g - gp) : 0.f) * a.grad_scale;
            if (wave == 0) {
                const size_t u = (size_t)N + BQ + b;
                __hip_atomic_store(part + 4 * u + 0, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + 4 * u + 1, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + 4 * u + 2, mc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the row's gradient leaves for the heads' weight gradient (a later grouped GEMM reads it from memory)
    for (int k = lane; k < NH; k += 64) t.d_out[(size_t)row * t.ld_dout + k] = dl[wave][k];
    // ---- backward tail: heads' input gradient -> decoder.norm backward -> norm3 backward (+ dropout3)
    float d[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) d[e] = 0.f;
#pragma unroll
    for (int k = 0; k < kTLHeads; ++k) {
        if (k < NH) {
            const float dk = dl[wave][k];
#pragma unroll
            for (int e = 0; e < EPL; ++e) d[e] += dk * whs[(size_t)k * H + cc[e]];
        }
    }
    float agF[EPL], abF[EPL], ag3[EPL], ab3[EPL], xh[EPL], gg[EPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool in = lane + 64 * e < H;
        xh[e] = in ? (y3[e] - mF) * rF : 0.f;
        const float dd = in ? d[e] : 0.f;
        agF[e] = dd * xh[e]; abF[e] = dd;
        gg[e] = dd * gF[e];
        s1 += gg[e]; s2 += gg[e] * xh[e];
    }
    s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
    float dt[EPL], u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) dt[e] = rF * (gg[e] - s1 - xh[e] * s2);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool in = lane + 64 * e < H;
        xh[e] = in ? (x[e] - m3) * r3 : 0.f;
        const float dd = in ? dt[e] : 0.f;
        ag3[e] = dd * xh[e]; ab3[e] = dd;
        gg[e] = dd * g3[e];
        u1 += gg[e]; u2 += gg[e] * xh[e];
    }
    u1 = wave_sum(u1) / (float)H; u2 = wave_sum(u2) / (float)H;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < H) {
            const float o = r3 * (gg[e] - u1 - xh[e] * u2);
            t.dx[(size_t)row * H + c] = o;
            t.dx2[(size_t)row * H + c] = o * keep[e];
            red[(wave * 4 + 0) * H + c] = agF[e]; red[(wave * 4 + 1) * H + c] = abF[e];
            red[(wave * 4 + 2) * H + c] = ag3[e]; red[(wave * 4 + 3) * H + c] = ab3[e];
        }
    }
    __syncthreads();
    // partials per 4 rows (waves 0-3 -> partial block 2b, waves 4-7 -> 2b+1): r3d_layernorm_bwd's layout for B*Q rows
    for (int i = threadIdx.x; i < 2 * 4 * H; i += 512) {
        const int half = i / (4 * H), j = i - half * 4 * H, which = j / H, c = j - which * H;
        const int w0 = 4 * half;
        const float s = (red[(w0 * 4 + which) * H + c] + red[((w0 + 1) * 4 + which) * H + c]) +
                        (red[((w0 + 2) * 4 + which) * H + c] + red[((w0 + 3) * 4 + which) * H + c]);
        float* ws = which < 2 ? t.wsF : t.ws3;
        ws[((size_t)(2 * b + half) * 2 + (which & 1)) * H + c] = s;
    }
}

// The common end of the loss launches: counters / deferred or last-arrival reduction of the loss partials.
__device__ __forceinline__ void tail_losses_finish(const r3d_tail_losses_args& t, const LossArgs& a, float* part,
                                                   unsigned* arrivals, int* is_last) {
    if (t.defer_finalize) {
        // the partials stay in `part` for a later launch (r3d_losses_finalize / the AdamW launch's extra workgroup): no
        // agent-scope arrival round trip and no last-workgroup pass at the end of this chain; only the counters tick here
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.tick_a) *a.tick_a += 1;
            if (a.tick_b) *a.tick_b += 1;
        }
        return;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0)
        *is_last = (__hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!*is_last) return;
    losses_finalize8(a, part);
    if (threadIdx.x == 0) *arrivals = 0u;
}

}  // namespace r3d
