// The small-shape attention core (Lk <= 64 keys, compile-time head width DH and query count LQ) as a per-wave device
// function: one wave handles one (clip, head) unit with LQ*DH + LQ*64 + 64*(DH+1) floats of LDS of its own, so the same
// code is the body of mha_fwd_small_kernel (attention.hip; one 64-thread workgroup per unit) and a ROLE of other launches
// (gemm_ln.hip: the parameter-only query self-attention of decoder layer 0 rides in the fuser's attn.proj + norm2 launch).
#pragma once
#include "common.h"

namespace r3d {

struct MhaArgs {
    const float* q; int ldq; const float* k; int ldk; const float* v; int ldv;
    const uint8_t* kpm;                 // [B][Lk], 1 = padded key (cross attention only), or NULL
    const int64_t* key_label; int pad_idx;   // alternative form: key j of clip b is padded iff key_label[b][j] == pad_idx
    float* probs;                       // [B][heads][Lq][Lk] softmax output BEFORE dropout (saved for backward)
    const uint8_t* drop; float drop_scale;   // [B][heads][Lq][Lk] keep mask, or NULL
    float* o; int ldo;                  // fwd: attention output rows (b*Lq + i), columns h*dh + d
    const float* d_o; int lddo;         // bwd
    float* dq; int lddq; float* dk; int lddk; float* dv; int lddv;
    int B, heads, Lq, Lk, dh; float scale;
};

constexpr int mha_small_lds_floats(int DH, int LQ) { return LQ * DH + LQ * 64 + 64 * (DH + 1); }

// WG = true: the unit is a whole workgroup (one wave) -> workgroup barrier.  WG = false: the unit is one wave of a larger
// workgroup; its LDS region is private to the wave and a wave's LDS operations execute in order, so only the compiler
// has to be kept from reordering.
template <bool WG> __device__ __forceinline__ void mha_unit_sync() {
    if (WG) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
}

template <int DH, int LQ, bool WG>
__device__ __forceinline__ void mha_fwd_small_unit(const MhaArgs& a, const int unit, float* lds) {
    float* qs = lds;                              // [LQ][DH], 16-byte aligned
    float* sc = lds + LQ * DH;                    // [LQ][64]
    float* vc = sc + LQ * 64;                     // [64][DH + 1]
    const int b = unit / a.heads, h = unit % a.heads;
    const int lane = threadIdx.x & 63, Lk = a.Lk;
    const int jc = lane < Lk ? lane : Lk - 1;
    const float* qb = a.q + (size_t)b * LQ * a.ldq + h * DH;
    const float4* kr = reinterpret_cast<const float4*>(a.k + ((size_t)b * Lk + jc) * a.ldk + h * DH);
    const float4* vr = reinterpret_cast<const float4*>(a.v + ((size_t)b * Lk + jc) * a.ldv + h * DH);
    const size_t pbase = ((size_t)(b * a.heads + h) * LQ) * Lk;
    float4 k4[DH / 4], v4[DH / 4];
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) { k4[t] = kr[t]; v4[t] = vr[t]; }
    float qv[(LQ * DH + 63) / 64];
#pragma unroll
    for (int t = 0; t < (LQ * DH + 63) / 64; ++t) {
        const int e = lane + 64 * t, ec = e < LQ * DH ? e : LQ * DH - 1;
        qv[t] = qb[(size_t)(ec / DH) * a.ldq + (ec % DH)];
    }
    // optional operands as UNCONDITIONAL loads (a load under a branch is waited for on the spot): an absent one reads a
    // valid address (the key row) and is discarded; the raw values are converted only after the staging barrier
    const bool has_drop = a.drop != nullptr, has_kpm = a.kpm != nullptr, has_lab = a.key_label != nullptr;
    const uint8_t* dp = has_drop ? a.drop + pbase + jc : reinterpret_cast<const uint8_t*>(kr);
    const size_t dst_ = has_drop ? (size_t)Lk : 0;
    uint8_t kraw[LQ];
#pragma unroll
    for (int i = 0; i < LQ; ++i) kraw[i] = dp[(size_t)i * dst_];
    const uint8_t kpm_raw = *(has_kpm ? a.kpm + (size_t)b * Lk + jc : reinterpret_cast<const uint8_t*>(kr));
    const int64_t lab_raw = *(has_lab ? a.key_label + (size_t)b * Lk + jc : reinterpret_cast<const int64_t*>(kr));
#pragma unroll
    for (int t = 0; t < (LQ * DH + 63) / 64; ++t) {
        const int e = lane + 64 * t;
        if (e < LQ * DH) qs[e] = qv[t];
    }
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) {
        vc[lane * (DH + 1) + 4 * t + 0] = v4[t].x; vc[lane * (DH + 1) + 4 * t + 1] = v4[t].y;
        vc[lane * (DH + 1) + 4 * t + 2] = v4[t].z; vc[lane * (DH + 1) + 4 * t + 3] = v4[t].w;
    }
    mha_unit_sync<WG>();
    float keep[LQ];
#pragma unroll
    for (int i = 0; i < LQ; ++i) keep[i] = has_drop ? a.drop_scale * (float)kraw[i] : 1.f;
    const bool masked = lane >= Lk || (has_kpm && kpm_raw != 0) || (has_lab && lab_raw == (int64_t)a.pad_idx);
    float p[LQ];
#pragma unroll
    for (int i = 0; i < LQ; ++i) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < DH / 4; ++t) {
            const float4 q4 = *reinterpret_cast<const float4*>(qs + i * DH + 4 * t);
            s += q4.x * k4[t].x + q4.y * k4[t].y + q4.z * k4[t].z + q4.w * k4[t].w;
        }
        s = masked ? -INFINITY : s * a.scale;
        const float m = wave_max(s);
        const float e = expf(s - m);                 // all keys masked: -inf - -inf = NaN, as in PyTorch
        const float sum = wave_sum(e);
        p[i] = e / sum;
        if (lane < Lk) a.probs[pbase + (size_t)i * Lk + lane] = p[i];
        sc[i * 64 + lane] = lane < Lk ? p[i] * keep[i] : 0.f;
    }
    mha_unit_sync<WG>();
#pragma unroll
    for (int t = 0; t < (LQ * DH + 63) / 64; ++t) {
        const int e = lane + 64 * t;
        if (e < LQ * DH) {
            const int i = e / DH, d = e % DH;
            float s = 0.f;
            for (int r = 0; r < Lk; ++r) s += sc[i * 64 + r] * vc[r * (DH + 1) + d];
            a.o[((size_t)b * LQ + i) * a.ldo + h * DH + d] = s;
        }
    }
}

constexpr int mha_small_bwd_lds_floats(int DH, int LQ) { return 2 * LQ * DH + LQ * 64 + 64 * (DH + 1); }

template <int DH, int LQ, bool WG>
__device__ __forceinline__ void mha_bwd_small_unit(const MhaArgs& a, const int unit, float* lds) {
    float* qs = lds;                              // [LQ][DH], 16-byte aligned
    float* dos = lds + LQ * DH;                   // [LQ][DH], 16-byte aligned
    float* ds = dos + LQ * DH;                    // [LQ][64]
    float* kc = ds + LQ * 64;                     // [64][DH + 1]
    const int b = unit / a.heads, h = unit % a.heads;
    const int lane = threadIdx.x & 63, Lk = a.Lk;
    const int jc = lane < Lk ? lane : Lk - 1;
    const float* qb = a.q + (size_t)b * LQ * a.ldq + h * DH;
    const float* dob = a.d_o + (size_t)b * LQ * a.lddo + h * DH;
    const float4* kr = reinterpret_cast<const float4*>(a.k + ((size_t)b * Lk + jc) * a.ldk + h * DH);
    const float4* vr = reinterpret_cast<const float4*>(a.v + ((size_t)b * Lk + jc) * a.ldv + h * DH);
    const size_t pbase = ((size_t)(b * a.heads + h) * LQ) * Lk;
    float4 k4[DH / 4], v4[DH / 4];
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) { k4[t] = kr[t]; v4[t] = vr[t]; }
    float qv[(LQ * DH + 63) / 64], dv_[(LQ * DH + 63) / 64];
#pragma unroll
    for (int t = 0; t < (LQ * DH + 63) / 64; ++t) {
        const int e = lane + 64 * t, ec = e < LQ * DH ? e : LQ * DH - 1;
        qv[t] = qb[(size_t)(ec / DH) * a.ldq + (ec % DH)];
        dv_[t] = dob[(size_t)(ec / DH) * a.lddo + (ec % DH)];
    }
    const bool has_drop = a.drop != nullptr;
    const uint8_t* dp = has_drop ? a.drop + pbase + jc : reinterpret_cast<const uint8_t*>(kr);
    const size_t dst_ = has_drop ? (size_t)Lk : 0;
    float P[LQ];
    uint8_t kraw[LQ];
#pragma unroll
    for (int i = 0; i < LQ; ++i) {
        P[i] = a.probs[pbase + (size_t)i * Lk + jc];
        kraw[i] = dp[(size_t)i * dst_];
    }
#pragma unroll
    for (int t = 0; t < (LQ * DH + 63) / 64; ++t) {
        const int e = lane + 64 * t;
        if (e < LQ * DH) { qs[e] = qv[t]; dos[e] = dv_[t]; }
    }
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) {
        kc[lane * (DH + 1) + 4 * t + 0] = k4[t].x; kc[lane * (DH + 1) + 4 * t + 1] = k4[t].y;
        kc[lane * (DH + 1) + 4 * t + 2] = k4[t].z; kc[lane * (DH + 1) + 4 * t + 3] = k4[t].w;
    }
    mha_unit_sync<WG>();
    float keep[LQ];
#pragma unroll
    for (int i = 0; i < LQ; ++i) keep[i] = has_drop ? a.drop_scale * (float)kraw[i] : 1.f;
    // dP = (dO . V^T) o keep ; softmax backward with the 1/sqrt(dh) folded in ; Pd = dropped probabilities
    float dS[LQ], Pd[LQ];
    const bool live = lane < Lk;
#pragma unroll
    for (int i = 0; i < LQ; ++i) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < DH / 4; ++t) {
            const float4 g4 = *reinterpret_cast<const float4*>(dos + i * DH + 4 * t);
            s += g4.x * v4[t].x + g4.y * v4[t].y + g4.z * v4[t].z + g4.w * v4[t].w;
        }
        const float dp = s * keep[i];
        const float pi = live ? P[i] : 0.f;
        const float tsum = wave_sum(pi * dp);
        dS[i] = live ? pi * (dp - tsum) * a.scale : 0.f;
        Pd[i] = pi * keep[i];
        ds[i * 64 + lane] = dS[i];
    }
    mha_unit_sync<WG>();
    // dq = dS . K  (contraction over the keys, through LDS)
#pragma unroll
    for (int t = 0; t < (LQ * DH + 63) / 64; ++t) {
        const int e = lane + 64 * t;
        if (e < LQ * DH) {
            const int i = e / DH, d = e % DH;
            float s = 0.f;
            for (int r = 0; r < Lk; ++r) s += ds[i * 64 + r] * kc[r * (DH + 1) + d];
            a.dq[((size_t)b * LQ + i) * a.lddq + h * DH + d] = s;
        }
    }
    // dk_j = sum_i dS[i][j] q_i ; dv_j = sum_i Pd[i][j] dO_i  (lane j, rows in registers, q / dO broadcast from LDS)
    if (live) {
        float4* dkr = reinterpret_cast<float4*>(a.dk + ((size_t)b * Lk + lane) * a.lddk + h * DH);
        float4* dvr = reinterpret_cast<float4*>(a.dv + ((size_t)b * Lk + lane) * a.lddv + h * DH);
#pragma unroll
        for (int t = 0; t < DH / 4; ++t) {
            float4 gk = make_float4(0.f, 0.f, 0.f, 0.f), gv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < LQ; ++i) {
                const float4 q4 = *reinterpret_cast<const float4*>(qs + i * DH + 4 * t);
                const float4 g4 = *reinterpret_cast<const float4*>(dos + i * DH + 4 * t);
                gk.x += dS[i] * q4.x; gk.y += dS[i] * q4.y; gk.z += dS[i] * q4.z; gk.w += dS[i] * q4.w;
                gv.x += Pd[i] * g4.x; gv.y += Pd[i] * g4.y; gv.z += Pd[i] * g4.z; gv.w += Pd[i] * g4.w;
            }
            dkr[t] = gk;
            dvr[t] = gv;
        }
    }
}

}  // namespace r3d
