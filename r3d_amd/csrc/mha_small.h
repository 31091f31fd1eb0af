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
    float keep[LQ];
#pragma unroll
    for (int i = 0; i < LQ; ++i) keep[i] = a.drop ? a.drop_scale * (float)a.drop[pbase + (size_t)i * Lk + jc] : 1.f;
    bool masked = lane >= Lk;
    if (a.kpm) masked = masked || a.kpm[(size_t)b * Lk + jc] != 0;
    if (a.key_label) masked = masked || a.key_label[(size_t)b * Lk + jc] == (int64_t)a.pad_idx;
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

}  // namespace r3d
