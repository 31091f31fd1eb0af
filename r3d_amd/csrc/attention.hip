// Multi-head attention core of the DETR-style decoder (nn.MultiheadAttention inside
// TransformerDecoderLayer.forward_post, model/extras/transformer.py:289-304):
//   scores = (q . k^T) * dh^-0.5, key_padding_mask -> -inf, softmax over keys, dropout(p) on the probabilities, P . v
// The decoder has only n_query (8) query rows per clip, so the core is tiny and latency-bound: one wave per
// (clip, head); the key chunk is staged in LDS ([64][dh+1], conflict-free lane-per-key dot products), the score
// matrix [Lq][Lk] lives in LDS, reductions are wave shuffles.  The in/out projections run on the MFMA GEMM.
#include "common.h"
#include "mha_small.h"
#include "../../include/r3d_hip.h"

namespace r3d {


// Coalesced copy of rows [row0, row0+nrows) x dh of a strided matrix into the padded LDS chunk [64][dh+1].
// All loads of the flat loop are independent, so the wave keeps many in flight (the tensors are tiny and the kernel
// is pure latency: nothing in the inner loops below touches global memory).
__device__ __forceinline__ void load_chunk(float* kc, const float* base, int ld, int row0, int nrows, int dh, int lane) {
    const int total = nrows * dh;
    for (int e = lane; e < total; e += 64) {
        const int r = e / dh, d = e - r * dh;
        kc[r * (dh + 1) + d] = base[(size_t)(row0 + r) * ld + d];
    }
}

template <int MAXE>
__global__ __launch_bounds__(64) void mha_fwd_kernel(const MhaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int lane = threadIdx.x;
    const int Lq = a.Lq, Lk = a.Lk, dh = a.dh;
    float* qs = lds;                       // [Lq][dh]
    float* sc = qs + Lq * dh;              // [Lq][Lk]
    float* kc = sc + Lq * Lk;              // [64][dh+1]  K chunk
    float* vc = kc + 64 * (dh + 1);        // [64][dh+1]  V chunk
    float* keep = vc + 64 * (dh + 1);      // [Lq][Lk]    dropout keep * 1/(1-p)  (1 when dropout is off)
    const float* qb = a.q + (size_t)b * Lq * a.ldq + h * dh;
    const float* kb = a.k + (size_t)b * Lk * a.ldk + h * dh;
    const float* vb = a.v + (size_t)b * Lk * a.ldv + h * dh;
    const size_t pbase = ((size_t)(b * a.heads + h) * Lq) * Lk;
    const bool single = Lk <= 64;          // one chunk: q, K and V are all fetched in ONE memory round trip
    for (int e = lane; e < Lq * dh; e += 64) {
        const int i = e / dh, d = e - i * dh;
        qs[e] = qb[(size_t)i * a.ldq + d];
    }
    for (int e = lane; e < Lq * Lk; e += 64) keep[e] = a.drop ? a.drop_scale * (float)a.drop[pbase + e] : 1.f;
    if (single) {
        load_chunk(kc, kb, a.ldk, 0, Lk, dh, lane);
        load_chunk(vc, vb, a.ldv, 0, Lk, dh, lane);
    }
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        const int nrows = min(64, Lk - c0);
        __syncthreads();
        if (!single) load_chunk(kc, kb, a.ldk, c0, nrows, dh, lane);
        __syncthreads();
        const int j = c0 + lane;
        if (j < Lk) {
            const bool masked = (a.kpm && a.kpm[(size_t)b * Lk + j]) ||
                                (a.key_label && a.key_label[(size_t)b * Lk + j] == (int64_t)a.pad_idx);
            for (int i = 0; i < Lq; ++i) {
                float s = 0.f;
                for (int d = 0; d < dh; ++d) s += qs[i * dh + d] * kc[lane * (dh + 1) + d];
                sc[i * Lk + j] = masked ? -INFINITY : s * a.scale;
            }
        }
    }
    __syncthreads();
    for (int i = 0; i < Lq; ++i) {
        float m = -INFINITY;
        for (int j = lane; j < Lk; j += 64) m = fmaxf(m, sc[i * Lk + j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < Lk; j += 64) {
            const float e = expf(sc[i * Lk + j] - m);       // all keys masked: -inf - -inf = NaN, as in PyTorch
            sc[i * Lk + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int j = lane; j < Lk; j += 64) {
            const float p = sc[i * Lk + j] / sum;
            a.probs[pbase + (size_t)i * Lk + j] = p;
            sc[i * Lk + j] = p * keep[i * Lk + j];
        }
    }
    // O = P_dropped . V : output element e = i*dh + d lives in lane e % 64, slot e / 64
    float acc[MAXE];
#pragma unroll
    for (int t = 0; t < MAXE; ++t) acc[t] = 0.f;
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        const int nrows = min(64, Lk - c0);
        __syncthreads();
        if (!single) load_chunk(vc, vb, a.ldv, c0, nrows, dh, lane);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MAXE; ++t) {
            const int e = lane + 64 * t;
            if (e < Lq * dh) {
                const int i = e / dh, d = e - i * dh;
                float s = 0.f;
                for (int r = 0; r < nrows; ++r) s += sc[i * Lk + c0 + r] * vc[r * (dh + 1) + d];
                acc[t] += s;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MAXE; ++t) {
        const int e = lane + 64 * t;
        if (e < Lq * dh) {
            const int i = e / dh, d = e - i * dh;
            a.o[((size_t)b * Lq + i) * a.ldo + h * dh + d] = acc[t];
        }
    }
}

template <int MAXE>
__global__ __launch_bounds__(64) void mha_bwd_kernel(const MhaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int lane = threadIdx.x;
    const int Lq = a.Lq, Lk = a.Lk, dh = a.dh;
    float* qs = lds;                       // [Lq][dh]
    float* dos = qs + Lq * dh;             // [Lq][dh]
    float* P = dos + Lq * dh;              // [Lq][Lk]  softmax probs, later the dropped probs
    float* dS = P + Lq * Lk;               // [Lq][Lk]
    float* kc = dS + Lq * Lk;              // [64][dh+1]  K chunk
    float* vc = kc + 64 * (dh + 1);        // [64][dh+1]  V chunk
    float* keep = vc + 64 * (dh + 1);      // [Lq][Lk]    dropout keep * 1/(1-p)
    const float* qb = a.q + (size_t)b * Lq * a.ldq + h * dh;
    const float* kb = a.k + (size_t)b * Lk * a.ldk + h * dh;
    const float* vb = a.v + (size_t)b * Lk * a.ldv + h * dh;
    const float* dob = a.d_o + (size_t)b * Lq * a.lddo + h * dh;
    const size_t pbase = ((size_t)(b * a.heads + h) * Lq) * Lk;
    const bool single = Lk <= 64;          // one chunk: every operand is fetched in ONE memory round trip
    for (int e = lane; e < Lq * dh; e += 64) {
        const int i = e / dh, d = e - i * dh;
        qs[e] = qb[(size_t)i * a.ldq + d];
        dos[e] = dob[(size_t)i * a.lddo + d];
    }
    for (int e = lane; e < Lq * Lk; e += 64) {
        P[e] = a.probs[pbase + e];
        keep[e] = a.drop ? a.drop_scale * (float)a.drop[pbase + e] : 1.f;
    }
    if (single) {
        load_chunk(vc, vb, a.ldv, 0, Lk, dh, lane);
        load_chunk(kc, kb, a.ldk, 0, Lk, dh, lane);
    }
    // dP = (dO . V^T) o keep*scale
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        const int nrows = min(64, Lk - c0);
        __syncthreads();
        if (!single) load_chunk(vc, vb, a.ldv, c0, nrows, dh, lane);
        __syncthreads();
        const int j = c0 + lane;
        if (j < Lk)
            for (int i = 0; i < Lq; ++i) {
                float s = 0.f;
                for (int d = 0; d < dh; ++d) s += dos[i * dh + d] * vc[lane * (dh + 1) + d];
                dS[i * Lk + j] = s * keep[i * Lk + j];
            }
    }
    __syncthreads();
    // softmax backward, with the 1/sqrt(dh) of the scores folded in; then P <- dropped probs for dV
    for (int i = 0; i < Lq; ++i) {
        float t = 0.f;
        for (int j = lane; j < Lk; j += 64) t += P[i * Lk + j] * dS[i * Lk + j];
        t = wave_sum(t);
        for (int j = lane; j < Lk; j += 64) {
            const float p = P[i * Lk + j];
            dS[i * Lk + j] = p * (dS[i * Lk + j] - t) * a.scale;
            P[i * Lk + j] = p * keep[i * Lk + j];
        }
    }
    // dq = dS . K ; dk = dS^T . q ; dv = P_dropped^T . dO   -- K chunk by chunk through LDS
    float acc[MAXE];
#pragma unroll
    for (int t = 0; t < MAXE; ++t) acc[t] = 0.f;
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        const int nrows = min(64, Lk - c0);
        __syncthreads();
        if (!single) load_chunk(kc, kb, a.ldk, c0, nrows, dh, lane);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MAXE; ++t) {
            const int e = lane + 64 * t;
            if (e < Lq * dh) {
                const int i = e / dh, d = e - i * dh;
                float s = 0.f;
                for (int r = 0; r < nrows; ++r) s += dS[i * Lk + c0 + r] * kc[r * (dh + 1) + d];
                acc[t] += s;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MAXE; ++t) {
        const int e = lane + 64 * t;
        if (e < Lq * dh) {
            const int i = e / dh, d = e - i * dh;
            a.dq[((size_t)b * Lq + i) * a.lddq + h * dh + d] = acc[t];
        }
    }
    for (int e = lane; e < Lk * dh; e += 64) {
        const int j = e / dh, d = e - j * dh;
        float gk = 0.f, gv = 0.f;
        for (int i = 0; i < Lq; ++i) {
            gk += dS[i * Lk + j] * qs[i * dh + d];
            gv += P[i * Lk + j] * dos[i * dh + d];
        }
        a.dk[((size_t)b * Lk + j) * a.lddk + h * dh + d] = gk;
        a.dv[((size_t)b * Lk + j) * a.lddv + h * dh + d] = gv;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small-shape kernels (Lk <= 64 keys, compile-time head width DH and query count LQ): lane j owns key j.  K and V rows
// stay in REGISTERS (one 16-byte load per 4 features, all issued up front together with q, the masks and, backward,
// dO and the probabilities: one memory round trip), q / dO are read from LDS as broadcasts, the score / softmax /
// dropout chain is per-lane arithmetic plus wave reductions over the keys, and only the two contractions over the key
// index (P.V forward, dS.K backward) go through LDS.  Everything is unrolled at compile time.
// ---------------------------------------------------------------------------------------------------------------
template <int DH, int LQ>
__global__ __launch_bounds__(64) void mha_fwd_small_kernel(const MhaArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[mha_small_lds_floats(DH, LQ)];
    mha_fwd_small_unit<DH, LQ, true>(a, blockIdx.x, lds);        // (mha_small.h)
}

template <int DH, int LQ>
__global__ __launch_bounds__(64) void mha_bwd_small_kernel(const MhaArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[mha_small_bwd_lds_floats(DH, LQ)];
    mha_bwd_small_unit<DH, LQ, true>(a, blockIdx.x, lds);        // (mha_small.h)
}

// small path: compile-time (dh, Lq) instance exists, Lk <= 64 and 16-byte aligned rows
static bool mha_small_ok(const MhaArgs& a, bool bwd) {
    if (a.Lk > 64 || a.Lq != 8 || !(a.dh == 16 || a.dh == 32 || a.dh == 64 || a.dh == 128)) return false;
    if ((a.ldk | a.ldv) & 3) return false;
    if (!r3d_aligned16(a.k) || !r3d_aligned16(a.v)) return false;
    if (bwd && (((a.lddk | a.lddv) & 3) || !r3d_aligned16(a.dk) || !r3d_aligned16(a.dv))) return false;
    return true;
}

template <bool BWD>
static void mha_small_launch(const MhaArgs& a, hipStream_t s) {
    const dim3 g(a.B * a.heads), blk(64);
    if (BWD) {
        if (a.dh == 16) hipLaunchKernelGGL((mha_bwd_small_kernel<16, 8>), g, blk, 0, s, a);
        else if (a.dh == 32) hipLaunchKernelGGL((mha_bwd_small_kernel<32, 8>), g, blk, 0, s, a);
        else if (a.dh == 64) hipLaunchKernelGGL((mha_bwd_small_kernel<64, 8>), g, blk, 0, s, a);
        else hipLaunchKernelGGL((mha_bwd_small_kernel<128, 8>), g, blk, 0, s, a);   // hidden 1024: one wave may use 512 registers
    } else {
        if (a.dh == 16) hipLaunchKernelGGL((mha_fwd_small_kernel<16, 8>), g, blk, 0, s, a);
        else if (a.dh == 32) hipLaunchKernelGGL((mha_fwd_small_kernel<32, 8>), g, blk, 0, s, a);
        else if (a.dh == 64) hipLaunchKernelGGL((mha_fwd_small_kernel<64, 8>), g, blk, 0, s, a);
        else hipLaunchKernelGGL((mha_fwd_small_kernel<128, 8>), g, blk, 0, s, a);
    }
}

static size_t mha_lds_bytes(int Lq, int Lk, int dh, bool bwd) {
    size_t f = (size_t)Lq * dh + (size_t)2 * Lq * Lk + (size_t)2 * 64 * (dh + 1);
    if (bwd) f += (size_t)Lq * dh + (size_t)Lq * Lk;
    return f * sizeof(float);
}

static int mha_check(const MhaArgs& a, bool bwd) {
    if (!a.q || !a.k || !a.v || !a.probs) return R3D_EINVAL;
    if (a.B <= 0 || a.heads <= 0 || a.Lq <= 0 || a.Lk <= 0 || a.dh <= 0) return R3D_EINVAL;
    const int H = a.heads * a.dh;
    if (a.ldq < H || a.ldk < H || a.ldv < H) return R3D_EINVAL;
    if (!bwd && (!a.o || a.ldo < H)) return R3D_EINVAL;
    if (bwd && (!a.d_o || !a.dq || !a.dk || !a.dv || a.lddo < H || a.lddq < H || a.lddk < H || a.lddv < H))
        return R3D_EINVAL;
    if (mha_lds_bytes(a.Lq, a.Lk, a.dh, bwd) > 160 * 1024) return R3D_EINVAL;
    if ((long)a.Lq * a.dh > 1024) return R3D_EINVAL;               // 16 output slots per lane
    return R3D_OK;
}

}  // namespace r3d

using namespace r3d;

R3D_EXPORT int r3d_mha_core_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const uint8_t* key_padding_mask, const int64_t* key_label, int pad_idx, float* probs,
                                const uint8_t* drop_mask,
                                float drop_scale, float* o, int ldo, int B, int heads, int Lq, int Lk, int dh,
                                void* stream) {
    MhaArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kpm = key_padding_mask; a.probs = probs;
    a.key_label = key_label; a.pad_idx = pad_idx;
    a.drop = drop_mask; a.drop_scale = drop_scale; a.o = o; a.ldo = ldo;
    a.B = B; a.heads = heads; a.Lq = Lq; a.Lk = Lk; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
    int rc = mha_check(a, false);
    if (rc != R3D_OK) return rc;
    if (mha_small_ok(a, false)) {
        mha_small_launch<false>(a, (hipStream_t)stream);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    const size_t lds = mha_lds_bytes(Lq, Lk, dh, false);
    const bool small = (long)Lq * dh <= 128;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(small ? (const void*)mha_fwd_kernel<2> : (const void*)mha_fwd_kernel<16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (small) hipLaunchKernelGGL(mha_fwd_kernel<2>, dim3(B * heads), dim3(64), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mha_fwd_kernel<16>, dim3(B * heads), dim3(64), lds, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_mha_core_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const float* probs, const uint8_t* drop_mask, float drop_scale, const float* d_o,
                                int lddo, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, int B, int heads,
                                int Lq, int Lk, int dh, void* stream) {
    MhaArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.probs = const_cast<float*>(probs);
    a.drop = drop_mask; a.drop_scale = drop_scale; a.d_o = d_o; a.lddo = lddo;
    a.dq = dq; a.lddq = lddq; a.dk = dk; a.lddk = lddk; a.dv = dv; a.lddv = lddv;
    a.B = B; a.heads = heads; a.Lq = Lq; a.Lk = Lk; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
    int rc = mha_check(a, true);
    if (rc != R3D_OK) return rc;
    if (mha_small_ok(a, true)) {
        mha_small_launch<true>(a, (hipStream_t)stream);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    const size_t lds = mha_lds_bytes(Lq, Lk, dh, true);
    const bool small = (long)Lq * dh <= 128;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(small ? (const void*)mha_bwd_kernel<2> : (const void*)mha_bwd_kernel<16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (small) hipLaunchKernelGGL(mha_bwd_kernel<2>, dim3(B * heads), dim3(64), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mha_bwd_kernel<16>, dim3(B * heads), dim3(64), lds, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
