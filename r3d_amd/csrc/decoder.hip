// Fused DETR-style decoder layer for the token-fusion path, ONE workgroup per clip.
//
// The decoder works on n_query (8) rows per clip; as separate launches one layer is ~17 dependent kernels forward
// and ~20 backward whose ~4.5 us launch floor each dwarfs their work (TransformerDecoderLayer.forward_post,
// model/extras/transformer.py:281-330, called from TransformerDecoder.forward :161-191).  Clips are independent, so
// a workgroup keeps one clip's whole layer in LDS and only streams the (L2-resident, shared) weights:
//   forward : [tgt+query_pos] -> self-attn in_proj -> 8x8 attention -> out_proj -> +res, LN1
//             -> q proj of (t1+query_pos); k,v proj of (memory+pos) -> masked cross attention -> out_proj -> +res, LN2
//             -> FFN(relu) -> +res, LN3  [-> final decoder LayerNorm -> fc|fc_len head]
//   backward: the adjoint chain of input gradients; every dY a weight gradient needs is written out so that ALL
//             weight/bias gradients of the step still run as one grouped GEMM launch (gemm_f32.hip).
// GEMMs: v_mfma_f32_16x16x4_f32 (exact fp32) on a 16-row block (the query rows, zero padded); the A operand comes
// from LDS, the B operand (weights) straight from global memory -- with 16 rows there is no reuse to stage for.
// The k order inside a 16-deep group is permuted identically on both operands so one 16-byte read feeds 4 MFMAs.
#include "common.h"
#include "../../include/r3d_hip.h"
#include <stdlib.h>

namespace r3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DEC_THREADS = 1024;     // 16 waves: the layer is a chain of latency-bound phases -- the weights stream
                                      // from L2 per k-step, so what matters is loads in flight, not FLOPs per wave
constexpr float kLnEpsD = 1e-5f;

// ---- Y[16 x N] = X[16 x K] . W[N x K]^T : epi(row, col, value) for every element of the 16 x N result ------------
// The phase is latency-bound (weights come from L2 per k-step), so the work is spread to keep every wave busy with many
// independent loads in flight:
//   * N/16 >= 2 * waves : each wave takes column tiles in PAIRS -- both tiles' weight loads are issued before the first
//     MFMA and the two accumulator chains interleave;
//   * N/16 * 2 <= waves  : two waves share a tile and split K; the upper half's partial sums go through `red` (LDS,
//     [waves/2][4][64] floats) -- needs the workgroup barrier inside, so EVERY wave must call this function.
template <class Epi>
__device__ __forceinline__ void mm16_nt(const float* X, int ldx, int K, const float* __restrict__ W, int ldw, int N,
                                        float* red, Epi epi) {
    constexpr int NW = DEC_THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int ntiles = (N + 15) >> 4;
    if (ntiles * 2 <= NW) {
        // ---- k-split: wave = (tile, half)
        const int tile = wave >> 1, half = wave & 1;
        const bool active = tile < ntiles;
        const int n = tile * 16 + i;
        const bool nok = active && n < N;
        const int kh = K >> 1;                                  // K is a multiple of 32 on this path (H, 4H)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        __syncthreads();                  // the lower halves of a PREVIOUS k-split call may still be reading `red`
        if (active) {
            const float* wrow = W + (size_t)(nok ? n : 0) * ldw + half * kh + 4 * kq;
            const float* xrow = X + i * ldx + half * kh + 4 * kq;
#pragma unroll 8
            for (int k0 = 0; k0 < kh; k0 += 16) {
                const float4 a = *reinterpret_cast<const float4*>(xrow + k0);
                float4 b = *reinterpret_cast<const float4*>(wrow + k0);
                if (!nok) b = make_float4(0.f, 0.f, 0.f, 0.f);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
            }
            if (half == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(tile * 4 + r) * 64 + lane] = acc[r];
            }
        }
        __syncthreads();
        if (active && half == 0 && nok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(kq * 4 + r, n, acc[r] + red[(tile * 4 + r) * 64 + lane]);
        }
        return;
    }
    for (int tile = wave; tile < ntiles; tile += 2 * NW) {
        const int tile2 = tile + NW;
        const bool has2 = tile2 < ntiles;
        const int n1 = tile * 16 + i, n2 = tile2 * 16 + i;
        const bool ok1 = n1 < N, ok2 = has2 && n2 < N;
        const float* w1 = W + (size_t)(ok1 ? n1 : 0) * ldw + 4 * kq;
        const float* w2 = W + (size_t)(ok2 ? n2 : 0) * ldw + 4 * kq;
        const float* xrow = X + i * ldx + 4 * kq;
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k0 = 0; k0 < K; k0 += 16) {
            const float4 a = *reinterpret_cast<const float4*>(xrow + k0);
            float4 b1 = *reinterpret_cast<const float4*>(w1 + k0);
            float4 b2 = *reinterpret_cast<const float4*>(w2 + k0);
            if (!ok1) b1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!ok2) b2 = make_float4(0.f, 0.f, 0.f, 0.f);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b2.x, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b2.y, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2.z, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b2.w, acc2, 0, 0, 0);
        }
        // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
        if (ok1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(kq * 4 + r, n1, acc1[r]);
        }
        if (ok2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(kq * 4 + r, n2, acc2[r]);
        }
    }
}

// ---- dX[16 x K] = dY[16 x N] . W[N x K] : contraction over the ROWS of W --------------------------------------------
template <class Epi>
__device__ __forceinline__ void mm16_nn(const float* dY, int ldy, int N, const float* __restrict__ W, int ldw, int K,
                                        Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int ktiles = (K + 15) >> 4;
    for (int tile = wave; tile < ktiles; tile += DEC_THREADS / 64) {
        const int kc = tile * 16 + i;                      // output column (an input feature of the linear layer)
        const bool kok = kc < K;
        const float* yrow = dY + i * ldy + 4 * kq;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int n0 = 0; n0 < N; n0 += 16) {
            const float4 a = *reinterpret_cast<const float4*>(yrow + n0);
            const float* wp = W + (size_t)(n0 + 4 * kq) * ldw + (kok ? kc : 0);
            float b0 = wp[0], b1 = wp[ldw], b2 = wp[2 * (size_t)ldw], b3 = wp[3 * (size_t)ldw];
            if (!kok) { b0 = 0.f; b1 = 0.f; b2 = 0.f; b3 = 0.f; }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc, 0, 0, 0);
        }
        if (kok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(kq * 4 + r, kc, acc[r]);
        }
    }
}

// LayerNorm over H of `rows` rows held in LDS (x, ld); one wave per row.  y may alias x.
template <class Out>
__device__ __forceinline__ void ln_rows(const float* x, int ld, int rows, int H, const float* g, const float* b,
                                        float* mean_out, float* rstd_out, Out out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < rows; r += DEC_THREADS / 64) {
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += x[r * ld + c];
        const float mean = wave_sum(s) / (float)H;
        float q = 0.f;
        for (int c = lane; c < H; c += 64) { const float d = x[r * ld + c] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + kLnEpsD);
        if (lane == 0 && mean_out) { mean_out[r] = mean; rstd_out[r] = rstd; }
        for (int c = lane; c < H; c += 64) out(r, c, (x[r * ld + c] - mean) * rstd * g[c] + b[c]);
    }
}

struct DecLayerParams {
    const float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b, *n1_g, *n1_b;
    const float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b, *n2_g, *n2_b;
    const float *l1_w, *l1_b, *l2_w, *l2_b, *n3_g, *n3_b;
};

struct DecFwdArgs {
    DecLayerParams p;
    const float* fused;          // [B*S, H] memory
    const float* pos;            // [S, H]
    const float* qpos;           // [Q, H]
    const float* tgt_in;         // [B*Q, H] or NULL (layer 0: tgt = 0)
    const int64_t* key_label; int pad_idx;     // key j of clip b padded iff key_label[b*S + j] == pad_idx (or NULL)
    const uint8_t *drop_sa_p, *drop_ca_p, *drop_d1, *drop_d2, *drop_d3, *drop_ff; float drop_scale;
    // saved activations (global)
    float *sa_qkv, *p_sa, *sa_o, *t1_pre, *t1, *m1, *r1, *caq, *cakv, *p_ca, *ca_o, *t2_pre, *t2, *m2, *r2, *ff1, *t3_pre,
        *t3, *m3, *r3;
    // optional tail of the LAST layer: decoder.norm + [fc ; fc_len] head
    const float *fin_g, *fin_b; float *tgtF, *mF, *rF;
    const float *head_w, *head_b; float* actdur; int n_head_out;
    int B, S, Q, H, heads;
};

__device__ __forceinline__ float keepf(const uint8_t* m, size_t idx, float scale) { return m ? scale * (float)m[idx] : 1.f; }

// LDS carve (floats).  LD = H + 4.
//   XA  [16][LD]      current 16-row block (rows >= Q are zero)      XB [16][LD]   second row block
//   WIDE[16][4H + 4]  qkv / ff1                                      KV [S16][2H+4] keys|values   KIN [S16][LD]
//   PR  [heads][Q][max(Q,S)]
__global__ __launch_bounds__(DEC_THREADS) void decoder_layer_fwd_kernel(const DecFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int S = a.S, Q = a.Q, H = a.H, heads = a.heads, dh = H / heads;
    const int LD = H + 4, LDW = 4 * H + 4, LDK = 2 * H + 4;
    const int S16 = (S + 15) & ~15;
    const int PL = (S > Q ? S : Q);
    float* XA = lds;
    float* XB = XA + 16 * LD;
    float* WIDE = XB + 16 * LD;
    float* KV = WIDE + 16 * LDW;
    float* KIN = KV + S16 * LDK;
    float* PR = KIN + S16 * LD;
    float* RED = PR + heads * Q * PL;                      // [waves/2][4][64] k-split partial sums
    float* KMASK = RED + (DEC_THREADS / 128) * 4 * 64;     // [S16] 1 = padded key
    const float dsc = a.drop_scale;
    const size_t rowQ = (size_t)b * Q;                     // first query row of this clip in [B*Q, .] tensors
    const size_t rowS = (size_t)b * S;

    // ---- XA = tgt + query_pos (zero padded to 16 rows);  XB = tgt (residual)
    for (int e = tid; e < 16 * H; e += DEC_THREADS) {
        const int r = e / H, c = e - r * H;
        float t = 0.f, qp = 0.f;
        if (r < Q) {
            qp = a.qpos[r * H + c];
            if (a.tgt_in) t = a.tgt_in[(rowQ + r) * H + c];
        }
        XA[r * LD + c] = t + qp;
        XB[r * LD + c] = t;
    }
    // KIN = memory + pos (zero padded)
    for (int e = tid; e < S16 * H; e += DEC_THREADS) {
        const int r = e / H, c = e - r * H;
        KIN[r * LD + c] = (r < S) ? a.fused[(rowS + r) * H + c] + a.pos[r * H + c] : 0.f;
    }
    for (int j = tid; j < S; j += DEC_THREADS)
        KMASK[j] = (a.key_label && a.key_label[rowS + j] == (int64_t)a.pad_idx) ? 1.f : 0.f;
    __syncthreads();
    // ---- self attention in_proj: WIDE[16][3H] = XA . Win^T + b
    mm16_nt(XA, LD, H, a.p.sa_in_w, H, 3 * H, RED, [&](int r, int c, float v) {
        v += a.p.sa_in_b[c];
        WIDE[r * LDW + c] = v;
        if (r < Q) a.sa_qkv[(rowQ + r) * 3 * H + c] = v;
    });
    __syncthreads();
    // ---- 8x8 attention per head: PR[h][i][j]
    const float scale = 1.0f / sqrtf((float)dh);
    for (int e = tid; e < heads * Q * Q; e += DEC_THREADS) {
        const int h = e / (Q * Q), i = (e / Q) % Q, j = e % Q;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += WIDE[i * LDW + h * dh + d] * WIDE[j * LDW + H + h * dh + d];
        PR[(h * Q + i) * PL + j] = s * scale;
    }
    __syncthreads();
    for (int e = tid; e < heads * Q; e += DEC_THREADS) {
        float* row = PR + e * PL;
        float m = -INFINITY;
        for (int j = 0; j < Q; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < Q; ++j) { const float ex = expf(row[j] - m); row[j] = ex; sum += ex; }
        const size_t pb = ((size_t)b * heads * Q + e) * Q;
        for (int j = 0; j < Q; ++j) {
            const float pj = row[j] / sum;
            a.p_sa[pb + j] = pj;
            row[j] = pj * keepf(a.drop_sa_p, pb + j, dsc);
        }
    }
    __syncthreads();
    // sa_o -> XA (rows >= Q stay zero)
    for (int e = tid; e < Q * H; e += DEC_THREADS) {
        const int i = e / H, c = e - i * H, h = c / dh;
        float s = 0.f;
        for (int j = 0; j < Q; ++j) s += PR[(h * Q + i) * PL + j] * WIDE[j * LDW + 2 * H + c];
        XA[i * LD + c] = s;
        a.sa_o[(rowQ + i) * H + c] = s;
    }
    __syncthreads();
    // ---- out_proj + dropout1 + residual -> t1_pre (in XB)
    mm16_nt(XA, LD, H, a.p.sa_out_w, H, H, RED, [&](int r, int c, float v) {
        if (r < Q) {
            v = (v + a.p.sa_out_b[c]) * keepf(a.drop_d1, (rowQ + r) * H + c, dsc) + XB[r * LD + c];
            XB[r * LD + c] = v;
            a.t1_pre[(rowQ + r) * H + c] = v;
        }
    });
    __syncthreads();
    // ---- LN1 -> t1 in XB (residual for the next sub-layer), XA = t1 + query_pos
    ln_rows(XB, LD, Q, H, a.p.n1_g, a.p.n1_b, a.m1 + rowQ, a.r1 + rowQ, [&](int r, int c, float v) {
        XB[r * LD + c] = v;
        XA[r * LD + c] = v + a.qpos[r * H + c];
        a.t1[(rowQ + r) * H + c] = v;
    });
    __syncthreads();
    // ---- cross attention: q = XA . Wq^T + bq -> WIDE[:, 0:H];  k|v rows = KIN . Wkv^T + bkv -> KV
    mm16_nt(XA, LD, H, a.p.ca_in_w, H, H, RED, [&](int r, int c, float v) {
        v += a.p.ca_in_b[c];
        WIDE[r * LDW + c] = v;
        if (r < Q) a.caq[(rowQ + r) * H + c] = v;
    });
    for (int rt = 0; rt < S16; rt += 16) {
        mm16_nt(KIN + rt * LD, LD, H, a.p.ca_in_w + (size_t)H * H, H, 2 * H, RED, [&](int r, int c, float v) {
            v += a.p.ca_in_b[H + c];
            KV[(rt + r) * LDK + c] = v;
            if (rt + r < S) a.cakv[(rowS + rt + r) * 2 * H + c] = v;
        });
    }
    __syncthreads();
    for (int e = tid; e < heads * Q * S; e += DEC_THREADS) {
        const int h = e / (Q * S), i = (e / S) % Q, j = e % S;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += WIDE[i * LDW + h * dh + d] * KV[j * LDK + h * dh + d];
        PR[(h * Q + i) * PL + j] = (KMASK[j] != 0.f) ? -INFINITY : s * scale;
    }
    __syncthreads();
    for (int e = tid; e < heads * Q; e += DEC_THREADS) {
        float* row = PR + e * PL;
        float m = -INFINITY;
        for (int j = 0; j < S; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < S; ++j) { const float ex = expf(row[j] - m); row[j] = ex; sum += ex; }
        const size_t pb = ((size_t)b * heads * Q + e) * S;
        for (int j = 0; j < S; ++j) {
            const float pj = row[j] / sum;                 // all keys masked -> NaN, as in PyTorch
            a.p_ca[pb + j] = pj;
            row[j] = pj * keepf(a.drop_ca_p, pb + j, dsc);
        }
    }
    __syncthreads();
    for (int e = tid; e < Q * H; e += DEC_THREADS) {
        const int i = e / H, c = e - i * H, h = c / dh;
        float s = 0.f;
        for (int j = 0; j < S; ++j) s += PR[(h * Q + i) * PL + j] * KV[j * LDK + H + c];
        XA[i * LD + c] = s;
        a.ca_o[(rowQ + i) * H + c] = s;
    }
    __syncthreads();
    mm16_nt(XA, LD, H, a.p.ca_out_w, H, H, RED, [&](int r, int c, float v) {
        if (r < Q) {
            v = (v + a.p.ca_out_b[c]) * keepf(a.drop_d2, (rowQ + r) * H + c, dsc) + XB[r * LD + c];
            XB[r * LD + c] = v;
            a.t2_pre[(rowQ + r) * H + c] = v;
        }
    });
    __syncthreads();
    ln_rows(XB, LD, Q, H, a.p.n2_g, a.p.n2_b, a.m2 + rowQ, a.r2 + rowQ, [&](int r, int c, float v) {
        XB[r * LD + c] = v;
        XA[r * LD + c] = v;
        a.t2[(rowQ + r) * H + c] = v;
    });
    __syncthreads();
    // ---- FFN: ff1 = drop(relu(t2 . W1^T + b1)) -> WIDE[16][4H];  t3_pre = t2 + drop3(ff1 . W2^T + b2)
    mm16_nt(XA, LD, H, a.p.l1_w, H, 4 * H, RED, [&](int r, int c, float v) {
        v = fmaxf(v + a.p.l1_b[c], 0.f);
        if (r < Q) {
            v *= keepf(a.drop_ff, (rowQ + r) * 4 * H + c, dsc);
            a.ff1[(rowQ + r) * 4 * H + c] = v;
        } else {
            v = 0.f;
        }
        WIDE[r * LDW + c] = v;
    });
    __syncthreads();
    mm16_nt(WIDE, LDW, 4 * H, a.p.l2_w, 4 * H, H, RED, [&](int r, int c, float v) {
        if (r < Q) {
            v = (v + a.p.l2_b[c]) * keepf(a.drop_d3, (rowQ + r) * H + c, dsc) + XB[r * LD + c];
            XB[r * LD + c] = v;
            a.t3_pre[(rowQ + r) * H + c] = v;
        }
    });
    __syncthreads();
    ln_rows(XB, LD, Q, H, a.p.n3_g, a.p.n3_b, a.m3 + rowQ, a.r3 + rowQ, [&](int r, int c, float v) {
        XB[r * LD + c] = v;
        a.t3[(rowQ + r) * H + c] = v;
    });
    if (!a.fin_g) return;
    __syncthreads();
    // ---- decoder.norm (transformer.py:182-183) and the [fc ; fc_len] head (futr_safuser_tokenfusion.py:219-226)
    ln_rows(XB, LD, Q, H, a.fin_g, a.fin_b, a.mF + rowQ, a.rF + rowQ, [&](int r, int c, float v) {
        XA[r * LD + c] = v;
        a.tgtF[(rowQ + r) * H + c] = v;
    });
    __syncthreads();
    mm16_nt(XA, LD, H, a.head_w, H, a.n_head_out, RED, [&](int r, int c, float v) {
        if (r < Q) a.actdur[(rowQ + r) * a.n_head_out + c] = v + a.head_b[c];
    });
}

static size_t dec_fwd_lds_bytes(int S, int Q, int H, int heads) {
    const int LD = H + 4, LDW = 4 * H + 4, LDK = 2 * H + 4, S16 = (S + 15) & ~15, PL = S > Q ? S : Q;
    return sizeof(float) * ((size_t)2 * 16 * LD + 16 * LDW + (size_t)S16 * LDK + (size_t)S16 * LD + (size_t)heads * Q * PL +
                            (size_t)(DEC_THREADS / 128) * 4 * 64 + S16);
}

}  // namespace r3d

using namespace r3d;

/* 1 when the fused per-clip decoder kernels support this shape (everything of one clip's layer fits one CU's LDS). */
R3D_EXPORT int r3d_decoder_fused_supported(int S, int Q, int H, int heads) {
    if (S <= 0 || Q <= 0 || Q > 16 || H <= 0 || heads <= 0 || (H % 32) != 0 || (H % heads) != 0) return 0;
    return dec_fwd_lds_bytes(S, Q, H, heads) <= 150 * 1024 ? 1 : 0;
}

/* One fused decoder layer forward for all clips.  `ptrs` is an array of device pointers in the order of
 * r3d_decoder_fwd_ptrs (include/r3d_hip.h); optional ones may be NULL. */
R3D_EXPORT int r3d_decoder_layer_fwd(const void* const* ptrs, int nptrs, int B, int S, int Q, int H, int heads, int pad_idx,
                                     float drop_scale, int n_head_out, void* stream) {
    R3D_REQUIRE(ptrs && nptrs == R3D_DEC_FWD_NPTRS);
    R3D_REQUIRE(r3d_decoder_fused_supported(S, Q, H, heads) && B > 0);
    DecFwdArgs a{};
    int k = 0;
    auto F = [&]() { return (const float*)ptrs[k++]; };
    auto M = [&]() { return (float*)const_cast<void*>(ptrs[k++]); };
    auto U = [&]() { return (const uint8_t*)ptrs[k++]; };
    a.p.sa_in_w = F(); a.p.sa_in_b = F(); a.p.sa_out_w = F(); a.p.sa_out_b = F(); a.p.n1_g = F(); a.p.n1_b = F();
    a.p.ca_in_w = F(); a.p.ca_in_b = F(); a.p.ca_out_w = F(); a.p.ca_out_b = F(); a.p.n2_g = F(); a.p.n2_b = F();
    a.p.l1_w = F(); a.p.l1_b = F(); a.p.l2_w = F(); a.p.l2_b = F(); a.p.n3_g = F(); a.p.n3_b = F();
    a.fused = F(); a.pos = F(); a.qpos = F(); a.tgt_in = F();
    a.key_label = (const int64_t*)ptrs[k++];
    a.drop_sa_p = U(); a.drop_ca_p = U(); a.drop_d1 = U(); a.drop_d2 = U(); a.drop_d3 = U(); a.drop_ff = U();
    a.sa_qkv = M(); a.p_sa = M(); a.sa_o = M(); a.t1_pre = M(); a.t1 = M(); a.m1 = M(); a.r1 = M(); a.caq = M(); a.cakv = M();
    a.p_ca = M(); a.ca_o = M(); a.t2_pre = M(); a.t2 = M(); a.m2 = M(); a.r2 = M(); a.ff1 = M(); a.t3_pre = M(); a.t3 = M();
    a.m3 = M(); a.r3 = M();
    a.fin_g = F(); a.fin_b = F(); a.tgtF = M(); a.mF = M(); a.rF = M(); a.head_w = F(); a.head_b = F(); a.actdur = M();
    if (k != R3D_DEC_FWD_NPTRS) return R3D_EINVAL;
    for (int i = 0; i < 21; ++i) R3D_REQUIRE(ptrs[i] != nullptr);                    // parameters, memory, pos, qpos
    for (int i = 29; i < 49; ++i) R3D_REQUIRE(ptrs[i] != nullptr);                   // saved activations
    if (a.fin_g) R3D_REQUIRE(a.fin_b && a.tgtF && a.mF && a.rF && a.head_w && a.head_b && a.actdur && n_head_out > 0);
    a.pad_idx = pad_idx; a.drop_scale = drop_scale; a.n_head_out = n_head_out;
    a.B = B; a.S = S; a.Q = Q; a.H = H; a.heads = heads;
    const size_t lds = dec_fwd_lds_bytes(S, Q, H, heads);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)decoder_layer_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(decoder_layer_fwd_kernel, dim3(B), dim3(DEC_THREADS), lds, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
