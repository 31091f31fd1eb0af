// fp32-exact MFMA GEMM family for gfx950 (MI355X).  See include/r3d_hip.h "GEMM family" for the contract.
//
// Why f32 MFMA: the depth projection contracts over K = 50176 (model/futr_safuser_tokenfusion.py:143,195);
// bf16 inputs cannot hold the 1e-3 parity budget there, and gfx950 has no TF32/xf32.  v_mfma_f32_32x32x2_f32 is an
// exact fp32 fma chain at the f32 vector peak (157 TFLOP/s) with 1 operand VGPR per lane per 4096 FLOP.
//
// Tile scheme (per workgroup):  BM x BN output, BK k per step, waves laid out WM x WN, each wave owning
// (BM/WM) x (BN/WN) as 32x32 MFMA tiles, operands staged through LDS:
//   - a K-contiguous global operand (nn.Linear weights / activations in NT) keeps its layout in LDS ([rows][BK+4]);
//     one ds_read_b128 then feeds FOUR MFMAs (the k order inside a step is permuted identically on both operands);
//   - an M/N-contiguous operand (NN's B, TN's A and B) is staged k-major ([BK][cols+4]) and read with ds_read_b32.
//   Both are copied with float4 global loads + ds_write_b128: no transposing scalar writes, no LDS bank conflicts.
// Software pipeline: global loads run TWO k-steps ahead of the MFMAs in two named register stages (the step is
// latency-bound: a 64-deep step is ~1 us of MFMA against ~2 us of HBM latency), LDS is double-buffered, one
// barrier per step.  Out-of-range rows / columns / k are zero-filled, so any M, N, K works; the float4 path needs
// 16-byte aligned bases and leading dimensions that are multiples of 4 (d.vec, decided on the host), otherwise
// element loads are used.
// The epilogue runs straight from the accumulators in per-operand stages (16 independent loads in flight each).
// gemm_grouped_kernel runs many independent problems of one layout/tile in ONE launch (all weight gradients of the
// step): the per-kernel floor of ~5 us dominates these latency-bound problems, not their FLOPs.
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------
// element-wise epilogue (split-K reducer)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gemm_epilogue(const r3d_gemm_desc& d, int m, int n, float acc) {
    m ^= d.c_row_xor;
    float v = d.alpha * acc;
    if (d.bias) v += d.bias[n];
    if (d.pre_out) d.pre_out[(size_t)m * d.ldpre + n] = v;
    if (d.act == 1) v = fmaxf(v, 0.0f);
    else if (d.act == 2) v = gelu_f(v);
    if (d.drop_mask) v *= d.drop_scale * (float)d.drop_mask[(size_t)m * d.lddrop + n];
    if (d.mul == 1) v = (d.aux[(size_t)m * d.ldaux + n] > 0.0f) ? v : 0.0f;
    else if (d.mul == 2) v *= gelu_grad_f(d.aux[(size_t)m * d.ldaux + n]);
    if (d.res1) v += d.res1[(size_t)m * d.ldr1 + n];
    if (d.res2) v += d.res2[(size_t)m * d.ldr2 + n];
    float* c = d.C + (size_t)m * d.ldc + n;
    if (d.accumulate) v += *c;
    *c = v;
}

// Staged epilogue of one 32x32 accumulator tile held by a wave.  C/D map of the 32x32 MFMA: col = lane & 31 (fixed
// per lane), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); mb already contains the lane's 4*(lane>>5) offset.
// Operands of the epilogue for NR output elements of one column n (rows given by the caller).  load() issues EVERY
// operand load unconditionally from clamped (always valid) rows before any is used: one memory round trip instead of one
// per operand -- and the small-tile kernel calls it BEFORE its k-loop, so the (cold: written by the previous kernel,
// possibly through another XCD's L2) residual / mask / aux reads travel under the main loop.
template <int NR>
struct EpiOps {
    float kk[NR], xa[NR], x1[NR], x2[NR], xc[NR];
    float bias;
    __device__ __forceinline__ void load(const r3d_gemm_desc& d, const size_t* ml, int n) {
        bias = d.bias ? d.bias[n] : 0.f;
        if (d.drop_mask) {
#pragma unroll
            for (int r = 0; r < NR; ++r) kk[r] = (float)d.drop_mask[ml[r] * d.lddrop + n];
        }
        if (d.mul) {
#pragma unroll
            for (int r = 0; r < NR; ++r) xa[r] = d.aux[ml[r] * d.ldaux + n];
        }
        if (d.res1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) x1[r] = d.res1[ml[r] * d.ldr1 + n];
        }
        if (d.res2) {
#pragma unroll
            for (int r = 0; r < NR; ++r) x2[r] = d.res2[ml[r] * d.ldr2 + n];
        }
        if (d.accumulate) {
#pragma unroll
            for (int r = 0; r < NR; ++r) xc[r] = d.C[ml[r] * d.ldc + n];
        }
    }
    // arithmetic in the reference's order, then the store; m[r] = destination row, ok[r] = row inside the matrix
    __device__ __forceinline__ void apply(const r3d_gemm_desc& d, const float* acc, const int* m, const bool* ok, int n) {
        float v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = d.alpha * acc[r] + bias;
        if (d.pre_out) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (ok[r]) d.pre_out[(size_t)m[r] * d.ldpre + n] = v[r];
        }
        if (d.act == 1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] = fmaxf(v[r], 0.0f);
        } else if (d.act == 2) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] = gelu_f(v[r]);
        }
        if (d.drop_mask) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] *= d.drop_scale * kk[r];
        }
        if (d.mul == 1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] = (xa[r] > 0.0f) ? v[r] : 0.0f;
        } else if (d.mul) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] *= gelu_grad_f(xa[r]);
        }
        if (d.res1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] += x1[r];
        }
        if (d.res2) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] += x2[r];
        }
        if (d.accumulate) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] += xc[r];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (ok[r]) d.C[(size_t)m[r] * d.ldc + n] = v[r];
    }
};

__device__ __forceinline__ void gemm_epilogue_tile(const r3d_gemm_desc& d, const f32x16& acc, int mb, int n, int split) {
    if (n >= d.N) return;
    int m[16];
    size_t ml[16];       // row used for operand LOADS: clamped into the matrix so that every load is unconditional
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = mb + (r & 3) + 8 * (r >> 2);
        ok[r] = mr < d.M;
        m[r] = mr ^ d.c_row_xor;
        ml[r] = (size_t)((ok[r] ? mr : d.M - 1) ^ d.c_row_xor);
    }
    if (d.splitk > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) d.partial[((size_t)split * d.M + (m[r] ^ d.c_row_xor)) * d.N + n] = acc[r];
        return;
    }
    // two halves of 8 rows: the operand arrays of a 16-row pass pushed the 128x128 tiles over 256 VGPRs (spills)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        EpiOps<8> e;
        e.load(d, ml + 8 * h, n);
        float a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = acc[8 * h + r];
        e.apply(d, a, m + 8 * h, ok + 8 * h, n);
    }
}

// ---------------------------------------------------------------------------------------------------------
// staging: one register stage of a tile (two of them are alive in the pipeline)
// ---------------------------------------------------------------------------------------------------------
// VEC: every load is an UNCONDITIONAL 16-byte load from a clamped (always valid) address; the element is zeroed (and
// the broadcast addend added) when the piece is written to LDS one step later.  A load under a lane-dependent branch
// makes hipcc wait vmcnt(0) right behind it, which serialises the memory round trips of the tile (seen in the ISA).
// Pieces (one float4 per thread each) can be loaded / stored one at a time so the main loop can slot them between
// MFMA groups.
// K-contiguous source: tile of R rows x BK k.  f indexes float4s: row = f/(BK/4), kq = f%(BK/4).
template <int R, int BK, int NT, bool VEC>
struct StageKC {
    static constexpr int Q4 = BK / 4;
    static constexpr int NLD = (R * Q4) / NT;
    static_assert(NLD >= 1 && (R * Q4) % NT == 0, "tile too small for the workgroup");
    float4 v[NLD], y[NLD];
    bool ok[NLD];
    __device__ __forceinline__ void load_one(int p, const float* __restrict__ base, int ld, int row0, int rows_total,
                                             int k0, int k_end, int k_total, int row_xor, const float* add, int add_mod,
                                             int add_ld) {
        const int f = threadIdx.x + p * NT;
        const int row = row0 + f / Q4;
        const int k = k0 + ((f % Q4) << 2);
        if (VEC) {
            const int vrow = row < rows_total ? row : 0;
            const int sk = k < k_total - 4 ? k : k_total - 4;
            v[p] = *reinterpret_cast<const float4*>(base + (size_t)(vrow ^ row_xor) * ld + sk);
            if (add) y[p] = *reinterpret_cast<const float4*>(add + (size_t)(vrow % add_mod) * add_ld + sk);
            else y[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            ok[p] = row < rows_total && k < k_end;           // K % 4 == 0: a float4 is all in or all out
        } else {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows_total && k < k_end) {
                const float* src = base + (size_t)(row ^ row_xor) * ld + k;
                const float* ad = add ? add + (size_t)(row % add_mod) * add_ld + k : nullptr;
                x.x = src[0] + (ad ? ad[0] : 0.f);
                if (k + 1 < k_end) x.y = src[1] + (ad ? ad[1] : 0.f);
                if (k + 2 < k_end) x.z = src[2] + (ad ? ad[2] : 0.f);
                if (k + 3 < k_end) x.w = src[3] + (ad ? ad[3] : 0.f);
            }
            v[p] = x;
            y[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            ok[p] = true;
        }
    }
    // LDS image [R][BK + 4] (row-major like the source: no transpose on the way in, one ds_write_b128 per float4)
    __device__ __forceinline__ void store_one(int p, float* __restrict__ s) const {
        constexpr int S = BK + 4;
        const int f = threadIdx.x + p * NT;
        const int row = f / Q4, kq = (f % Q4) << 2;
        float4 x = make_float4(v[p].x + y[p].x, v[p].y + y[p].y, v[p].z + y[p].z, v[p].w + y[p].w);
        if (!ok[p]) x = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(s + row * S + kq) = x;
    }
};

// M/N-contiguous source: tile of BK k-rows x R columns.  f indexes float4s: krow = f/(R/4), cq = f%(R/4).
template <int R, int BK, int NT, bool VEC>
struct StageMC {
    static constexpr int NLD = (R * BK / 4) / NT;
    static_assert(NLD >= 1 && (R * BK / 4) % NT == 0, "tile too small for the workgroup");
    float4 v[NLD], y[NLD];
    bool ok[NLD];
    __device__ __forceinline__ void load_one(int p, const float* __restrict__ base, int ld, int col0, int cols_total,
                                             int k0, int k_end, int k_total, const float* add, int add_mod, int add_ld) {
        const int f = threadIdx.x + p * NT;
        const int k = k0 + f / (R / 4);
        const int c = col0 + ((f % (R / 4)) << 2);
        if (VEC) {
            const int sk = k < k_total ? k : k_total - 1;
            const int sc = c < cols_total - 4 ? c : cols_total - 4;
            v[p] = *reinterpret_cast<const float4*>(base + (size_t)sk * ld + sc);
            if (add) y[p] = *reinterpret_cast<const float4*>(add + (size_t)(sk % add_mod) * add_ld + sc);   // B'[k,:] += add[k%mod,:]
            else y[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            ok[p] = k < k_end && c < cols_total;             // cols % 4 == 0
        } else {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < k_end && c < cols_total) {
                const float* src = base + (size_t)k * ld + c;
                const float* ad = add ? add + (size_t)(k % add_mod) * add_ld + c : nullptr;
                x.x = src[0] + (ad ? ad[0] : 0.f);
                if (c + 1 < cols_total) x.y = src[1] + (ad ? ad[1] : 0.f);
                if (c + 2 < cols_total) x.z = src[2] + (ad ? ad[2] : 0.f);
                if (c + 3 < cols_total) x.w = src[3] + (ad ? ad[3] : 0.f);
            }
            v[p] = x;
            y[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            ok[p] = true;
        }
    }
    // LDS image [BK][R + 4]
    __device__ __forceinline__ void store_one(int p, float* __restrict__ s) const {
        constexpr int S = R + 4;
        const int f = threadIdx.x + p * NT;
        const int krow = f / (R / 4), cq = (f % (R / 4)) << 2;
        float4 x = make_float4(v[p].x + y[p].x, v[p].y + y[p].y, v[p].z + y[p].z, v[p].w + y[p].w);
        if (!ok[p]) x = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(s + krow * S + cq) = x;
    }
};

// Both operands of one k-step.  Piece index q runs over the A pieces (even slots) and B pieces (odd slots) so that a
// partial [q0, q1) touches both operands evenly.
template <int LA, int LB, int BM, int BN, int BK, int NT, bool VEC>
struct Stage {
    StageKC<BM, BK, NT, VEC> a_kc;
    StageMC<BM, BK, NT, VEC> a_mc;
    StageKC<BN, BK, NT, VEC> b_kc;
    StageMC<BN, BK, NT, VEC> b_mc;
    static constexpr int NA = (LA == 0) ? StageKC<BM, BK, NT, VEC>::NLD : StageMC<BM, BK, NT, VEC>::NLD;
    static constexpr int NB = (LB == 0) ? StageKC<BN, BK, NT, VEC>::NLD : StageMC<BN, BK, NT, VEC>::NLD;
    static constexpr int NP = NA + NB;
    __device__ __forceinline__ void load_piece(int q, const r3d_gemm_desc& d, int m0, int n0, int k0, int k_end) {
        if (q < NA) {
            if (LA == 0) a_kc.load_one(q, d.A, d.lda, m0, d.M, k0, k_end, d.K, d.a_row_xor, d.a_add, d.a_add_mod, d.a_add_ld);
            else a_mc.load_one(q, d.A, d.lda, m0, d.M, k0, k_end, d.K, nullptr, 1, 0);
        } else {
            if (LB == 0) b_kc.load_one(q - NA, d.B, d.ldb, n0, d.N, k0, k_end, d.K, 0, nullptr, 1, 0);
            else b_mc.load_one(q - NA, d.B, d.ldb, n0, d.N, k0, k_end, d.K, d.b_add, d.b_add_mod, d.b_add_ld);
        }
    }
    __device__ __forceinline__ void store_piece(int q, float* as, float* bs) const {
        if (q < NA) { if (LA == 0) a_kc.store_one(q, as); else a_mc.store_one(q, as); }
        else { if (LB == 0) b_kc.store_one(q - NA, bs); else b_mc.store_one(q - NA, bs); }
    }
    __device__ __forceinline__ void load(const r3d_gemm_desc& d, int m0, int n0, int k0, int k_end) {
#pragma unroll
        for (int q = 0; q < NP; ++q) load_piece(q, d, m0, n0, k0, k_end);
    }
    __device__ __forceinline__ void store(float* as, float* bs) const {
#pragma unroll
        for (int q = 0; q < NP; ++q) store_piece(q, as, bs);
    }
};

// ---------------------------------------------------------------------------------------------------------
// one output tile of one problem
// ---------------------------------------------------------------------------------------------------------
template <int LA, int LB, int BM, int BN, int BK>
constexpr int gemm_lds_floats() {
    return 2 * ((LA == 0 ? BM * (BK + 4) : BK * (BM + 4)) + (LB == 0 ? BN * (BK + 4) : BK * (BN + 4)));
}

// WK > 1: WK groups of WM x WN waves share the output tile and split every k-step between them (intra-workgroup
// split-K, reduced through LDS at the end): two waves per SIMD cover each other's LDS / barrier / load-issue stalls.  The step's many tiny GEMMs have fewer tiles
// than the chip has CUs; what they need is more loads in flight and a shorter MFMA chain per wave, not more tiles.
template <int LA, int LB, int BM, int BN, int BK, int WM, int WN, int WK, bool VEC>
__device__ __forceinline__ void gemm_body(const r3d_gemm_desc& d, int tile, int split, float* smem) {
    constexpr int NT = 64 * WM * WN * WK;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // LDS images: K-contiguous operand -> [rows][BK+4] (fragment = one ds_read_b128 of 4 consecutive k, conflict-free:
    // row stride 68 floats puts the 16 lanes of a b128 group on 16 distinct 16-byte slots);
    // M/N-contiguous operand -> [BK][cols+4] (fragment = ds_read_b32 of 32 consecutive floats).
    constexpr int SA = (LA == 0) ? BK + 4 : BM + 4;
    constexpr int SB = (LB == 0) ? BK + 4 : BN + 4;
    constexpr int A_FLOATS = (LA == 0) ? BM * SA : BK * SA;
    constexpr int B_FLOATS = (LB == 0) ? BN * SB : BK * SB;
    static_assert(A_FLOATS % 4 == 0 && B_FLOATS % 4 == 0, "LDS carve must stay 16-byte aligned");
    float* As0 = smem;
    float* As1 = smem + A_FLOATS;
    float* Bs0 = smem + 2 * A_FLOATS;
    float* Bs1 = smem + 2 * A_FLOATS + B_FLOATS;

    const int tiles_n = (d.N + BN - 1) / BN;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int k_begin = (d.splitk > 1) ? split * d.k_per_split : 0;
    const int k_end = (d.splitk > 1) ? min(d.K, k_begin + d.k_per_split) : d.K;
    const int nk = (k_end - k_begin + BK - 1) / BK;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_mn = wave % (WM * WN), wk = wave / (WM * WN);     // WK groups of WM x WN waves split every k-step
    const int wm_off = (wave_mn / WN) * (BM / WM);
    const int wn_off = (wave_mn % WN) * (BN / WN);
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int G_PER_WAVE = BK / 8 / WK;             // groups of 8 k (4 MFMAs) per wave and k-step
    static_assert(G_PER_WAVE >= 1, "BK too small for the k-split");
    const int g0 = wk * G_PER_WAVE;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float asum[TM];                    // TN only: column sums of A = the bias gradient (d.bias_grad)
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;

    // One group = 8 consecutive k.  MFMA t of a group (t = 0..3) contracts k = 8g + 4*(lane>>5) + t on BOTH operands:
    // the k order inside a step is permuted (any order is a valid dot product) so that a K-contiguous operand feeds
    // four MFMAs from one 16-byte LDS read.
    auto group = [&](const float* as, const float* bs, int gq) {
        const int kb = (g0 + gq) * 8 + 4 * lhi;
        float a[TM][4], b[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (LA == 0) {
                const float4 t4 = *reinterpret_cast<const float4*>(as + (wm_off + i * 32 + l31) * SA + kb);
                a[i][0] = t4.x; a[i][1] = t4.y; a[i][2] = t4.z; a[i][3] = t4.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a[i][t] = as[(kb + t) * SA + wm_off + i * 32 + l31];
                    asum[i] += a[i][t];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (LB == 0) {
                const float4 t4 = *reinterpret_cast<const float4*>(bs + (wn_off + j * 32 + l31) * SB + kb);
                b[j][0] = t4.x; b[j][1] = t4.y; b[j][2] = t4.z; b[j][3] = t4.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[j][t] = bs[(kb + t) * SB + wn_off + j * 32 + l31];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    };

    // One k-step: issue the global loads of tile kt+2 (into `ld_st`), run the MFMA groups of the current LDS buffer,
    // then write tile kt+1 (from `st_st`) into the other buffer.  (Slotting the pieces BETWEEN the MFMA groups was
    // measured 1.7x slower: hipcc then puts a vmcnt wait in front of every group.)
    using StageT = Stage<LA, LB, BM, BN, BK, NT, VEC>;
    auto step = [&](const float* as, const float* bs, StageT& ld_st, bool do_load, int k_next, StageT& st_st, bool do_store,
                    float* as_o, float* bs_o) {
        if (do_load) ld_st.load(d, m0, n0, k_next, k_end);
#pragma unroll
        for (int gq = 0; gq < G_PER_WAVE; ++gq) group(as, bs, gq);
        if (do_store) st_st.store(as_o, bs_o);
    };

    // Small-tile kernel (one 32x32 tile, the 4 waves split k): after the k-loop every wave finishes 4 of the 16 row
    // groups, so the epilogue (loads, erf of the GELU, stores) is spread over the 4 waves and its operand reads are
    // issued HERE, before the k-loop.
    constexpr bool QUAD = (WK == 4 && WM == 1 && WN == 1 && TM == 1 && TN == 1);
    EpiOps<4> qe;
    int qm[4];
    bool qok[4];
    const int qn = n0 + l31;
    if (QUAD) {
        size_t qml[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mr = m0 + 4 * lhi + q + 8 * wk;            // row of accumulator register r = 4 * wk + q
            qok[q] = mr < d.M;
            qm[q] = mr ^ d.c_row_xor;
            qml[q] = (size_t)((qok[q] ? mr : d.M - 1) ^ d.c_row_xor);
        }
        if (d.splitk <= 1) qe.load(d, qml, qn < d.N ? qn : d.N - 1);
    }

    StageT st0, st1;
    if (nk > 0) st0.load(d, m0, n0, k_begin, k_end);
    if (nk > 1) st1.load(d, m0, n0, k_begin + BK, k_end);
    if (nk > 0) st0.store(As0, Bs0);
    __syncthreads();
    for (int kt = 0; kt < nk;) {
        // even step: tile kt is in buffer 0, tile kt+1 in register stage 1; stage 0 is free for tile kt+2
        step(As0, Bs0, st0, kt + 2 < nk, k_begin + (kt + 2) * BK, st1, kt + 1 < nk, As1, Bs1);
        __syncthreads();
        if (++kt >= nk) break;
        // odd step: roles swapped
        step(As1, Bs1, st1, kt + 2 < nk, k_begin + (kt + 2) * BK, st0, kt + 1 < nk, As0, Bs0);
        __syncthreads();
        ++kt;
    }

    if (WK == 1 && d.adam_m) {
        // AdamW epilogue: the output tile (= the gradient) goes through LDS so that parameter and moments are streamed
        // row-contiguously with 16-byte accesses (the MFMA layout gives a lane one column: 4-byte accesses, six
        // arrays -- measured slower than the separate AdamW launch).  Same arithmetic and order as adamw_kernel.
        constexpr int SO = BN + 4;
        static_assert(BM * SO <= gemm_lds_floats<LA, LB, BM, BN, BK>(), "output tile does not fit the staging LDS");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[(wm_off + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * SO + wn_off + j * 32 + l31] = acc[i][j][r];
        __syncthreads();
        const float lr = *d.adam_lr;
        const double stepd = (double)*d.adam_step;
        const float bc1 = (float)(1.0 - pow((double)d.adam_beta1, stepd));
        const float bc2_sqrt = (float)sqrt(1.0 - pow((double)d.adam_beta2, stepd));
        const float decay = 1.0f - lr * d.adam_wd;
        const float step_size = lr / bc1;
        const float b2 = d.adam_beta2, eps = d.adam_eps, gs = d.alpha * d.adam_gscale;
        const float omb1 = 1.0f - d.adam_beta1, omb2 = 1.0f - b2;
        for (int f = threadIdx.x; f < BM * (BN / 4); f += NT) {
            const int row = f / (BN / 4), c4 = f % (BN / 4);
            const int gm = m0 + row, gn = n0 + 4 * c4;
            if (gm >= d.M || gn >= d.N) continue;
            const float4 gg = *reinterpret_cast<const float4*>(smem + row * SO + 4 * c4);
            const size_t o = (size_t)gm * d.ldc + gn;
            float4 pp = *reinterpret_cast<float4*>(d.C + o);
            float4 mm = *reinterpret_cast<float4*>(d.adam_m + o);
            float4 vv = *reinterpret_cast<float4*>(d.adam_v + o);
#define R3D_ADAM_E(c)                                              \
            {                                                      \
                const float gr = gg.c * gs;                        \
                pp.c *= decay;                                     \
                mm.c = mm.c + (gr - mm.c) * omb1;                  \
                vv.c = vv.c * b2 + gr * gr * omb2;                 \
                const float den = sqrtf(vv.c) / bc2_sqrt + eps;    \
                pp.c -= step_size * (mm.c / den);                  \
            }
            R3D_ADAM_E(x) R3D_ADAM_E(y) R3D_ADAM_E(z) R3D_ADAM_E(w)
#undef R3D_ADAM_E
            *reinterpret_cast<float4*>(d.C + o) = pp;
            *reinterpret_cast<float4*>(d.adam_m + o) = mm;
            *reinterpret_cast<float4*>(d.adam_v + o) = vv;
        }
        return;
    }
    if (QUAD) {
        constexpr int PER_WAVE = 17 * 64;
        static_assert(4 * PER_WAVE <= gemm_lds_floats<LA, LB, BM, BN, BK>(), "k-split reduction does not fit");
        float* mine = smem + wk * PER_WAVE;
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = acc[0][0][r];
        mine[16 * 64 + lane] = asum[0];
        __syncthreads();
        float v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wk + q;
            v4[q] = (smem[r * 64 + lane] + smem[PER_WAVE + r * 64 + lane]) +
                    (smem[2 * PER_WAVE + r * 64 + lane] + smem[3 * PER_WAVE + r * 64 + lane]);
        }
        if (LA == 1 && d.bias_grad && n0 == 0 && wk == 0) {
            float t = (smem[16 * 64 + lane] + smem[PER_WAVE + 16 * 64 + lane]) +
                      (smem[2 * PER_WAVE + 16 * 64 + lane] + smem[3 * PER_WAVE + 16 * 64 + lane]);
            t += __shfl_xor(t, 32, 64);
            const int m = m0 + l31;
            if (lhi == 0 && m < d.M) d.bias_grad[m] = t;
        }
        if (qn >= d.N) return;
        if (d.splitk > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (qok[q]) d.partial[((size_t)split * d.M + (qm[q] ^ d.c_row_xor)) * d.N + qn] = v4[q];
            return;
        }
        qe.apply(d, v4, qm, qok, qn);
        return;
    }
    if (WK > 1) {
        // reduce the WK partial accumulators (and bias-gradient sums) into the wk == 0 waves through LDS (the staging
        // buffers are free: the loop ended with a barrier)
        constexpr int PER_WAVE = (TM * TN * 16 + TM) * 64;
        static_assert((WK - 1) * WM * WN * PER_WAVE <= gemm_lds_floats<LA, LB, BM, BN, BK>(), "k-split reduction does not fit");
        float* red = smem + ((wk - 1) * (WM * WN) + wave_mn) * PER_WAVE;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
                red[(TM * TN * 16 + i) * 64 + lane] = asum[i];
            }
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w = 1; w < WK; ++w) {
            const float* rw = smem + ((w - 1) * (WM * WN) + wave_mn) * PER_WAVE;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += rw[((i * TN + j) * 16 + r) * 64 + lane];
                asum[i] += rw[(TM * TN * 16 + i) * 64 + lane];
            }
        }
    }
    if (LA == 1 && d.bias_grad && n0 == 0 && wn_off == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float t = asum[i] + __shfl_xor(asum[i], 32, 64);
            const int m = m0 + wm_off + i * 32 + l31;
            if (lhi == 0 && m < d.M) d.bias_grad[m] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            gemm_epilogue_tile(d, acc[i][j], m0 + wm_off + i * 32 + 4 * lhi, n0 + wn_off + j * 32 + l31, split);
}


template <int LA, int LB, int BM, int BN, int BK, int WM, int WN, int WK, bool VEC>
__global__ __launch_bounds__(64 * WM * WN * WK) void gemm_f32_kernel(const r3d_gemm_desc d, const int mode, const int G,
                                                                     const int NG) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_lds_floats<LA, LB, BM, BN, BK>()];
    // XCD-aware placement.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one L2), so workgroups
    // that stream the SAME operand panel must have equal linear id mod 8 or every one of them pulls its own copy over
    // the fabric (measured on the depth projection: FETCH_SIZE = 2.0x the algorithmic bytes with the 2-D grid).
    // A group = G workgroups sharing a panel; group g goes to XCD g % 8, its members sit at stride 8.
    //   mode 1: group = one K-split, members = its output tiles
    //   mode 2: group = one column of tiles (shares the B panel), mode 3: one row of tiles (shares the A panel)
    int tile = blockIdx.x, split = blockIdx.y;
    if (mode != 0) {
        const int p = blockIdx.x, idx = p >> 3;
        const int g = (idx / G) * 8 + (p & 7), mem = idx % G;
        if (g >= NG) return;
        const int tiles_n = (d.N + BN - 1) / BN;
        if (mode == 1) { tile = mem; split = g; }
        else if (mode == 2) { tile = mem * tiles_n + g; split = 0; }
        else { tile = g * tiles_n + mem; split = 0; }
    }
    gemm_body<LA, LB, BM, BN, BK, WM, WN, WK, VEC>(d, tile, split, smem);
}

// Many independent problems (same layout and tile config, splitk == 1) in one launch.  prefix[p] = first workgroup
// of problem p, prefix[n] = grid size; descriptors live in device memory (uploaded once per shape by the host).
// Groups of up to kGroupArgs problems carry the prefix table IN THE KERNEL ARGUMENTS (pf.v[p] = first workgroup of problem
// p, INT_MAX beyond the last): the problem of a workgroup is found with scalar compares.  Searching the table in device
// memory is a chain of log2(n) dependent scalar loads -- one L2 round trip each, in front of the descriptor load and the
// operand loads of every workgroup of every grouped launch (nine per step).
constexpr int kGroupArgs = 32;
struct GroupPrefix { int v[kGroupArgs]; };

template <int LA, int LB, int BM, int BN, int BK, int WM, int WN, int WK>
__global__ __launch_bounds__(64 * WM * WN * WK) void gemm_grouped_kernel(const r3d_gemm_desc* __restrict__ descs,
                                                                    const int* __restrict__ prefix, int n,
                                                                    const GroupPrefix pf, const int use_pf) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_lds_floats<LA, LB, BM, BN, BK>()];
    int lo = 0, base = 0;                    // largest p with prefix[p] <= blockIdx.x, and that prefix
    if (use_pf) {
#pragma unroll
        for (int p = 1; p < kGroupArgs; ++p)
            if (pf.v[p] <= (int)blockIdx.x) { lo = p; base = pf.v[p]; }
    } else {
        int hi = n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
        }
        base = prefix[lo];
    }
    const r3d_gemm_desc& d = descs[lo];
    if (d.vec) gemm_body<LA, LB, BM, BN, BK, WM, WN, WK, true>(d, (int)blockIdx.x - base, 0, smem);
    else gemm_body<LA, LB, BM, BN, BK, WM, WN, WK, false>(d, (int)blockIdx.x - base, 0, smem);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const r3d_gemm_desc d, int nsplit) {
    const size_t total = (size_t)d.M * d.N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int p = 0; p < nsplit; ++p) s += d.partial[(size_t)p * total + e];
        gemm_epilogue(d, (int)(e / d.N), (int)(e % d.N), s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of a WIDE layer from FEW rows: C[M, N] = A^T . B with A [K, M], B [K, N], K <= 128 rows (tokens),
// M <= 128 (hidden), N huge (depth_projection: 50176 pixels).  The tiled kernel gives each 64x64 output tile its own
// workgroup, whose whole life is two k-steps: prologue latency, not MFMA, sets its time (33 us, 32 % of peak).  Here a
// workgroup is PERSISTENT over 64-column panels of B:
//   * A^T (all of dY, 64 KB) is read once per workgroup; every wave keeps its 32-row slice of A for all K in REGISTERS
//     (64 VGPRs) -- the MFMA A-operand never touches LDS again;
//   * B panels [K][64] stream through a double-buffered LDS image (k-major, 32-lane-contiguous ds_read_b32), the loads
//     of panel p+1 are issued before the MFMAs of panel p;
//   * 8 waves = 4 (M) x 2 (N) wave tiles of 32x32: one panel is 64 MFMAs per wave = the CU's MFMA pipes for ~4 us;
//     784 panels = 196 workgroups x 4 panels, perfectly balanced (measured 26 us = 63 TFLOP/s; the 60 idle CUs and
//     the 3 us prologue are what is left).
// ---------------------------------------------------------------------------------------------------------
constexpr int kPanelN = 64;

__global__ __launch_bounds__(512, 1) void wgrad_panel_kernel(const r3d_gemm_desc d, const int npanels,
                                                             const int panels_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SB = kPanelN + 4;                       // [K][68]: a row of 64 floats + pad
    const int K = d.K, M = d.M, N = d.N;
    float* Bs0 = lds;
    float* Bs1 = lds + 128 * SB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 wave tiles of 32 x 32
    const int m_row = wm * 32 + l31;
    const bool m_ok = m_row < M;
    // ---- this wave's A fragments for every k: areg[s] = A[k = 2 s + lhi][m_row]  (coalesced 128-byte row segments)
    float areg[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int k = 2 * s + lhi;
        areg[s] = (k < K && m_ok) ? d.A[(size_t)k * d.lda + m_row] : 0.f;
    }
    const int p_begin = blockIdx.x * panels_per_wg;
    const int p_end = min(npanels, p_begin + panels_per_wg);
    // B panel loader: 128 rows x 16 float4 = 2048 float4 / 512 threads = 4 each
    float4 breg[4];
    auto load_panel = [&](int p) {
        const int n0 = p * kPanelN;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int f = tid + 512 * t, k = f >> 4, c4 = f & 15;
            const int kc = k < K ? k : K - 1;
            const int nc = n0 + 4 * c4 < N ? n0 + 4 * c4 : 0;    // N % 4 == 0 (validated): a float4 is all-in or all-out
            breg[t] = *reinterpret_cast<const float4*>(d.B + (size_t)kc * d.ldb + nc);
        }
    };
    auto store_panel = [&](float* bs) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int f = tid + 512 * t, k = f >> 4, c4 = f & 15;
            *reinterpret_cast<float4*>(bs + k * SB + 4 * c4) = (k < K) ? breg[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (p_begin >= p_end) return;
    load_panel(p_begin);
    store_panel(Bs0);
    __syncthreads();
    // The 16 stores of a finished panel are issued after the MFMAs of the NEXT one (fire-and-forget: nothing waits for
    // them before the kernel ends).
    f32x16 prev;
    int prev_n = -1;
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[r] = 0.f;
    for (int p = p_begin; p < p_end; ++p) {
        float* bs = ((p - p_begin) & 1) ? Bs1 : Bs0;
        float* bo = ((p - p_begin) & 1) ? Bs0 : Bs1;
        const bool more = p + 1 < p_end;
        if (more) load_panel(p + 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bcol = bs + wn * 32 + l31;
        const bool st = prev_n >= 0 && prev_n < N;
        // 64 MFMAs on one accumulator, B operands from LDS.  Left alone, hipcc issues every ds_read right in front of its
        // MFMA and waits lgkmcnt(0) (seen in the ISA), so the schedule is pinned: 16 operands in flight first, then
        // 8 MFMAs / 8 more operands alternately.
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const float b = bcol[(2 * s + lhi) * SB];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], b, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);          // 8 x ds_read2_b32 = 16 operands ahead
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // 8 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // 4 x ds_read2_b32 = 8 operands
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        if (more) store_panel(bo);           // before the C stores: on this ISA stores count in vmcnt like the B loads
        if (st) {
            // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < M) d.C[(size_t)m * d.ldc + prev_n] = d.alpha * prev[r];
            }
        }
        prev = acc;
        prev_n = p * kPanelN + wn * 32 + l31;
        __syncthreads();
    }
    if (prev_n >= 0 && prev_n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < M) d.C[(size_t)m * d.ldc + prev_n] = d.alpha * prev[r];
        }
    }
}

int launch_wgrad_panel_bf3(const r3d_gemm_desc& d, hipStream_t s);         // gemm_bf3.hip
int launch_gemm_bf3_nt(const r3d_gemm_desc& d, hipStream_t s);
int launch_gemm_bf3_nt_pair(const r3d_gemm_desc& d, const r3d_gemm_desc& e, hipStream_t s);
bool gemm_bf3_nt_ok(const r3d_gemm_desc& d);
int launch_gemm_bf3_tn(const r3d_gemm_desc& d, hipStream_t s);
bool gemm_bf3_tn_ok(const r3d_gemm_desc& d);

static bool wgrad_panel_ok(const r3d_gemm_desc& d) {
    if (d.layout != R3D_GEMM_TN || d.K > 128 || d.M > 128 || d.N < 2048 || (d.N & 3) || (d.ldb & 3)) return false;
    if (d.splitk > 1 || d.b_add || d.bias || d.pre_out || d.act || d.drop_mask || d.mul || d.res1 || d.res2) return false;
    if (d.accumulate || d.bias_grad || d.c_row_xor || d.adam_m) return false;
    return r3d_aligned16(d.B);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static const int kTileSz[11] = {0, 32, 64, 128, 64, 128, 64, 64, 64, 128, 128};

template <int LA, int LB, int BM, int BN, int BK, int WM, int WN, int WK>
static int launch_cfg(const r3d_gemm_desc& d, int nsplit, hipStream_t s) {
    const int tm = r3d_cdiv(d.M, BM), tn = r3d_cdiv(d.N, BN);
    dim3 grid(tm * tn, nsplit, 1);
    int mode = 0, G = 1, NG = 1;
    if (nsplit > 1 && tm * tn > 1 && tm * tn <= 32) { mode = 1; G = tm * tn; NG = nsplit; }
    else if (nsplit == 1 && tm >= 2 && tm <= 16 && tn >= 16) { mode = 2; G = tm; NG = tn; }
    else if (nsplit == 1 && tn >= 2 && tn <= 16 && tm >= 16) { mode = 3; G = tn; NG = tm; }
    if (mode) grid = dim3(8 * G * r3d_cdiv(NG, 8), 1, 1);
    if (d.vec) hipLaunchKernelGGL((gemm_f32_kernel<LA, LB, BM, BN, BK, WM, WN, WK, true>), grid, dim3(64 * WM * WN * WK), 0, s, d, mode, G, NG);
    else hipLaunchKernelGGL((gemm_f32_kernel<LA, LB, BM, BN, BK, WM, WN, WK, false>), grid, dim3(64 * WM * WN * WK), 0, s, d, mode, G, NG);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

template <int LA, int LB>
static int launch_layout(const r3d_gemm_desc& d, int nsplit, hipStream_t s) {
    switch (d.tile) {
        case 1: return launch_cfg<LA, LB, 32, 32, 64, 1, 1, 4>(d, nsplit, s);
        case 2: return launch_cfg<LA, LB, 64, 64, 64, 2, 2, 1>(d, nsplit, s);
        case 3: return launch_cfg<LA, LB, 128, 128, 32, 2, 2, 1>(d, nsplit, s);
        case 4: return launch_cfg<LA, LB, 64, 64, 64, 2, 2, 2>(d, nsplit, s);
        case 5: return launch_cfg<LA, LB, 128, 128, 32, 2, 2, 2>(d, nsplit, s);
        default: return R3D_EINVAL;
    }
}

template <int LA, int LB>
static int launch_grouped(const r3d_gemm_desc* descs, const int* prefix, const int32_t* host_prefix, int n, int total,
                          int tile, hipStream_t s) {
    GroupPrefix pf;
    const int use_pf = (host_prefix && n <= kGroupArgs) ? 1 : 0;
    for (int p = 0; p < kGroupArgs; ++p) pf.v[p] = (use_pf && p < n) ? host_prefix[p] : 0x7fffffff;
    switch (tile) {
        case 1:
            hipLaunchKernelGGL((gemm_grouped_kernel<LA, LB, 32, 32, 64, 1, 1, 4>), dim3(total), dim3(256), 0, s, descs, prefix, n,
                               pf, use_pf);
            break;
        case 2:
            hipLaunchKernelGGL((gemm_grouped_kernel<LA, LB, 64, 64, 64, 2, 2, 1>), dim3(total), dim3(256), 0, s, descs, prefix, n,
                               pf, use_pf);
            break;
        default: return R3D_EINVAL;
    }
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

static int nsplits_of(const r3d_gemm_desc& d) {
    return (d.splitk > 1) ? r3d_cdiv(d.K, d.k_per_split) : 1;
}

static bool gemm_can_vec(const r3d_gemm_desc& d) {
    bool vec = r3d_aligned16(d.A) && r3d_aligned16(d.B) && (d.lda % 4 == 0) && (d.ldb % 4 == 0);
    // the branch-free 16-byte path needs whole float4s along each operand's contiguous dimension
    const int a_contig = (d.layout == R3D_GEMM_TN) ? d.M : d.K;
    const int b_contig = (d.layout == R3D_GEMM_NT) ? d.K : d.N;
    vec = vec && (a_contig % 4 == 0) && (b_contig % 4 == 0) && a_contig >= 4 && b_contig >= 4;
    if (d.a_add) vec = vec && r3d_aligned16(d.a_add) && (d.a_add_ld % 4 == 0);
    if (d.b_add) vec = vec && r3d_aligned16(d.b_add) && (d.b_add_ld % 4 == 0);
    return vec;
}

static int gemm_validate(const r3d_gemm_desc* d) {
    if (!d || !d->A || !d->B || !d->C) return R3D_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return R3D_EINVAL;
    if (d->layout < 0 || d->layout > 2) return R3D_EINVAL;
    if (d->tile < 1 || d->tile > 12) return R3D_EINVAL;
    const int a_min = (d->layout == R3D_GEMM_TN) ? d->M : d->K;
    const int b_min = (d->layout == R3D_GEMM_NT) ? d->K : d->N;
    if (d->lda < a_min || d->ldb < b_min || d->ldc < d->N) return R3D_EINVAL;
    if (d->layout == R3D_GEMM_TN && (d->a_add || d->a_row_xor)) return R3D_EINVAL;
    if (d->a_add && (d->a_add_mod <= 0 || d->a_add_ld < d->K)) return R3D_EINVAL;
    if (d->b_add && (d->layout == R3D_GEMM_NT || d->b_add_mod <= 0 || d->b_add_ld < d->N)) return R3D_EINVAL;
    if ((d->a_row_xor || d->c_row_xor) && (d->M & 1)) return R3D_EINVAL;   // pair swap needs an even row count
    if (d->c_row_xor < 0 || d->c_row_xor > 1) return R3D_EINVAL;
    if (d->a_row_xor < 0 || d->a_row_xor > 1) return R3D_EINVAL;
    if (d->mul && !d->aux) return R3D_EINVAL;
    if (d->bias_grad && (d->layout != R3D_GEMM_TN || d->splitk > 1)) return R3D_EINVAL;
    if (d->adam_m) {                     // AdamW epilogue: nothing else may want the product
        if (!d->adam_v || !d->adam_lr || !d->adam_step || d->splitk > 1) return R3D_EINVAL;
        if (d->tile != 2 && d->tile != 3 && d->tile != 10 && d->tile != 12) return R3D_EINVAL;   // tiles without k-split waves; the bf16x3 TN tile
        if ((d->N & 3) || (d->ldc & 3) || d->c_row_xor || d->alpha == 0.f) return R3D_EINVAL;
        if (!r3d_aligned16(d->C) || !r3d_aligned16(d->adam_m) || !r3d_aligned16(d->adam_v)) return R3D_EALIGN;
        if (d->bias || d->pre_out || d->act || d->drop_mask || d->mul || d->res1 || d->res2 || d->accumulate) return R3D_EINVAL;
    }
    if (d->splitk > 1) {
        if (!d->partial || d->k_per_split <= 0 || (d->k_per_split % 16) != 0) return R3D_EINVAL;
        if (r3d_cdiv(d->K, d->k_per_split) > d->splitk) return R3D_EINVAL;
    }
    return R3D_OK;
}

}  // namespace r3d

using namespace r3d;

R3D_EXPORT int64_t r3d_gemm_partial_floats(int32_t M, int32_t N, int32_t splitk) {
    return (splitk > 1) ? (int64_t)splitk * M * N : 0;
}

// Cost model for a 256-CU / 1024-SIMD part.  These problems are latency-bound long before they are FLOP-bound:
// a workgroup pays ~2 us per 64-deep k-step until enough independent waves share a CU, plus ~5 us per extra launch
// (the split-K reducer).  Larger tiles cut the L2->LDS operand traffic (each element is re-fetched tiles_n or
// tiles_m times), more splits add slab traffic.
R3D_EXPORT int r3d_gemm_plan(r3d_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return R3D_EINVAL;
    // Constants fitted (tools/gemm_fit.py, log-space least squares, rms error 7 %, mean regret of the pick 1.6 %) to
    // 1893 hipGraph-timed launches of tools/gemm_sweep.py on MI355X: 81 shapes x 5 tiles x split counts.
    // Per unit of 64 k-elements a workgroup needs max(lat, load * thr) us, load = workgroups resident per CU.
    static const double lat[6] = {0, 0.525, 1.394, 4.768, 1.309, 4.224};
    static const double thr[6] = {0, 0.429, 1.387, 4.815, 1.291, 4.418};
    static const double epi[6] = {0, 3.563, 3.284, 5.613, 2.852, 5.837};     // prologue + epilogue of one round
    static const int occ[6] = {0, 8, 4, 1, 2, 1};
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 49, 61, 64, 96, 98, 122, 128, 192, 196, 256, 392, 512};
    double best = 1e300;
    int bt = 1, bs = 1, bk = d->K;
    for (int t = 1; t <= 5; ++t) {
        const long tiles = (long)r3d_cdiv(d->M, kTileSz[t]) * r3d_cdiv(d->N, kTileSz[t]);
        double lt = 1.0;
        // (TN: the sweep puts the 64x64 k-split tile ahead of both 128x128 tiles at every measured shape)
        if (d->layout == R3D_GEMM_TN) lt = (t == 5) ? 1.15 : (t == 3) ? 1.3 : 0.996;
        else if (d->layout == R3D_GEMM_NN) lt = 1.074;
        int last_ns = 0;
        for (int sk : cand) {
            const int kps = (sk == 1) ? d->K : r3d_cdiv(r3d_cdiv(d->K, sk), 64) * 64;
            if (sk > 1 && kps < 128) break;
            const int ns = r3d_cdiv(d->K, kps);
            if (ns == last_ns || (sk > 1 && ns < 2)) continue;
            last_ns = ns;
            const long wgs = tiles * ns;
            const long ncu = (wgs + 255) / 256;
            const double load = ncu <= 4 ? (double)ncu : (double)wgs / 256.0;
            const double unit = lat[t] > load * thr[t] * lt ? lat[t] : load * thr[t] * lt;
            const double rounds = (double)((ncu + occ[t] - 1) / occ[t]);
            double t_us = 1.118 + r3d_cdiv(kps, 64) * unit + rounds * epi[t];
            if (ns > 1) t_us += 5.0 + 2.0 * 4.0 * ns * (double)d->M * d->N / 4.0e6;     // slabs out and back + the reducer
            if (t_us < best) { best = t_us; bt = t; bs = ns; bk = kps; }
        }
    }
    d->tile = bt;
    d->splitk = bs;
    d->k_per_split = (bs > 1) ? bk : d->K;
    if (d->prec == 1 && d->layout == R3D_GEMM_NT && (d->K >= 8192 || (d->K >= 2048 && (long)d->M * d->N >= 256L * 256L)) && (d->K & 7) == 0 &&
        !d->a_add && !d->a_row_xor &&
        d->alpha == 1.0f && (d->lda & 3) == 0 && (d->ldb & 3) == 0 && r3d_aligned16(d->A) && r3d_aligned16(d->B)) {
        // long-K NT product on the bf16 matrix cores (gemm_bf3.hip): 64 x 64 tiles while they are few, 128 x 128 beyond;
        // as many K-splits as fill the chip once (one workgroup per CU), each a multiple of 64 deep
        const long t64 = (long)r3d_cdiv(d->M, 64) * r3d_cdiv(d->N, 64);
        const int tl = t64 <= 16 ? 8 : 9;
        const long tiles = tl == 8 ? t64 : (long)r3d_cdiv(d->M, 128) * r3d_cdiv(d->N, 128);
        int ns = (int)(256 / (tiles < 256 ? tiles : 256));
        if (ns < 2) ns = 2;
        int kps = r3d_cdiv(r3d_cdiv(d->K, ns), 64) * 64;
        if (kps < 256) kps = 256;
        ns = r3d_cdiv(d->K, kps);
        if (ns >= 2) { d->tile = tl; d->splitk = ns; d->k_per_split = kps; return R3D_OK; }
    }
    {                                    // few-row weight gradient of a wide layer: the persistent panel kernel
        r3d_gemm_desc t = *d;
        t.splitk = 1;
        // ... and beyond its 128 x 128 limit, the tiled kernel on the bf16 matrix cores (wide N, enough rows to amortise the split)
        if (d->prec == 1 && !wgrad_panel_ok(t) && d->N >= 2048 && d->K >= 64 && (long)d->M * d->K >= 128 * 128 && gemm_bf3_tn_ok(t)) {
            d->tile = 12; d->splitk = 1; d->k_per_split = d->K;     // (12: tile 10's kernel with one LDS stage, two workgroups per CU)
            return R3D_OK;
        }
        if (wgrad_panel_ok(t)) {
            const bool bf3 = d->prec == 1 && (d->ldc & 3) == 0 && r3d_aligned16(d->C);    // (its C tiles leave as float4)
            d->tile = bf3 ? 7 : 6; d->splitk = 1; d->k_per_split = d->K;
        }
    }
    return R3D_OK;
}

/* Two split-K NT products with the same M x N (both planned onto tile 8: prec == 1, splitk > 1, partial set, raw slabs left
 * for the caller's reducer) in ONE launch: the second product's K-splits take the workgroups after the first's. */
R3D_EXPORT int r3d_gemm_bf3_nt_pair(const r3d_gemm_desc* first, const r3d_gemm_desc* second, void* stream) {
    int rc = gemm_validate(first);
    if (rc != R3D_OK) return rc;
    rc = gemm_validate(second);
    if (rc != R3D_OK) return rc;
    return launch_gemm_bf3_nt_pair(*first, *second, (hipStream_t)stream);
}

R3D_EXPORT int r3d_gemm_f32(const r3d_gemm_desc* dp, void* stream) {
    int rc = gemm_validate(dp);
    if (rc != R3D_OK) return rc;
    r3d_gemm_desc d = *dp;
    d.vec = gemm_can_vec(d) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int ns = nsplits_of(d);
    if (d.tile == 8 || d.tile == 9 || d.tile == 11) return launch_gemm_bf3_nt(d, s);          // long-K NT split-K on the bf16 matrix cores
    if (d.tile == 10 || d.tile == 12) return launch_gemm_bf3_tn(d, s);                        // wide TN (weight gradient) on the same
    if (d.tile == 7) {                   // the same panel kernel on the bf16 matrix cores (exact 3-way operand split)
        if (!wgrad_panel_ok(d) || (d.ldc & 3) || !r3d_aligned16(d.C)) return R3D_EINVAL;
        return launch_wgrad_panel_bf3(d, s);
    }
    if (d.tile == 6) {                   // persistent panel kernel (few-row weight gradient of a wide layer)
        if (!wgrad_panel_ok(d)) return R3D_EINVAL;
        const int npanels = r3d_cdiv(d.N, kPanelN);
        const int per = r3d_cdiv(npanels, 256);                 // panels per workgroup: balanced over <= 256 workgroups
        const int wgs = r3d_cdiv(npanels, per);
        const size_t lds = (size_t)2 * 128 * (kPanelN + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)wgrad_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL(wgrad_panel_kernel, dim3(wgs), dim3(512), lds, s, d, npanels, per);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    switch (d.layout) {
        case R3D_GEMM_NT: return launch_layout<0, 0>(d, ns, s);
        case R3D_GEMM_NN: return launch_layout<0, 1>(d, ns, s);
        default: return launch_layout<1, 1>(d, ns, s);
    }
}

R3D_EXPORT int r3d_splitk_reduce(const r3d_gemm_desc* dp, void* stream) {
    int rc = gemm_validate(dp);
    if (rc != R3D_OK) return rc;
    if (dp->splitk <= 1) return R3D_EINVAL;
    const size_t total = (size_t)dp->M * dp->N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *dp, nsplits_of(*dp));
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Host-side preparation of a problem group: validates the n descriptors (same layout, splitk == 1), fills their
 * `tile` (all get `tile`) and `vec` fields in place and writes prefix[0..n] (first workgroup of each problem).
 * The caller uploads descs and prefix to device memory once and replays r3d_gemm_grouped_launch every step; passing the
 * HOST copy of prefix as well (host_prefix, optional) lets groups of <= 32 problems carry the table in the kernel
 * arguments instead of searching it in device memory. */
R3D_EXPORT int r3d_gemm_grouped_prepare(r3d_gemm_desc* descs, int n, int tile, int32_t* prefix) {
    if (!descs || !prefix || n <= 0 || (tile != 1 && tile != 2)) return R3D_EINVAL;
    int total = 0;
    for (int p = 0; p < n; ++p) {
        r3d_gemm_desc& d = descs[p];
        d.tile = tile;
        d.splitk = 1;
        d.k_per_split = d.K;
        d.partial = nullptr;
        int rc = gemm_validate(&d);
        if (rc != R3D_OK) return rc;
        if (d.layout != descs[0].layout) return R3D_EINVAL;
        d.vec = gemm_can_vec(d) ? 1 : 0;
        prefix[p] = total;
        total += r3d_cdiv(d.M, kTileSz[tile]) * r3d_cdiv(d.N, kTileSz[tile]);
    }
    prefix[n] = total;
    return R3D_OK;
}

R3D_EXPORT int r3d_gemm_grouped_launch(const r3d_gemm_desc* dev_descs, const int32_t* dev_prefix, const int32_t* host_prefix,
                                       int n, int total_tiles, int layout, int tile, void* stream) {
    if (!dev_descs || !dev_prefix || n <= 0 || total_tiles <= 0) return R3D_EINVAL;
    if (host_prefix && (host_prefix[0] != 0 || host_prefix[n] != total_tiles)) return R3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (layout) {
        case R3D_GEMM_NT: return launch_grouped<0, 0>(dev_descs, dev_prefix, host_prefix, n, total_tiles, tile, s);
        case R3D_GEMM_NN: return launch_grouped<0, 1>(dev_descs, dev_prefix, host_prefix, n, total_tiles, tile, s);
        case R3D_GEMM_TN: return launch_grouped<1, 1>(dev_descs, dev_prefix, host_prefix, n, total_tiles, tile, s);
        default: return R3D_EINVAL;
    }
}
