// Effective rank of the fused token matrix via a batched one-sided (Hestenes) Jacobi SVD.
//
// Build-side addition: the reference has NO SVD / effective-rank code; README.md:8-14 only describes
// the quantity.  Definition used everywhere in this repo (Roy & Vetterli): sigma = svd(X[R,C]),
// p = sigma / sum(sigma), erank = exp(-sum p log p).  The checker is torch.linalg.svdvals on the oracle's fused
// features (oracle/futr_oracle.py: effective_rank).
//
// LDS-resident kernel: one workgroup (16 waves) per matrix.  The C column vectors (length R, zero-padded to a multiple
// of 4) are held TRANSPOSED in LDS ([C][Rp] floats: coalesced HBM column sweeps on load/store, 16-byte LDS accesses in
// the sweep).  A round of the round-robin tournament pairs all columns disjointly; a pair is rotated by a GROUP of G
// lanes (G = 16: one DPP row, four pairs per wave, 64 pairs per pass -- a whole round of a 128-column matrix at once;
// G = 64 when there are fewer pairs than that).  Each lane keeps its rows of both columns in registers between the dot
// product and the rotation; the squared column norms are cached in LDS and updated by the rotation
// (alpha' = alpha - t gamma, beta' = beta + t gamma; recomputed exactly at the start of every sweep), so a pair costs ONE
// reduction (gamma), done with v_add_f32_dpp row_ror butterflies -- no LDS crossbar traffic.  A pair is rotated when
// |gamma| > tol * sqrt(alpha * beta), tol = sqrt(R) * eps_f32 (the sgesvj criterion); sweeps repeat until one starts with
// every coupling below kStop (see there), or max_sweeps.
// For power-of-two C <= 128 (the training step's [B S, H] tokens) the pairs of a sweep are visited LEVEL BY LEVEL instead
// (HALVE, see erank_jacobi_kernel): a lane group of 8 keeps the first column of its pair in registers for a whole level and
// only the partner goes through LDS -- the kernel is bound by the LDS write rate and the rotation's latency chain, and this
// halves the former.
// Outputs: singular values (unsorted), entropy, erank, sweep count and optionally the rotated columns
// Af^T = (X V)^T [C][R], from which the backward  dX = Af diag(g / sigma^3) (Af^T X)  is two MFMA GEMMs
// (U diag(g) V^T with V^T = Sigma^-2 Af^T X; no accumulation of V in the sweep).
// Warm start: `v0` (optional, [C][C]) is an orthogonal matrix from an earlier decomposition of a nearby matrix; the
// caller passes X V0 as x (one MFMA GEMM), whose columns are already almost orthogonal -- 3-5 sweeps instead of 10-11.
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

constexpr int kJacThreads = 1024;
// Sweeps stop after the first sweep in which no pair's coupling |gamma| / sqrt(alpha beta) exceeded kStop (measured before
// its rotation; every pair above tol is still rotated in that sweep).  Jacobi converges quadratically at the end: a sweep
// that starts below kStop leaves couplings of order kStop^2, i.e. singular values good to ~kStop^4 and singular
// directions to ~kStop^2 -- the sweep that would only confirm "nothing left to rotate" (and the one before it, whose
// rotations are below what erank, +-0.5, and its gradient, 1e-2, can see) is not run.
#ifndef R3D_ERANK_STOP
#define R3D_ERANK_STOP 1e-2f
#endif
constexpr float kStop2 = R3D_ERANK_STOP * R3D_ERANK_STOP;

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// every lane of a 16-lane DPP row receives the row's sum (row_ror:8,4,2,1 butterfly -> 4 v_add_f32_dpp)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0x128>(v);
    v += dpp_f<0x124>(v);
    v += dpp_f<0x122>(v);
    v += dpp_f<0x121>(v);
    return v;
}
template <int G> __device__ __forceinline__ float group_sum(float v) {
    if (G == 8) {                               // quad butterflies + the mirror inside a half row (lane i <-> 7 - i)
        v += dpp_f<0xB1>(v);                    // quad_perm [1,0,3,2]
        v += dpp_f<0x4E>(v);                    // quad_perm [2,3,0,1]
        v += dpp_f<0x141>(v);                   // row_half_mirror
        return v;
    }
    v = row16_sum(v);
    if (G == 64) {
        const int x = __builtin_bit_cast(int, v);
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    }
    return v;
}
__device__ __forceinline__ float dot4(const float4 u, const float4 v) { return u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w; }

// pair k of round t of a round-robin tournament among n players (n even): player n-1 stays, the others rotate
__device__ __forceinline__ void rr_pair(int n, int t, int k, int& i, int& j) {
    if (k == 0) { i = n - 1; j = t; }
    else {
        i = t + k; if (i >= n - 1) i -= n - 1;
        j = t - k + (n - 1); if (j >= n - 1) j -= n - 1;
    }
    if (i > j) { const int s = i; i = j; j = s; }
}

template <int G> __device__ __forceinline__ float col_norm2(const float4* col, int Rp4, int lg) {
    float s = 0.f;
    for (int ch = lg; ch < Rp4; ch += G) { const float4 u = col[ch]; s += dot4(u, u); }
    return group_sum<G>(s);
}

// One Hestenes rotation of columns (i, j) by a group of G lanes (lane lg of the group).  A column is S4 16-byte chunks
// long; the dot product runs over its first D4 chunks only (the matrix rows), the rotation over all of them (the tail
// holds the column's slice of the accumulated right singular basis V when a warm start is kept, see ErankArgs::vt_out).
// NCH > 0: every lane holds its NCH chunks of both columns in registers (needs NCH * G >= S4; EXACT: NCH * G == S4 and
// D4 == nchd * G, no bounds checks, all LDS reads issued before the first use); NCH == 0: any length, two passes over
// LDS.  Returns true when the pair's coupling exceeded the CONTINUE threshold gamma^2 > stop2 alpha beta (it is rotated
// whenever it exceeds the much smaller tol2).
template <int G, int NCH, bool EXACT>
__device__ __forceinline__ bool jac_pair(float4* A4, float* nrm, int S4, int D4, int nchd, int i, int j, int lg, float tol2,
                                         float negl, float stop2) {
    float4* ai = A4 + (size_t)i * S4;
    float4* aj = A4 + (size_t)j * S4;
    float4 u[NCH > 0 ? NCH : 1], v[NCH > 0 ? NCH : 1];
    float ga = 0.f;
    if (NCH > 0) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int ch = lg + G * q;
            if (EXACT || ch < S4) { u[q] = ai[ch]; v[q] = aj[ch]; }
            else { u[q] = make_float4(0.f, 0.f, 0.f, 0.f); v[q] = u[q]; }
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q)
            if (EXACT ? (q < nchd) : (lg + G * q < D4)) ga += dot4(u[q], v[q]);
    } else {
        for (int ch = lg; ch < D4; ch += G) ga += dot4(ai[ch], aj[ch]);
    }
    ga = group_sum<G>(ga);
    const float al = nrm[i], be = nrm[j];
    // |gamma| > tol sqrt(alpha beta), squared (no v_sqrt on the critical path)
    const bool rot = ga * ga > tol2 * al * be && al > negl && be > negl;
    if (rot) {                                  // the same decision in every lane of the group
        // t = tan(theta) = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (beta - alpha) / (2 gamma), written with
        // tau = beta - alpha, d = 2 gamma as  sign(tau) d / (|tau| + sqrt(tau^2 + d^2)):  one v_sqrt + one v_rcp
        const float tau = be - al, d = 2.f * ga;
        float t = d * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(tau * tau + d * d));
        t = tau >= 0.f ? t : -t;
        const float c = __builtin_amdgcn_rsqf(1.f + t * t), s = c * t;
        if (NCH > 0) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int ch = lg + G * q;
                if (EXACT || ch < S4) {
                    const float4 a = u[q], b = v[q];
                    ai[ch] = make_float4(c * a.x - s * b.x, c * a.y - s * b.y, c * a.z - s * b.z, c * a.w - s * b.w);
                    aj[ch] = make_float4(s * a.x + c * b.x, s * a.y + c * b.y, s * a.z + c * b.z, s * a.w + c * b.w);
                }
            }
        } else {
            for (int ch = lg; ch < S4; ch += G) {
                const float4 a = ai[ch], b = aj[ch];
                ai[ch] = make_float4(c * a.x - s * b.x, c * a.y - s * b.y, c * a.z - s * b.z, c * a.w - s * b.w);
                aj[ch] = make_float4(s * a.x + c * b.x, s * a.y + c * b.y, s * a.z + c * b.z, s * a.w + c * b.w);
            }
        }
        if (lg == 0) { nrm[i] = fmaxf(al - t * ga, 0.f); nrm[j] = be + t * ga; }
    }
    return rot && ga * ga > stop2 * al * be;
}

// The same rotation with column i RESIDENT IN REGISTERS (u, its cached squared norm al) -- only the partner column j goes
// through LDS: half the LDS traffic of jac_pair.  G = 16 or 8, EXACT layout (NCH chunks per lane, all of them matrix rows).
// G = 8: the 16 lanes the LDS serves together belong to two pairs whose columns start on the same bank, so the odd group
// visits its chunks in the order q ^ 1 (qx): at any instruction the two groups sit on different halves of the bank row.
template <int G, int NCH>
__device__ __forceinline__ bool jac_pair_fixed(float4 (&u)[NCH], float& al, float4* aj, float* nrm_j, int lg, float tol2,
                                               float negl, float stop2, int qx) {
    float4 v[NCH];
    float ga = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) v[q] = aj[lg + G * (q ^ qx)];
#pragma unroll
    for (int q = 0; q < NCH; ++q) ga += dot4(u[q], v[q]);
    ga = group_sum<G>(ga);
    const float be = *nrm_j;
    const bool rot = ga * ga > tol2 * al * be && al > negl && be > negl;
    if (rot) {
        const float tau = be - al, d = 2.f * ga;
        float t = d * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(tau * tau + d * d));
        t = tau >= 0.f ? t : -t;
        const float c = __builtin_amdgcn_rsqf(1.f + t * t), s = c * t;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float4 a = u[q], b = v[q];
            u[q] = make_float4(c * a.x - s * b.x, c * a.y - s * b.y, c * a.z - s * b.z, c * a.w - s * b.w);
            aj[lg + G * (q ^ qx)] = make_float4(s * a.x + c * b.x, s * a.y + c * b.y, s * a.z + c * b.z, s * a.w + c * b.w);
        }
        if (lg == 0) *nrm_j = be + t * ga;
        const float bigger = ga * ga > stop2 * al * be;
        al = fmaxf(al - t * ga, 0.f);
        return bigger;
    }
    return false;
}

// The same rotation with BOTH columns in registers (u: cached squared norm al, v: be).  Returns the pair's "still moving"
// flag like jac_pair; every lane of the group takes the same branch.
template <int G, int NCH>
__device__ __forceinline__ bool jac_rot_rr(float4 (&u)[NCH], float& al, float4 (&v)[NCH], float& be, float tol2, float negl,
                                           float stop2) {
    float ga = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) ga += dot4(u[q], v[q]);
    ga = group_sum<G>(ga);
    const bool rot = ga * ga > tol2 * al * be && al > negl && be > negl;
    if (rot) {
        const float tau = be - al, d = 2.f * ga;
        float t = d * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(tau * tau + d * d));
        t = tau >= 0.f ? t : -t;
        const float c = __builtin_amdgcn_rsqf(1.f + t * t), sn = c * t;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float4 a = u[q], b = v[q];
            u[q] = make_float4(c * a.x - sn * b.x, c * a.y - sn * b.y, c * a.z - sn * b.z, c * a.w - sn * b.w);
            v[q] = make_float4(sn * a.x + c * b.x, sn * a.y + c * b.y, sn * a.z + c * b.z, sn * a.w + c * b.w);
        }
        const bool bigger = ga * ga > stop2 * al * be;
        be = be + t * ga;
        al = fmaxf(al - t * ga, 0.f);
        return bigger;
    }
    return false;
}

struct ErankArgs {
    const float* x; int ld; long long batch_stride;      // [batch][R][ld]
    int R, C;
    float* sigma;          // [batch][C]
    float* af_t;           // [batch][C][R] or NULL
    float* stats;          // [batch][4] = {erank, entropy, sum sigma, sweeps}
    int sqrt_out;          // input is a Gram matrix: report sqrt of its singular values
    int max_sweeps;
    const float* vt_in;    // warm start (batch == 1): V0^T [C][C]; x already holds X V0.  NULL: V0 = identity
    float* vt_out;         // (V0 V')^T [C][C]: the right singular basis after the sweeps (rotations applied to V0's rows too)
};

// sigma -> {erank, entropy, sum sigma}; sig: C floats in LDS; wred: 16 floats of LDS scratch; all threads of the block
__device__ __forceinline__ void erank_stats_block(const float* sig, int C, float* wred, float* st, float sweeps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    float part = 0.f;
    for (int c = tid; c < C; c += blockDim.x) part += sig[c];
    part = wave_sum(part);
    if (lane == 0) wred[wave] = part;
    __syncthreads();
    float total = 0.f;
    for (int w = 0; w < nw; ++w) total += wred[w];
    __syncthreads();
    float ent = 0.f;
    for (int c = tid; c < C; c += blockDim.x) {
        const float p = sig[c] / total;
        if (p > 0.f) ent -= p * logf(p);
    }
    ent = wave_sum(ent);
    if (lane == 0) wred[wave] = ent;
    __syncthreads();
    if (tid == 0) {
        float H = 0.f;
        for (int w = 0; w < nw; ++w) H += wred[w];
        st[0] = expf(H); st[1] = H; st[2] = total; st[3] = sweeps;
    }
}

// HALVE (C a power of two <= 128, G = 16, EXACT, no basis carried): the pairs of a sweep are visited level by level --
// level s = C/2, C/4, .., 1 splits every block of 2s columns into halves and runs s rounds, round r pairing column kk of the
// first half with column (kk + r) mod s of the second: C/2 disjoint pairs in every one of the (C/2 + C/4 + .. + 1) = C - 1
// rounds, every pair once per sweep, like the round-robin order (same or fewer sweeps: simulated on CPU and measured).
// Unlike it, a group keeps the SAME first column for a whole level, so that column stays in registers and only the
// partner travels through LDS: per round 1 column read + 1 written per pair instead of 2 + 2.  The round is bound by the
// LDS (ds_write_b128 ~83 B/clk/CU, reads ~244) plus the dot -> rcp / sqrt / rsq -> rotate latency chain, not by issue.
// MODE 2 (round 3): the level order in 2 x 2 register blocks.  A group owns TWO adjacent first-half columns for a whole level
// and, per super-round, fetches two adjacent second-half columns, rotates all four cross pairs in registers -- (i1,j1),
// (i2,j2), then (i1,j2), (i2,j1): two dependent stages of two independent rotations -- and writes the two partners back:
// per rotation HALF the LDS traffic and HALF the workgroup barriers of MODE 1 (the round is bound by exactly those two:
// 64 KB of LDS traffic + one barrier + the dot -> reduce -> rcp / sqrt / rsq -> rotate chain per 64 rotations).  The last
// level (blocks of two columns) runs as one plain round.  Every pair is still visited once per sweep.
template <int G, int NCH, bool EXACT, int HALVE = 0>
__global__ __launch_bounds__(kJacThreads) void erank_jacobi_kernel(const ErankArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // columns [Cp][Sp] (Sp = Rp [+ Cv]), nrm [Cp]
    __shared__ int rotated;
    __shared__ float wred[kJacThreads / 64];
    const int R = a.R, C = a.C;
    const int Rp = (R + 3) & ~3, Rp4 = Rp >> 2;
    const int Cp = (C + 1) & ~1;                                   // even number of players (last may be a dummy)
    const int Cv = a.vt_out ? (C + 3) & ~3 : 0;                    // rows of V^T kept behind every column
    const int Sp = Rp + Cv, S4 = Sp >> 2;
    const int nchd = Rp4 / G;                                      // (EXACT: chunks of the dot product per lane)
    float* A = smem;
    float4* A4 = reinterpret_cast<float4*>(smem);
    float* nrm = smem + (size_t)Cp * Sp;
    const int tid = threadIdx.x;
    const int slot = tid / G, lg = tid % G, nslots = kJacThreads / G;
    const float* X = a.x + (size_t)blockIdx.x * a.batch_stride;

    // transposing fill: consecutive lanes take consecutive rows of one column (conflict-free LDS stores; the strided
    // HBM reads of the [R, C] matrix hit each 64-byte sector once, the rest comes from L2)
    for (int e = tid; e < Cp * Rp; e += kJacThreads) {
        const int c = e / Rp, r = e - c * Rp;
        A[(size_t)c * Sp + r] = (c < C && r < R) ? X[(size_t)r * a.ld + c] : 0.f;
    }
    if (Cv) {                                                       // column c carries row c of V^T = column c of V
        for (int e = tid; e < Cp * Cv; e += kJacThreads) {
            const int c = e / Cv, k = e - c * Cv;
            float v = 0.f;
            if (c < C && k < C) v = a.vt_in ? a.vt_in[(size_t)c * C + k] : (c == k ? 1.f : 0.f);
            A[(size_t)c * Sp + Rp + k] = v;
        }
    }
    __syncthreads();

    const float tol2 = (float)R * (1.1920929e-7f * 1.1920929e-7f);      // tol = sqrt(R) eps, squared
    const int npairs = Cp / 2, nrounds = Cp - 1;
    float negl = 0.f;
    int sweeps = 0;
    for (; sweeps < a.max_sweeps; ++sweeps) {
        // exact squared norms (the cached ones drift by rounding over a sweep of updates)
        for (int c = slot; c < Cp; c += nslots) {
            const float s2 = col_norm2<G>(A4 + (size_t)c * S4, Rp4, lg);
            if (lg == 0) nrm[c] = s2;
        }
        if (tid == 0) rotated = 0;
        __syncthreads();
        if (sweeps == 0) {
            // ||X||_F^2: columns whose squared norm falls below (1e-6 ||X||_F)^2 are numerically zero (rank-deficient
            // input, e.g. R < C) and are not rotated against each other -- their mutual "angles" are rounding noise.
            float f = 0.f;
            for (int c = 0; c < C; ++c) f += nrm[c];
            negl = f * 1e-12f;
        }
        bool my_rot = false;
        if constexpr (HALVE == 2) {
            constexpr int N2 = NCH > 0 ? NCH : 1;
            const int ngrp = npairs / 2;                           // groups: one per two first-half columns
            const bool act = slot < ngrp;
            for (int hs = npairs; hs >= 2; hs >>= 1) {
                const int hg = hs >> 1;                            // super-columns (pairs of columns) per half block
                const int bp = slot / hg, kk = slot - bp * hg;
                const int i1 = bp * 2 * hs + 2 * kk, jb = bp * 2 * hs + hs;
                float4 u1[N2], u2[N2];
                float al1 = 0.f, al2 = 0.f;
                if (act) {
#pragma unroll
                    for (int q = 0; q < N2; ++q) {
                        u1[q] = A4[(size_t)i1 * S4 + lg + G * q];
                        u2[q] = A4[(size_t)(i1 + 1) * S4 + lg + G * q];
                    }
                    al1 = nrm[i1]; al2 = nrm[i1 + 1];
                }
                for (int r = 0; r < hg; ++r) {
                    if (act) {
                        const int j1 = jb + 2 * ((kk + r) & (hg - 1));
                        float4* c1 = A4 + (size_t)j1 * S4;
                        float4* c2 = c1 + S4;
                        float4 v1[N2], v2[N2];
#pragma unroll
                        for (int q = 0; q < N2; ++q) { v1[q] = c1[lg + G * q]; v2[q] = c2[lg + G * q]; }
                        float be1 = nrm[j1], be2 = nrm[j1 + 1];
                        my_rot |= jac_rot_rr<G, N2>(u1, al1, v1, be1, tol2, negl, kStop2);
                        my_rot |= jac_rot_rr<G, N2>(u2, al2, v2, be2, tol2, negl, kStop2);
                        my_rot |= jac_rot_rr<G, N2>(u1, al1, v2, be2, tol2, negl, kStop2);
                        my_rot |= jac_rot_rr<G, N2>(u2, al2, v1, be1, tol2, negl, kStop2);
#pragma unroll
                        for (int q = 0; q < N2; ++q) { c1[lg + G * q] = v1[q]; c2[lg + G * q] = v2[q]; }
                        if (lg == 0) { nrm[j1] = be1; nrm[j1 + 1] = be2; }
                    }
                    __syncthreads();
                }
                if (act) {
#pragma unroll
                    for (int q = 0; q < N2; ++q) {
                        A4[(size_t)i1 * S4 + lg + G * q] = u1[q];
                        A4[(size_t)(i1 + 1) * S4 + lg + G * q] = u2[q];
                    }
                    if (lg == 0) { nrm[i1] = al1; nrm[i1 + 1] = al2; }
                }
                __syncthreads();
            }
            // last level: blocks of two columns (2k, 2k + 1), one round; group g takes pairs 2g and 2g + 1
            if (act) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * (2 * slot + h);
                    float4* ci = A4 + (size_t)i * S4;
                    float4* cj = ci + S4;
                    float4 u[N2], v[N2];
#pragma unroll
                    for (int q = 0; q < N2; ++q) { u[q] = ci[lg + G * q]; v[q] = cj[lg + G * q]; }
                    float al = nrm[i], be = nrm[i + 1];
                    my_rot |= jac_rot_rr<G, N2>(u, al, v, be, tol2, negl, kStop2);
#pragma unroll
                    for (int q = 0; q < N2; ++q) { ci[lg + G * q] = u[q]; cj[lg + G * q] = v[q]; }
                    if (lg == 0) { nrm[i] = al; nrm[i + 1] = be; }
                }
            }
            __syncthreads();
        } else if constexpr (HALVE == 1) {
            const bool act = slot < npairs;                        // (npairs <= 64 <= nslots)
            const int qx = (G == 8) ? (slot & 1) : 0;
            for (int hs = npairs; hs >= 1; hs >>= 1) {
                const int bp = slot / hs, kk = slot - bp * hs;
                const int i = bp * 2 * hs + kk, jb = bp * 2 * hs + hs;
                float4 u[NCH > 0 ? NCH : 1];
                float al = 0.f;
                if (act) {
#pragma unroll
                    for (int q = 0; q < NCH; ++q) u[q] = A4[(size_t)i * S4 + lg + G * (q ^ qx)];
                    al = nrm[i];
                }
                for (int r = 0; r < hs; ++r) {
                    if (act) {
                        const int j = jb + ((kk + r) & (hs - 1));
                        my_rot |= jac_pair_fixed<G, (NCH > 0 ? NCH : 1)>(u, al, A4 + (size_t)j * S4, nrm + j, lg, tol2, negl, kStop2, qx);
                    }
                    __syncthreads();
                }
                if (act) {
#pragma unroll
                    for (int q = 0; q < NCH; ++q) A4[(size_t)i * S4 + lg + G * (q ^ qx)] = u[q];
                    if (lg == 0) nrm[i] = al;
                }
                __syncthreads();
            }
        } else {
        for (int rd = 0; rd < nrounds; ++rd) {
            for (int k = slot; k < npairs; k += nslots) {
                int i, j;
                rr_pair(Cp, rd, k, i, j);
                my_rot |= jac_pair<G, NCH, EXACT>(A4, nrm, S4, Rp4, nchd, i, j, lg, tol2, negl, kStop2);
            }
            __syncthreads();
        }
        }
        if (my_rot) rotated = 1;
        __syncthreads();
        const int any = rotated;
        __syncthreads();
        if (!any) { ++sweeps; break; }
    }

    // singular values (from the columns themselves, not the cached norms), entropy, erank
    float* sig = a.sigma + (size_t)blockIdx.x * C;
    for (int c = slot; c < C; c += nslots) {
        const float s2 = col_norm2<G>(A4 + (size_t)c * S4, Rp4, lg);
        float s = s2 > negl ? sqrtf(s2) : 0.f;       // numerically zero columns (rank-deficient input) report sigma = 0
        if (a.sqrt_out) s = sqrtf(s);
        if (lg == 0) { nrm[c] = s; sig[c] = s; }
    }
    __syncthreads();
    erank_stats_block(nrm, C, wred, a.stats + (size_t)blockIdx.x * 4, (float)sweeps);
    if (a.af_t) {
        float* out = a.af_t + (size_t)blockIdx.x * C * R;
        if ((R & 3) == 0) {
            float4* o4 = reinterpret_cast<float4*>(out);
            for (int e = tid; e < C * Rp4; e += kJacThreads) {
                const int c = e / Rp4, q = e - c * Rp4;
                o4[e] = A4[(size_t)c * S4 + q];
            }
        } else {
            for (int e = tid; e < R * C; e += kJacThreads) {
                const int c = e / R, r = e - c * R;
                out[e] = A[(size_t)c * Sp + r];
            }
        }
    }
    if (Cv) {
        for (int e = tid; e < C * C; e += kJacThreads) {
            const int c = e / C, k = e - c * C;
            a.vt_out[e] = A[(size_t)c * Sp + Rp + k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Matrices that do not fit one CU's LDS: two-level (block) one-sided Jacobi.  The columns live transposed in HBM
// (At [Cpad][Rp], a few MB: L2 / Infinity-Cache resident), grouped in nblk blocks of b columns.  One sweep =
//   1 launch  "within": workgroup p orthogonalises the b columns of block p against each other,
//   nblk-1 launches "cross": round-robin over block pairs (p,q); a workgroup holds both blocks in LDS (2b columns,
//                    16-byte coalesced column sweeps from / to HBM) and rotates every (i in p, j in q) pair: b inner
//                    rounds of b disjoint pairs, one wave (G = 64) per pair, same rotation code as above.
// so every column pair is visited exactly once per sweep (a cyclic ordering -> the usual quadratic convergence).
// The launches of all sweeps are enqueued up front; a device-side flag turns the remainder into no-ops once a sweep
// made no rotation.  ctrl (ints): [0] done, [1] sweeps run, [2] float bits of ||X||_F^2, [4+s] rotations in sweep s.
struct ErankBlk {
    float* at; int R, Rp, C, b, nblk;  // nblk even (the last block may be a dummy one: index >= nreal)
    int nreal;
    int* ctrl;
};

// xt != 0: x holds the TRANSPOSE of the matrix to decompose ([C][ld], row c = column c of the matrix) -- the layout
// `at` wants, so the copy needs no transposition (the engine decomposes fused^T when the token matrix is wide).
__global__ __launch_bounds__(256) void erank_blk_init_kernel(const float* __restrict__ x, int ld, ErankBlk g, int xt) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    float part = 0.f;
    if (xt) {
        for (int k = ty; k < 32; k += 8) {
            const int c = c0 + k, r = r0 + tx;
            const float v = (r < g.R && c < g.C) ? x[(size_t)c * ld + r] : 0.f;
            if (c < g.nreal * g.b && r < g.Rp) g.at[(size_t)c * g.Rp + r] = v;
            part += v * v;
        }
    } else {
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        const float v = (r < g.R && c < g.C) ? x[(size_t)r * ld + c] : 0.f;
        tile[k][tx] = v;
        part += v * v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < g.nreal * g.b && r < g.Rp) g.at[(size_t)c * g.Rp + r] = tile[tx][k];      // rows R..Rp-1: zeros
    }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<float*>(g.ctrl + 2), red[0] + red[1] + red[2] + red[3]);
}

// mode 0: within-block (grid = nreal), mode 1: cross, round rd of the block tournament (grid = nblk / 2)
template <int NCH, bool EXACT>
__global__ __launch_bounds__(kJacThreads) void erank_blk_round_kernel(ErankBlk g, int mode, int rd, int sweep) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // A [2b][Rp], nrm [2b]
    if (g.ctrl[0]) return;
    constexpr int G = 64;
    const int b = g.b, Rp = g.Rp, Rp4 = Rp >> 2;
    const int tid = threadIdx.x, slot = tid / G, lg = tid % G, nslots = blockDim.x / G;
    int p, q;
    if (mode == 0) { p = blockIdx.x; q = -1; }
    else {
        rr_pair(g.nblk, rd, blockIdx.x, p, q);
        if (q >= g.nreal) return;                                  // paired with the dummy block
    }
    const int ncol = (mode == 0) ? b : 2 * b;
    float4* A4 = reinterpret_cast<float4*>(smem);
    float* nrm = smem + (size_t)ncol * Rp;
    const float4* P4 = reinterpret_cast<const float4*>(g.at + (size_t)p * b * Rp);
    const float4* Q4 = reinterpret_cast<const float4*>(g.at + (size_t)(q < 0 ? p : q) * b * Rp);
    const int half = b * Rp4;
    for (int e = tid; e < ncol * Rp4; e += blockDim.x) A4[e] = e < half ? P4[e] : Q4[e - half];
    __syncthreads();
    for (int c = slot; c < ncol; c += nslots) {
        const float s2 = col_norm2<G>(A4 + (size_t)c * Rp4, Rp4, lg);
        if (lg == 0) nrm[c] = s2;
    }
    __syncthreads();
    const float negl = __int_as_float(g.ctrl[2]) * 1e-12f;
    const float tol2 = (float)g.R * (1.1920929e-7f * 1.1920929e-7f);
    bool my_rot = false;
    const int npairs = (mode == 0) ? b / 2 : b;
    const int nrounds = (mode == 0) ? b - 1 : b;
    if (EXACT && NCH > 0 && mode == 1 && nslots >= b) {
        // cross rounds: wave k keeps column k of block p for all b rounds -- resident in registers, only the partner
        // column of block q goes through LDS (see jac_pair_fixed)
        const bool act = slot < b;
        float4 u[NCH > 0 ? NCH : 1];
        float al = 0.f;
        if (act) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) u[q] = A4[(size_t)slot * Rp4 + lg + G * q];
            al = nrm[slot];
        }
        for (int t = 0; t < b; ++t) {
            if (act) {
                int j = slot + t; if (j >= b) j -= b; j += b;
                my_rot |= jac_pair_fixed<G, (NCH > 0 ? NCH : 1)>(u, al, A4 + (size_t)j * Rp4, nrm + j, lg, tol2, negl, kStop2, 0);
            }
            __syncthreads();
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) A4[(size_t)slot * Rp4 + lg + G * q] = u[q];
        }
        __syncthreads();
    } else {
    for (int t = 0; t < nrounds; ++t) {
        for (int k = slot; k < npairs; k += nslots) {
            int i, j;
            if (mode == 0) rr_pair(b, t, k, i, j);
            else { i = k; j = k + t; if (j >= b) j -= b; j += b; }
            my_rot |= jac_pair<G, NCH, EXACT>(A4, nrm, Rp4, Rp4, Rp4 / G, i, j, lg, tol2, negl, kStop2);
        }
        __syncthreads();
    }
    }
    float4* Pw = reinterpret_cast<float4*>(g.at + (size_t)p * b * Rp);
    float4* Qw = reinterpret_cast<float4*>(g.at + (size_t)(q < 0 ? p : q) * b * Rp);
    for (int e = tid; e < ncol * Rp4; e += blockDim.x) { if (e < half) Pw[e] = A4[e]; else Qw[e - half] = A4[e]; }
    if (my_rot && lg == 0) atomicAdd(g.ctrl + 4 + sweep, 1);
}

__global__ void erank_blk_sweep_end_kernel(ErankBlk g, int sweep) {
    if (g.ctrl[0]) return;
    g.ctrl[1] = sweep + 1;
    if (g.ctrl[4 + sweep] == 0) g.ctrl[0] = 1;
}

// sigma[c] = ||column c||: one wave per column, 16 columns per workgroup
__global__ __launch_bounds__(kJacThreads) void erank_blk_sigma_kernel(ErankBlk g, float* sig) {
    constexpr int G = 64;
    const int tid = threadIdx.x, slot = tid / G, lg = tid % G;
    const int c = blockIdx.x * (kJacThreads / G) + slot;
    if (c >= g.C) return;
    const float s2 = col_norm2<G>(reinterpret_cast<const float4*>(g.at + (size_t)c * g.Rp), g.Rp >> 2, lg);
    if (lg == 0) sig[c] = s2 > __int_as_float(g.ctrl[2]) * 1e-12f ? sqrtf(s2) : 0.f;   // (numerically zero columns: 0)
}

__global__ __launch_bounds__(kJacThreads) void erank_blk_stats_kernel(ErankBlk g, const float* sig, float* stats) {
    __shared__ float wred[kJacThreads / 64];
    erank_stats_block(sig, g.C, wred, stats, (float)g.ctrl[1]);
}

// coef[k] = gout * d erank / d sigma_k / sigma_k^3, zero where sigma_k is negligible (rank-deficient directions
// carry no defined singular vectors; p log p -> 0 there: SURVEY.md Appendix A.11)
__global__ __launch_bounds__(256) void erank_coef_kernel(const float* sigma, const float* stats, const float* gout,
                                                         float* coef, int C, int max_rank) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C) return;
    const float er = stats[0], H = stats[1], total = stats[2];
    const float s = sigma[k];
    float smax = 0.f;
    int larger = 0;                      // singular values above this one: only the min(R, C) largest are real
    for (int j = 0; j < C; ++j) {
        smax = fmaxf(smax, sigma[j]);
        larger += (sigma[j] > s || (sigma[j] == s && j < k)) ? 1 : 0;
    }
    float c = 0.f;
    if (s > 1e-6f * smax && s > 0.f && (max_rank <= 0 || larger < max_rank)) {
        const float p = s / total;
        const float g = -er * (logf(p) + H) / total;
        c = (gout ? *gout : 1.f) * g / (s * s * s);
    }
    coef[k] = c;
}

// The well-conditioned form of the backward (see r3d_erank_bwd_coef2): cg[k] = gout * (d erank / d sigma_k) / sigma_k and
// inv[k] = 1 / sigma_k, both zero where erank_coef_kernel's coefficient is zero.
__global__ __launch_bounds__(256) void erank_coef2_kernel(const float* sigma, const float* stats, const float* gout,
                                                          float* cg, float* inv, int C, int max_rank) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C) return;
    const float er = stats[0], H = stats[1], total = stats[2];
    const float s = sigma[k];
    float smax = 0.f;
    int larger = 0;
    for (int j = 0; j < C; ++j) {
        smax = fmaxf(smax, sigma[j]);
        larger += (sigma[j] > s || (sigma[j] == s && j < k)) ? 1 : 0;
    }
    float c = 0.f, iv = 0.f;
    if (s > 1e-6f * smax && s > 0.f && (max_rank <= 0 || larger < max_rank)) {
        const float p = s / total;
        const float g = -er * (logf(p) + H) / total;
        c = (gout ? *gout : 1.f) * g / s;
        iv = 1.f / s;
    }
    cg[k] = c;
    inv[k] = iv;
}

// w[r, :] = cg[r] * (2 w[r, :] - p[r, :])
__global__ __launch_bounds__(256) void erank_bwd_fix_kernel(float* w, const float* p, const float* cg, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256)
        w[e] = cg[e / cols] * (2.f * w[e] - p[e]);
}

__global__ __launch_bounds__(256) void scale_rows_kernel(float* x, int ld, int rows, int cols, const float* coef) {
    const size_t total = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        x[(size_t)r * ld + c] *= coef[r];
    }
}

// one Newton-Schulz step towards the nearest orthogonal matrix: V^T <- 1.5 V^T - 0.5 (V^T V) V^T  (gv = (V^T V^T^T) V^T)
__global__ __launch_bounds__(256) void erank_vt_polish_kernel(const float* __restrict__ vt_raw, const float* __restrict__ gv,
                                                              float* __restrict__ vt, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
        vt[e] = 1.5f * vt_raw[e] - 0.5f * gv[e];
}

}  // namespace r3d

using namespace r3d;

R3D_EXPORT int r3d_erank_jacobi_warm(const float* x, int ld, int64_t batch_stride, int batch, int R, int C, int gram,
                                     float* sigma, float* af_t, float* stats, int max_sweeps, const float* vt_in,
                                     float* vt_out, void* stream);

static inline int erank_rp(int R) { return (R + 3) & ~3; }

/* LDS bytes of r3d_erank_jacobi for an [R, C] matrix; keep_v != 0: with the C x C right singular basis riding along. */
R3D_EXPORT int64_t r3d_erank_lds_bytes(int R, int C) { return ((int64_t)((C + 1) & ~1) * erank_rp(R) + ((C + 1) & ~1)) * 4; }
R3D_EXPORT int64_t r3d_erank_lds_bytes_v(int R, int C) {
    return ((int64_t)((C + 1) & ~1) * (erank_rp(R) + ((C + 3) & ~3)) + ((C + 1) & ~1)) * 4;
}

template <int G, int NCH, bool EXACT>
static int erank_launch2(const ErankArgs& a, int batch, int64_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)erank_jacobi_kernel<G, NCH, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((erank_jacobi_kernel<G, NCH, EXACT>), dim3(batch), dim3(kJacThreads), (size_t)lds, st, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
template <int G, int NCH>
static int erank_launch(const ErankArgs& a, int batch, int64_t lds, hipStream_t st) {
    const int rp4 = erank_rp(a.R) / 4, s4 = rp4 + (a.vt_out ? ((a.C + 3) & ~3) / 4 : 0);
    if (NCH > 0 && s4 == G * NCH && rp4 % G == 0) return erank_launch2<G, NCH, true>(a, batch, lds, st);
    return erank_launch2<G, NCH, false>(a, batch, lds, st);
}

/* Batched effective rank.  x: [batch][R][ld] (row-major matrices X[R,C]); sigma [batch][C] (unsorted);
 * af_t (optional) [batch][C][R] rotated columns for the backward; stats [batch][4] = {erank, entropy, sum sigma,
 * sweeps}.  gram != 0: x holds Gram matrices X^T X (R == C) and sigma reports sqrt of their singular values.
 * Returns R3D_EINVAL when the matrix does not fit the 160 KB LDS of one CU (see r3d_erank_lds_bytes). */
R3D_EXPORT int r3d_erank_jacobi(const float* x, int ld, int64_t batch_stride, int batch, int R, int C, int gram,
                                float* sigma, float* af_t, float* stats, int max_sweeps, void* stream) {
    return r3d_erank_jacobi_warm(x, ld, batch_stride, batch, R, C, gram, sigma, af_t, stats, max_sweeps, nullptr, nullptr,
                                 stream);
}

/* r3d_erank_jacobi that also carries the right singular basis: every rotation of two columns of X V0 is applied to
 * the same two rows of V^T, so vt_out [C][C] = (V0 V')^T with X (V0 V') = the rotated columns.  vt_in: V0^T of an
 * earlier decomposition of a NEARBY matrix (the caller passes x = X V0, one GEMM; NULL = identity): the columns start
 * almost orthogonal and the sweep count drops from 10-11 to 3-5.  batch must be 1; the matrix and the basis must fit
 * the LDS together (r3d_erank_lds_bytes_v).  vt_out == NULL: plain r3d_erank_jacobi. */
R3D_EXPORT int r3d_erank_jacobi_warm(const float* x, int ld, int64_t batch_stride, int batch, int R, int C, int gram,
                                     float* sigma, float* af_t, float* stats, int max_sweeps, const float* vt_in,
                                     float* vt_out, void* stream) {
    R3D_REQUIRE(x && sigma && stats && batch > 0 && R > 0 && C > 0 && ld >= C);
    R3D_REQUIRE(!gram || R == C);
    R3D_REQUIRE(!vt_in || vt_out);
    R3D_REQUIRE(!vt_out || batch == 1);
    const int64_t lds = vt_out ? r3d_erank_lds_bytes_v(R, C) : r3d_erank_lds_bytes(R, C);
    R3D_REQUIRE(lds <= 160 * 1024 - 256);
    if (af_t && (R & 3) == 0) R3D_REQUIRE(r3d_aligned16(af_t));
    ErankArgs a{x, ld, (long long)batch_stride, R, C, sigma, af_t, stats, gram, max_sweeps > 0 ? max_sweeps : 30, vt_in, vt_out};
    hipStream_t st = (hipStream_t)stream;
    const int npairs = ((C + 1) & ~1) / 2;
    const int s4 = erank_rp(R) / 4 + (vt_out ? ((C + 3) & ~3) / 4 : 0);       // 16-byte chunks per column
#ifndef R3D_JAC_HALVING
#define R3D_JAC_HALVING 1
#endif
    if (R3D_JAC_HALVING && !vt_out && C >= 32 && C <= 128 && (C & (C - 1)) == 0 && (R & 63) == 0 && R <= 512) {
        // power-of-two column count, whole chunks: the level order with the first column of every pair in registers
        auto go = [&](auto kern) {
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL(kern, dim3(batch), dim3(kJacThreads), (size_t)lds, st, a);
            R3D_LAUNCH_CHECK();
            return (int)R3D_OK;
        };
#ifndef R3D_JAC_LANES
#define R3D_JAC_LANES 8
#endif
#ifndef R3D_JAC_BLOCK2
#define R3D_JAC_BLOCK2 0        // measured (round 3): 90 us per sweep and 8 sweeps against 81 us and 7 for MODE 1 at [128, 128]:
#endif                          // half the LDS traffic and barriers, but each super-round is TWO dependent rotation stages and
                                // the round was a latency chain already (dot -> reduce -> rcp / sqrt / rsq -> rotate), not
                                // LDS- or barrier-bound; kept under the macro, tests pass in both modes
        if (R3D_JAC_BLOCK2 && C >= 8) {                // the level order in 2 x 2 register blocks (16 lanes per group)
            switch (R / 64) {
                case 1: return go(erank_jacobi_kernel<16, 1, true, 2>);
                case 2: return go(erank_jacobi_kernel<16, 2, true, 2>);
                case 4: return go(erank_jacobi_kernel<16, 4, true, 2>);
                default: break;
            }
        }
        if (R3D_JAC_LANES == 8) {
            switch (R / 64) {
                case 1: return go(erank_jacobi_kernel<8, 2, true, 1>);
                case 2: return go(erank_jacobi_kernel<8, 4, true, 1>);
                case 4: return go(erank_jacobi_kernel<8, 8, true, 1>);
                default: break;
            }
        }
        switch (R / 64) {
            case 1: return go(erank_jacobi_kernel<16, 1, true, 1>);
            case 2: return go(erank_jacobi_kernel<16, 2, true, 1>);
            case 4: return go(erank_jacobi_kernel<16, 4, true, 1>);
            case 8: return go(erank_jacobi_kernel<16, 8, true, 1>);
            default: break;
        }
    }
    if (npairs > 16) {                          // 16 lanes per pair: 64 pairs per pass
        if (s4 <= 16) return erank_launch<16, 1>(a, batch, lds, st);
        if (s4 <= 32) return erank_launch<16, 2>(a, batch, lds, st);
        if (s4 <= 64) return erank_launch<16, 4>(a, batch, lds, st);
        if (s4 <= 128) return erank_launch<16, 8>(a, batch, lds, st);
        return erank_launch<16, 0>(a, batch, lds, st);
    }
    if (s4 <= 64) return erank_launch<64, 1>(a, batch, lds, st);
    if (s4 <= 128) return erank_launch<64, 2>(a, batch, lds, st);
    if (s4 <= 256) return erank_launch<64, 4>(a, batch, lds, st);
    if (s4 <= 512) return erank_launch<64, 8>(a, batch, lds, st);
    return erank_launch<64, 0>(a, batch, lds, st);
}

/* Block size (columns per block) of the two-level Jacobi for column length R: two blocks must fit one CU's LDS. */
static int erank_blk_b(int R) {
    int b = 16;
    while (b > 2 && ((int64_t)2 * b * erank_rp(R) + 2 * b) * 4 > 150 * 1024) b >>= 1;
    return b;
}

/* Sizes for r3d_erank_blocked: out[0] = floats of af_t ([Cpad][Rp], Cpad = C rounded up to the block size, Rp = R
 * rounded up to 4), out[1] = ints of ctrl, out[2] = Rp (row stride of af_t), out[3] = block size b.
 * Returns R3D_EINVAL when even two 2-column blocks exceed the LDS (R > ~9500). */
R3D_EXPORT int r3d_erank_blocked_sizes(int R, int C, int max_sweeps, int64_t* out) {
    R3D_REQUIRE(out && R > 0 && C > 0);
    const int b = erank_blk_b(R);
    R3D_REQUIRE(((int64_t)2 * b * erank_rp(R) + 2 * b) * 4 <= 150 * 1024);
    const int nreal = r3d_cdiv(C, b);
    out[0] = (int64_t)nreal * b * erank_rp(R);
    out[1] = 4 + (max_sweeps > 0 ? max_sweeps : 16);
    out[2] = erank_rp(R);
    out[3] = b;
    return R3D_OK;
}

template <int NCH, bool EXACT>
static int erank_blk_sweeps2(const ErankBlk& g, int ms, int64_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)erank_blk_round_kernel<NCH, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int b = g.b;
    const int tw = 64 * (b / 2 < 1 ? 1 : (b / 2 > 16 ? 16 : b / 2)), tc = 64 * (b > 16 ? 16 : b);
    for (int s = 0; s < ms; ++s) {
        hipLaunchKernelGGL((erank_blk_round_kernel<NCH, EXACT>), dim3(g.nreal), dim3(tw), (size_t)lds, st, g, 0, 0, s);
        if (g.nblk > 1)
            for (int rd = 0; rd < g.nblk - 1; ++rd)
                hipLaunchKernelGGL((erank_blk_round_kernel<NCH, EXACT>), dim3(g.nblk / 2), dim3(tc), (size_t)lds, st, g, 1, rd, s);
        hipLaunchKernelGGL(erank_blk_sweep_end_kernel, dim3(1), dim3(1), 0, st, g, s);
    }
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

template <int NCH>
static int erank_blk_sweeps(const ErankBlk& g, int ms, int64_t lds, hipStream_t st) {
    if (NCH > 0 && (g.Rp >> 2) == 64 * NCH) return erank_blk_sweeps2<NCH, true>(g, ms, lds, st);
    return erank_blk_sweeps2<NCH, false>(g, ms, lds, st);
}

/* Effective rank of ONE matrix X[R, C] (row-major, leading dimension ld) of any size: two-level one-sided Jacobi
 * with the columns in HBM (see above).  af_t [Cpad][Rp] receives the rotated columns (X V)^T (rows >= C and the
 * columns R..Rp-1 of every row are zero padding); ctrl is integer scratch; sizes from r3d_erank_blocked_sizes.
 * sigma [C], stats [4] as r3d_erank_jacobi.  Enqueues 3 + max_sweeps * (nblk + 1) launches on the stream (default
 * max_sweeps 16 -- 8 to 10 are used at the BASELINE shapes; launches after convergence return at once but still cost ~3 us each), never synchronises. */
R3D_EXPORT int r3d_erank_blocked_t(const float* x, int ld, int x_transposed, int R, int C, float* sigma, float* af_t,
                                   int* ctrl, float* stats, int max_sweeps, void* stream);
R3D_EXPORT int r3d_erank_blocked(const float* x, int ld, int R, int C, float* sigma, float* af_t, int* ctrl, float* stats,
                                 int max_sweeps, void* stream) {
    return r3d_erank_blocked_t(x, ld, 0, R, C, sigma, af_t, ctrl, stats, max_sweeps, stream);
}

/* r3d_erank_blocked with the input optionally given transposed: x_transposed != 0 -> x is [C][ld] and holds X^T (row c
 * = column c of the [R, C] matrix that is decomposed).  The training step uses it to decompose fused^T when the token
 * matrix has fewer rows than columns (the Jacobi works on the orientation with the fewer columns). */
R3D_EXPORT int r3d_erank_blocked_t(const float* x, int ld, int x_transposed, int R, int C, float* sigma, float* af_t,
                                   int* ctrl, float* stats, int max_sweeps, void* stream) {
    R3D_REQUIRE(x && sigma && af_t && ctrl && stats && R > 0 && C > 0 && ld >= (x_transposed ? R : C));
    R3D_REQUIRE(r3d_aligned16(af_t));
    const int b = erank_blk_b(R);
    const int Rp = erank_rp(R);
    const int64_t lds = ((int64_t)2 * b * Rp + 2 * b) * 4;
    R3D_REQUIRE(lds <= 150 * 1024);
    const int ms = max_sweeps > 0 ? max_sweeps : 16;
    const int nreal = r3d_cdiv(C, b);
    const int nblk = (nreal + 1) & ~1;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ctrl, 0, sizeof(int) * (4 + ms), st);
    if (e != hipSuccess) return (int)e;
    ErankBlk g{af_t, R, Rp, C, b, nblk, nreal, ctrl};
    hipLaunchKernelGGL(erank_blk_init_kernel, dim3(r3d_cdiv(nreal * b, 32), r3d_cdiv(Rp, 32)), dim3(256), 0, st, x, ld, g, x_transposed ? 1 : 0);
    const int rp4 = Rp / 4;
    int rc;
    if (rp4 <= 64) rc = erank_blk_sweeps<1>(g, ms, lds, st);
    else if (rp4 <= 128) rc = erank_blk_sweeps<2>(g, ms, lds, st);
    else if (rp4 <= 256) rc = erank_blk_sweeps<4>(g, ms, lds, st);
    else if (rp4 <= 512) rc = erank_blk_sweeps<8>(g, ms, lds, st);
    else rc = erank_blk_sweeps<0>(g, ms, lds, st);
    if (rc != R3D_OK) return rc;
    hipLaunchKernelGGL(erank_blk_sigma_kernel, dim3(r3d_cdiv(C, kJacThreads / 64)), dim3(kJacThreads), 0, st, g, sigma);
    hipLaunchKernelGGL(erank_blk_stats_kernel, dim3(1), dim3(kJacThreads), 0, st, g, (const float*)sigma, stats);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* coef[k] = (*gout) * (d erank / d sigma_k) / sigma_k^3  for the backward  dX = Af diag(coef) (Af^T X). */
R3D_EXPORT int r3d_erank_bwd_coef(const float* sigma, const float* stats, const float* gout, float* coef, int C,
                                  int max_rank, void* stream) {
    R3D_REQUIRE(sigma && stats && coef && C > 0);
    hipLaunchKernelGGL(erank_coef_kernel, dim3(r3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sigma, stats, gout,
                       coef, C, max_rank);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* The backward in its well-conditioned form.  With U = (X V) Sigma^-1 (the rotated columns, normalised) the gradient of a
 * function of the singular values is U diag(g) V^T; the sweep does not carry V, and V^T = Sigma^-1 U^T X picks up, for a
 * small sigma_i, the residual coupling e_ij of column i with every large column j amplified by sigma_j / sigma_i (the
 * two-GEMM form Af diag(g / sigma^3) Af^T X has exactly that error: 0.4 of the gradient's scale with LAPACK's fp32
 * vectors at the BASELINE shapes).  One Neumann term of (U^T U)^-1 removes it:  V^T = Sigma^-1 (2 I - U^T U) U^T X up to
 * e^2 sigma_j / sigma_i.  The host composes: scale_rows(af_t, inv) -> W = U^T X -> G = U^T U -> P = G W ->
 * r3d_erank_bwd_fix: W <- diag(cg) (2 W - P) -> dX = U W.  r3d_erank_bwd_coef2 fills cg[k] = gout * (d erank / d sigma_k)
 * / sigma_k and inv[k] = 1 / sigma_k (both zero where r3d_erank_bwd_coef's coefficient is). */
R3D_EXPORT int r3d_erank_bwd_coef2(const float* sigma, const float* stats, const float* gout, float* cg, float* inv, int C,
                                   int max_rank, void* stream) {
    R3D_REQUIRE(sigma && stats && cg && inv && C > 0);
    hipLaunchKernelGGL(erank_coef2_kernel, dim3(r3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sigma, stats, gout, cg,
                       inv, C, max_rank);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* w[r, :] = cg[r] * (2 w[r, :] - p[r, :]); w and p dense [rows, cols]. */
R3D_EXPORT int r3d_erank_bwd_fix(float* w, const float* p, const float* cg, int rows, int cols, void* stream) {
    R3D_REQUIRE(w && p && cg && rows > 0 && cols > 0);
    const size_t total = (size_t)rows * cols;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(erank_bwd_fix_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, p, cg, rows, cols);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* x[r, :] *= coef[r] */
R3D_EXPORT int r3d_scale_rows(float* x, int ld, int rows, int cols, const float* coef, void* stream) {
    R3D_REQUIRE(x && coef && rows > 0 && cols > 0 && ld >= cols);
    const size_t total = (size_t)rows * cols;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, coef);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* vt <- 1.5 vt_raw - 0.5 gv (n elements): the Newton-Schulz orthogonality polish of the warm-start basis. */
R3D_EXPORT int r3d_erank_vt_polish(const float* vt_raw, const float* gv, float* vt, int64_t n, void* stream) {
    R3D_REQUIRE(vt_raw && gv && vt && n > 0);
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(erank_vt_polish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, vt_raw, gv, vt, (size_t)n);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
