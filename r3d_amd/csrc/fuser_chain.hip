// The SA-Fuser block and everything row-local around it as ONE launch per direction (hidden = 128).
//
// Between the embedding seam (embed.hip) and the decoder's cross attention every operation of the step is local to a
// frame's two modality tokens: V projection + pair swap (the closed form of the masked 2-token attention),
// attn.proj + residual, norm2, fc1 + GELU, fc2 + residuals, fuser.norm, the mean over the two tokens
// (model/extras/transformerblock.py:118-135, model/futr_safuser_tokenfusion.py:83-94), then the segmentation head and
// the layer-0 key/value projection of the decoder on the fused row (model/futr_safuser_tokenfusion.py:228-232,
// model/extras/transformer.py:300-302).  As separate launches that chain is 5 dependent latency-bound launches
// (~37 us of the 245 us step, MFMA utilisation 1-2 %); here a workgroup owns 16 complete token rows (8 frames), keeps the
// activations in LDS from stage to stage and streams every weight matrix through LDS exactly once:
//
//   wave w of 8 computes the 16 x 16 output tiles {w, w + 8, ...} of every stage with v_mfma_f32_16x16x4_f32 (exact
//   fp32); a tile's 16 weight rows travel global -> registers -> a wave-private LDS region -> MFMA operand layout in
//   128-deep chunks (the product loop of gemm_ln.hip), and the chunk pipeline (two register stages ahead of the chunk
//   being multiplied) runs ACROSS the stage boundaries: the next stage's weights are in flight while the current stage's
//   epilogue, LayerNorm reductions and barriers execute -- weights depend on nothing in the chain.
//
// The backward kernel mirrors it (fuser.norm backward -> fc2 / GELU' / fc1 input gradients -> norm2 backward -> attn.proj
// and V input gradients -> norm1 backward -> token-exchange backward -> depth LayerNorm + ReLU backward), with the
// decoder's memory-side input gradients (key/value projection, segmentation head) as its first stage.
//
// A second ROLE of the same launches (extra workgroups, 16 query rows = 2 clips each) runs the layer-0 query
// self-attention sub-layer (model/extras/transformer.py:289-293,300): with tgt = 0 it depends on parameters and dropout
// masks only, so it needs no launches of its own.
//
// Intermediates the weight-gradient launch and the backward need (vsw, x1, h2, u, f1, x3, row statistics, ...) are stored
// from the accumulator layout as they are produced.
#include "chain_bf3.h"
#include "mha_small.h"
#include "../../include/r3d_hip.h"

namespace r3d {

constexpr int kFcBufH = 0;
constexpr int kFcBufV = kFcBufH + kFcRows * kFcP1;
constexpr int kFcBufF = kFcBufV + kFcRows * kFcP1;
constexpr int kFcWl = kFcBufF + kFcRows * kFcP4;
constexpr int kFcRed = kFcWl + 8 * 16 * kFcWP;
constexpr int kFcLdsFloats = kFcRed + 2 * 8 * kFcRows;
constexpr int kFcLdsBytes = kFcLdsFloats * 4;
static_assert(mha_small_lds_floats(16, 8) <= 16 * kFcWP, "the query role's attention units borrow the wave's weight region");
static_assert(mha_small_bwd_lds_floats(16, 8) <= 16 * kFcWP, "the query role's attention units borrow the wave's weight region");

typedef r3d_fuser_chain_fwd_args FcFwd;

// ---------------------------------------------------------------------------------------------------------------
// forward, fuser role
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fc_fwd_fuser(const FcFwd& A, const int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    float* bufH = lds + kFcBufH;
    float* bufV = lds + kFcBufV;
    float* bufF = lds + kFcBufF;
    float* wl = lds + kFcWl + wave * (16 * kFcWP);
    float (*red)[8][kFcRows] = reinterpret_cast<float (*)[8][kFcRows]>(lds + kFcRed);
    const int row0 = wg * kFcRows;
    const int c = wave * 16 + li;                         // this lane's column of a 128-wide stage
    constexpr int H = kFcH;
    const int nseg_t = (A.K + 15) >> 4;                   // stage-5 tiles: 16 key/value tiles, then the segmentation head's
    R3D_CHAIN_MARK(A.timeline, wg == 0, 0);
    const int nt5 = 16 + nseg_t;

    // ---- weight chunks 0, 1 and the A rows of stage 1
    FcW s0, s1;
    fc_wload(s0, A.wv + (size_t)(wave * 16) * H, H, 16, lane);                      // c0: V projection
    fc_wload(s1, A.wproj + (size_t)(wave * 16) * H, H, 16, lane);                   // c1: attn.proj
    const f32x4 a_in = *reinterpret_cast<const f32x4*>(A.h1 + (size_t)(row0 + (tid >> 5)) * H + 4 * (tid & 31));
    // ---- epilogue operands, all requested up front (they are as cold as the tiles)
    float x0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x0[i] = A.x0[(size_t)(row0 + 4 * q + i) * H + c];
    const float b_proj = A.bproj[c], g2 = A.g2[c], be2 = A.be2[c], b_2 = A.b2[c], gF = A.gf[c], beF = A.bef[c];
    float b_1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b_1[t] = A.b1[(4 * wave + t) * 16 + li];
    const int f0 = (row0 >> 1) + 2 * q;                   // the two frames whose token pairs this lane finishes
    const float pos0 = A.pos[(size_t)(f0 % A.S) * H + c], pos1 = A.pos[(size_t)((f0 + 1) % A.S) * H + c];
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(bufH + (tid >> 5) * kFcP1 + 4 * (tid & 31)) = a_in;
    fc_wstore(s0, wl, lane);
    fc_wload(s0, A.w1 + (size_t)((4 * wave + 0) * 16) * H, H, 16, lane);             // c2: fc1 tile 0
    __syncthreads();

    R3D_CHAIN_MARK(A.timeline, wg == 0, 1);
    const float* wr = wl + li * kFcWP + 4 * q;
    FcOp a;
    f32x4 acc0, acc1;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // ---- stage 1: V = h1 . Wv^T, stored pair-swapped (the OTHER modality token's value: attention == swap)
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fc_chunk(a, wr, acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s1, wl, lane);
    fc_wload(s1, A.w1 + (size_t)((4 * wave + 1) * 16) * H, H, 16, lane);             // c3
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = acc0[i] + acc1[i];
        const int rs = 4 * q + (i ^ 1);
        A.vsw[(size_t)(row0 + rs) * H + c] = v;
        bufV[rs * kFcP1 + c] = v;
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 2);
    // ---- stage 2: x1 = x0 + vsw . Wproj^T + b ; h2 = norm2(x1)
    fc_opload(a, bufV + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fc_chunk(a, wr, acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s0, wl, lane);
    fc_wload(s0, A.w1 + (size_t)((4 * wave + 2) * 16) * H, H, 16, lane);             // c4
    __builtin_amdgcn_sched_barrier(0);
    float x1[4], mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x1[i] = (acc0[i] + acc1[i]) + b_proj + x0[i];
        A.x1[(size_t)(row0 + 4 * q + i) * H + c] = x1[i];
    }
    fc_layernorm(x1, red, wave, li, q, mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = (x1[i] - mean[i]) * rstd[i] * g2 + be2;
        const int r = 4 * q + i;
        A.h2[(size_t)(row0 + r) * H + c] = h;
        bufH[r * kFcP1 + c] = h;                          // (stage 1's reads of bufH are behind two barriers)
        if (wave == 0 && li == 0) { A.m2[row0 + r] = mean[i]; A.r2[row0 + r] = rstd[i]; }
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 3);
    // ---- stage 3: u = h2 . W1^T + b1 ; f1 = GELU(u)   (4 column tiles per wave, A operand read once)
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        if (t < 2) R3D_CHAIN_MARK(A.timeline, wg == 0, 8 + 4 * t);
        fc_chunk(a, wr, acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        if (t < 2) R3D_CHAIN_MARK(A.timeline, wg == 0, 9 + 4 * t);
        // chunk t + 3 of the stream into the wave's region, chunk t + 5 requested
#if defined(R3D_FC_PROBE) && (R3D_FC_PROBE & 8)
        if (t == 0) { fc_wstore(s1, wl, lane); }
        if (t == 1) { fc_wstore(s0, wl, lane); }
        if (t == 2) { fc_wstore(s1, wl, lane); }
        if (t == 3) { fc_wstore(s0, wl, lane); }
#elif defined(R3D_FC_PROBE) && (R3D_FC_PROBE & 16)
        if (t == 0) { fc_wload(s1, A.w1 + (size_t)((4 * wave + 3) * 16) * H, H, 16, lane); }       // c5
        if (t == 1) { fc_wload(s0, A.w2 + (size_t)(wave * 16) * (4 * H) + 0, 4 * H, 16, lane); }   // c6
        if (t == 2) { fc_wload(s1, A.w2 + (size_t)(wave * 16) * (4 * H) + 128, 4 * H, 16, lane); } // c7
        if (t == 3) { fc_wload(s0, A.w2 + (size_t)(wave * 16) * (4 * H) + 256, 4 * H, 16, lane); } // c8
#else
        if (t == 0) { fc_wstore(s1, wl, lane); fc_wload(s1, A.w1 + (size_t)((4 * wave + 3) * 16) * H, H, 16, lane); }       // c5
        if (t == 1) { fc_wstore(s0, wl, lane); fc_wload(s0, A.w2 + (size_t)(wave * 16) * (4 * H) + 0, 4 * H, 16, lane); }   // c6
        if (t == 2) { fc_wstore(s1, wl, lane); fc_wload(s1, A.w2 + (size_t)(wave * 16) * (4 * H) + 128, 4 * H, 16, lane); } // c7
        if (t == 3) { fc_wstore(s0, wl, lane); fc_wload(s0, A.w2 + (size_t)(wave * 16) * (4 * H) + 256, 4 * H, 16, lane); } // c8
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (t < 2) R3D_CHAIN_MARK(A.timeline, wg == 0, 10 + 4 * t);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u = (acc0[i] + acc1[i]) + b_1[t];
#if defined(R3D_FC_PROBE) && (R3D_FC_PROBE & 4)
            const float f = u;
#else
            const float f = gelu_f(u);
#endif
            const size_t o = (size_t)(row0 + 4 * q + i) * (4 * H) + cu;
#if !(defined(R3D_FC_PROBE) && (R3D_FC_PROBE & 1))
            A.u[o] = u;
            A.f1[o] = f;
#endif
#if !(defined(R3D_FC_PROBE) && (R3D_FC_PROBE & 2))
            bufF[(4 * q + i) * kFcP4 + cu] = f;
#else
            if (u == 123.456f) bufF[(4 * q + i) * kFcP4 + cu] = f;
#endif
        }
        if (t < 2) R3D_CHAIN_MARK(A.timeline, wg == 0, 11 + 4 * t);
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 4);
    // ---- stage 4: x3 = x1 (+ x0) + f1 . W2^T + b2 ; y = fuser.norm(x3) ; fused = mean over the token pair
    acc0 = zero; acc1 = zero;
    {
        const float* ar = bufF + li * kFcP4 + 4 * q;
        fc_opload(a, ar);
        fc_chunk(a, wr, acc0, acc1);                                                    // c6
        __builtin_amdgcn_sched_barrier(0);
        fc_wstore(s1, wl, lane);
        fc_wload(s1, A.w2 + (size_t)(wave * 16) * (4 * H) + 384, 4 * H, 16, lane);     // c9
        __builtin_amdgcn_sched_barrier(0);
        fc_opload(a, ar + 128);
        fc_chunk(a, wr, acc0, acc1);                                                    // c7
        __builtin_amdgcn_sched_barrier(0);
        fc_wstore(s0, wl, lane);
        {                                                                               // c10: stage-5 tile `wave`
            const int t5 = wave;                                                        // (< 16: a key/value tile)
            fc_wload(s0, A.wkv + (size_t)(t5 * 16) * H, H, 16, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        fc_opload(a, ar + 256);
        fc_chunk(a, wr, acc0, acc1);                                                    // c8
        __builtin_amdgcn_sched_barrier(0);
        fc_wstore(s1, wl, lane);
        {                                                                               // c11: stage-5 tile wave + 8
            const int t5 = wave + 8;
            fc_wload(s1, A.wkv + (size_t)(t5 * 16) * H, H, 16, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        fc_opload(a, ar + 384);
        fc_chunk(a, wr, acc0, acc1);                                                    // c9
        __builtin_amdgcn_sched_barrier(0);
    }
    fc_wstore(s0, wl, lane);                                                            // c10 in the region
    const int t5c = wave + 16;                                                          // c12: a segmentation-head tile
    const bool has12 = t5c < nt5;
    {
        const int ts = has12 ? t5c - 16 : 0;
        const int nv = A.K - 16 * ts;
        fc_wload(s0, A.wseg + (size_t)(ts * 16) * H, H, nv < 16 ? nv : 16, lane);
    }
    const float bkv0 = A.bkv[wave * 16 + li], bkv1 = A.bkv[(wave + 8) * 16 + li];
    const int cseg = (has12 ? t5c - 16 : 0) * 16 + li;
    const float bsg = A.bseg[cseg < A.K ? cseg : A.K - 1];
    __builtin_amdgcn_sched_barrier(0);
    R3D_CHAIN_MARK(A.timeline, wg == 0, 5);
    float x3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x3[i] = (acc0[i] + acc1[i]) + b_2 + x1[i] + (A.add_xres ? x0[i] : 0.f);
        A.x3[(size_t)(row0 + 4 * q + i) * H + c] = x3[i];
    }
    fc_layernorm(x3, red, wave, li, q, mean, rstd);
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y[i] = (x3[i] - mean[i]) * rstd[i] * gF + beF;
        if (A.y) A.y[(size_t)(row0 + 4 * q + i) * H + c] = y[i];
        if (wave == 0 && li == 0) { A.mf[row0 + 4 * q + i] = mean[i]; A.rf[row0 + 4 * q + i] = rstd[i]; }
    }
    {
        const float fu0 = (y[0] + y[1]) * 0.5f, fu1 = (y[2] + y[3]) * 0.5f;
        A.fused[(size_t)f0 * H + c] = fu0;
        A.fused[(size_t)(f0 + 1) * H + c] = fu1;
        // stage-5 A operand: rows 0-7 = fused (segmentation head), rows 8-15 = fused + pos (key = value = memory + pos)
        bufV[(2 * q) * kFcP1 + c] = fu0;
        bufV[(2 * q + 1) * kFcP1 + c] = fu1;
        bufV[(8 + 2 * q) * kFcP1 + c] = fu0 + pos0;
        bufV[(8 + 2 * q + 1) * kFcP1 + c] = fu1 + pos1;
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 6);
    // ---- stage 5: cakv = (fused + pos) . Wkv^T + b (output rows 8-15) ; seg = fused . Wseg^T + b (output rows 0-7)
    fc_opload(a, bufV + li * kFcP1 + 4 * q);
    const int fr = (row0 >> 1) + ((4 * q) & 7);            // frame of this lane's first output row in either half
    acc0 = zero; acc1 = zero;
    fc_chunk(a, wr, acc0, acc1);                                                        // c10
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s1, wl, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (q >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A.cakv[(size_t)(fr + i) * (2 * H) + wave * 16 + li] = (acc0[i] + acc1[i]) + bkv0;
    }
    acc0 = zero; acc1 = zero;
    fc_chunk(a, wr, acc0, acc1);                                                        // c11
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s0, wl, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (q >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A.cakv[(size_t)(fr + i) * (2 * H) + (wave + 8) * 16 + li] = (acc0[i] + acc1[i]) + bkv1;
    }
    if (has12) {                                           // (wave-uniform)
        acc0 = zero; acc1 = zero;
        fc_chunk(a, wr, acc0, acc1);                                                    // c12
        if (q < 2 && cseg < A.K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) A.seg[(size_t)(fr + i) * A.K + cseg] = (acc0[i] + acc1[i]) + bsg;
        }
    }
    R3D_CHAIN_MARK(A.timeline, wg == 0, 7);
    // segmentation heads with more than 128 classes would need tiles beyond wave + 16: refused by the host
}

// ---------------------------------------------------------------------------------------------------------------
// forward, query role: self-attention sub-layer of decoder layer 0 on tgt = 0 (transformer.py:289-293) and the
// cross-attention's query projection (:300); 16 rows = 2 clips x 8 queries per workgroup
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fc_fwd_query(const FcFwd& A, const int wgq, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    float* bufH = lds + kFcBufH;
    float* bufV = lds + kFcBufV;
    float* wl = lds + kFcWl + wave * (16 * kFcWP);
    float (*red)[8][kFcRows] = reinterpret_cast<float (*)[8][kFcRows]>(lds + kFcRed);
    constexpr int H = kFcH;
    const int row0 = wgq * kFcRows;
    const int c = wave * 16 + li;
    R3D_CHAIN_MARK(A.timeline, wgq == 0, 16);
    FcW s0, s1;
    fc_wload(s0, A.w_in + (size_t)(wave * 16) * H, H, 16, lane);                      // c0..c2: in_proj tiles w, w+8, w+16
    fc_wload(s1, A.w_in + (size_t)((wave + 8) * 16) * H, H, 16, lane);
    // A rows: tgt (= 0) + query_pos, the same 8 rows for every clip
    const f32x4 a_in = *reinterpret_cast<const f32x4*>(A.qpos + (size_t)((tid >> 5) & 7) * H + 4 * (tid & 31));
    float b_in[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) b_in[t] = A.b_in[(wave + 8 * t) * 16 + li];
    const float b_out = A.b_out[c], g1 = A.g1[c], be1 = A.be1[c], b_q = A.bq[c];
    const FcMaskSrc md1(A.drop_d1, A.g1, H);
    uint8_t kb[4];
    float qp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        kb[i] = md1.raw(r, c);
        qp[i] = A.qpos[(size_t)((4 * q + i) & 7) * H + c];
    }
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(bufH + (tid >> 5) * kFcP1 + 4 * (tid & 31)) = a_in;
    fc_wstore(s0, wl, lane);
    fc_wload(s0, A.w_in + (size_t)((wave + 16) * 16) * H, H, 16, lane);
    __syncthreads();
    const float* wr = wl + li * kFcWP + 4 * q;
    FcOp a;
    f32x4 acc0, acc1;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        acc0 = zero; acc1 = zero;
        fc_chunk(a, wr, acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) { fc_wstore(s1, wl, lane); fc_wload(s1, A.w_out + (size_t)(wave * 16) * H, H, 16, lane); }     // c3
        if (t == 1) { fc_wstore(s0, wl, lane); fc_wload(s0, A.wq + (size_t)(wave * 16) * H, H, 16, lane); }        // c4
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            A.sa_qkv[(size_t)(row0 + 4 * q + i) * (3 * H) + (wave + 8 * t) * 16 + li] = (acc0[i] + acc1[i]) + b_in[t];
    }
    R3D_CHAIN_MARK(A.timeline, wgq == 0, 17);
    __syncthreads();                                        // q / k / v rows of both clips are written (workgroup scope)
    // ---- attention core: (clip, head) units, head = wave
    {
        MhaArgs m{};
        m.q = A.sa_qkv; m.ldq = 3 * H; m.k = A.sa_qkv + H; m.ldk = 3 * H; m.v = A.sa_qkv + 2 * H; m.ldv = 3 * H;
        m.probs = A.p_sa; m.drop = A.drop_sa; m.drop_scale = A.drop_scale; m.o = A.sa_o; m.ldo = H;
        m.B = A.B; m.heads = 8; m.Lq = 8; m.Lk = 8; m.dh = 16; m.scale = 0.25f;
        mha_fwd_small_unit<16, 8, false>(m, (2 * wgq) * 8 + wave, wl);
        __builtin_amdgcn_wave_barrier();
        mha_fwd_small_unit<16, 8, false>(m, (2 * wgq + 1) * 8 + wave, wl);
    }
    __syncthreads();
    {
        const f32x4 so = *reinterpret_cast<const f32x4*>(A.sa_o + (size_t)(row0 + (tid >> 5)) * H + 4 * (tid & 31));
        *reinterpret_cast<f32x4*>(bufV + (tid >> 5) * kFcP1 + 4 * (tid & 31)) = so;
    }
    R3D_CHAIN_MARK(A.timeline, wgq == 0, 18);
    fc_wstore(s1, wl, lane);                                // c3 (the attention units are done with the region)
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wgq == 0, 19);
    // ---- out_proj -> dropout -> (+ tgt = 0) -> norm1
    fc_opload(a, bufV + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fc_chunk(a, wr, acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s0, wl, lane);                                // c4
    __builtin_amdgcn_sched_barrier(0);
    float t1p[4], mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t1p[i] = ((acc0[i] + acc1[i]) + b_out) * md1.keep(kb[i], A.drop_scale);
        A.t1_pre[(size_t)(row0 + 4 * q + i) * H + c] = t1p[i];
    }
    fc_layernorm(t1p, red, wave, li, q, mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * q + i;
        const float t1 = (t1p[i] - mean[i]) * rstd[i] * g1 + be1;
        A.t1[(size_t)(row0 + r) * H + c] = t1;
        bufH[r * kFcP1 + c] = t1 + qp[i];
        if (wave == 0 && li == 0) { A.m1[row0 + r] = mean[i]; A.r1[row0 + r] = rstd[i]; }
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wgq == 0, 20);
    // ---- cross-attention query projection: caq = (t1 + query_pos) . Wq^T + b
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fc_chunk(a, wr, acc0, acc1);
#pragma unroll
    for (int i = 0; i < 4; ++i) A.caq[(size_t)(row0 + 4 * q + i) * H + c] = (acc0[i] + acc1[i]) + b_q;
    R3D_CHAIN_MARK(A.timeline, wgq == 0, 21);
}

// ---------------------------------------------------------------------------------------------------------------
// forward, fuser role on the bf16 matrix cores (chain_bf3.h): the same stages, weights from the operand-order planes
// (A.pl_*), activations handed from stage to stage as three bf16 planes in LDS
// ---------------------------------------------------------------------------------------------------------------
constexpr int kF3P1 = kFcH + 8;                              // bf16 pitch of a [16][128] image
constexpr int kF3P4 = 4 * kFcH + 8;                          // ... of the [16][512] one
constexpr int kF3ImgH = 0;                                   // (offsets in bf16 elements)
constexpr int kF3ImgV = kF3ImgH + 3 * 16 * kF3P1;
constexpr int kF3ImgF = kF3ImgV + 3 * 16 * kF3P1;
constexpr int kF3RedBytes = (kF3ImgF + 3 * 16 * kF3P4) * 2;
constexpr int kF3LdsBytes = kF3RedBytes + 2 * 2 * 8 * kFcRows * 4;

__device__ __forceinline__ void fc3_fwd_fuser(const FcFwd& A, const int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    unsigned short* img = reinterpret_cast<unsigned short*>(lds);
    unsigned short* imgH = img + kF3ImgH;
    unsigned short* imgV = img + kF3ImgV;
    unsigned short* imgF = img + kF3ImgF;
    float (*red)[8][kFcRows] = reinterpret_cast<float (*)[8][kFcRows]>(reinterpret_cast<unsigned char*>(lds) + kF3RedBytes);
    const int row0 = wg * kFcRows;
    const int c = wave * 16 + li;
    constexpr int H = kFcH;
    const int nseg_t = (A.K + 15) >> 4;
    const bool has12 = wave < nseg_t;
    R3D_CHAIN_MARK(A.timeline, wg == 0, 0);
    // ---- weight chunks c0..c2 (chunk c lives in register set c % 3; a set is refilled with chunk c + 3 once c is multiplied)
    Bf3B b0, b1, b2;
    bf3_bload<4>(b0, A.pl_wv, 4, wave, 0, lane);                                       // c0: V projection, tile w
    bf3_bload<4>(b1, A.pl_wproj, 4, wave, 0, lane);                                    // c1: attn.proj
    bf3_bload<4>(b2, A.pl_w1, 4, 4 * wave + 0, 0, lane);                               // c2: fc1 tile 4w
    const f32x4 a_in = *reinterpret_cast<const f32x4*>(A.h1 + (size_t)(row0 + (tid >> 5)) * H + 4 * (tid & 31));
    float x0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x0[i] = A.x0[(size_t)(row0 + 4 * q + i) * H + c];
    const float b_proj = A.bproj[c], g2 = A.g2[c], be2 = A.be2[c], b_2 = A.b2[c], gF = A.gf[c], beF = A.bef[c];
    float b_1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b_1[t] = A.b1[(4 * wave + t) * 16 + li];
    const int f0 = (row0 >> 1) + 2 * q;
    const float pos0 = A.pos[(size_t)(f0 % A.S) * H + c], pos1 = A.pos[(size_t)((f0 + 1) % A.S) * H + c];
    const float bkv0 = A.bkv[wave * 16 + li], bkv1 = A.bkv[(wave + 8) * 16 + li];
    const int cseg = (has12 ? wave : 0) * 16 + li;
    const float bsg = A.bseg[cseg < A.K ? cseg : A.K - 1];
    __builtin_amdgcn_sched_barrier(0);
    bf3_stage_tile(imgH, kF3P1, 0, a_in, tid);
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 1);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0 = zero, acc1 = zero;
    // ---- stage 1: V = h1 . Wv^T, stored pair-swapped
    bf3_chunk<4>(imgH, kF3P1, li, q, 0, b0, acc0, acc1);                               // c0
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b0, A.pl_w1, 4, 4 * wave + 1, 0, lane);                               // c3
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = acc0[i] + acc1[i];
        const int rs = 4 * q + (i ^ 1);
        A.vsw[(size_t)(row0 + rs) * H + c] = v;
        bf3_store1(imgV, kF3P1, rs, c, v);
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 2);
    // ---- stage 2: x1 = x0 + vsw . Wproj^T + b ; h2 = norm2(x1)
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgV, kF3P1, li, q, 0, b1, acc0, acc1);                               // c1
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b1, A.pl_w1, 4, 4 * wave + 2, 0, lane);                               // c4
    __builtin_amdgcn_sched_barrier(0);
    float x1[4], mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x1[i] = (acc0[i] + acc1[i]) + b_proj + x0[i];
        A.x1[(size_t)(row0 + 4 * q + i) * H + c] = x1[i];
    }
    fc_layernorm(x1, red, wave, li, q, mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = (x1[i] - mean[i]) * rstd[i] * g2 + be2;
        const int r = 4 * q + i;
        A.h2[(size_t)(row0 + r) * H + c] = h;
        bf3_store1(imgH, kF3P1, r, c, h);
        if (wave == 0 && li == 0) { A.m2[row0 + r] = mean[i]; A.r2[row0 + r] = rstd[i]; }
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 3);
    // ---- stage 3: u = h2 . W1^T + b1 ; f1 = GELU(u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        if (t == 0) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b2, acc0, acc1);               // c2
        if (t == 1) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b0, acc0, acc1);               // c3
        if (t == 2) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b1, acc0, acc1);               // c4
        if (t == 3) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b2, acc0, acc1);               // c5
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) bf3_bload<4>(b2, A.pl_w1, 4, 4 * wave + 3, 0, lane);               // c5
        if (t == 1) bf3_bload<4>(b0, A.pl_w2, 16, wave, 0, lane);                      // c6: fc2, k-steps 0..3
        if (t == 2) bf3_bload<4>(b1, A.pl_w2, 16, wave, 4, lane);                      // c7
        if (t == 3) bf3_bload<4>(b2, A.pl_w2, 16, wave, 8, lane);                      // c8
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u = (acc0[i] + acc1[i]) + b_1[t];
            const float f = gelu_f(u);
            const size_t o = (size_t)(row0 + 4 * q + i) * (4 * H) + cu;
            A.u[o] = u;
            A.f1[o] = f;
            bf3_store1(imgF, kF3P4, 4 * q + i, cu, f);
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 4);
    // ---- stage 4: x3 = x1 (+ x0) + f1 . W2^T + b2 ; y = fuser.norm(x3) ; fused = mean over the token pair
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgF, kF3P4, li, q, 0, b0, acc0, acc1);                               // c6
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b0, A.pl_w2, 16, wave, 12, lane);                                     // c9
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 4, b1, acc0, acc1);                               // c7
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b1, A.pl_wkv, 4, wave, 0, lane);                                      // c10
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 8, b2, acc0, acc1);                               // c8
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b2, A.pl_wkv, 4, wave + 8, 0, lane);                                  // c11
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 12, b0, acc0, acc1);                              // c9
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b0, A.pl_wseg, 4, has12 ? wave : 0, 0, lane);                         // c12
    __builtin_amdgcn_sched_barrier(0);
    R3D_CHAIN_MARK(A.timeline, wg == 0, 5);
    float x3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x3[i] = (acc0[i] + acc1[i]) + b_2 + x1[i] + (A.add_xres ? x0[i] : 0.f);
        A.x3[(size_t)(row0 + 4 * q + i) * H + c] = x3[i];
    }
    fc_layernorm(x3, red, wave, li, q, mean, rstd);
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y[i] = (x3[i] - mean[i]) * rstd[i] * gF + beF;
        if (A.y) A.y[(size_t)(row0 + 4 * q + i) * H + c] = y[i];
        if (wave == 0 && li == 0) { A.mf[row0 + 4 * q + i] = mean[i]; A.rf[row0 + 4 * q + i] = rstd[i]; }
    }
    {
        const float fu0 = (y[0] + y[1]) * 0.5f, fu1 = (y[2] + y[3]) * 0.5f;
        A.fused[(size_t)f0 * H + c] = fu0;
        A.fused[(size_t)(f0 + 1) * H + c] = fu1;
        bf3_store1(imgV, kF3P1, 2 * q, c, fu0);
        bf3_store1(imgV, kF3P1, 2 * q + 1, c, fu1);
        bf3_store1(imgV, kF3P1, 8 + 2 * q, c, fu0 + pos0);
        bf3_store1(imgV, kF3P1, 8 + 2 * q + 1, c, fu1 + pos1);
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 6);
    // ---- stage 5: cakv (output rows 8-15) ; seg (output rows 0-7)
    const int fr = (row0 >> 1) + ((4 * q) & 7);
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgV, kF3P1, li, q, 0, b1, acc0, acc1);                               // c10
    if (q >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A.cakv[(size_t)(fr + i) * (2 * H) + wave * 16 + li] = (acc0[i] + acc1[i]) + bkv0;
    }
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgV, kF3P1, li, q, 0, b2, acc0, acc1);                               // c11
    if (q >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A.cakv[(size_t)(fr + i) * (2 * H) + (wave + 8) * 16 + li] = (acc0[i] + acc1[i]) + bkv1;
    }
    if (has12) {
        acc0 = zero; acc1 = zero;
        bf3_chunk<4>(imgV, kF3P1, li, q, 0, b0, acc0, acc1);                           // c12
        if (q < 2 && cseg < A.K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) A.seg[(size_t)(fr + i) * A.K + cseg] = (acc0[i] + acc1[i]) + bsg;
        }
    }
    R3D_CHAIN_MARK(A.timeline, wg == 0, 7);
}

__global__ __launch_bounds__(512) void fuser_chain_fwd_bf3_kernel(const FcFwd A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    const int nf = (2 * A.N) / kFcRows;
    if ((int)blockIdx.x < nf) fc3_fwd_fuser(A, (int)blockIdx.x, lds);
    else fc_fwd_query(A, (int)blockIdx.x - nf, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// operand-order bf16x3 planes of the chain weights (chain_bf3.h): one wave per (tile, k-step) block of one job
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_planes_kernel(const r3d_plane_job* __restrict__ jobs, int njobs, int total) {
    weight_planes_block(jobs, njobs, total, (int)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

__global__ __launch_bounds__(512) void fuser_chain_fwd_kernel(const FcFwd A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // The two waves a SIMD hosts (w and w + 4) run the same stage sequence in lockstep and would want the matrix core, the
    // LDS and the VALU at the same moments; a fixed priority difference lets one burst its MFMAs while the other is in its
    // epilogue / staging, so the phases of the pair interleave instead of colliding.
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    const int nf = (2 * A.N) / kFcRows;
    if ((int)blockIdx.x < nf) fc_fwd_fuser(A, (int)blockIdx.x, lds);
    else fc_fwd_query(A, (int)blockIdx.x - nf, lds);
}


// ===============================================================================================================
// backward
// ===============================================================================================================
// Input-gradient products are NN (dX = dY . W, W as nn.Linear stores it: [out, in] = [k, n]): a tile's weight chunk is
// 128 k-rows x 16 columns, staged k-major in the wave's region (pitch 20 floats: the four k groups of an MFMA operand
// read land on disjoint banks) and read back with 32 scalar LDS loads in the same (k group, k) register layout the NT
// chunks use, so the MFMA sequence is shared.
constexpr int kFbWl = kFcBufF + kFcRows * kFcP4;
constexpr int kFbRed = kFbWl + 8 * 128 * kFbWP;
constexpr int kFbLdsFloats = kFbRed + 2 * 2 * 8 * kFcRows;
constexpr int kFbLdsBytes = kFbLdsFloats * 4;
static_assert(mha_small_bwd_lds_floats(16, 8) <= 128 * kFbWP, "attention units borrow the wave's weight region");
static_assert(kFbLdsBytes <= 160 * 1024, "LDS");

typedef r3d_fuser_chain_bwd_args FcBwd;

__device__ __forceinline__ void fc_bwd_fuser(const FcBwd& A, const int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    float* bufH = lds + kFcBufH;
    float* bufV = lds + kFcBufV;
    float* bufF = lds + kFcBufF;
    float* wl = lds + kFbWl + wave * (128 * kFbWP);
    float* redA = lds + kFbRed;
    float* redB = redA + 2 * 8 * kFcRows;
    constexpr int H = kFcH;
    const int row0 = wg * kFcRows;
    const int c = wave * 16 + li;
    const int n0 = wave * 16;                              // first column of this wave's tile in a 128-wide stage
    const int f0 = (row0 >> 1) + 2 * q;                    // the two frames of this lane's rows 4q .. 4q + 3
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    R3D_CHAIN_MARK(A.timeline, wg == 0, 0);
    // ---- weight chunks c0, c1 and the A operand of stage 0: row r <- [d_cakv[frame r >> 1] | d_seg[frame] | 0]
    FcW s0, s1;
    fb_wload(s0, A.wkv + n0, H, 128, lane);                                          // c0: Wkv k-rows 0..127
    fb_wload(s1, A.wkv + (size_t)128 * H + n0, H, 128, lane);                        // c1: Wkv k-rows 128..255
    {
        const int r = tid >> 5, c4 = tid & 31, fr = (row0 >> 1) + (r >> 1);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(A.d_cakv + (size_t)fr * (2 * H) + 4 * c4);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(A.d_cakv + (size_t)fr * (2 * H) + 128 + 4 * c4);
        f32x4 v2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * c4 + e;
            const float t = A.d_seg[(size_t)fr * A.K + (k < A.K ? k : A.K - 1)];
            v2[e] = k < A.K ? t : 0.f;
        }
        float* dst = bufF + r * kFcP4 + 4 * c4;
        *reinterpret_cast<f32x4*>(dst) = v0;
        *reinterpret_cast<f32x4*>(dst + 128) = v1;
        *reinterpret_cast<f32x4*>(dst + 256) = v2;
    }
    // ---- operands of the fuser.norm backward
    float x3v[4], mfv[4], rfv[4], dex[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        x3v[i] = A.x3[(size_t)r * H + c]; mfv[i] = A.mf[r]; rfv[i] = A.rf[r];
    }
    const float gF = A.gf[c];
    {
        const float* ex = A.d_extra ? A.d_extra : A.gf;     // (absent: reads something harmless, discarded)
        const int ld = A.d_extra ? H : 0;
        dex[0] = ex[(size_t)f0 * ld + c]; dex[1] = ex[(size_t)(f0 + 1) * ld + c];
        if (!A.d_extra) { dex[0] = 0.f; dex[1] = 0.f; }
    }
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);
    fb_wload(s0, A.wseg + n0, H, A.K, lane);                                          // c2: Wseg k-rows (K valid)
    __syncthreads();
    FcOp a;
    f32x4 acc0 = zero, acc1 = zero;
    const float* ar = bufF + li * kFcP4 + 4 * q;
    R3D_CHAIN_MARK(A.timeline, wg == 0, 1);
    // ---- stage 0: d(memory + pos) = d_cakv . Wkv  (kept: pos_embedding's gradient), then + d_seg . Wseg
    fc_opload(a, ar);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c0
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);
    fb_wload(s1, A.w2 + (4 * wave + 0) * 16, 4 * H, 128, lane);                       // c3: fc2 weight, column tile 4w
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 128);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c1
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);
    fb_wload(s0, A.w2 + (4 * wave + 1) * 16, 4 * H, 128, lane);                       // c4
    __builtin_amdgcn_sched_barrier(0);
    A.d_fused[(size_t)f0 * H + c] = acc0[0] + acc1[0];                                // rows 4q, 4q+1 are frame f0's copies
    A.d_fused[(size_t)(f0 + 1) * H + c] = acc0[2] + acc1[2];
    fc_opload(a, ar + 256);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c2
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);
    fb_wload(s1, A.w2 + (4 * wave + 2) * 16, 4 * H, 128, lane);                       // c5
    __builtin_amdgcn_sched_barrier(0);
    R3D_CHAIN_MARK(A.timeline, wg == 0, 2);
    // ---- stage 1: fuser.norm backward; each token row of a pair receives half of the frame's gradient (:94)
    float dx3[4];
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = 0.5f * ((acc0[i] + acc1[i]) + dex[i >> 1]);
            xh[i] = (x3v[i] - mfv[i]) * rfv[i];
            g[i] = d[i] * gF;
            gx[i] = g[i] * xh[i];
        }
        float* pp = A.part_nf + (size_t)(wg * 4 + q) * (2 * H);
        pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
        pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dx3[i] = rfv[i] * (g[i] - s1v[i] - xh[i] * s2v[i]);
            A.d_x3[(size_t)(row0 + 4 * q + i) * H + c] = dx3[i];
            bufH[(4 * q + i) * kFcP1 + c] = dx3[i];
        }
    }
    // operands of stage 2 (GELU') and of the norm2 backward
    float uv[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) uv[t][i] = A.u[(size_t)(row0 + 4 * q + i) * (4 * H) + (4 * wave + t) * 16 + li];
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 3);
    // ---- stage 2: d_u = (d_x3 . W2) * GELU'(u)
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        fb_chunk(a, wl, li, q, acc0, acc1);                                           // c3 + t
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) { fb_wstore(s0, wl, lane); fb_wload(s0, A.w2 + (4 * wave + 3) * 16, 4 * H, 128, lane); }        // c6
        if (t == 1) { fb_wstore(s1, wl, lane); fb_wload(s1, A.w1 + n0, H, 128, lane); }                             // c7
        if (t == 2) { fb_wstore(s0, wl, lane); fb_wload(s0, A.w1 + (size_t)128 * H + n0, H, 128, lane); }           // c8
        if (t == 3) { fb_wstore(s1, wl, lane); fb_wload(s1, A.w1 + (size_t)256 * H + n0, H, 128, lane); }           // c9
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float du = (acc0[i] + acc1[i]) * gelu_grad_f(uv[t][i]);
            A.d_u[(size_t)(row0 + 4 * q + i) * (4 * H) + cu] = du;
            bufF[(4 * q + i) * kFcP4 + cu] = du;
        }
    }
    float x1v[4], m2v[4], r2v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        x1v[i] = A.x1[(size_t)r * H + c]; m2v[i] = A.m2[r]; r2v[i] = A.r2[r];
    }
    const float g2 = A.g2[c];
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 4);
    // ---- stage 3: d_h2 = d_u . W1 ; norm2 backward ; d_x1 = that + d_x3 (the residual x3 = x1 + ...)
    acc0 = zero; acc1 = zero;
    fc_opload(a, ar);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c7
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);
    fb_wload(s0, A.w1 + (size_t)384 * H + n0, H, 128, lane);                          // c10
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 128);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c8
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);
    fb_wload(s1, A.wproj + n0, H, 128, lane);                                         // c11
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 256);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c9
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);
    fb_wload(s0, A.wv + n0, H, 128, lane);                                            // c12
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 384);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c10
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);                                                          // c11 in the region
    __builtin_amdgcn_sched_barrier(0);
    float dx1[4];
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = acc0[i] + acc1[i];
            if (A.d_h2) A.d_h2[(size_t)(row0 + 4 * q + i) * H + c] = d[i];
            xh[i] = (x1v[i] - m2v[i]) * r2v[i];
            g[i] = d[i] * g2;
            gx[i] = g[i] * xh[i];
        }
        float* pp = A.part_n2 + (size_t)(wg * 4 + q) * (2 * H);
        pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
        pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        fb_rowsum2<4>(g, gx, redB, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dx1[i] = r2v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]) + dx3[i];
            A.d_x1[(size_t)(row0 + 4 * q + i) * H + c] = dx1[i];
            bufV[(4 * q + i) * kFcP1 + c] = dx1[i];
        }
    }
    // operands of the seam's backward (norm1, embd_drop, exchange, depth LayerNorm + ReLU)
    const FcMaskSrc mx0(A.drop_x0, A.g1n, H);
    float x0v[4], m1v[4], r1v[4];
    uint8_t kb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        x0v[i] = A.x0[(size_t)r * H + c]; m1v[i] = A.m1[r]; r1v[i] = A.r1[r];
        kb[i] = mx0.raw(r, c);
    }
    const float g1n = A.g1n[c], mr = A.m_rgb[c], md = A.m_dep[c], gd = A.lnd_g[c], bd = A.lnd_b[c];
    float rgbv[2], dpv[2], mdv[2], rdv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rgbv[j] = A.rgb[(size_t)(f0 + j) * H + c]; dpv[j] = A.dep_pre[(size_t)(f0 + j) * H + c];
        mdv[j] = A.mean_d[f0 + j]; rdv[j] = A.rstd_d[f0 + j];
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 5);
    // ---- stage 4: d_vsw = d_x1 . Wproj, un-swapped into d_v (gradient of the V projection's output rows)
    fc_opload(a, bufV + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c11
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);                                                          // c12
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = acc0[i] + acc1[i];
        const int rs = 4 * q + (i ^ 1);
        A.d_v[(size_t)(row0 + rs) * H + c] = v;
        bufH[rs * kFcP1 + c] = v;
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 6);
    // ---- stage 5: d_h1 = d_v . Wv ; norm1 backward + both residual gradients ; embd_drop ; exchange ; depth LN + ReLU
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c12
    float gx0[4];
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = acc0[i] + acc1[i];
            if (A.d_h1) A.d_h1[(size_t)(row0 + 4 * q + i) * H + c] = d[i];
            xh[i] = (x0v[i] - m1v[i]) * r1v[i];
            g[i] = d[i] * g1n;
            gx[i] = g[i] * xh[i];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                       // one (dgamma, dbeta) partial per frame
            float* pp = A.part_n1 + (size_t)(f0 + j) * (2 * H);
            pp[c] = d[2 * j] * xh[2 * j] + d[2 * j + 1] * xh[2 * j + 1];
            pp[H + c] = d[2 * j] + d[2 * j + 1];
        }
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gx0[i] = (r1v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]) + dx1[i] + (A.add_xres ? dx3[i] : 0.f)) *
                     mx0.keep(kb[i], A.drop_scale);
    }
    {
        float g[2], gx[2], xh[2], s1v[2], s2v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float g0 = gx0[2 * j], g1 = gx0[2 * j + 1];
            const float drgb = ((mr != 0.f ? 0.f : g0) + (md != 0.f ? g1 : 0.f)) * (rgbv[j] > 0.f ? 1.f : 0.f);
            A.d_rgb_pre[(size_t)(f0 + j) * H + c] = drgb;
            float dd = (mr != 0.f ? g0 : 0.f) + (md != 0.f ? 0.f : g1);
            xh[j] = (dpv[j] - mdv[j]) * rdv[j];
            if (!(xh[j] * gd + bd > 0.f)) dd = 0.f;
            float* pp = A.part_dep + (size_t)(f0 + j) * (2 * H);
            pp[c] = dd * xh[j];
            pp[H + c] = dd;
            g[j] = dd * gd;
            gx[j] = g[j] * xh[j];
        }
        fb_rowsum2<2>(g, gx, redB, wave, li, q, s1v, s2v);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            A.d_dep_pre[(size_t)(f0 + j) * H + c] = rdv[j] * (g[j] - s1v[j] - xh[j] * s2v[j]);
    }
    R3D_CHAIN_MARK(A.timeline, wg == 0, 7);
}

// query role: backward of the layer-0 query self-attention sub-layer and the cross-attention's query projection.  With
// tgt = 0 nothing flows further back than the parameters; sain (gradient w.r.t. tgt + query_pos) feeds query_embed.
__device__ __forceinline__ void fc_bwd_query(const FcBwd& A, const int wgq, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    float* bufH = lds + kFcBufH;
    float* bufV = lds + kFcBufV;
    float* bufF = lds + kFcBufF;
    float* wl = lds + kFbWl + wave * (128 * kFbWP);
    float* redA = lds + kFbRed;
    constexpr int H = kFcH;
    const int row0 = wgq * kFcRows;
    const int c = wave * 16 + li, n0 = wave * 16;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    FcW s0, s1;
    fb_wload(s0, A.wq + n0, H, 128, lane);                                            // c0
    fb_wload(s1, A.w_out + n0, H, 128, lane);                                         // c1
    const f32x4 a_in = *reinterpret_cast<const f32x4*>(A.d_caq + (size_t)(row0 + (tid >> 5)) * H + 4 * (tid & 31));
    const FcMaskSrc md1(A.drop_d1, A.g1d, H);
    float t2p[4], tpre[4], m1v[4], r1v[4];
    uint8_t kb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        t2p[i] = A.d_t1_res[(size_t)r * H + c];
        tpre[i] = A.t1_pre[(size_t)r * H + c];
        m1v[i] = A.m1d[r]; r1v[i] = A.r1d[r];
        kb[i] = md1.raw(r, c);
    }
    const float g1 = A.g1d[c];
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(bufH + (tid >> 5) * kFcP1 + 4 * (tid & 31)) = a_in;
    fb_wstore(s0, wl, lane);
    fb_wload(s0, A.w_in + n0, H, 128, lane);                                          // c2: in_proj k-rows 0..127
    __syncthreads();
    FcOp a;
    f32x4 acc0 = zero, acc1 = zero;
    // ---- caqin = d_caq . Wq ; decoder norm1 backward (dy = caqin + the residual gradient) ; dropout
    fc_opload(a, bufH + li * kFcP1 + 4 * q);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c0
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);
    fb_wload(s1, A.w_in + (size_t)128 * H + n0, H, 128, lane);                        // c3
    __builtin_amdgcn_sched_barrier(0);
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float cq = acc0[i] + acc1[i];
            A.caqin[(size_t)(row0 + 4 * q + i) * H + c] = cq;
            d[i] = cq + t2p[i];
            xh[i] = (tpre[i] - m1v[i]) * r1v[i];
            g[i] = d[i] * g1;
            gx[i] = g[i] * xh[i];
        }
        float* pp = A.part_d1 + (size_t)(wgq * 4 + q) * (2 * H);
        pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
        pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float o = r1v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]);
            const size_t e = (size_t)(row0 + 4 * q + i) * H + c;
            const float ok = o * md1.keep(kb[i], A.drop_scale);
            if (A.t1pre_out) A.t1pre_out[e] = o;
            A.sap[e] = ok;
            bufV[(4 * q + i) * kFcP1 + c] = ok;
        }
    }
    __syncthreads();
    // ---- sao = sap . Wout
    fc_opload(a, bufV + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c1
#pragma unroll
    for (int i = 0; i < 4; ++i) A.sao[(size_t)(row0 + 4 * q + i) * H + c] = acc0[i] + acc1[i];
    __syncthreads();                                        // sao rows of both clips are written (workgroup scope)
    {
        MhaArgs m{};
        m.q = A.sa_qkv; m.ldq = 3 * H; m.k = A.sa_qkv + H; m.ldk = 3 * H; m.v = A.sa_qkv + 2 * H; m.ldv = 3 * H;
        m.probs = const_cast<float*>(A.p_sa); m.drop = A.drop_sa; m.drop_scale = A.drop_scale; m.d_o = A.sao; m.lddo = H;
        m.dq = A.saqkv; m.lddq = 3 * H; m.dk = A.saqkv + H; m.lddk = 3 * H; m.dv = A.saqkv + 2 * H; m.lddv = 3 * H;
        m.B = A.B; m.heads = 8; m.Lq = 8; m.Lk = 8; m.dh = 16; m.scale = 0.25f;
        mha_bwd_small_unit<16, 8, false>(m, (2 * wgq) * 8 + wave, wl);
        __builtin_amdgcn_wave_barrier();
        mha_bwd_small_unit<16, 8, false>(m, (2 * wgq + 1) * 8 + wave, wl);
    }
    __syncthreads();
    {
        const int r = tid >> 5, c4 = tid & 31;
        const float* src = A.saqkv + (size_t)(row0 + r) * (3 * H) + 4 * c4;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 128),
                    v2 = *reinterpret_cast<const f32x4*>(src + 256);
        float* dst = bufF + r * kFcP4 + 4 * c4;
        *reinterpret_cast<f32x4*>(dst) = v0;
        *reinterpret_cast<f32x4*>(dst + 128) = v1;
        *reinterpret_cast<f32x4*>(dst + 256) = v2;
    }
    fb_wstore(s0, wl, lane);                                                          // c2
    fb_wload(s0, A.w_in + (size_t)256 * H + n0, H, 128, lane);                        // c4
    __syncthreads();
    // ---- sain = saqkv . Win
    const float* ar = bufF + li * kFcP4 + 4 * q;
    acc0 = zero; acc1 = zero;
    fc_opload(a, ar);
    fb_chunk(a, wl, li, q, acc0, acc1);                                               // c2
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);                                                          // c3
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 128);
    fb_chunk(a, wl, li, q, acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);                                                          // c4
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 256);
    fb_chunk(a, wl, li, q, acc0, acc1);
#pragma unroll
    for (int i = 0; i < 4; ++i) A.sain[(size_t)(row0 + 4 * q + i) * H + c] = acc0[i] + acc1[i];
}

// ---------------------------------------------------------------------------------------------------------------
// backward, fuser role on the bf16 matrix cores: with the planes of the TRANSPOSED weights every input-gradient product is a
// y = x B^T product like the forward ones (no k-major staging); stages as fc_bwd_fuser
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fc3_bwd_fuser(const FcBwd& A, const int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    unsigned short* img = reinterpret_cast<unsigned short*>(lds);
    unsigned short* imgH = img + kF3ImgH;
    unsigned short* imgV = img + kF3ImgV;
    unsigned short* imgF = img + kF3ImgF;
    float* redA = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + kF3RedBytes);
    float* redB = redA + 2 * 8 * kFcRows;
    constexpr int H = kFcH;
    const int row0 = wg * kFcRows;
    const int c = wave * 16 + li;
    const int f0 = (row0 >> 1) + 2 * q;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    R3D_CHAIN_MARK(A.timeline, wg == 0, 0);
    Bf3B b0, b1, b2;
    bf3_bload<4>(b0, A.pl_wkv_t, 8, wave, 0, lane);                                    // c0: (Wkv)^T, k-steps 0..3
    bf3_bload<4>(b1, A.pl_wkv_t, 8, wave, 4, lane);                                    // c1: k-steps 4..7
    bf3_bload<1>(b2, A.pl_wseg_t, 1, wave, 0, lane);                                   // c2: (Wseg)^T, K <= 32
    {
        // A operand of stage 0: row r <- [d_cakv[frame r >> 1] (256) | d_seg[frame] zero-padded to 32]
        const int r = tid >> 5, c4 = tid & 31, fr = (row0 >> 1) + (r >> 1);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(A.d_cakv + (size_t)fr * (2 * H) + 4 * c4);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(A.d_cakv + (size_t)fr * (2 * H) + 128 + 4 * c4);
        f32x4 v2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * (c4 & 7) + e;
            const float t = A.d_seg[(size_t)fr * A.K + (k < A.K ? k : A.K - 1)];
            v2[e] = k < A.K ? t : 0.f;
        }
        bf3_store4(imgF, kF3P4, r, 4 * c4, v0);
        bf3_store4(imgF, kF3P4, r, 128 + 4 * c4, v1);
        if (c4 < 8) bf3_store4(imgF, kF3P4, r, 256 + 4 * c4, v2);
    }
    float x3v[4], mfv[4], rfv[4], dex[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        x3v[i] = A.x3[(size_t)r * H + c]; mfv[i] = A.mf[r]; rfv[i] = A.rf[r];
    }
    const float gF = A.gf[c];
    {
        const float* ex = A.d_extra ? A.d_extra : A.gf;
        const int ld = A.d_extra ? H : 0;
        dex[0] = ex[(size_t)f0 * ld + c]; dex[1] = ex[(size_t)(f0 + 1) * ld + c];
        if (!A.d_extra) { dex[0] = 0.f; dex[1] = 0.f; }
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 1);
    f32x4 acc0 = zero, acc1 = zero;
    // ---- stage 0: d(memory + pos) = d_cakv . Wkv (kept: pos_embedding's gradient), then + d_seg . Wseg
    bf3_chunk<4>(imgF, kF3P4, li, q, 0, b0, acc0, acc1);                               // c0
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b0, A.pl_w2_t, 4, 4 * wave + 0, 0, lane);                             // c3: (W2)^T tile 4w
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 4, b1, acc0, acc1);                               // c1
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b1, A.pl_w2_t, 4, 4 * wave + 1, 0, lane);                             // c4
    __builtin_amdgcn_sched_barrier(0);
    A.d_fused[(size_t)f0 * H + c] = acc0[0] + acc1[0];
    A.d_fused[(size_t)(f0 + 1) * H + c] = acc0[2] + acc1[2];
    bf3_chunk<1>(imgF, kF3P4, li, q, 8, b2, acc0, acc1);                               // c2
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b2, A.pl_w2_t, 4, 4 * wave + 2, 0, lane);                             // c5
    __builtin_amdgcn_sched_barrier(0);
    R3D_CHAIN_MARK(A.timeline, wg == 0, 2);
    // ---- stage 1: fuser.norm backward
    float dx3[4];
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = 0.5f * ((acc0[i] + acc1[i]) + dex[i >> 1]);
            xh[i] = (x3v[i] - mfv[i]) * rfv[i];
            g[i] = d[i] * gF;
            gx[i] = g[i] * xh[i];
        }
        float* pp = A.part_nf + (size_t)(wg * 4 + q) * (2 * H);
        pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
        pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dx3[i] = rfv[i] * (g[i] - s1v[i] - xh[i] * s2v[i]);
            A.d_x3[(size_t)(row0 + 4 * q + i) * H + c] = dx3[i];
            bf3_store1(imgH, kF3P1, 4 * q + i, c, dx3[i]);
        }
    }
    float uv[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) uv[t][i] = A.u[(size_t)(row0 + 4 * q + i) * (4 * H) + (4 * wave + t) * 16 + li];
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 3);
    // ---- stage 2: d_u = (d_x3 . W2) * GELU'(u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        if (t == 0) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b0, acc0, acc1);               // c3
        if (t == 1) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b1, acc0, acc1);               // c4
        if (t == 2) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b2, acc0, acc1);               // c5
        if (t == 3) bf3_chunk<4>(imgH, kF3P1, li, q, 0, b0, acc0, acc1);               // c6
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) bf3_bload<4>(b0, A.pl_w2_t, 4, 4 * wave + 3, 0, lane);             // c6
        if (t == 1) bf3_bload<4>(b1, A.pl_w1_t, 16, wave, 0, lane);                    // c7: (W1)^T, k-steps 0..3
        if (t == 2) bf3_bload<4>(b2, A.pl_w1_t, 16, wave, 4, lane);                    // c8
        if (t == 3) bf3_bload<4>(b0, A.pl_w1_t, 16, wave, 8, lane);                    // c9
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float du = (acc0[i] + acc1[i]) * gelu_grad_f(uv[t][i]);
            A.d_u[(size_t)(row0 + 4 * q + i) * (4 * H) + cu] = du;
            bf3_store1(imgF, kF3P4, 4 * q + i, cu, du);
        }
    }
    float x1v[4], m2v[4], r2v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        x1v[i] = A.x1[(size_t)r * H + c]; m2v[i] = A.m2[r]; r2v[i] = A.r2[r];
    }
    const float g2 = A.g2[c];
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 4);
    // ---- stage 3: d_h2 = d_u . W1 ; norm2 backward ; d_x1 = that + d_x3
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgF, kF3P4, li, q, 0, b1, acc0, acc1);                               // c7
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b1, A.pl_w1_t, 16, wave, 12, lane);                                   // c10
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 4, b2, acc0, acc1);                               // c8
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b2, A.pl_wproj_t, 4, wave, 0, lane);                                  // c11
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 8, b0, acc0, acc1);                               // c9
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b0, A.pl_wv_t, 4, wave, 0, lane);                                     // c12
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kF3P4, li, q, 12, b1, acc0, acc1);                              // c10
    __builtin_amdgcn_sched_barrier(0);
    float dx1[4];
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = acc0[i] + acc1[i];
            if (A.d_h2) A.d_h2[(size_t)(row0 + 4 * q + i) * H + c] = d[i];
            xh[i] = (x1v[i] - m2v[i]) * r2v[i];
            g[i] = d[i] * g2;
            gx[i] = g[i] * xh[i];
        }
        float* pp = A.part_n2 + (size_t)(wg * 4 + q) * (2 * H);
        pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
        pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        fb_rowsum2<4>(g, gx, redB, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dx1[i] = r2v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]) + dx3[i];
            A.d_x1[(size_t)(row0 + 4 * q + i) * H + c] = dx1[i];
            bf3_store1(imgV, kF3P1, 4 * q + i, c, dx1[i]);
        }
    }
    const FcMaskSrc mx0(A.drop_x0, A.g1n, H);
    float x0v[4], m1v[4], r1v[4];
    uint8_t kb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        x0v[i] = A.x0[(size_t)r * H + c]; m1v[i] = A.m1[r]; r1v[i] = A.r1[r];
        kb[i] = mx0.raw(r, c);
    }
    const float g1n = A.g1n[c], mr = A.m_rgb[c], md = A.m_dep[c], gd = A.lnd_g[c], bd = A.lnd_b[c];
    float rgbv[2], dpv[2], mdv[2], rdv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rgbv[j] = A.rgb[(size_t)(f0 + j) * H + c]; dpv[j] = A.dep_pre[(size_t)(f0 + j) * H + c];
        mdv[j] = A.mean_d[f0 + j]; rdv[j] = A.rstd_d[f0 + j];
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 5);
    // ---- stage 4: d_vsw = d_x1 . Wproj, un-swapped into d_v
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgV, kF3P1, li, q, 0, b2, acc0, acc1);                               // c11
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = acc0[i] + acc1[i];
        const int rs = 4 * q + (i ^ 1);
        A.d_v[(size_t)(row0 + rs) * H + c] = v;
        bf3_store1(imgH, kF3P1, rs, c, v);
    }
    __syncthreads();
    R3D_CHAIN_MARK(A.timeline, wg == 0, 6);
    // ---- stage 5: d_h1 = d_v . Wv ; norm1 backward + both residual gradients ; embd_drop ; exchange ; depth LN + ReLU
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgH, kF3P1, li, q, 0, b0, acc0, acc1);                               // c12
    float gx0[4];
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = acc0[i] + acc1[i];
            if (A.d_h1) A.d_h1[(size_t)(row0 + 4 * q + i) * H + c] = d[i];
            xh[i] = (x0v[i] - m1v[i]) * r1v[i];
            g[i] = d[i] * g1n;
            gx[i] = g[i] * xh[i];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* pp = A.part_n1 + (size_t)(f0 + j) * (2 * H);
            pp[c] = d[2 * j] * xh[2 * j] + d[2 * j + 1] * xh[2 * j + 1];
            pp[H + c] = d[2 * j] + d[2 * j + 1];
        }
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gx0[i] = (r1v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]) + dx1[i] + (A.add_xres ? dx3[i] : 0.f)) *
                     mx0.keep(kb[i], A.drop_scale);
    }
    {
        float g[2], gx[2], xh[2], s1v[2], s2v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float g0 = gx0[2 * j], g1 = gx0[2 * j + 1];
            const float drgb = ((mr != 0.f ? 0.f : g0) + (md != 0.f ? g1 : 0.f)) * (rgbv[j] > 0.f ? 1.f : 0.f);
            A.d_rgb_pre[(size_t)(f0 + j) * H + c] = drgb;
            float dd = (mr != 0.f ? g0 : 0.f) + (md != 0.f ? 0.f : g1);
            xh[j] = (dpv[j] - mdv[j]) * rdv[j];
            if (!(xh[j] * gd + bd > 0.f)) dd = 0.f;
            float* pp = A.part_dep + (size_t)(f0 + j) * (2 * H);
            pp[c] = dd * xh[j];
            pp[H + c] = dd;
            g[j] = dd * gd;
            gx[j] = g[j] * xh[j];
        }
        fb_rowsum2<2>(g, gx, redB, wave, li, q, s1v, s2v);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            A.d_dep_pre[(size_t)(f0 + j) * H + c] = rdv[j] * (g[j] - s1v[j] - xh[j] * s2v[j]);
    }
    R3D_CHAIN_MARK(A.timeline, wg == 0, 7);
}

__global__ __launch_bounds__(512) void fuser_chain_bwd_bf3_kernel(const FcBwd A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    const int nf = (2 * A.N) / kFcRows;
    if ((int)blockIdx.x < nf) fc3_bwd_fuser(A, (int)blockIdx.x, lds);
    else fc_bwd_query(A, (int)blockIdx.x - nf, lds);
}

__global__ __launch_bounds__(512) void fuser_chain_bwd_kernel(const FcBwd A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // The two waves a SIMD hosts (w and w + 4) run the same stage sequence in lockstep and would want the matrix core, the
    // LDS and the VALU at the same moments; a fixed priority difference lets one burst its MFMAs while the other is in its
    // epilogue / staging, so the phases of the pair interleave instead of colliding.
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    const int nf = (2 * A.N) / kFcRows;
    if ((int)blockIdx.x < nf) fc_bwd_fuser(A, (int)blockIdx.x, lds);
    else fc_bwd_query(A, (int)blockIdx.x - nf, lds);
}

}  // namespace r3d

/* 1 when the shapes fit the row-local chain kernels: hidden 128, whole 16-row workgroups on both roles, 8 heads of 16, 8
 * queries, a segmentation head of <= 128 classes. */
R3D_EXPORT int r3d_fuser_chain_supported(int N, int H, int K, int B, int Q, int heads) {
    return (H == r3d::kFcH && N > 0 && (2 * N) % r3d::kFcRows == 0 && K > 0 && K <= 128 && B > 0 && Q == 8 && heads == 8 &&
            (B * Q) % r3d::kFcRows == 0) ? 1 : 0;
}

R3D_EXPORT int r3d_fuser_chain_fwd(const r3d_fuser_chain_fwd_args* a, void* stream) {
    R3D_REQUIRE(a);
    R3D_REQUIRE(r3d_fuser_chain_supported(a->N, a->H, a->K, a->B, a->Q, a->heads));
    R3D_REQUIRE(a->S > 0 && a->N % a->S == 0);
    R3D_REQUIRE(a->x0 && a->h1 && a->wv && a->wproj && a->bproj && a->g2 && a->be2 && a->w1 && a->b1 && a->w2 && a->b2 && a->gf &&
                a->bef && a->pos && a->wkv && a->bkv && a->wseg && a->bseg);
    R3D_REQUIRE(a->vsw && a->x1 && a->h2 && a->m2 && a->r2 && a->u && a->f1 && a->x3 && a->mf && a->rf && a->fused && a->seg &&
                a->cakv);
    R3D_REQUIRE(a->qpos && a->w_in && a->b_in && a->w_out && a->b_out && a->g1 && a->be1 && a->wq && a->bq);
    R3D_REQUIRE(a->sa_qkv && a->p_sa && a->sa_o && a->t1_pre && a->t1 && a->m1 && a->r1 && a->caq);
    const void* al[] = {a->h1, a->wv, a->wproj, a->w1, a->w2, a->wkv, a->wseg, a->qpos, a->w_in, a->w_out, a->wq, a->sa_o,
                        a->sa_qkv};
    for (const void* p : al)
        if (!r3d_aligned16(p)) return R3D_EALIGN;
    hipError_t e = hipFuncSetAttribute((const void*)r3d::fuser_chain_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       r3d::kFcLdsBytes);
    if (e != hipSuccess) return (int)e;
    const int grid = (2 * a->N) / r3d::kFcRows + (a->B * a->Q) / r3d::kFcRows;
    if (a->pl_wv) {               // weights pre-split into operand-order bf16 planes: the fuser role on the bf16 matrix cores
        R3D_REQUIRE(a->pl_wproj && a->pl_w1 && a->pl_w2 && a->pl_wkv && a->pl_wseg);
        const int ldsb = r3d::kFcLdsBytes > r3d::kF3LdsBytes ? r3d::kFcLdsBytes : r3d::kF3LdsBytes;     // (query role: fp32 layout)
        e = hipFuncSetAttribute((const void*)r3d::fuser_chain_fwd_bf3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(r3d::fuser_chain_fwd_bf3_kernel, dim3(grid), dim3(512), (size_t)ldsb, (hipStream_t)stream, *a);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    hipLaunchKernelGGL(r3d::fuser_chain_fwd_kernel, dim3(grid), dim3(512), (size_t)r3d::kFcLdsBytes, (hipStream_t)stream, *a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* bf16 elements of the operand-order planes of an N x K matrix (three planes, zero-padded to whole 16 x 32 blocks). */
R3D_EXPORT int64_t r3d_weight_plane_elems(int N, int K) { return N > 0 && K > 0 ? (int64_t)r3d::bf3_plane_elems(N, K) : 0; }

/* Writes the bf16x3 operand-order planes of njobs matrices (chain_bf3.h) in one launch.  jobs: DEVICE array; first_block of
 * job j = sum over i < j of ceil(N_i / 16) * ceil(K_i / 32); total_blocks = that sum over all jobs. */
R3D_EXPORT int r3d_weight_planes(const r3d_plane_job* jobs_device, int njobs, int total_blocks, void* stream) {
    R3D_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0);
    hipLaunchKernelGGL(r3d::weight_planes_kernel, dim3(r3d_cdiv(total_blocks, 4)), dim3(256), 0, (hipStream_t)stream, jobs_device,
                       njobs, total_blocks);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_fuser_chain_bwd(const r3d_fuser_chain_bwd_args* a, void* stream) {
    R3D_REQUIRE(a);
    R3D_REQUIRE(r3d_fuser_chain_supported(a->N, a->H, a->K, a->B, a->Q, a->heads));
    R3D_REQUIRE(a->d_cakv && a->d_seg && a->wkv && a->wseg && a->x3 && a->mf && a->rf && a->gf && a->w2 && a->u && a->w1 && a->x1 &&
                a->m2 && a->r2 && a->g2 && a->wproj && a->wv && a->x0 && a->m1 && a->r1 && a->g1n && a->m_rgb && a->m_dep &&
                a->rgb && a->dep_pre && a->mean_d && a->rstd_d && a->lnd_g && a->lnd_b);
    R3D_REQUIRE(a->d_fused && a->d_x3 && a->d_u && a->d_x1 && a->d_v && a->d_rgb_pre && a->d_dep_pre && a->part_nf && a->part_n2 &&
                a->part_n1 && a->part_dep);
    R3D_REQUIRE(a->d_caq && a->d_t1_res && a->wq && a->t1_pre && a->m1d && a->r1d && a->g1d && a->w_out && a->sa_qkv && a->p_sa &&
                a->w_in && a->caqin && a->sap && a->sao && a->saqkv && a->sain && a->part_d1);
    const void* al[] = {a->d_cakv, a->wkv, a->wseg, a->w2, a->w1, a->wproj, a->wv, a->d_caq, a->wq, a->w_out, a->w_in, a->saqkv,
                        a->sa_qkv, a->sao};
    for (const void* p : al)
        if (!r3d_aligned16(p)) return R3D_EALIGN;
    hipError_t e = hipFuncSetAttribute((const void*)r3d::fuser_chain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       r3d::kFbLdsBytes);
    if (e != hipSuccess) return (int)e;
    const int grid = (2 * a->N) / r3d::kFcRows + (a->B * a->Q) / r3d::kFcRows;
    if (a->pl_wkv_t) {            // planes of the transposed weights: the fuser role on the bf16 matrix cores
        R3D_REQUIRE(a->pl_wseg_t && a->pl_w2_t && a->pl_w1_t && a->pl_wproj_t && a->pl_wv_t && a->K <= 32);
        const int ldsb = r3d::kFbLdsBytes > r3d::kF3LdsBytes ? r3d::kFbLdsBytes : r3d::kF3LdsBytes;     // (query role: fp32 layout)
        e = hipFuncSetAttribute((const void*)r3d::fuser_chain_bwd_bf3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(r3d::fuser_chain_bwd_bf3_kernel, dim3(grid), dim3(512), (size_t)ldsb, (hipStream_t)stream, *a);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    hipLaunchKernelGGL(r3d::fuser_chain_bwd_kernel, dim3(grid), dim3(512), (size_t)r3d::kFbLdsBytes, (hipStream_t)stream, *a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
