// The decoder layer's query side, from the cross-attention core to the layer's last residual -- and, in the training
// step, the decoder tail, the three losses and the whole way back to the cross-attention's input gradients -- as ONE launch
// (hidden = 128, 8 queries, 8 heads of 16, <= 64 keys).
//
// Everything after the key/value projection of the decoder layer is local to a clip: its 8 query rows attend to the clip's
// S memory rows (model/extras/transformer.py:300-304), then out_proj -> dropout -> + tgt -> norm2 (:304-306), linear1 ->
// ReLU -> dropout -> linear2 -> dropout -> + tgt (:325-328), norm3 -> decoder.norm -> heads (:329,:182-183,
// model/futr_safuser_tokenfusion.py:219-226), the losses (train/train_proposed_depth.py:171-213) and the adjoint of all of
// it.  As separate launches that is 4 + 1 + 5 dependent latency-bound launches (66 us of the 212 us step at the bench
// shape).  Here one workgroup owns one clip: the attention core runs one (clip, head) unit per wave (mha_small.h), the
// products are the row-complete MFMA stages of chain_common.h with 8 of the 16 tile rows in use (the matrix cores do not
// care: a workgroup's time is set by its chain of weight chunks, not by the rows), the activations stay in LDS from stage to
// stage, and the tail + losses of the clip (losses_dev.h: tail_clip_body) run between the forward and the backward half.
//
// phases (bit mask): 1 = forward half, 2 = tail + losses + tail backward (needs the loss workgroups: grid = B + N/8),
// 4 = backward half.  The engine launches 7 when forward, losses and backward follow each other (training step), 1 and 4
// separately otherwise (inference; tests that stop between the calls).
#include "chain_bf3.h"
#include "mha_small.h"
#include "losses_dev.h"

namespace r3d {

constexpr int kDcBufA = 0;
constexpr int kDcBufB = kDcBufA + kFcRows * kFcP1;
constexpr int kDcBufF = kDcBufB + kFcRows * kFcP1;
constexpr int kDcWl = kDcBufF + kFcRows * kFcP4;
constexpr int kDcWlWave = 128 * kFbWP;                      // per wave: the larger of the two chunk layouts
constexpr int kDcRed = kDcWl + 8 * kDcWlWave;
constexpr int kDcLdsFloats = kDcRed + 2 * 2 * 8 * kFcRows;
constexpr int kDcLdsBytes = kDcLdsFloats * 4;
static_assert(16 * kFcWP <= kDcWlWave && mha_small_lds_floats(16, 8) <= kDcWlWave && mha_small_bwd_lds_floats(16, 8) <= kDcWlWave,
              "wave region");
static_assert(kDcLdsBytes + (2 * 8 * (kTLHeads + 8) + 8 * 4 * 128 + 64) * 4 <= 160 * 1024, "LDS (dynamic + the tail's static arrays)");

typedef r3d_decoder_chain_args DcArgs;

// rows 0..7 of a 16-row LDS tile <- 8 rows of a dense [*, 128] matrix starting at row0; rows 8..15 <- 0
__device__ __forceinline__ void dc_stage_rows(float* buf, const float* src, int row0, int tid) {
    const int r = tid >> 5, c4 = tid & 31;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + (r & 7)) * kFcH + 4 * c4);
    *reinterpret_cast<f32x4*>(buf + r * kFcP1 + 4 * c4) = r < 8 ? v : z;
}

__device__ __forceinline__ void dc_fwd(const DcArgs& D, const int b, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    float* bufA = lds + kDcBufA;
    float* bufB = lds + kDcBufB;
    float* bufF = lds + kDcBufF;
    float* wl = lds + kDcWl + wave * kDcWlWave;
    float (*red)[8][kFcRows] = reinterpret_cast<float (*)[8][kFcRows]>(lds + kDcRed);
    constexpr int H = kFcH;
    const int row0 = b * 8;
    const int c = wave * 16 + li;
    const bool live = q < 2;                               // tile rows 0..7 are the clip's queries
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    R3D_CHAIN_MARK(D.timeline, b == 0, 0);
    // ---- weight chunks c0 (out_proj tile w), c1 (linear1 tile 4w) and the epilogue operands: all requested up front
    FcW s0, s1;
    fc_wload(s0, D.wo + (size_t)(wave * 16) * H, H, 16, lane);
    fc_wload(s1, D.w1 + (size_t)((4 * wave + 0) * 16) * H, H, 16, lane);
    const FcMaskSrc md2(D.drop_d2, D.g2, H), md3(D.drop_d3, D.g2, H), mff(D.drop_ff, D.g2, 4 * H);
    float t1v[4];
    uint8_t kb2[4], kb3[4], kbf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t r = (size_t)(row0 + ((4 * q + i) & 7));
        t1v[i] = D.t1[r * H + c];
        kb2[i] = md2.raw(r, c);
        kb3[i] = md3.raw(r, c);
#pragma unroll
        for (int t = 0; t < 4; ++t) kbf[t][i] = mff.raw(r, (4 * wave + t) * 16 + li);
    }
    const float b_o = D.bo[c], g2 = D.g2[c], be2 = D.be2[c], b_2 = D.b2[c];
    float b_1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b_1[t] = D.b1[(4 * wave + t) * 16 + li];
    R3D_CHAIN_MARK(D.timeline, b == 0, 1);
    // ---- cross-attention core: one (clip, head) unit per wave (transformer.py:300-304; get_pad_mask :243)
    {
        MhaArgs m{};
        m.q = D.caq; m.ldq = H; m.k = D.cakv; m.ldk = 2 * H; m.v = D.cakv + H; m.ldv = 2 * H;
        m.key_label = D.key_label; m.pad_idx = D.pad_idx; m.probs = D.p_ca; m.drop = D.drop_ca; m.drop_scale = D.drop_scale;
        m.o = D.ca_o; m.ldo = H; m.B = D.B; m.heads = 8; m.Lq = 8; m.Lk = D.S; m.dh = 16; m.scale = 0.25f;
        mha_fwd_small_unit<16, 8, false>(m, b * 8 + wave, wl);
    }
    R3D_CHAIN_MARK(D.timeline, b == 0, 2);
    __syncthreads();                                        // ca_o rows of the clip are written (workgroup scope)
    dc_stage_rows(bufA, D.ca_o, row0, tid);
    fc_wstore(s0, wl, lane);                                                            // c0
    fc_wload(s0, D.w1 + (size_t)((4 * wave + 1) * 16) * H, H, 16, lane);               // c2
    __syncthreads();
    const float* wr = wl + li * kFcWP + 4 * q;
    FcOp a;
    f32x4 acc0 = zero, acc1 = zero;
    R3D_CHAIN_MARK(D.timeline, b == 0, 3);
    // ---- out_proj -> dropout -> + t1 -> norm2 (transformer.py:304-306)
    fc_opload(a, bufA + li * kFcP1 + 4 * q);
    fc_chunk(a, wr, acc0, acc1);                                                        // c0
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s1, wl, lane);                                                            // c1
    fc_wload(s1, D.w1 + (size_t)((4 * wave + 2) * 16) * H, H, 16, lane);               // c3
    __builtin_amdgcn_sched_barrier(0);
    float t2p[4], mean[4], rstd[4], t2v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t2p[i] = ((acc0[i] + acc1[i]) + b_o) * md2.keep(kb2[i], D.drop_scale) + t1v[i];
        if (live) D.t2_pre[(size_t)(row0 + 4 * q + i) * H + c] = t2p[i];
    }
    fc_layernorm(t2p, red, wave, li, q, mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t2v[i] = (t2p[i] - mean[i]) * rstd[i] * g2 + be2;
        const int r = 4 * q + i;
        if (live) {
            D.t2[(size_t)(row0 + r) * H + c] = t2v[i];
            if (wave == 0 && li == 0) { D.m2[row0 + r] = mean[i]; D.r2[row0 + r] = rstd[i]; }
        }
        bufB[r * kFcP1 + c] = live ? t2v[i] : 0.f;
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 4);
    // ---- linear1 -> ReLU -> dropout (transformer.py:327)
    fc_opload(a, bufB + li * kFcP1 + 4 * q);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        fc_chunk(a, wr, acc0, acc1);                                                    // c1 + t
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) { fc_wstore(s0, wl, lane); fc_wload(s0, D.w1 + (size_t)((4 * wave + 3) * 16) * H, H, 16, lane); }       // c4
        if (t == 1) { fc_wstore(s1, wl, lane); fc_wload(s1, D.w2 + (size_t)(wave * 16) * (4 * H) + 0, 4 * H, 16, lane); }   // c5
        if (t == 2) { fc_wstore(s0, wl, lane); fc_wload(s0, D.w2 + (size_t)(wave * 16) * (4 * H) + 128, 4 * H, 16, lane); } // c6
        if (t == 3) { fc_wstore(s1, wl, lane); fc_wload(s1, D.w2 + (size_t)(wave * 16) * (4 * H) + 256, 4 * H, 16, lane); } // c7
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float f = fmaxf((acc0[i] + acc1[i]) + b_1[t], 0.f) * mff.keep(kbf[t][i], D.drop_scale);
            if (live) D.ff1[(size_t)(row0 + 4 * q + i) * (4 * H) + cu] = f;
            bufF[(4 * q + i) * kFcP4 + cu] = live ? f : 0.f;
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 5);
    // ---- linear2 -> dropout -> + t2 (transformer.py:327-328)
    acc0 = zero; acc1 = zero;
    const float* ar = bufF + li * kFcP4 + 4 * q;
    fc_opload(a, ar);
    fc_chunk(a, wr, acc0, acc1);                                                        // c5
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s0, wl, lane);                                                            // c6
    fc_wload(s0, D.w2 + (size_t)(wave * 16) * (4 * H) + 384, 4 * H, 16, lane);         // c8
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 128);
    fc_chunk(a, wr, acc0, acc1);                                                        // c6
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s1, wl, lane);                                                            // c7
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 256);
    fc_chunk(a, wr, acc0, acc1);                                                        // c7
    __builtin_amdgcn_sched_barrier(0);
    fc_wstore(s0, wl, lane);                                                            // c8
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 384);
    fc_chunk(a, wr, acc0, acc1);                                                        // c8
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            D.t3_pre[(size_t)(row0 + 4 * q + i) * H + c] = ((acc0[i] + acc1[i]) + b_2) * md3.keep(kb3[i], D.drop_scale) + t2v[i];
    }
}

__device__ __forceinline__ void dc_bwd(const DcArgs& D, const int b, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    float* bufA = lds + kDcBufA;
    float* bufF = lds + kDcBufF;
    float* wl = lds + kDcWl + wave * kDcWlWave;
    float* redA = lds + kDcRed;
    constexpr int H = kFcH;
    const int row0 = b * 8;
    const int c = wave * 16 + li, n0 = wave * 16;
    const bool live = q < 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    R3D_CHAIN_MARK(D.timeline, b == 0, 10);
    // ---- weight chunks c0, c1 (linear2's weight, column tiles 4w, 4w + 1), the A rows (d ff2) and the epilogue operands
    FcW s0, s1;
    fb_wload(s0, D.w2 + (4 * wave + 0) * 16, 4 * H, 128, lane);
    fb_wload(s1, D.w2 + (4 * wave + 1) * 16, 4 * H, 128, lane);
    dc_stage_rows(bufA, D.d_ff2, row0, tid);
    const FcMaskSrc md2(D.drop_d2, D.g2, H), mff(D.drop_ff, D.g2, 4 * H);
    float ffv[4][4], res[4], t2pv[4], m2v[4], r2v[4];
    uint8_t kb2[4], kbf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t r = (size_t)(row0 + ((4 * q + i) & 7));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ffv[t][i] = D.ff1[r * (4 * H) + (4 * wave + t) * 16 + li];
            kbf[t][i] = mff.raw(r, (4 * wave + t) * 16 + li);
        }
        res[i] = D.d_t3pre[r * H + c];
        t2pv[i] = D.t2_pre[r * H + c];
        m2v[i] = D.m2[r]; r2v[i] = D.r2[r];
        kb2[i] = md2.raw(r, c);
    }
    const float g2 = D.g2[c];
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);                                                            // c0
    fb_wload(s0, D.w2 + (4 * wave + 2) * 16, 4 * H, 128, lane);                         // c2
    __syncthreads();
    FcOp a;
    f32x4 acc0, acc1;
    R3D_CHAIN_MARK(D.timeline, b == 0, 11);
    // ---- d ff1 = (d ff2 . W2) * ReLU' * dropout
    fc_opload(a, bufA + li * kFcP1 + 4 * q);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        fb_chunk(a, wl, li, q, acc0, acc1);                                             // c0 + t
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) { fb_wstore(s1, wl, lane); fb_wload(s1, D.w2 + (4 * wave + 3) * 16, 4 * H, 128, lane); }        // c3
        if (t == 1) { fb_wstore(s0, wl, lane); fb_wload(s0, D.w1 + n0, H, 128, lane); }                             // c4
        if (t == 2) { fb_wstore(s1, wl, lane); fb_wload(s1, D.w1 + (size_t)128 * H + n0, H, 128, lane); }           // c5
        if (t == 3) { fb_wstore(s0, wl, lane); fb_wload(s0, D.w1 + (size_t)256 * H + n0, H, 128, lane); }           // c6
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // ReLU' (the stored activation is post-ReLU, post-dropout) x dropout'
            const float gv = ffv[t][i] > 0.f ? (acc0[i] + acc1[i]) * mff.keep(kbf[t][i], D.drop_scale) : 0.f;
            if (live) D.d_ff1[(size_t)(row0 + 4 * q + i) * (4 * H) + cu] = gv;
            bufF[(4 * q + i) * kFcP4 + cu] = live ? gv : 0.f;
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 12);
    // ---- d t2 = d ff1 . W1 + d t3_pre (the residual) ; norm2 backward ; dropout2'
    acc0 = zero; acc1 = zero;
    const float* ar = bufF + li * kFcP4 + 4 * q;
    fc_opload(a, ar);
    fb_chunk(a, wl, li, q, acc0, acc1);                                                 // c4
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);                                                            // c5
    fb_wload(s1, D.w1 + (size_t)384 * H + n0, H, 128, lane);                            // c7
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 128);
    fb_chunk(a, wl, li, q, acc0, acc1);                                                 // c5
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);                                                            // c6
    fb_wload(s0, D.wo + n0, H, 128, lane);                                              // c8
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 256);
    fb_chunk(a, wl, li, q, acc0, acc1);                                                 // c6
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s1, wl, lane);                                                            // c7
    __builtin_amdgcn_sched_barrier(0);
    fc_opload(a, ar + 384);
    fb_chunk(a, wl, li, q, acc0, acc1);                                                 // c7
    __builtin_amdgcn_sched_barrier(0);
    fb_wstore(s0, wl, lane);                                                            // c8
    __builtin_amdgcn_sched_barrier(0);
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = live ? (acc0[i] + acc1[i]) + res[i] : 0.f;
            xh[i] = (t2pv[i] - m2v[i]) * r2v[i];
            g[i] = d[i] * g2;
            gx[i] = g[i] * xh[i];
        }
        if (live) {                                         // one (dgamma, dbeta) partial per 4 rows: blocks 2b, 2b + 1
            float* pp = D.part_d2 + (size_t)(2 * b + q) * (2 * H);
            pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
            pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        }
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float o = r2v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]);
            const size_t e = (size_t)(row0 + 4 * q + i) * H + c;
            const float ok = o * md2.keep(kb2[i], D.drop_scale);
            if (live) { D.d_t2pre[e] = o; D.d_cap[e] = ok; }
            bufA[(4 * q + i) * kFcP1 + c] = live ? ok : 0.f;
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 13);
    // ---- d ca_o = d cap . Wo, then the attention core's backward
    fc_opload(a, bufA + li * kFcP1 + 4 * q);
    acc0 = zero; acc1 = zero;
    fb_chunk(a, wl, li, q, acc0, acc1);                                                 // c8
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) D.d_cao[(size_t)(row0 + 4 * q + i) * H + c] = acc0[i] + acc1[i];
    }
    R3D_CHAIN_MARK(D.timeline, b == 0, 14);
    __syncthreads();                                        // d ca_o rows of the clip are written (workgroup scope)
    {
        MhaArgs m{};
        m.q = D.caq; m.ldq = H; m.k = D.cakv; m.ldk = 2 * H; m.v = D.cakv + H; m.ldv = 2 * H;
        m.probs = D.p_ca; m.drop = D.drop_ca; m.drop_scale = D.drop_scale; m.d_o = D.d_cao; m.lddo = H;
        m.dq = D.d_caq; m.lddq = H; m.dk = D.d_cakv; m.lddk = 2 * H; m.dv = D.d_cakv + H; m.lddv = 2 * H;
        m.B = D.B; m.heads = 8; m.Lq = 8; m.Lk = D.S; m.dh = 16; m.scale = 0.25f;
        mha_bwd_small_unit<16, 8, false>(m, b * 8 + wave, wl);
    }
    R3D_CHAIN_MARK(D.timeline, b == 0, 15);
}

// ---------------------------------------------------------------------------------------------------------------
// the two halves on the bf16 matrix cores (chain_bf3.h): weights from the operand-order planes (D.pl_*), activations as bf16
// planes in LDS; same stages, same stored tensors
// ---------------------------------------------------------------------------------------------------------------
constexpr int kD3P1 = kFcH + 8, kD3P4 = 4 * kFcH + 8;
constexpr int kD3ImgA = 0;                                   // (bf16 elements)
constexpr int kD3ImgB = kD3ImgA + 3 * 16 * kD3P1;
constexpr int kD3ImgF = kD3ImgB + 3 * 16 * kD3P1;
constexpr int kD3ScrBytes = (kD3ImgF + 3 * 16 * kD3P4) * 2;  // attention scratch: one region per wave
constexpr int kD3ScrWave = 1856;                             // floats (mha_small_bwd_lds_floats(16, 8))
constexpr int kD3RedBytes = kD3ScrBytes + 8 * kD3ScrWave * 4;
constexpr int kD3LdsBytes = kD3RedBytes + 2 * 2 * 8 * kFcRows * 4;
static_assert(mha_small_lds_floats(16, 8) <= kD3ScrWave && mha_small_bwd_lds_floats(16, 8) <= kD3ScrWave, "attention scratch");
static_assert(kD3LdsBytes + (2 * 8 * (kTLHeads + 8) + 8 * 4 * 128 + 64) * 4 <= 160 * 1024, "LDS (dynamic + the tail's static arrays)");

// rows 0..7 of a bf16x3 image <- 8 rows of a dense [*, 128] fp32 matrix; rows 8..15 <- 0
__device__ __forceinline__ void dc3_stage_rows(unsigned short* img, const float* src, int row0, int tid) {
    const int r = tid >> 5, c4 = tid & 31;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + (r & 7)) * kFcH + 4 * c4);
    bf3_store4(img, kD3P1, r, 4 * c4, r < 8 ? v : z);
}

__device__ __forceinline__ void dc3_fwd(const DcArgs& D, const int b, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    unsigned short* img = reinterpret_cast<unsigned short*>(lds);
    unsigned short* imgA = img + kD3ImgA;
    unsigned short* imgB = img + kD3ImgB;
    unsigned short* imgF = img + kD3ImgF;
    float* scr = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + kD3ScrBytes) + wave * kD3ScrWave;
    float (*red)[8][kFcRows] = reinterpret_cast<float (*)[8][kFcRows]>(reinterpret_cast<unsigned char*>(lds) + kD3RedBytes);
    constexpr int H = kFcH;
    const int row0 = b * 8;
    const int c = wave * 16 + li;
    const bool live = q < 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    R3D_CHAIN_MARK(D.timeline, b == 0, 0);
    Bf3B b0, b1, b2;
    bf3_bload<4>(b0, D.pl_wo, 4, wave, 0, lane);                                       // c0: out_proj tile w
    bf3_bload<4>(b1, D.pl_w1, 4, 4 * wave + 0, 0, lane);                               // c1: linear1 tile 4w
    bf3_bload<4>(b2, D.pl_w1, 4, 4 * wave + 1, 0, lane);                               // c2
    const FcMaskSrc md2(D.drop_d2, D.g2, H), md3(D.drop_d3, D.g2, H), mff(D.drop_ff, D.g2, 4 * H);
    float t1v[4];
    uint8_t kb2[4], kb3[4], kbf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t r = (size_t)(row0 + ((4 * q + i) & 7));
        t1v[i] = D.t1[r * H + c];
        kb2[i] = md2.raw(r, c);
        kb3[i] = md3.raw(r, c);
#pragma unroll
        for (int t = 0; t < 4; ++t) kbf[t][i] = mff.raw(r, (4 * wave + t) * 16 + li);
    }
    const float b_o = D.bo[c], g2 = D.g2[c], be2 = D.be2[c], b_2 = D.b2[c];
    float b_1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b_1[t] = D.b1[(4 * wave + t) * 16 + li];
    R3D_CHAIN_MARK(D.timeline, b == 0, 1);
    {
        MhaArgs m{};
        m.q = D.caq; m.ldq = H; m.k = D.cakv; m.ldk = 2 * H; m.v = D.cakv + H; m.ldv = 2 * H;
        m.key_label = D.key_label; m.pad_idx = D.pad_idx; m.probs = D.p_ca; m.drop = D.drop_ca; m.drop_scale = D.drop_scale;
        m.o = D.ca_o; m.ldo = H; m.B = D.B; m.heads = 8; m.Lq = 8; m.Lk = D.S; m.dh = 16; m.scale = 0.25f;
        mha_fwd_small_unit<16, 8, false>(m, b * 8 + wave, scr);
    }
    R3D_CHAIN_MARK(D.timeline, b == 0, 2);
    __syncthreads();                                        // ca_o rows of the clip are written (workgroup scope)
    dc3_stage_rows(imgA, D.ca_o, row0, tid);
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 3);
    f32x4 acc0 = zero, acc1 = zero;
    // ---- out_proj -> dropout -> + t1 -> norm2
    bf3_chunk<4>(imgA, kD3P1, li, q, 0, b0, acc0, acc1);                               // c0
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b0, D.pl_w1, 4, 4 * wave + 2, 0, lane);                               // c3
    __builtin_amdgcn_sched_barrier(0);
    float t2p[4], mean[4], rstd[4], t2v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t2p[i] = ((acc0[i] + acc1[i]) + b_o) * md2.keep(kb2[i], D.drop_scale) + t1v[i];
        if (live) D.t2_pre[(size_t)(row0 + 4 * q + i) * H + c] = t2p[i];
    }
    fc_layernorm(t2p, red, wave, li, q, mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t2v[i] = (t2p[i] - mean[i]) * rstd[i] * g2 + be2;
        const int r = 4 * q + i;
        if (live) {
            D.t2[(size_t)(row0 + r) * H + c] = t2v[i];
            if (wave == 0 && li == 0) { D.m2[row0 + r] = mean[i]; D.r2[row0 + r] = rstd[i]; }
        }
        bf3_store1(imgB, kD3P1, r, c, live ? t2v[i] : 0.f);
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 4);
    // ---- linear1 -> ReLU -> dropout
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        if (t == 0) bf3_chunk<4>(imgB, kD3P1, li, q, 0, b1, acc0, acc1);               // c1
        if (t == 1) bf3_chunk<4>(imgB, kD3P1, li, q, 0, b2, acc0, acc1);               // c2
        if (t == 2) bf3_chunk<4>(imgB, kD3P1, li, q, 0, b0, acc0, acc1);               // c3
        if (t == 3) bf3_chunk<4>(imgB, kD3P1, li, q, 0, b1, acc0, acc1);               // c4
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) bf3_bload<4>(b1, D.pl_w1, 4, 4 * wave + 3, 0, lane);               // c4
        if (t == 1) bf3_bload<4>(b2, D.pl_w2, 16, wave, 0, lane);                      // c5: linear2, k-steps 0..3
        if (t == 2) bf3_bload<4>(b0, D.pl_w2, 16, wave, 4, lane);                      // c6
        if (t == 3) bf3_bload<4>(b1, D.pl_w2, 16, wave, 8, lane);                      // c7
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float f = fmaxf((acc0[i] + acc1[i]) + b_1[t], 0.f) * mff.keep(kbf[t][i], D.drop_scale);
            if (live) D.ff1[(size_t)(row0 + 4 * q + i) * (4 * H) + cu] = f;
            bf3_store1(imgF, kD3P4, 4 * q + i, cu, live ? f : 0.f);
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 5);
    // ---- linear2 -> dropout -> + t2
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgF, kD3P4, li, q, 0, b2, acc0, acc1);                               // c5
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b2, D.pl_w2, 16, wave, 12, lane);                                     // c8
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kD3P4, li, q, 4, b0, acc0, acc1);                               // c6
    bf3_chunk<4>(imgF, kD3P4, li, q, 8, b1, acc0, acc1);                               // c7
    bf3_chunk<4>(imgF, kD3P4, li, q, 12, b2, acc0, acc1);                              // c8
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            D.t3_pre[(size_t)(row0 + 4 * q + i) * H + c] =
                ((acc0[i] + acc1[i]) + b_2) * md3.keep(kb3[i], D.drop_scale) + t2v[i];
    }
}

__device__ __forceinline__ void dc3_bwd(const DcArgs& D, const int b, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    unsigned short* img = reinterpret_cast<unsigned short*>(lds);
    unsigned short* imgA = img + kD3ImgA;
    unsigned short* imgF = img + kD3ImgF;
    float* scr = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + kD3ScrBytes) + wave * kD3ScrWave;
    float* redA = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + kD3RedBytes);
    constexpr int H = kFcH;
    const int row0 = b * 8;
    const int c = wave * 16 + li;
    const bool live = q < 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    R3D_CHAIN_MARK(D.timeline, b == 0, 10);
    Bf3B b0, b1, b2;
    bf3_bload<4>(b0, D.pl_w2_t, 4, 4 * wave + 0, 0, lane);                             // c0: (linear2.weight)^T tile 4w
    bf3_bload<4>(b1, D.pl_w2_t, 4, 4 * wave + 1, 0, lane);                             // c1
    bf3_bload<4>(b2, D.pl_w2_t, 4, 4 * wave + 2, 0, lane);                             // c2
    dc3_stage_rows(imgA, D.d_ff2, row0, tid);
    const FcMaskSrc md2(D.drop_d2, D.g2, H), mff(D.drop_ff, D.g2, 4 * H);
    float ffv[4][4], res[4], t2pv[4], m2v[4], r2v[4];
    uint8_t kb2[4], kbf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t r = (size_t)(row0 + ((4 * q + i) & 7));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ffv[t][i] = D.ff1[r * (4 * H) + (4 * wave + t) * 16 + li];
            kbf[t][i] = mff.raw(r, (4 * wave + t) * 16 + li);
        }
        res[i] = D.d_t3pre[r * H + c];
        t2pv[i] = D.t2_pre[r * H + c];
        m2v[i] = D.m2[r]; r2v[i] = D.r2[r];
        kb2[i] = md2.raw(r, c);
    }
    const float g2 = D.g2[c];
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 11);
    f32x4 acc0, acc1;
    // ---- d ff1 = (d ff2 . W2) * ReLU' * dropout
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0 = zero; acc1 = zero;
        if (t == 0) bf3_chunk<4>(imgA, kD3P1, li, q, 0, b0, acc0, acc1);               // c0
        if (t == 1) bf3_chunk<4>(imgA, kD3P1, li, q, 0, b1, acc0, acc1);               // c1
        if (t == 2) bf3_chunk<4>(imgA, kD3P1, li, q, 0, b2, acc0, acc1);               // c2
        if (t == 3) bf3_chunk<4>(imgA, kD3P1, li, q, 0, b0, acc0, acc1);               // c3
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) bf3_bload<4>(b0, D.pl_w2_t, 4, 4 * wave + 3, 0, lane);             // c3
        if (t == 1) bf3_bload<4>(b1, D.pl_w1_t, 16, wave, 0, lane);                    // c4: (linear1.weight)^T, k-steps 0..3
        if (t == 2) bf3_bload<4>(b2, D.pl_w1_t, 16, wave, 4, lane);                    // c5
        if (t == 3) bf3_bload<4>(b0, D.pl_w1_t, 16, wave, 8, lane);                    // c6
        __builtin_amdgcn_sched_barrier(0);
        const int cu = (4 * wave + t) * 16 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gv = ffv[t][i] > 0.f ? (acc0[i] + acc1[i]) * mff.keep(kbf[t][i], D.drop_scale) : 0.f;
            if (live) D.d_ff1[(size_t)(row0 + 4 * q + i) * (4 * H) + cu] = gv;
            bf3_store1(imgF, kD3P4, 4 * q + i, cu, live ? gv : 0.f);
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 12);
    // ---- d t2 = d ff1 . W1 + d t3_pre ; norm2 backward ; dropout2'
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgF, kD3P4, li, q, 0, b1, acc0, acc1);                               // c4
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b1, D.pl_w1_t, 16, wave, 12, lane);                                   // c7
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kD3P4, li, q, 4, b2, acc0, acc1);                               // c5
    __builtin_amdgcn_sched_barrier(0);
    bf3_bload<4>(b2, D.pl_wo_t, 4, wave, 0, lane);                                     // c8
    __builtin_amdgcn_sched_barrier(0);
    bf3_chunk<4>(imgF, kD3P4, li, q, 8, b0, acc0, acc1);                               // c6
    bf3_chunk<4>(imgF, kD3P4, li, q, 12, b1, acc0, acc1);                              // c7
    {
        float d[4], xh[4], g[4], gx[4], s1v[4], s2v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = live ? (acc0[i] + acc1[i]) + res[i] : 0.f;
            xh[i] = (t2pv[i] - m2v[i]) * r2v[i];
            g[i] = d[i] * g2;
            gx[i] = g[i] * xh[i];
        }
        if (live) {
            float* pp = D.part_d2 + (size_t)(2 * b + q) * (2 * H);
            pp[c] = (d[0] * xh[0] + d[1] * xh[1]) + (d[2] * xh[2] + d[3] * xh[3]);
            pp[H + c] = (d[0] + d[1]) + (d[2] + d[3]);
        }
        fb_rowsum2<4>(g, gx, redA, wave, li, q, s1v, s2v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float o = r2v[i] * (g[i] - s1v[i] - xh[i] * s2v[i]);
            const size_t e = (size_t)(row0 + 4 * q + i) * H + c;
            const float ok = o * md2.keep(kb2[i], D.drop_scale);
            if (live) { D.d_t2pre[e] = o; D.d_cap[e] = ok; }
            bf3_store1(imgA, kD3P1, 4 * q + i, c, live ? ok : 0.f);
        }
    }
    __syncthreads();
    R3D_CHAIN_MARK(D.timeline, b == 0, 13);
    // ---- d ca_o = d cap . Wo, then the attention core's backward
    acc0 = zero; acc1 = zero;
    bf3_chunk<4>(imgA, kD3P1, li, q, 0, b2, acc0, acc1);                               // c8
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) D.d_cao[(size_t)(row0 + 4 * q + i) * H + c] = acc0[i] + acc1[i];
    }
    R3D_CHAIN_MARK(D.timeline, b == 0, 14);
    __syncthreads();                                        // d ca_o rows of the clip are written (workgroup scope)
    {
        MhaArgs m{};
        m.q = D.caq; m.ldq = H; m.k = D.cakv; m.ldk = 2 * H; m.v = D.cakv + H; m.ldv = 2 * H;
        m.probs = D.p_ca; m.drop = D.drop_ca; m.drop_scale = D.drop_scale; m.d_o = D.d_cao; m.lddo = H;
        m.dq = D.d_caq; m.lddq = H; m.dk = D.d_cakv; m.lddk = 2 * H; m.dv = D.d_cakv + H; m.lddv = 2 * H;
        m.B = D.B; m.heads = 8; m.Lq = 8; m.Lk = D.S; m.dh = 16; m.scale = 0.25f;
        mha_bwd_small_unit<16, 8, false>(m, b * 8 + wave, scr);
    }
    R3D_CHAIN_MARK(D.timeline, b == 0, 15);
}

__global__ __launch_bounds__(512) void decoder_chain_bf3_kernel(const DcArgs D, const r3d_tail_losses_args t, const LossArgs a,
                                                                float* part, unsigned* arrivals) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float lg[8][kTLHeads + 8];
    __shared__ float dl[8][kTLHeads + 8];
    __shared__ float red[8][4][128];
    __shared__ int is_last;
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < D.B) {
        const int b = (int)blockIdx.x;
        if (D.phases & 1) dc3_fwd(D, b, lds);
        R3D_CHAIN_MARK(D.timeline, b == 0, 6);
        if (D.phases & 2) {
            __syncthreads();
            tail_clip_body(t, a, part, b, lg, dl, red);
        }
        R3D_CHAIN_MARK(D.timeline, b == 0, 7);
        if (D.phases & 4) {
            __syncthreads();
            dc3_bwd(D, b, lds);
        }
    } else if (D.phases & 2) {
        const int u = ((int)blockIdx.x - D.B) * 8 + wave;
        if (u < a.B * a.S) losses_unit(a, part, u, lane);
    }
    if (D.phases & 2) tail_losses_finish(t, a, part, arrivals, &is_last);
}

__global__ __launch_bounds__(512) void decoder_chain_kernel(const DcArgs D, const r3d_tail_losses_args t, const LossArgs a,
                                                            float* part, unsigned* arrivals) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // The two waves a SIMD hosts (w and w + 4) run the same stage sequence in lockstep and would want the matrix core, the
    // LDS and the VALU at the same moments; a fixed priority difference lets one burst its MFMAs while the other is in its
    // epilogue / staging, so the phases of the pair interleave instead of colliding.
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    __shared__ float lg[8][kTLHeads + 8];
    __shared__ float dl[8][kTLHeads + 8];
    __shared__ float red[8][4][128];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < D.B) {
        const int b = (int)blockIdx.x;
        if (D.phases & 1) dc_fwd(D, b, lds);
        R3D_CHAIN_MARK(D.timeline, b == 0, 6);
        if (D.phases & 2) {
            __syncthreads();                                // t3_pre rows of the clip are written (workgroup scope)
            tail_clip_body(t, a, part, b, lg, dl, red);
        }
        R3D_CHAIN_MARK(D.timeline, b == 0, 7);
        if (D.phases & 4) {
            __syncthreads();                                // ... and the tail's two gradients
            dc_bwd(D, b, lds);
        }
    } else if (D.phases & 2) {
        const int u = ((int)blockIdx.x - D.B) * 8 + wave;
        if (u < a.B * a.S) losses_unit(a, part, u, lane);
    }
    if (D.phases & 2) tail_losses_finish(t, a, part, arrivals, &is_last);
}

}  // namespace r3d

using namespace r3d;

R3D_EXPORT int r3d_decoder_tail_losses_supported(int H, int n_head, int Q, int rows);

/* 1 when one clip's query side fits the chain kernel: hidden 128, 8 queries, 8 heads, <= 64 keys. */
R3D_EXPORT int r3d_decoder_chain_supported(int H, int Q, int heads, int S) {
    return (H == kFcH && Q == 8 && heads == 8 && S > 0 && S <= 64) ? 1 : 0;
}

R3D_EXPORT int r3d_decoder_chain(const r3d_decoder_chain_args* d, const r3d_tail_losses_args* tail, float* ws, void* stream) {
    R3D_REQUIRE(d && (d->phases & 7) && !(d->phases & ~7));
    R3D_REQUIRE(r3d_decoder_chain_supported(d->H, d->Q, d->heads, d->S) && d->B > 0);
    R3D_REQUIRE(d->caq && d->cakv && d->p_ca && d->wo && d->w1 && d->w2 && d->t2_pre && d->m2 && d->r2 && d->g2 && d->ff1);
    if (d->phases & 1)
        R3D_REQUIRE(d->ca_o && d->bo && d->t1 && d->be2 && d->t2 && d->b1 && d->b2 && d->t3_pre);
    if (d->phases & 4)
        R3D_REQUIRE(d->d_t3pre && d->d_ff2 && d->d_ff1 && d->d_t2pre && d->d_cap && d->d_cao && d->d_caq && d->d_cakv && d->part_d2);
    const void* al[] = {d->caq, d->cakv, d->wo, d->w1, d->w2, d->ca_o, d->d_ff2, d->d_cao, d->d_cakv};
    for (const void* p : al)
        if (p && !r3d_aligned16(p)) return R3D_EALIGN;
    r3d_tail_losses_args t{};
    LossArgs a{};
    int grid = d->B;
    unsigned* arrivals = nullptr;
    if (d->phases & 2) {
        R3D_REQUIRE(tail && ws);
        t = *tail;
        R3D_REQUIRE(t.x && t.g3 && t.b3 && t.gF && t.bF && t.w_head && t.b_head && t.t3 && t.m3 && t.r3 && t.tgtF && t.mF && t.rF);
        R3D_REQUIRE(t.out && t.seg && t.past_label && t.target && t.target_dur && t.d_seg && t.d_out && t.loss_out && t.counts);
        R3D_REQUIRE(t.dx && t.dx2 && t.wsF && t.ws3);
        R3D_REQUIRE(t.B == d->B && t.S == d->S && t.Q == d->Q && t.H == d->H && t.K > 0 && t.n_head == t.K + 1);
        R3D_REQUIRE(t.ld_out >= t.n_head && t.ld_dout >= t.n_head && t.ld_seg >= t.K && t.ld_dseg >= t.K);
        if (!r3d_decoder_tail_losses_supported(t.H, t.n_head, t.Q, t.B * t.Q)) return R3D_EINVAL;
        if (!r3d_aligned16(ws)) return R3D_EALIGN;
        if (d->phases & 1) R3D_REQUIRE(t.x == d->t3_pre);
        if (d->phases & 4) R3D_REQUIRE(t.dx == d->d_t3pre && t.dx2 == d->d_ff2);
        a = LossArgs{t.seg, t.ld_seg, t.out, t.ld_out, t.out + t.K, t.ld_out, t.past_label, t.target, t.target_dur, t.B, t.S, t.Q,
                     t.K, t.pad_idx, t.exclude_idx, 0, t.dur_den, t.grad_scale, t.d_seg, t.ld_dseg, t.d_out, t.ld_dout,
                     t.d_out + t.K, t.ld_dout, t.loss_out, t.counts, t.tick_a, t.tick_b};
        const int units = t.B * t.S + t.B * t.Q + t.B;
        grid = t.B + r3d_cdiv(t.B * t.S, 8);
        arrivals = reinterpret_cast<unsigned*>(ws + 4 * (size_t)units);
    }
    if (d->pl_wo) {               // operand-order bf16x3 planes of the weights (and their transposes): bf16 matrix cores
        R3D_REQUIRE(d->pl_w1 && d->pl_w2 && d->pl_w2_t && d->pl_w1_t && d->pl_wo_t);
        hipError_t e3 = hipFuncSetAttribute((const void*)decoder_chain_bf3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kD3LdsBytes);
        if (e3 != hipSuccess) return (int)e3;
        hipLaunchKernelGGL(decoder_chain_bf3_kernel, dim3(grid), dim3(512), (size_t)kD3LdsBytes, (hipStream_t)stream, *d, t, a, ws,
                           arrivals);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    hipError_t e = hipFuncSetAttribute((const void*)decoder_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDcLdsBytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(decoder_chain_kernel, dim3(grid), dim3(512), (size_t)kDcLdsBytes, (hipStream_t)stream, *d, t, a, ws, arrivals);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
