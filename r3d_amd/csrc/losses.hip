// The three losses of the training step, their counters and their gradients in ONE launch, with no host sync:
//   segmentation CE   cal_loss            utils.py:449-490   (called at train/train_proposed_depth.py:181)
//   anticipation CE   cal_weighted_loss   utils.py:410-447   (train_proposed_depth.py:195, weights from
//                     get_last_non_padding_labels :28-50)
//   duration MSE      normalize_duration  utils.py:325-328 + train_proposed_depth.py:204-207
//   n_correct/n_word  cal_performance     utils.py:368-376
// The reference issues >= 16 device->host syncs per step here (.item() x8 + a python loop over clips); this kernel
// leaves 4 losses and 4 counters in device memory: a row-parallel launch (one wave per logits row / clip) and a
// one-workgroup finaliser with a fixed reduction order -> bitwise reproducible.
#include "losses_dev.h"
#include "loss_finalize.h"

namespace r3d {

// One launch: every workgroup finishes its 4 units and publishes their partials with agent-scope write-through stores
// (drained with vmcnt(0) before its arrival is counted); the LAST workgroup to arrive reads them back with agent-scope
// loads and adds them up in a fixed order -- bitwise reproducible, no second dependent launch (~5 us on this part) for a
// result the backward does not even consume, and no whole-L2 write-back fence (measured 13 us with __threadfence).
__global__ __launch_bounds__(256) void losses_kernel(const LossArgs a, float* part, unsigned* arrivals) {
    __shared__ int is_last;
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u < a.B * a.S + a.B * a.Q + a.B) losses_unit(a, part, u, lane);
    __builtin_amdgcn_s_waitcnt(0);                     // this wave's write-through stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = (__hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!is_last) return;
    losses_finalize(a, part);
    if (threadIdx.x == 0) *arrivals = 0u;             // ready for the next launch
}


// ---------------------------------------------------------------------------------------------------------
// The decoder's tail, the losses and the tail's backward in ONE launch (training step, hidden <= 128, <= 24 head outputs,
// 8 queries per clip): three dependent row-local launches (r3d_decoder_tail_fwd -> r3d_losses_fwd_bwd ->
// r3d_decoder_tail_bwd, 8.7 + 9.2 + 6.6 us) whose only cross-row coupling is inside a clip (the duration normalisation
// over its Q queries) or a scalar every workgroup can recompute (the duration-mask count).
//   workgroups [0, B)        : one clip each, wave q = query row b*Q + q:
//        norm3 -> decoder.norm -> heads (logits also kept in LDS)  | barrier |  action CE + this row's duration gradient
//        (from the clip's Q duration logits in LDS)  ->  heads' input gradient -> decoder.norm backward -> norm3 backward
//        (+ dropout3), LayerNorm parameter partials folded per 4 rows = r3d_layernorm_bwd's layout for B*Q rows.
//   workgroups [B, B + N/8)  : the segmentation rows, one wave each (losses_unit).
//   the last workgroup to arrive adds the loss partials in a fixed order (as losses_kernel).
// Same arithmetic, same order per row as the three kernels it replaces.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void tail_losses_kernel(const r3d_tail_losses_args t, const LossArgs a, float* part,
                                                          unsigned* arrivals) {
    __shared__ float lg[8][kTLHeads + 8];            // logits of the clip's rows; column n_head-1 = duration
    __shared__ float dl[8][kTLHeads + 8];            // their gradients
    __shared__ float red[8][4][128];                 // LayerNorm parameter partials per wave
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < a.B) {
        tail_clip_body(t, a, part, (int)blockIdx.x, lg, dl, red);
    } else {
        const int u = ((int)blockIdx.x - a.B) * 8 + wave;
        if (u < a.B * a.S) losses_unit(a, part, u, lane);
    }
    tail_losses_finish(t, a, part, arrivals, &is_last);
}

// The same launch for hidden sizes 129 .. 512 (tail_clip_body_wide<8>): head weights and LayerNorm partials in dynamic LDS.
__global__ __launch_bounds__(512) void tail_losses_wide_kernel(const r3d_tail_losses_args t, const LossArgs a, float* part,
                                                               unsigned* arrivals) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];       // [n_head][H] head weights | [8][4][H] partials
    __shared__ float lg[8][kTLHeads + 8];
    __shared__ float dl[8][kTLHeads + 8];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < a.B) {
        float* whs = dyn;
        float* red = dyn + (size_t)t.n_head * t.H;
        for (int i = threadIdx.x; i < t.n_head * t.H; i += 512) whs[i] = t.w_head[i];
        __syncthreads();
        tail_clip_body_wide<8>(t, a, part, (int)blockIdx.x, lg, dl, whs, red);
    } else {
        const int u = ((int)blockIdx.x - a.B) * 8 + wave;
        if (u < a.B * a.S) losses_unit(a, part, u, lane);
    }
    tail_losses_finish(t, a, part, arrivals, &is_last);
}

__global__ __launch_bounds__(256) void losses_finalize_kernel(const r3d_loss_finalize_job j) {
    __shared__ double red[4][3][3];
    loss_finalize_block<4>(j, red);
}

}  // namespace r3d

using namespace r3d;

/* loss_out[4] = {seg CE, weighted action CE, duration MSE, total}; counts[4] = {seg correct, seg total, action
 * correct, action total}.  Gradient buffers are optional (validate()); they receive d(total)/d(logit) * grad_scale.
 * dur_den (device scalar, optional) overrides the local duration-mask sum: under data parallelism the caller passes
 * (global mask sum / world size) so that the average of rank gradients equals the gradient of the reference's
 * global-batch loss.  val_mode=1 reproduces validate()'s unmasked duration target
 * (train_proposed_depth.py:98-99) and ignores seg when seg == NULL. */
/* ws: r3d_losses_ws_floats(B, S, Q) floats of scratch, 16-byte aligned, ZERO before the first call (its last 16 bytes
 * hold the arrival counter, which every launch leaves at zero again).  tick_a / tick_b (optional device int64): both
 * are incremented once per call -- the caller's step counter and dropout offset ride along instead of costing a launch. */
R3D_EXPORT int64_t r3d_losses_ws_floats(int B, int S, int Q) { return 4ll * ((int64_t)B * S + (int64_t)B * Q + B) + 4; }

R3D_EXPORT int r3d_losses_fwd_bwd(const float* seg_logits, int ld_seg, const float* act_logits, int ld_act,
                                  const float* dur, int ld_dur, const int64_t* past_label, const int64_t* target,
                                  const float* target_dur, int B, int S, int Q, int K, int pad_idx, int exclude_idx,
                                  int val_mode, const float* dur_den, float grad_scale, float* d_seg, int ld_dseg,
                                  float* d_act, int ld_dact, float* d_dur, int ld_ddur, float* loss_out,
                                  int64_t* counts, float* ws, int64_t* tick_a, int64_t* tick_b, void* stream) {
    R3D_REQUIRE(act_logits && dur && past_label && target && target_dur && loss_out && counts && ws);
    if (!r3d_aligned16(ws)) return R3D_EALIGN;
    R3D_REQUIRE(B > 0 && S > 0 && Q > 0 && K > 0);
    R3D_REQUIRE(ld_act >= K && (!seg_logits || ld_seg >= K) && ld_dur >= 1);
    R3D_REQUIRE(!d_seg || (seg_logits && ld_dseg >= K));
    R3D_REQUIRE(!d_act || ld_dact >= K);
    R3D_REQUIRE(!d_dur || ld_ddur >= 1);
    LossArgs a{seg_logits, ld_seg, act_logits, ld_act, dur, ld_dur, past_label, target, target_dur, B, S, Q, K, pad_idx,
               exclude_idx, val_mode, dur_den, grad_scale, d_seg, ld_dseg, d_act, ld_dact, d_dur, ld_ddur, loss_out,
               counts, tick_a, tick_b};
    const int units = B * S + B * Q + B;
    hipLaunchKernelGGL(losses_kernel, dim3(r3d_cdiv(units, 4)), dim3(256), 0, (hipStream_t)stream, a, ws,
                       reinterpret_cast<unsigned*>(ws + 4 * (size_t)units));
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Training step only: r3d_decoder_tail_fwd + r3d_losses_fwd_bwd + r3d_decoder_tail_bwd as one launch (see
 * tail_losses_kernel).  Supported when r3d_decoder_tail_losses_supported(...) != 0; ws as for r3d_losses_fwd_bwd. */
R3D_EXPORT int r3d_decoder_tail_losses_supported(int H, int n_head, int Q, int rows) {
    return (H > 0 && H <= 512 && n_head > 0 && n_head <= kTLHeads && Q == 8 && rows > 0 && rows <= 1024 && rows % 8 == 0) ? 1 : 0;
}

R3D_EXPORT int r3d_decoder_tail_losses(const r3d_tail_losses_args* p, float* ws, void* stream) {
    R3D_REQUIRE(p && ws);
    const r3d_tail_losses_args& t = *p;
    R3D_REQUIRE(t.x && t.g3 && t.b3 && t.gF && t.bF && t.w_head && t.b_head && t.t3 && t.m3 && t.r3 && t.tgtF && t.mF && t.rF);
    R3D_REQUIRE(t.out && t.seg && t.past_label && t.target && t.target_dur && t.d_seg && t.d_out && t.loss_out && t.counts);
    R3D_REQUIRE(t.dx && t.dx2 && t.wsF && t.ws3);
    R3D_REQUIRE(t.B > 0 && t.S > 0 && t.K > 0 && t.n_head == t.K + 1 && t.ld_out >= t.n_head && t.ld_dout >= t.n_head);
    R3D_REQUIRE(t.ld_seg >= t.K && t.ld_dseg >= t.K);
    if (!r3d_decoder_tail_losses_supported(t.H, t.n_head, t.Q, t.B * t.Q)) return R3D_EINVAL;
    if (!r3d_aligned16(ws)) return R3D_EALIGN;
    LossArgs a{t.seg, t.ld_seg, t.out, t.ld_out, t.out + t.K, t.ld_out, t.past_label, t.target, t.target_dur, t.B, t.S, t.Q,
               t.K, t.pad_idx, t.exclude_idx, 0, t.dur_den, t.grad_scale, t.d_seg, t.ld_dseg, t.d_out, t.ld_dout,
               t.d_out + t.K, t.ld_dout, t.loss_out, t.counts, t.tick_a, t.tick_b};
    const int units = t.B * t.S + t.B * t.Q + t.B;
    const int grid = t.B + r3d_cdiv(t.B * t.S, 8);
    if (t.H > 128) {                      // hidden 129 .. 512: head weights + LayerNorm partials in dynamic LDS
        const size_t lds = ((size_t)t.n_head * t.H + (size_t)8 * 4 * t.H) * sizeof(float);
        hipError_t e = hipFuncSetAttribute((const void*)tail_losses_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(tail_losses_wide_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, t, a, ws,
                           reinterpret_cast<unsigned*>(ws + 4 * (size_t)units));
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    hipLaunchKernelGGL(tail_losses_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, t, a, ws,
                       reinterpret_cast<unsigned*>(ws + 4 * (size_t)units));
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

static int loss_finalize_job_ok(const r3d_loss_finalize_job* j) {
    return j && j->part && j->loss_out && j->counts && j->B > 0 && j->S > 0 && j->Q > 0;
}

R3D_EXPORT int r3d_losses_finalize(const r3d_loss_finalize_job* job, void* stream) {
    R3D_REQUIRE(loss_finalize_job_ok(job));
    hipLaunchKernelGGL(losses_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *job);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
