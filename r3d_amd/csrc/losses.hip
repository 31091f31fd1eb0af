// The three losses of the training step, their counters and their gradients in ONE launch, with no host sync:
//   segmentation CE   cal_loss            utils.py:449-490   (called at train/train_proposed_depth.py:181)
//   anticipation CE   cal_weighted_loss   utils.py:410-447   (train_proposed_depth.py:195, weights from
//                     get_last_non_padding_labels :28-50)
//   duration MSE      normalize_duration  utils.py:325-328 + train_proposed_depth.py:204-207
//   n_correct/n_word  cal_performance     utils.py:368-376
// The reference issues >= 16 device->host syncs per step here (.item() x8 + a python loop over clips); this kernel
// leaves 4 losses and 4 counters in device memory.  One 512-thread workgroup (rows are few: N + B*Q), wave per row,
// fixed reduction order -> bitwise reproducible.
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
};

// CE of one row held across a wave.  Returns loss contribution; writes gradient (softmax - onehot) * gscale if dl.
__device__ __forceinline__ float ce_row(const float* logit, int K, int64_t label, bool valid, int pad_idx,
                                        float gscale, float* dl, int lane, int* argmax_out) {
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < K; c += 64) {
        const float x = logit[c];
        if (x > m || (x == m && c < am)) { m = x; am = c; }
    }
    // wave arg-max: largest value, smallest index among equals (first occurrence, as torch.max on CPU)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64);
        const int oa = __shfl_xor(am, off, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
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

__global__ __launch_bounds__(512) void losses_kernel(const LossArgs a) {
    __shared__ double red[8][4];
    __shared__ long long cred[8][4];
    __shared__ float wclip[4096];
    __shared__ int msum_sh;
    __shared__ int wmask[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.B * a.S, BQ = a.B * a.Q;

    // ---- pre-pass: per-clip CE weight (last observed label vs first future label) and the duration mask sum
    for (int b = wave; b < a.B; b += 8) {
        int last = -1;
        for (int s = lane; s < a.S; s += 64)
            if (a.past_label[(size_t)b * a.S + s] != (int64_t)a.pad_idx) last = s;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
        const int64_t ref = (last >= 0) ? a.past_label[(size_t)b * a.S + last] : (int64_t)a.pad_idx;
        if (lane == 0) wclip[b] = (ref == a.target[(size_t)b * a.Q]) ? 1.0f : 10.0f;
    }
    int mcount = 0;
    for (int e = threadIdx.x; e < BQ; e += 512) mcount += (a.target_dur[e] != (float)a.pad_idx) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mcount += __shfl_xor(mcount, off, 64);
    if (lane == 0) wmask[wave] = mcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 8; ++w) t += wmask[w];
        msum_sh = t;
    }
    __syncthreads();
    const float msum = (float)msum_sh;
    const float dur_den = a.dur_den ? *a.dur_den : msum;

    double l_seg = 0.0, l_act = 0.0, l_dur = 0.0;
    long long c_seg = 0, n_seg = 0, c_act = 0, n_act = 0;

    // ---- segmentation rows
    if (a.seg) {
        const float inv = 1.0f / (float)N;
        for (int r = wave; r < N; r += 8) {
            const int64_t lab = a.past_label[r];
            // labels outside [0,K) that are neither pad nor excluded would raise in PyTorch; here they are ignored
            const bool valid = (lab != (int64_t)a.pad_idx) && (lab != (int64_t)a.exclude_idx) && lab >= 0 && lab < a.K;
            int am;
            float l = ce_row(a.seg + (size_t)r * a.ld_seg, a.K, lab, valid, a.pad_idx, inv * a.grad_scale,
                             a.d_seg ? a.d_seg + (size_t)r * a.ld_dseg : nullptr, lane, &am);
            if (valid && am == a.pad_idx) l += 2.0f;          // penalty term of cal_loss (utils.py:481-486)
            if (lane == 0) {
                l_seg += (double)l;
                n_seg += valid ? 1 : 0;
                c_seg += (valid && (int64_t)am == lab) ? 1 : 0;
            }
        }
    }
    // ---- anticipation rows
    {
        const float inv = 1.0f / (float)BQ;
        for (int r = wave; r < BQ; r += 8) {
            const int64_t lab = a.target[r];
            const bool valid = (lab != (int64_t)a.pad_idx) && (lab != (int64_t)a.exclude_idx) && lab >= 0 && lab < a.K;
            const float w = wclip[r / a.Q];
            int am;
            const float l = ce_row(a.act + (size_t)r * a.ld_act, a.K, lab, valid, a.pad_idx, w * inv * a.grad_scale,
                                   a.d_act ? a.d_act + (size_t)r * a.ld_dact : nullptr, lane, &am);
            if (lane == 0) {
                l_act += (double)(l * w);
                n_act += valid ? 1 : 0;
                c_act += (valid && (int64_t)am == lab) ? 1 : 0;
            }
        }
    }
    // ---- duration: one wave per clip, lane = query
    for (int b = wave; b < a.B; b += 8) {
        float sq = 0.f, gp = 0.f;
        // pass 1: normaliser
        float ssum = 0.f;
        for (int q = lane; q < a.Q; q += 64) {
            const float td = a.target_dur[(size_t)b * a.Q + q];
            const float mk = (td != (float)a.pad_idx) ? 1.f : 0.f;
            ssum += fabsf(expf(a.dur[((size_t)b * a.Q + q) * a.ld_dur]) * mk);
        }
        ssum = wave_sum(ssum);
        const float den = fmaxf(ssum, 1e-12f);
        // pass 2: loss and sum_i g_i p_i
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
        if (lane == 0) l_dur += (double)sq;
    }
    if (lane == 0) {
        red[wave][0] = l_seg; red[wave][1] = l_act; red[wave][2] = l_dur; red[wave][3] = 0.0;
        cred[wave][0] = c_seg; cred[wave][1] = n_seg; cred[wave][2] = c_act; cred[wave][3] = n_act;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0;
        long long k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        for (int w = 0; w < 8; ++w) {
            s0 += red[w][0]; s1 += red[w][1]; s2 += red[w][2];
            k0 += cred[w][0]; k1 += cred[w][1]; k2 += cred[w][2]; k3 += cred[w][3];
        }
        const float ls = a.seg ? (float)(s0 / (double)N) : 0.f;
        const float la = (float)(s1 / (double)BQ);
        const float ld = (float)(s2 / (double)dur_den);
        a.loss_out[0] = ls; a.loss_out[1] = la; a.loss_out[2] = ld; a.loss_out[3] = ls + la + ld;
        a.counts[0] = k0; a.counts[1] = k1; a.counts[2] = k2; a.counts[3] = k3;
    }
}

}  // namespace r3d

using namespace r3d;

/* loss_out[4] = {seg CE, weighted action CE, duration MSE, total}; counts[4] = {seg correct, seg total, action
 * correct, action total}.  Gradient buffers are optional (validate()); they receive d(total)/d(logit) * grad_scale.
 * dur_den (device scalar, optional) overrides the local duration-mask sum: under data parallelism the caller passes
 * (global mask sum / world size) so that the average of rank gradients equals the gradient of the reference's
 * global-batch loss.  val_mode=1 reproduces validate()'s unmasked duration target
 * (train_proposed_depth.py:98-99) and ignores seg when seg == NULL. */
R3D_EXPORT int r3d_losses_fwd_bwd(const float* seg_logits, int ld_seg, const float* act_logits, int ld_act,
                                  const float* dur, int ld_dur, const int64_t* past_label, const int64_t* target,
                                  const float* target_dur, int B, int S, int Q, int K, int pad_idx, int exclude_idx,
                                  int val_mode, const float* dur_den, float grad_scale, float* d_seg, int ld_dseg,
                                  float* d_act, int ld_dact, float* d_dur, int ld_ddur, float* loss_out,
                                  int64_t* counts, void* stream) {
    R3D_REQUIRE(act_logits && dur && past_label && target && target_dur && loss_out && counts);
    R3D_REQUIRE(B > 0 && B <= 4096 && S > 0 && Q > 0 && K > 0);
    R3D_REQUIRE(ld_act >= K && (!seg_logits || ld_seg >= K) && ld_dur >= 1);
    R3D_REQUIRE(!d_seg || (seg_logits && ld_dseg >= K));
    R3D_REQUIRE(!d_act || ld_dact >= K);
    R3D_REQUIRE(!d_dur || ld_ddur >= 1);
    LossArgs a{seg_logits, ld_seg, act_logits, ld_act, dur, ld_dur, past_label, target, target_dur, B, S, Q, K, pad_idx,
               exclude_idx, val_mode, dur_den, grad_scale, d_seg, ld_dseg, d_act, ld_dact, d_dur, ld_ddur, loss_out,
               counts};
    hipLaunchKernelGGL(losses_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
