// The reduction of the per-unit loss partials into loss_out[4] / counts[4] (train/train_proposed_depth.py:182-213 as
// composed from utils.py:446,489): shared by the last-arriving workgroup of the loss kernels (losses.hip) and by the extra
// workgroup of the AdamW launch (optim.hip).  One workgroup of NW waves; red: LDS [NW][3][3] doubles.
#pragma once
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

template <int NW>
__device__ __forceinline__ void loss_finalize_block(const r3d_loss_finalize_job& j, double (*red)[3][3]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = j.B * j.S, BQ = j.B * j.Q;
    double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int u = threadIdx.x; u < N + BQ + j.B; u += 64 * NW) {
        const float vx = j.part[4 * (size_t)u + 0], vy = j.part[4 * (size_t)u + 1], vz = j.part[4 * (size_t)u + 2];
        const int g = u < N ? 0 : (u < N + BQ ? 1 : 2);
        acc[g][0] += vx; acc[g][1] += vy; acc[g][2] += vz;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double t = wave_sum_d(acc[g][k]);
            if (lane == 0) red[wave][g][k] = t;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3][3];
        for (int g = 0; g < 3; ++g)
            for (int k = 0; k < 3; ++k) {
                double s = 0.0;
                for (int w = 0; w < NW; ++w) s += red[w][g][k];
                t[g][k] = s;
            }
        const double msum = t[2][2] / (double)j.B;
        const double dur_den = j.dur_den ? (double)*j.dur_den : msum;
        const float ls = j.has_seg ? (float)(t[0][0] / (double)N) : 0.f;
        const float la = (float)(t[1][0] / (double)BQ);
        const float ld = (float)(t[2][0] / dur_den);
        j.loss_out[0] = ls; j.loss_out[1] = la; j.loss_out[2] = ld; j.loss_out[3] = ls + la + ld;
        j.counts[0] = (int64_t)(t[0][1] + 0.5); j.counts[1] = (int64_t)(t[0][2] + 0.5);
        j.counts[2] = (int64_t)(t[1][1] + 0.5); j.counts[3] = (int64_t)(t[1][2] + 0.5);
        if (j.acc_loss)
            for (int i = 0; i < 4; ++i) j.acc_loss[i] += (double)j.loss_out[i];
        if (j.acc_counts)
            for (int i = 0; i < 4; ++i) j.acc_counts[i] += j.counts[i];
    }
}

}  // namespace r3d
