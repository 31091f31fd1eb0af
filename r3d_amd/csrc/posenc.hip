// Row-local kernels of the depth-as-query model (model/futr_unsupervised_depth.py of the reference):
//   * sinusoidal PositionalEncoding + dropout (model/extras/position.py:29-35):  y = drop(x + pos_table[s]),
//     applied to the RGB embedding (futr_unsupervised_depth.py:99) and to the depth embedding that becomes the decoder's
//     query (:115); backward = the dropout mask (and, for the RGB branch, the ReLU gate of :97 in the same pass);
//   * F.adaptive_avg_pool1d over the S decoder outputs of a clip down to n_query rows (:134): window of output q is
//     [floor(q S / Q), ceil((q + 1) S / Q)), and its adjoint.
// Rows are ordered (clip, frame) b-major like everywhere in this library: row = b * S + s.
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

__global__ __launch_bounds__(256) void posenc_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ table,
                                                         int ldt, int S, const uint8_t* __restrict__ drop, float drop_scale,
                                                         float* __restrict__ y, int ldy, int rows, int H) {
    const size_t total = (size_t)rows * H;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / H), c = (int)(e - (size_t)r * H);
        float v = x[(size_t)r * ldx + c] + table[(size_t)(r % S) * ldt + c];
        if (drop) v = drop[e] ? v * drop_scale : 0.f;
        y[(size_t)r * ldy + c] = v;
    }
}

// dx = dy * mask * scale, gated by (gate > 0) when gate != NULL (gate = the post-ReLU embedding)
__global__ __launch_bounds__(256) void posenc_bwd_kernel(const float* __restrict__ dy, int lddy, const uint8_t* __restrict__ drop,
                                                         float drop_scale, const float* __restrict__ gate, int ldg,
                                                         float* __restrict__ dx, int lddx, int rows, int H) {
    const size_t total = (size_t)rows * H;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / H), c = (int)(e - (size_t)r * H);
        float v = dy[(size_t)r * lddy + c];
        if (drop) v = drop[e] ? v * drop_scale : 0.f;
        if (gate && !(gate[(size_t)r * ldg + c] > 0.f)) v = 0.f;
        dx[(size_t)r * lddx + c] = v;
    }
}

__device__ __forceinline__ int pool_start(int q, int S, int Q) { return (int)(((long long)q * S) / Q); }
__device__ __forceinline__ int pool_end(int q, int S, int Q) { return (int)((((long long)(q + 1)) * S + Q - 1) / Q); }

// y[b*Q + q, :] = mean_{s in window(q)} x[b*S + s, :]
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                          int B, int S, int Q, int H) {
    const size_t total = (size_t)B * Q * H;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int row = (int)(e / H), c = (int)(e - (size_t)row * H);
        const int b = row / Q, q = row - b * Q;
        const int s0 = pool_start(q, S, Q), s1 = pool_end(q, S, Q);
        float acc = 0.f;
        for (int s = s0; s < s1; ++s) acc += x[((size_t)b * S + s) * ldx + c];
        y[(size_t)row * ldy + c] = acc / (float)(s1 - s0);
    }
}

// dx[b*S + s, :] = sum_{q : s in window(q)} dy[b*Q + q, :] / len(q)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                          int B, int S, int Q, int H) {
    const size_t total = (size_t)B * S * H;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int row = (int)(e / H), c = (int)(e - (size_t)row * H);
        const int b = row / S, s = row - b * S;
        float acc = 0.f;
        for (int q = 0; q < Q; ++q) {
            const int s0 = pool_start(q, S, Q), s1 = pool_end(q, S, Q);
            if (s >= s0 && s < s1) acc += dy[((size_t)b * Q + q) * lddy + c] / (float)(s1 - s0);
        }
        dx[(size_t)row * lddx + c] = acc;
    }
}

// Label-index queries of model/futr_proposed.py:103-106: out[r, :] = weight[idx[r], :] + table[r % S, :] (nn.Embedding lookup +
// the sinusoidal table), and the lookup's adjoint d_weight[e, :] = sum over rows with idx[r] == e of d_out[r, :] -- one
// workgroup per embedding row scanning the (few hundred) indices: deterministic, no atomics.
__global__ __launch_bounds__(256) void embed_gather_fwd_kernel(const float* __restrict__ weight, int n_embed, const int64_t* __restrict__ idx,
                                                               const float* __restrict__ table, int ldt, int S, float* __restrict__ out,
                                                               int ldo, int rows, int H) {
    const size_t total = (size_t)rows * H;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / H), c = (int)(e - (size_t)r * H);
        int64_t i = idx[r];
        i = i < 0 ? 0 : (i >= n_embed ? n_embed - 1 : i);
        out[(size_t)r * ldo + c] = weight[(size_t)i * H + c] + table[(size_t)(r % S) * ldt + c];
    }
}

__global__ __launch_bounds__(256) void embed_gather_bwd_kernel(const float* __restrict__ d_out, int ldd, const int64_t* __restrict__ idx,
                                                               float* __restrict__ d_weight, int rows, int H) {
    const int e = blockIdx.x;
    for (int c = threadIdx.x; c < H; c += 256) {
        float acc = 0.f;
        for (int r = 0; r < rows; ++r)
            if (idx[r] == (int64_t)e) acc += d_out[(size_t)r * ldd + c];
        d_weight[(size_t)e * H + c] = acc;
    }
}

static inline int ew_blocks(size_t total) { return (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048); }

}  // namespace r3d

using namespace r3d;

/* y[r, :] = dropout(x[r, :] + table[r % S, :]) -- PositionalEncoding.forward (position.py:29-35) on b-major rows.
 * table: [>= S][ldt] sinusoid buffer (pos_table); drop: NULL or rows*H keep-bytes (contiguous), scale 1/(1-p). */
R3D_EXPORT int r3d_posenc_fwd(const float* x, int ldx, const float* table, int ldt, int S, const uint8_t* drop, float drop_scale,
                              float* y, int ldy, int rows, int H, void* stream) {
    R3D_REQUIRE(x && table && y && rows > 0 && H > 0 && S > 0 && ldx >= H && ldy >= H && ldt >= H);
    hipLaunchKernelGGL(posenc_fwd_kernel, dim3(ew_blocks((size_t)rows * H)), dim3(256), 0, (hipStream_t)stream, x, ldx, table,
                       ldt, S, drop, drop_scale, y, ldy, rows, H);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* dx = dy * keep * scale, zeroed where gate <= 0 (gate: NULL or the post-ReLU activation the encoding was added to). */
R3D_EXPORT int r3d_posenc_bwd(const float* dy, int lddy, const uint8_t* drop, float drop_scale, const float* gate, int ldg,
                              float* dx, int lddx, int rows, int H, void* stream) {
    R3D_REQUIRE(dy && dx && rows > 0 && H > 0 && lddy >= H && lddx >= H && (!gate || ldg >= H));
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3(ew_blocks((size_t)rows * H)), dim3(256), 0, (hipStream_t)stream, dy, lddy, drop,
                       drop_scale, gate, ldg, dx, lddx, rows, H);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* F.adaptive_avg_pool1d over the S rows of each of B clips down to Q rows (futr_unsupervised_depth.py:134). */
R3D_EXPORT int r3d_avgpool_rows_fwd(const float* x, int ldx, float* y, int ldy, int B, int S, int Q, int H, void* stream) {
    R3D_REQUIRE(x && y && B > 0 && S > 0 && Q > 0 && H > 0 && ldx >= H && ldy >= H);
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ew_blocks((size_t)B * Q * H)), dim3(256), 0, (hipStream_t)stream, x, ldx, y,
                       ldy, B, S, Q, H);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_avgpool_rows_bwd(const float* dy, int lddy, float* dx, int lddx, int B, int S, int Q, int H, void* stream) {
    R3D_REQUIRE(dy && dx && B > 0 && S > 0 && Q > 0 && H > 0 && lddy >= H && lddx >= H);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks((size_t)B * S * H)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx,
                       lddx, B, S, Q, H);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* out[r, :] = weight[idx[r], :] + table[r % S, :] -- nn.Embedding lookup + sinusoidal table (futr_proposed.py:103-106);
 * indices are clamped into [0, n_embed). */
R3D_EXPORT int r3d_embed_gather_fwd(const float* weight, int n_embed, const int64_t* idx, const float* table, int ldt, int S,
                                    float* out, int ldo, int rows, int H, void* stream) {
    R3D_REQUIRE(weight && idx && table && out && n_embed > 0 && rows > 0 && H > 0 && S > 0 && ldt >= H && ldo >= H);
    hipLaunchKernelGGL(embed_gather_fwd_kernel, dim3(ew_blocks((size_t)rows * H)), dim3(256), 0, (hipStream_t)stream, weight,
                       n_embed, idx, table, ldt, S, out, ldo, rows, H);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* d_weight[e, :] = sum_{r : idx[r] == e} d_out[r, :]  for every embedding row e (rows never looked up get zeros). */
R3D_EXPORT int r3d_embed_gather_bwd(const float* d_out, int ldd, const int64_t* idx, float* d_weight, int n_embed, int rows,
                                    int H, void* stream) {
    R3D_REQUIRE(d_out && idx && d_weight && n_embed > 0 && rows > 0 && H > 0 && ldd >= H);
    hipLaunchKernelGGL(embed_gather_bwd_kernel, dim3(n_embed), dim3(256), 0, (hipStream_t)stream, d_out, ldd, idx, d_weight, rows, H);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
