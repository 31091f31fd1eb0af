// Row-wise kernels: LayerNorm forward / backward (with the fusions the token-fusion step needs), column sums for
// bias gradients, row-modulo sums for broadcast-parameter gradients.  HBM-bound, one wave (64 lanes) per row,
// lane c owns columns c, c+64, ... (256-B coalesced wave accesses); reductions are wave shuffles -> no LDS traffic
// on the row path and bitwise run-to-run reproducible results (no float atomics anywhere).
#include "common.h"
#include "mha_small.h"
#include "../../include/r3d_hip.h"

namespace r3d {

constexpr float kLnEps = 1e-5f;   // nn.LayerNorm default; SURVEY.md Appendix A.2

typedef r3d_ln_fwd_job LnFwdArgs;      // public struct (include/r3d_hip.h): the multi-job entry point takes arrays of it

// Every load below is UNCONDITIONAL from a clamped column (a load under a lane-dependent branch makes hipcc wait
// vmcnt(0) right behind it) and all loads of a workgroup's rows are issued before the first reduction: the kernels are
// pure latency (their inputs were just written by the previous kernel), so what counts is ONE memory round trip.
template <int EPL>
__device__ __forceinline__ void ln_fwd_load(const LnFwdArgs& a, int row, int lane, float (&v)[EPL]) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const int cc = c < a.H ? c : a.H - 1;
        float t;
        if (a.nsplit > 0) {
            // split-K slabs: 8 independent partial sums keep 8 loads in flight (a serial chain pays one
            // memory round trip per slab)
            const float* px = a.x + (size_t)row * a.H + cc;
            const size_t stride = (size_t)a.rows * a.H;
            float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int p = 0;
            for (; p + 8 <= a.nsplit; p += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) s8[q] += px[(size_t)(p + q) * stride];
            }
            for (; p < a.nsplit; ++p) s8[0] += px[(size_t)p * stride];
            t = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
            if (a.bias) t += a.bias[cc];
            if (a.pre_out && c < a.H) a.pre_out[(size_t)row * a.H + c] = t;
        } else {
            t = a.x[(size_t)row * a.ldx + cc];
        }
        v[e] = c < a.H ? t : 0.f;
    }
}

template <int EPL>
__device__ __forceinline__ void ln_fwd_finish(const LnFwdArgs& a, int row, int lane, const float (&v)[EPL],
                                              const float (&gam)[EPL], const float (&bet)[EPL], float (&yv)[EPL]) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += v[e];
    const float mean = wave_sum(s) / (float)a.H;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const float dlt = (c < a.H) ? v[e] - mean : 0.f;
        q += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.H + kLnEps);
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        float o = 0.f;
        if (c < a.H) {
            o = (v[e] - mean) * rstd * gam[e] + bet[e];
            if (a.relu) o = fmaxf(o, 0.f);
            a.y[(size_t)row * a.ldy + c] = o;
        }
        yv[e] = o;
    }
}

struct LnFwdMulti { LnFwdArgs j[4]; };
struct LnBwdMulti;

template <int EPL>
__device__ __forceinline__ void ln_fwd_block(const LnFwdArgs& a);

template <int EPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a) { ln_fwd_block<EPL>(a); }

// Independent LayerNorm sites of equal width in ONE launch (blockIdx.y = site): what is saved is the dependent launch.
template <int EPL>
__global__ __launch_bounds__(256) void ln_fwd_multi_kernel(const LnFwdMulti m) { ln_fwd_block<EPL>(m.j[blockIdx.y]); }

template <int EPL>
__device__ __forceinline__ void ln_fwd_block(const LnFwdArgs& a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * 4 + wave;
    float gam[EPL], bet[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const int cc = c < a.H ? c : a.H - 1;
        gam[e] = a.gamma[cc];
        bet[e] = a.beta[cc];
    }
    float y0[EPL], v0[EPL];
    if (a.pair_out) {
        if (2 * unit >= a.rows) return;
        float y1[EPL], v1[EPL];
        ln_fwd_load<EPL>(a, 2 * unit, lane, v0);
        ln_fwd_load<EPL>(a, 2 * unit + 1, lane, v1);
        ln_fwd_finish<EPL>(a, 2 * unit, lane, v0, gam, bet, y0);
        ln_fwd_finish<EPL>(a, 2 * unit + 1, lane, v1, gam, bet, y1);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < a.H) a.pair_out[(size_t)unit * a.H + c] = (y0[e] + y1[e]) * 0.5f;
        }
    } else {
        if (unit >= a.rows) return;
        ln_fwd_load<EPL>(a, unit, lane, v0);
        ln_fwd_finish<EPL>(a, unit, lane, v0, gam, bet, y0);
    }
}

typedef r3d_ln_bwd_job LnBwdArgs;      // public struct; rows_per_block / nblocks are filled by the library
struct LnBwdMulti { LnBwdArgs j[4]; };

template <int EPL>
__device__ __forceinline__ void ln_bwd_block(const LnBwdArgs& a, float* red);

template <int EPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4][2][H]
    ln_bwd_block<EPL>(a, red);
}

template <int EPL>
__global__ __launch_bounds__(256) void ln_bwd_multi_kernel(const LnBwdMulti m) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4][2][H]
    ln_bwd_block<EPL>(m.j[blockIdx.y], red);
}

// ln_bwd_multi_kernel with a RIDER: blockIdx.y == njobs runs an independent small attention backward (dh 16, 8 queries,
// <= 64 keys; mha_small.h), 4 (clip, head) units per workgroup.  The backward of decoder layer 0's query self-attention
// feeds parameter gradients only, so it rides beside the fuser's norm2 backward instead of costing a launch.
struct LnBwdMultiMha { LnBwdArgs j[4]; int njobs; MhaArgs mha; int mha_units; };

template <int EPL>
__global__ __launch_bounds__(256) void ln_bwd_multi_mha_kernel(const LnBwdMultiMha m) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // max([4][2][H], 4 x attention unit)
    if ((int)blockIdx.y < m.njobs) { ln_bwd_block<EPL>(m.j[blockIdx.y], red); return; }
    const int wave = threadIdx.x >> 6;
    const int unit = (int)blockIdx.x * 4 + wave;
    if (unit < m.mha_units) mha_bwd_small_unit<16, 8, false>(m.mha, unit, red + wave * mha_small_bwd_lds_floats(16, 8));
}

template <int EPL>
__device__ __forceinline__ void ln_bwd_block(const LnBwdArgs& a, float* red) {
    if ((int)blockIdx.x >= a.nblocks) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_begin = blockIdx.x * a.rows_per_block;
    const int r_end = min(a.rows, r_begin + a.rows_per_block);
    float dg[EPL], db[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { dg[e] = 0.f; db[e] = 0.f; }
    float gam[EPL], bet[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        const int cc = c < a.H ? c : a.H - 1;
        gam[e] = a.gamma[cc];
        bet[e] = a.relu ? a.beta[cc] : 0.f;
    }
    for (int row = r_begin + wave; row < r_end; row += 4) {
        // phase 1: every load of the row, unconditional from clamped columns (see ln_fwd_load)
        const float mean = a.mean[row], rstd = a.rstd[row];
        float xv[EPL], dv[EPL], d2[EPL], a1[EPL], a2[EPL], km[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            const int cc = c < a.H ? c : a.H - 1;
            xv[e] = a.x[(size_t)row * a.ldx + cc];
            dv[e] = a.pair_in ? 0.5f * a.dy[(size_t)(row >> 1) * a.lddy + cc] : a.dy[(size_t)row * a.lddy + cc];
            d2[e] = a.dy2 ? (a.pair_in ? 0.5f * a.dy2[(size_t)(row >> 1) * a.lddy2 + cc] : a.dy2[(size_t)row * a.lddy2 + cc])
                          : 0.f;
            a1[e] = a.add1 ? a.add1[(size_t)row * a.ldadd1 + cc] : 0.f;
            a2[e] = a.add2 ? a.add2[(size_t)row * a.ldadd2 + cc] : 0.f;
            km[e] = (a.dx2 && a.drop_mask) ? a.drop_scale * (float)a.drop_mask[(size_t)row * a.lddrop + cc] : 1.f;
        }
        float xh[EPL], g[EPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            float xhat = 0.f, gg = 0.f;
            if (c < a.H) {
                xhat = (xv[e] - mean) * rstd;
                float d = dv[e] + d2[e];
                if (a.relu && !(xhat * gam[e] + bet[e] > 0.f)) d = 0.f;
                dg[e] += d * xhat;
                db[e] += d;
                gg = d * gam[e];
            }
            xh[e] = xhat;
            g[e] = gg;
            s1 += gg;
            s2 += gg * xhat;
        }
        s1 = wave_sum(s1) / (float)a.H;
        s2 = wave_sum(s2) / (float)a.H;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < a.H) {
                const float o = rstd * (g[e] - s1 - xh[e] * s2) + a1[e] + a2[e];
                a.dx[(size_t)row * a.lddx + c] = o;
                if (a.dx2) a.dx2[(size_t)row * a.lddx2 + c] = o * km[e];
            }
        }
    }
    if (!a.dgamma) return;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < a.H) { red[(wave * 2 + 0) * a.H + c] = dg[e]; red[(wave * 2 + 1) * a.H + c] = db[e]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * a.H; c += 256) {
        const int which = c / a.H, col = c % a.H;
        const float s = red[(0 * 2 + which) * a.H + col] + red[(1 * 2 + which) * a.H + col] +
                        red[(2 * 2 + which) * a.H + col] + red[(3 * 2 + which) * a.H + col];
        if (a.nblocks == 1) (which == 0 ? a.dgamma : a.dbeta)[col] = s;
        else a.ws[((size_t)blockIdx.x * 2 + which) * a.H + col] = s;
    }
}

// out[i] (+)= sum_{p < nchunks} ws[p * len + i]
__global__ __launch_bounds__(256) void chunk_sum_kernel(const float* ws, int nchunks, int len, float* out, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    float s = 0.f;
    for (int p = 0; p < nchunks; ++p) s += ws[(size_t)p * len + i];
    if (accumulate) s += out[i];
    out[i] = s;
}

// ws: [blocks][2][H] -> dgamma[H], dbeta[H].  Block = 64 consecutive entries of the 2H outputs; its 4 waves split
// the partial range (fixed order -> reproducible), lane = output entry (coalesced 256-B reads).
__global__ __launch_bounds__(256) void ln_param_finalize_kernel(const float* ws, int blocks, int H, float* dgamma,
                                                                float* dbeta) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < 2 * H)
        for (int p = wave; p < blocks; p += 4) s += ws[(size_t)p * 2 * H + i];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < 2 * H) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        (i < H ? dgamma : dbeta)[i < H ? i : i - H] = t;
    }
}

__global__ __launch_bounds__(256) void ln_param_finalize_batched_kernel(const r3d_ln_finalize_job* jobs) {
    __shared__ float red[4][64];
    const r3d_ln_finalize_job j = jobs[blockIdx.y];
    int rpb = ((j.rows + 255) / 256 + 3) / 4 * 4;
    if (rpb < 4) rpb = 4;
    // rows < 0: the producer wrote exactly -rows partial pairs (r3d_embed_fuse_bwd: one per frame)
    const int blocks = j.rows < 0 ? -j.rows : (j.rows + rpb - 1) / rpb;
    if (blocks <= 1 && j.rows > 0) return;                 // the backward kernel already wrote the final values
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < 2 * j.H) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};           // independent chains: 4 loads in flight per lane
        int p = wave;
        for (; p + 12 < blocks; p += 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] += j.ws[(size_t)(p + 4 * q) * 2 * j.H + i];
        }
        for (; p < blocks; p += 4) s4[0] += j.ws[(size_t)p * 2 * j.H + i];
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < 2 * j.H) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        (i < j.H ? j.dgamma : j.dbeta)[i < j.H ? i : i - j.H] = t;
    }
}

// column sums of x[rows, cols]: grid (colblocks of 64, rowchunks); wave w of the block takes rows w, w+4, ...
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int ld, int rows, int cols, int rows_per_chunk,
                                                     float* out, float* ws, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + wave; r < r1; r += 4) s += x[(size_t)r * ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) {
        float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (gridDim.y == 1) { if (accumulate) t += out[c]; out[c] = t; }
        else ws[(size_t)blockIdx.y * cols + c] = t;
    }
}

// out[r, c] (+)= sum_{rows with row % mod == r} x[row, c]      (gradient of a parameter broadcast over clips)
__global__ __launch_bounds__(256) void rowmod_sum_kernel(const float* x, int ld, int rows, int cols, int mod,
                                                         float* out, int ldo, int accumulate) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int row = r; row < rows; row += mod) s += x[(size_t)row * ld + c];
    float* o = out + (size_t)r * ldo + c;
    if (accumulate) s += *o;
    *o = s;
}

// out[r, :] = x[r, :] + add[r % mod, :]   (x may be NULL: pure broadcast)
__global__ __launch_bounds__(256) void add_rowbcast_kernel(const float* x, int ldx, const float* add, int ldadd, int mod,
                                                           float* out, int ldo, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        float v = add[(size_t)(r % mod) * ldadd + c];
        if (x) v += x[(size_t)r * ldx + c];
        out[(size_t)r * ldo + c] = v;
    }
}

// Batched version: job j sums up to two sources over rows with (row % mod) == r into out[r, :].  mod == 1 is a column
// sum (bias gradient).  One launch covers every broadcast-parameter / bias gradient left at the end of the backward.
__global__ __launch_bounds__(256) void rowmod_sum_batched_kernel(const r3d_rowsum_job* jobs) {
    // A workgroup owns 64 columns of one output row: 16 row-lanes x 16 float4 column-lanes, so the rows of a residue
    // class are read 16 at a time with independent 16-byte loads (a serial walk over N rows was 30 us at N = 128),
    // then the row-lanes are folded through LDS.
    __shared__ float4 red[16][16];
    const r3d_rowsum_job j = jobs[blockIdx.z];
    const int r = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    if (r >= j.mod || c0 >= j.cols) return;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const bool vec = ((j.cols | j.ld1 | j.ldd | (j.src2 ? j.ld2 : 0)) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(j.src1) | reinterpret_cast<uintptr_t>(j.dst) |
                       reinterpret_cast<uintptr_t>(j.src2)) & 15) == 0;
    const int c = c0 + cq * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
        if (c < j.cols) {
            for (int row = r + rl * j.mod; row < j.rows; row += 16 * j.mod) {
                const float4 v = *reinterpret_cast<const float4*>(j.src1 + (size_t)row * j.ld1 + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (j.src2)
                for (int row = r + rl * j.mod; row < j.rows; row += 16 * j.mod) {
                    const float4 v = *reinterpret_cast<const float4*>(j.src2 + (size_t)row * j.ld2 + c);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
        }
    } else {
        float* sp = reinterpret_cast<float*>(&s);
        for (int row = r + rl * j.mod; row < j.rows; row += 16 * j.mod)
            for (int u = 0; u < 4; ++u)
                if (c + u < j.cols) {
                    sp[u] += j.src1[(size_t)row * j.ld1 + c + u];
                    if (j.src2) sp[u] += j.src2[(size_t)row * j.ld2 + c + u];
                }
    }
    red[rl][cq] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int q = threadIdx.x >> 2, u = threadIdx.x & 3;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += reinterpret_cast<const float*>(&red[k][q])[u];
        const int cc = c0 + threadIdx.x;
        if (cc < j.cols) j.dst[(size_t)r * j.ldd + cc] = t;
    }
}

__global__ void tick_kernel(int64_t* a, int64_t* b) {
    if (threadIdx.x == 0) {
        if (a) *a += 1;
        if (b) *b += 1;
    }
}

template <typename K, typename A>
static int launch_epl(K k32, K k16, K k8, K k2, int H, dim3 grid, size_t shmem, hipStream_t s, const A& a) {
    if (H <= 128) hipLaunchKernelGGL(k2, grid, dim3(256), shmem, s, a);
    else if (H <= 512) hipLaunchKernelGGL(k8, grid, dim3(256), shmem, s, a);
    else if (H <= 1024) hipLaunchKernelGGL(k16, grid, dim3(256), shmem, s, a);
    else hipLaunchKernelGGL(k32, grid, dim3(256), shmem, s, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

}  // namespace r3d

using namespace r3d;

R3D_EXPORT int r3d_layernorm_fwd(const float* x, int ldx, int nsplit, const float* bias, float* pre_out,
                                 const float* gamma, const float* beta, float* y, int ldy, float* mean, float* rstd,
                                 float* pair_out, int rows, int H, int relu, void* stream) {
    R3D_REQUIRE(x && gamma && beta && y && mean && rstd);
    R3D_REQUIRE(rows > 0 && H > 0 && H <= 2048 && ldy >= H && nsplit >= 0);
    R3D_REQUIRE(nsplit > 0 || ldx >= H);
    R3D_REQUIRE(!pair_out || (rows % 2) == 0);
    LnFwdArgs a{x, ldx, nsplit, bias, pre_out, gamma, beta, y, ldy, mean, rstd, pair_out, rows, H, relu};
    const int units = pair_out ? rows / 2 : rows;
    return launch_epl(ln_fwd_kernel<32>, ln_fwd_kernel<16>, ln_fwd_kernel<8>, ln_fwd_kernel<2>, H,
                      dim3(r3d_cdiv(units, 4)), 0, (hipStream_t)stream, a);
}

// One row per wave (4 rows per block) up to 1024 rows -- the row path is latency-bound, so rows must not queue
// behind each other inside a wave; beyond that, at most 256 blocks of partial parameter gradients.
static int ln_bwd_rows_per_block(int rows) {
    const int rpb = (r3d_cdiv(rows, 256) + 3) / 4 * 4;
    return rpb < 4 ? 4 : rpb;
}

R3D_EXPORT int64_t r3d_layernorm_bwd_ws_floats(int rows, int H) {
    const int rpb = ln_bwd_rows_per_block(rows);
    const int blocks = r3d_cdiv(rows, rpb);
    return blocks > 1 ? (int64_t)blocks * 2 * H : 0;
}

R3D_EXPORT int r3d_layernorm_bwd(const float* dy, int lddy, int pair_in, const float* dy2, int lddy2, const float* x,
                                 int ldx, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int relu,
                                 const float* add1, int ldadd1, const float* add2, int ldadd2, float* dx, int lddx,
                                 float* dx2, int lddx2, const uint8_t* drop_mask, int lddrop, float drop_scale,
                                 float* dgamma, float* dbeta, float* ws, int rows, int H, int defer_finalize,
                                 void* stream) {
    R3D_REQUIRE(dy && x && mean && rstd && gamma && beta && dx);
    R3D_REQUIRE(rows > 0 && H > 0 && H <= 2048 && ldx >= H && lddx >= H && lddy >= H);
    R3D_REQUIRE((dgamma == nullptr) == (dbeta == nullptr));
    R3D_REQUIRE(!pair_in || (rows % 2) == 0);
    R3D_REQUIRE(!dy2 || lddy2 >= H);                  /* pair_in applies to dy2 as well */
    const int rpb = ln_bwd_rows_per_block(rows);
    const int blocks = r3d_cdiv(rows, rpb);
    R3D_REQUIRE(blocks == 1 || !dgamma || ws);
    LnBwdArgs a{dy, lddy, pair_in, dy2, lddy2, x, ldx, mean, rstd, gamma, beta, relu, add1, ldadd1, add2, ldadd2, dx, lddx,
                dx2, lddx2, drop_mask, lddrop, drop_scale, dgamma, dbeta, ws, rows, H, rpb, blocks};
    hipStream_t s = (hipStream_t)stream;
    const size_t shmem = (size_t)8 * H * sizeof(float);
    int rc = launch_epl(ln_bwd_kernel<32>, ln_bwd_kernel<16>, ln_bwd_kernel<8>, ln_bwd_kernel<2>, H, dim3(blocks),
                        shmem, s, a);
    if (rc != R3D_OK) return rc;
    if (dgamma && blocks > 1 && !defer_finalize) {
        hipLaunchKernelGGL(ln_param_finalize_kernel, dim3(r3d_cdiv(2 * H, 64)), dim3(256), 0, s, ws, blocks, H,
                           dgamma, dbeta);
        R3D_LAUNCH_CHECK();
    }
    return R3D_OK;
}

R3D_EXPORT int r3d_layernorm_fwd_multi(const r3d_ln_fwd_job* jobs, int njobs, void* stream) {
    R3D_REQUIRE(jobs && njobs >= 1 && njobs <= 4);
    LnFwdMulti m{};
    int units = 0, H = jobs[0].H;
    for (int i = 0; i < njobs; ++i) {
        const r3d_ln_fwd_job& a = jobs[i];
        R3D_REQUIRE(a.x && a.gamma && a.beta && a.y && a.mean && a.rstd);
        R3D_REQUIRE(a.rows > 0 && a.H == H && H > 0 && H <= 2048 && a.ldy >= H && a.nsplit >= 0);
        R3D_REQUIRE(a.nsplit > 0 || a.ldx >= H);
        R3D_REQUIRE(!a.pair_out || (a.rows % 2) == 0);
        const int u = a.pair_out ? a.rows / 2 : a.rows;
        units = u > units ? u : units;
        m.j[i] = a;
    }
    return launch_epl(ln_fwd_multi_kernel<32>, ln_fwd_multi_kernel<16>, ln_fwd_multi_kernel<8>, ln_fwd_multi_kernel<2>, H,
                      dim3(r3d_cdiv(units, 4), njobs), 0, (hipStream_t)stream, m);
}

R3D_EXPORT int r3d_layernorm_bwd_multi(r3d_ln_bwd_job* jobs, int njobs, void* stream) {
    R3D_REQUIRE(jobs && njobs >= 1 && njobs <= 4);
    LnBwdMulti m{};
    int maxb = 0;
    const int H = jobs[0].H;
    for (int i = 0; i < njobs; ++i) {
        r3d_ln_bwd_job& a = jobs[i];
        R3D_REQUIRE(a.dy && a.x && a.mean && a.rstd && a.gamma && a.beta && a.dx);
        R3D_REQUIRE(a.rows > 0 && a.H == H && H > 0 && H <= 2048 && a.ldx >= H && a.lddx >= H && a.lddy >= H);
        R3D_REQUIRE((a.dgamma == nullptr) == (a.dbeta == nullptr));
        R3D_REQUIRE(!a.pair_in || (a.rows % 2) == 0);
        R3D_REQUIRE(!a.dy2 || a.lddy2 >= H);
        a.rows_per_block = ln_bwd_rows_per_block(a.rows);
        a.nblocks = r3d_cdiv(a.rows, a.rows_per_block);
        R3D_REQUIRE(a.nblocks == 1 || !a.dgamma || a.ws);
        maxb = a.nblocks > maxb ? a.nblocks : maxb;
        m.j[i] = a;
    }
    return launch_epl(ln_bwd_multi_kernel<32>, ln_bwd_multi_kernel<16>, ln_bwd_multi_kernel<8>, ln_bwd_multi_kernel<2>, H,
                      dim3(maxb, njobs), (size_t)8 * H * sizeof(float), (hipStream_t)stream, m);
}

R3D_EXPORT int r3d_layernorm_bwd_multi_mha(r3d_ln_bwd_job* jobs, int njobs, const r3d_mha_bwd_job* mha, void* stream) {
    R3D_REQUIRE(jobs && njobs >= 1 && njobs <= 4 && mha);
    LnBwdMultiMha m{};
    int maxb = 0;
    const int H = jobs[0].H;
    R3D_REQUIRE(H > 0 && H <= 128);                        /* (the EPL = 2 instance; wider models launch the two separately) */
    for (int i = 0; i < njobs; ++i) {
        r3d_ln_bwd_job& a = jobs[i];
        R3D_REQUIRE(a.dy && a.x && a.mean && a.rstd && a.gamma && a.beta && a.dx);
        R3D_REQUIRE(a.rows > 0 && a.H == H && a.ldx >= H && a.lddx >= H && a.lddy >= H);
        R3D_REQUIRE((a.dgamma == nullptr) == (a.dbeta == nullptr));
        R3D_REQUIRE(!a.pair_in || (a.rows % 2) == 0);
        R3D_REQUIRE(!a.dy2 || a.lddy2 >= H);
        a.rows_per_block = ln_bwd_rows_per_block(a.rows);
        a.nblocks = r3d_cdiv(a.rows, a.rows_per_block);
        R3D_REQUIRE(a.nblocks == 1 || !a.dgamma || a.ws);
        maxb = a.nblocks > maxb ? a.nblocks : maxb;
        m.j[i] = a;
    }
    m.njobs = njobs;
    R3D_REQUIRE(mha->heads > 0 && mha->Lq == 8 && mha->dh == 16 && mha->Lk > 0 && mha->Lk <= 64 && mha->B > 0);
    R3D_REQUIRE(mha->q && mha->k && mha->v && mha->probs && mha->d_o && mha->dq && mha->dk && mha->dv);
    const int Hm = mha->heads * mha->dh;
    R3D_REQUIRE(mha->ldq >= Hm && mha->ldk >= Hm && mha->ldv >= Hm && mha->lddo >= Hm && mha->lddq >= Hm &&
                mha->lddk >= Hm && mha->lddv >= Hm);
    if (((mha->ldk | mha->ldv | mha->lddk | mha->lddv) & 3) || !r3d_aligned16(mha->k) || !r3d_aligned16(mha->v) ||
        !r3d_aligned16(mha->dk) || !r3d_aligned16(mha->dv))
        return R3D_EALIGN;
    MhaArgs& g = m.mha;
    g.q = mha->q; g.ldq = mha->ldq; g.k = mha->k; g.ldk = mha->ldk; g.v = mha->v; g.ldv = mha->ldv;
    g.probs = const_cast<float*>(mha->probs); g.drop = mha->drop_mask; g.drop_scale = mha->drop_scale;
    g.d_o = mha->d_o; g.lddo = mha->lddo; g.dq = mha->dq; g.lddq = mha->lddq; g.dk = mha->dk; g.lddk = mha->lddk;
    g.dv = mha->dv; g.lddv = mha->lddv;
    g.B = mha->B; g.heads = mha->heads; g.Lq = mha->Lq; g.Lk = mha->Lk; g.dh = mha->dh;
    g.scale = 1.0f / sqrtf((float)mha->dh);
    m.mha_units = mha->B * mha->heads;
    const int rb = r3d_cdiv(m.mha_units, 4);
    const size_t lds_ln = (size_t)8 * H * sizeof(float), lds_mha = (size_t)4 * mha_small_bwd_lds_floats(16, 8) * sizeof(float);
    hipLaunchKernelGGL(ln_bwd_multi_mha_kernel<2>, dim3(maxb > rb ? maxb : rb, njobs + 1), dim3(256),
                       lds_ln > lds_mha ? lds_ln : lds_mha, (hipStream_t)stream, m);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

static int colsum_chunks(int rows) { int c = r3d_cdiv(rows, 64); return c < 1 ? 1 : (c > 32 ? 32 : c); }

R3D_EXPORT int64_t r3d_colsum_ws_floats(int rows, int cols) {
    const int ch = colsum_chunks(rows);
    return ch > 1 ? (int64_t)ch * cols : 0;
}

/* out[c] (+)= sum_r x[r, c] -- bias gradients (the db of every nn.Linear autograd derives for the reference). */
R3D_EXPORT int r3d_colsum(const float* x, int ld, int rows, int cols, float* out, float* ws, int accumulate,
                          void* stream) {
    R3D_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols);
    const int ch = colsum_chunks(rows);
    R3D_REQUIRE(ch == 1 || ws);
    hipStream_t s = (hipStream_t)stream;
    const int rpc = r3d_cdiv(rows, ch);
    hipLaunchKernelGGL(colsum_kernel, dim3(r3d_cdiv(cols, 64), ch), dim3(256), 0, s, x, ld, rows, cols, rpc, out, ws,
                       accumulate);
    R3D_LAUNCH_CHECK();
    if (ch > 1) {
        hipLaunchKernelGGL(chunk_sum_kernel, dim3(r3d_cdiv(cols, 256)), dim3(256), 0, s, ws, ch, cols, out, accumulate);
        R3D_LAUNCH_CHECK();
    }
    return R3D_OK;
}

/* out[r, c] (+)= sum over rows with (row % mod) == r of x[row, c] -- gradient of parameters that the forward
 * broadcasts over clips: query_embed (futr_safuser_tokenfusion.py:205-209) and pos_embedding[:, :S] (:190). */
R3D_EXPORT int r3d_rowmod_sum(const float* x, int ld, int rows, int cols, int mod, float* out, int ldo,
                              int accumulate, void* stream) {
    R3D_REQUIRE(x && out && rows > 0 && cols > 0 && mod > 0 && ld >= cols && ldo >= cols);
    hipLaunchKernelGGL(rowmod_sum_kernel, dim3(r3d_cdiv(cols, 256), mod), dim3(256), 0, (hipStream_t)stream, x, ld,
                       rows, cols, mod, out, ldo, accumulate);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* out[r,:] = x[r,:] + add[r % mod,:] -- with_pos_embed of the decoder (model/extras/transformer.py:278-279,289,300-302):
 * tgt + query_pos (mod = n_query) and memory + pos (mod = S).  x == NULL gives the pure broadcast (layer 0: tgt = 0). */
R3D_EXPORT int r3d_add_rowbcast(const float* x, int ldx, const float* add, int ldadd, int mod, float* out, int ldo,
                                int rows, int cols, void* stream) {
    R3D_REQUIRE(add && out && rows > 0 && cols > 0 && mod > 0 && ldadd >= cols && ldo >= cols && (!x || ldx >= cols));
    const size_t total = (size_t)rows * cols;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(add_rowbcast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, add, ldadd, mod, out,
                       ldo, rows, cols);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* Second half of r3d_layernorm_bwd(..., defer_finalize=1): reduces the per-block partial parameter gradients left in
 * ws.  Separate so the host can put it on a stream off the backward's critical path.  No-op when rows <= 4. */
R3D_EXPORT int r3d_layernorm_bwd_finalize(const float* ws, int rows, int H, float* dgamma, float* dbeta, void* stream) {
    R3D_REQUIRE(dgamma && dbeta && rows > 0 && H > 0 && H <= 2048);
    const int blocks = r3d_cdiv(rows, ln_bwd_rows_per_block(rows));
    if (blocks <= 1) return R3D_OK;
    R3D_REQUIRE(ws);
    hipLaunchKernelGGL(ln_param_finalize_kernel, dim3(r3d_cdiv(2 * H, 64)), dim3(256), 0, (hipStream_t)stream, ws, blocks, H,
                       dgamma, dbeta);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_layernorm_bwd_finalize_batched(const r3d_ln_finalize_job* dev_jobs, int njobs, int max_H, void* stream) {
    R3D_REQUIRE(dev_jobs && njobs > 0 && max_H > 0 && max_H <= 2048);
    hipLaunchKernelGGL(ln_param_finalize_batched_kernel, dim3(r3d_cdiv(2 * max_H, 64), njobs), dim3(256), 0,
                       (hipStream_t)stream, dev_jobs);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_rowmod_sum_batched(const r3d_rowsum_job* dev_jobs, int njobs, int max_cols, int max_mod, void* stream) {
    R3D_REQUIRE(dev_jobs && njobs > 0 && max_cols > 0 && max_mod > 0);
    hipLaunchKernelGGL(rowmod_sum_batched_kernel, dim3(r3d_cdiv(max_cols, 64), max_mod, njobs), dim3(256), 0,
                       (hipStream_t)stream, dev_jobs);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* ++*a, ++*b (either may be NULL): the step counter and the dropout offset, one launch. */
R3D_EXPORT int r3d_tick(int64_t* a, int64_t* b, void* stream) {
    R3D_REQUIRE(a || b);
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
