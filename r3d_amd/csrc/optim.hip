// Fused multi-tensor AdamW over a flat parameter arena, and the Philox keep-masks for dropout.
//   AdamW: torch.optim.AdamW(lr, weight_decay) as constructed at main_darai.py:135 and stepped at
//          train/train_proposed_depth.py:215 -- the largest HBM term of the step (28 B/param).
//          One launch over [p | g | m | v] arenas, float4 per lane, grid-stride; hyper-parameters that change while a
//          hipGraph is replayed (lr, step) are read from device memory.
//   Dropout: nn.Dropout(0.1) sites of the path (futr_safuser_tokenfusion.py:26,83; transformer.py:268-276 and the
//          attention-probability dropout inside nn.MultiheadAttention).  RNG parity with PyTorch is impossible, so the
//          masks come from Philox4x32-10 keyed by (seed, per-step offset read from device memory).
#include <cstdlib>
#include "common.h"
#include "../../include/r3d_hip.h"
#include "loss_finalize.h"

namespace r3d {

__device__ __forceinline__ void adamw_body(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                           float4* __restrict__ v, size_t n4, const float* lr_ptr, const int64_t* step_ptr,
                                           float b1, float b2, float eps, float wd, float gscale, unsigned bid, unsigned nb) {
    // No fused multiply-add contraction in here: hipcc contracted the two-stride main loop and its one-stride tail
    // differently, so WHICH elements fell into the tail (the start and length of the range) changed their last bit.  The
    // same parameter must get the same update whether it is reached by one launch over the arena or by two over its parts
    // (found when a split launch stopped matching the single one bitwise from the second step on).
#pragma clang fp contract(off)
    const float lr = *lr_ptr;
    const double step = (double)*step_ptr;
    const float bc1 = (float)(1.0 - pow((double)b1, step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    const float decay = 1.0f - lr * wd;
    const float step_size = lr / bc1;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#define R3D_ADAM1(c)                                               \
        {                                                          \
            const float gr = gg.c * gscale;                        \
            pp.c *= decay;                                         \
            mm.c = mm.c + (gr - mm.c) * omb1;                      \
            vv.c = vv.c * b2 + gr * gr * omb2;                     \
            const float den = sqrtf(vv.c) / bc2_sqrt + eps;        \
            pp.c -= step_size * (mm.c / den);                      \
        }
    // two grid-strides per iteration: eight 16-byte loads in flight per lane before the first use (arenas beyond the
    // 256 MiB Infinity Cache are a pure HBM stream: more bytes in flight per CU, measured at the cfg4 / cfg5 arena sizes;
    // consecutive 4 KB pieces per workgroup instead of grid strides measured 10 % slower there)
    const size_t stride = (size_t)nb * 256;
    size_t i = (size_t)bid * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const size_t j = i + stride;
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float4 pq = p[j], gq = g[j], mq = m[j], vq = v[j];
        R3D_ADAM1(x) R3D_ADAM1(y) R3D_ADAM1(z) R3D_ADAM1(w)
        p[i] = pp; m[i] = mm; v[i] = vv;
        pp = pq; gg = gq; mm = mq; vv = vq;
        R3D_ADAM1(x) R3D_ADAM1(y) R3D_ADAM1(z) R3D_ADAM1(w)
        p[j] = pp; m[j] = mm; v[j] = vv;
    }
    if (i < n4) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        R3D_ADAM1(x) R3D_ADAM1(y) R3D_ADAM1(z) R3D_ADAM1(w)
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
#undef R3D_ADAM1
}

__global__ __launch_bounds__(256) void adamw_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                    float4* __restrict__ m, float4* __restrict__ v, size_t n4,
                                                    const float* lr_ptr, const int64_t* step_ptr, float b1, float b2,
                                                    float eps, float wd, float gscale) {
    adamw_body(p, g, m, v, n4, lr_ptr, step_ptr, b1, b2, eps, wd, gscale, blockIdx.x, gridDim.x);
}

// The same update on a [rows x cols] block of a row-major matrix with leading dimension ld: a COLUMN shard of
// depth_projection.weight [H, 50176] when the projection is tensor-parallel over pixels (r3d_amd/parallel.py).
__global__ __launch_bounds__(256) void adamw_2d_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int rows, int cols4,
                                                       int ld4, const float* lr_ptr, const int64_t* step_ptr, float b1,
                                                       float b2, float eps, float wd, float gscale) {
    const float lr = *lr_ptr;
    const double step = (double)*step_ptr;
    const float bc1 = (float)(1.0 - pow((double)b1, step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    const float decay = 1.0f - lr * wd;
    const float step_size = lr / bc1;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const size_t total = (size_t)rows * cols4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = (e / cols4) * ld4 + (e % cols4);
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define R3D_ADAM2(c)                                               \
        {                                                          \
            const float gr = gg.c * gscale;                        \
            pp.c *= decay;                                         \
            mm.c = mm.c + (gr - mm.c) * omb1;                      \
            vv.c = vv.c * b2 + gr * gr * omb2;                     \
            const float den = sqrtf(vv.c) / bc2_sqrt + eps;        \
            pp.c -= step_size * (mm.c / den);                      \
        }
        R3D_ADAM2(x) R3D_ADAM2(y) R3D_ADAM2(z) R3D_ADAM2(w)
#undef R3D_ADAM2
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = mulhi32(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

// mask[i] = 1 with probability (1-p).  4 elements per Philox call.
__device__ __forceinline__ void dropout_body(uint8_t* mask, size_t n, uint32_t thresh, uint64_t seed,
                                             const int64_t* offset_ptr, unsigned bid, unsigned nb) {
    const uint64_t off = offset_ptr ? (uint64_t)*offset_ptr : 0ull;
    const size_t n4 = (n + 3) / 4;
    for (size_t i = (size_t)bid * 256 + threadIdx.x; i < n4; i += (size_t)nb * 256) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)off, (uint32_t)(off >> 32)),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i * 4 + j < n) mask[i * 4 + j] = (u[j] >= thresh) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* mask, size_t n, uint32_t thresh, uint64_t seed,
                                                           const int64_t* offset_ptr) {
    dropout_body(mask, n, thresh, seed, offset_ptr, blockIdx.x, gridDim.x);
}

// AdamW and the NEXT step's dropout masks in one launch: the first nb_adam workgroups stream the arenas, the rest fill
// the mask pool (it is free once the backward has run) -- a dependent launch at the head of every step less.
__global__ __launch_bounds__(256) void adamw_dropout_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                            float4* __restrict__ m, float4* __restrict__ v, size_t n4,
                                                            const float* lr_ptr, const int64_t* step_ptr, float b1, float b2,
                                                            float eps, float wd, float gscale, unsigned nb_adam,
                                                            uint8_t* mask, size_t n_mask, uint32_t thresh, uint64_t seed,
                                                            const int64_t* offset_ptr) {
    if (blockIdx.x < nb_adam) adamw_body(p, g, m, v, n4, lr_ptr, step_ptr, b1, b2, eps, wd, gscale, blockIdx.x, nb_adam);
    else dropout_body(mask, n_mask, thresh, seed, offset_ptr, blockIdx.x - nb_adam, gridDim.x - nb_adam);
}

// ... and the reduction of the step's loss partials as ONE more workgroup (r3d_decoder_tail_losses(defer_finalize)):
// host-visible statistics nothing on the device waits for, so they leave the loss kernel's chain and hide here.
__global__ __launch_bounds__(256) void adamw_dropout_fin_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                                float4* __restrict__ m, float4* __restrict__ v, size_t n4,
                                                                const float* lr_ptr, const int64_t* step_ptr, float b1,
                                                                float b2, float eps, float wd, float gscale, unsigned nb_adam,
                                                                uint8_t* mask, size_t n_mask, uint32_t thresh, uint64_t seed,
                                                                const int64_t* offset_ptr, const r3d_loss_finalize_job fin) {
    __shared__ double red[4][3][3];
    // workgroup 0 (dispatched first): its load -> reduce -> store chain then runs under the streaming workgroups instead of
    // trailing the launch (as the LAST workgroup it lengthened the kernel by ~2 us)
    if (blockIdx.x == 0) { loss_finalize_block<4>(fin, red); return; }
    const unsigned bid = blockIdx.x - 1;
    if (bid < nb_adam) adamw_body(p, g, m, v, n4, lr_ptr, step_ptr, b1, b2, eps, wd, gscale, bid, nb_adam);
    else dropout_body(mask, n_mask, thresh, seed, offset_ptr, bid - nb_adam, gridDim.x - 1 - nb_adam);
}

__global__ __launch_bounds__(256) void adamw_fin_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                        float4* __restrict__ m, float4* __restrict__ v, size_t n4,
                                                        const float* lr_ptr, const int64_t* step_ptr, float b1, float b2,
                                                        float eps, float wd, float gscale, const r3d_loss_finalize_job fin) {
    __shared__ double red[4][3][3];
    if (blockIdx.x == 0) { loss_finalize_block<4>(fin, red); return; }
    adamw_body(p, g, m, v, n4, lr_ptr, step_ptr, b1, b2, eps, wd, gscale, blockIdx.x - 1, gridDim.x - 1);
}

}  // namespace r3d

using namespace r3d;

// Workgroups of the flat AdamW launch.  Swept 256 .. 16384 at the cfg2 / cfg4 / cfg5 arena sizes on two boxes
// (tools/r02_profile.py adamw, R3D_ADAMW_BLOCKS overrides): beyond the 256 MiB Infinity Cache (4 arrays x 4 B x n) three
// workgroups per CU stream best -- 0.65 - 0.73 of 8 TB/s against 0.60 - 0.63 with 4096 workgroups, whose 28 concurrent
// streams per CU scatter over more DRAM pages; inside the cache 4096 stay 0.5 us ahead (more requests in flight).
static inline int adam_blocks(size_t n4) {
    // (the override is read once: an entry point must not call getenv per launch; a value that is not a positive number --
    //  "0", "abc" -- is ignored instead of turning into a 0-block launch)
    static const int forced = [] {
        const char* e = getenv("R3D_ADAMW_BLOCKS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 0;
    }();
    const int cap = forced ? forced : ((n4 * 64 > ((size_t)200 << 20)) ? 768 : 4096);
    const size_t want = (n4 + 255) / 256;
    const size_t blocks = want < (size_t)cap ? want : (size_t)cap;
    return (int)(blocks > 0 ? blocks : 1);
}

R3D_EXPORT int r3d_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, const int64_t* step,
                              float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    R3D_REQUIRE(p && g && m && v && lr && step && n > 0);
    R3D_REQUIRE((n % 4) == 0);
    if (!(r3d_aligned16(p) && r3d_aligned16(g) && r3d_aligned16(m) && r3d_aligned16(v))) return R3D_EALIGN;
    const size_t n4 = (size_t)n / 4;
    const int blocks = adam_blocks(n4);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, n4, lr, step, beta1, beta2, eps, weight_decay, grad_scale);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_adamw_flat_fin(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, const int64_t* step,
                                  float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                  const r3d_loss_finalize_job* fin, void* stream) {
    if (!fin) return r3d_adamw_flat(p, g, m, v, n, lr, step, beta1, beta2, eps, weight_decay, grad_scale, stream);
    R3D_REQUIRE(p && g && m && v && lr && step && n > 0);
    R3D_REQUIRE((n % 4) == 0);
    R3D_REQUIRE(fin->part && fin->loss_out && fin->counts && fin->B > 0 && fin->S > 0 && fin->Q > 0);
    if (!(r3d_aligned16(p) && r3d_aligned16(g) && r3d_aligned16(m) && r3d_aligned16(v))) return R3D_EALIGN;
    const size_t n4 = (size_t)n / 4;
    const int blocks = adam_blocks(n4);
    hipLaunchKernelGGL(adamw_fin_kernel, dim3(blocks + 1), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, n4, lr, step, beta1, beta2, eps, weight_decay, grad_scale, *fin);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* r3d_adamw_flat and r3d_dropout_mask(mask, n_mask, p_drop, seed, offset) in ONE launch (the masks are the next step's). */
R3D_EXPORT int r3d_adamw_flat_dropout(float* p, const float* g, float* m, float* v, int64_t n, const float* lr,
                                      const int64_t* step, float beta1, float beta2, float eps, float weight_decay,
                                      float grad_scale, uint8_t* mask, int64_t n_mask, float p_drop, uint64_t seed,
                                      const int64_t* offset, void* stream) {
    R3D_REQUIRE(p && g && m && v && lr && step && n > 0 && mask && n_mask > 0 && p_drop >= 0.f && p_drop < 1.f);
    R3D_REQUIRE((n % 4) == 0);
    if (!(r3d_aligned16(p) && r3d_aligned16(g) && r3d_aligned16(m) && r3d_aligned16(v))) return R3D_EALIGN;
    const size_t n4 = (size_t)n / 4;
    const unsigned nb_adam = (unsigned)adam_blocks(n4);
    const double t = (double)p_drop * 4294967296.0;
    const uint32_t thresh = (uint32_t)(t >= 4294967295.0 ? 4294967295.0 : t);
    const size_t m4 = ((size_t)n_mask + 3) / 4;
    const unsigned nb_drop = (unsigned)((m4 + 255) / 256 < 512 ? (m4 + 255) / 256 : 512);
    hipLaunchKernelGGL(adamw_dropout_kernel, dim3(nb_adam + nb_drop), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, n4, lr, step, beta1, beta2, eps, weight_decay, grad_scale,
                       nb_adam, mask, (size_t)n_mask, thresh, seed, offset);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* r3d_adamw_flat_dropout + the deferred loss reduction (fin; NULL = plain r3d_adamw_flat_dropout). */
R3D_EXPORT int r3d_adamw_flat_dropout_fin(float* p, const float* g, float* m, float* v, int64_t n, const float* lr,
                                          const int64_t* step, float beta1, float beta2, float eps, float weight_decay,
                                          float grad_scale, uint8_t* mask, int64_t n_mask, float p_drop, uint64_t seed,
                                          const int64_t* offset, const r3d_loss_finalize_job* fin, void* stream) {
    if (!fin)
        return r3d_adamw_flat_dropout(p, g, m, v, n, lr, step, beta1, beta2, eps, weight_decay, grad_scale, mask, n_mask,
                                      p_drop, seed, offset, stream);
    R3D_REQUIRE(p && g && m && v && lr && step && n > 0 && mask && n_mask > 0 && p_drop >= 0.f && p_drop < 1.f);
    R3D_REQUIRE((n % 4) == 0);
    R3D_REQUIRE(fin->part && fin->loss_out && fin->counts && fin->B > 0 && fin->S > 0 && fin->Q > 0);
    if (!(r3d_aligned16(p) && r3d_aligned16(g) && r3d_aligned16(m) && r3d_aligned16(v))) return R3D_EALIGN;
    const size_t n4 = (size_t)n / 4;
    const unsigned nb_adam = (unsigned)adam_blocks(n4);
    const double t = (double)p_drop * 4294967296.0;
    const uint32_t thresh = (uint32_t)(t >= 4294967295.0 ? 4294967295.0 : t);
    const size_t m4 = ((size_t)n_mask + 3) / 4;
    const unsigned nb_drop = (unsigned)((m4 + 255) / 256 < 512 ? (m4 + 255) / 256 : 512);
    hipLaunchKernelGGL(adamw_dropout_fin_kernel, dim3(nb_adam + nb_drop + 1), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, n4, lr, step, beta1, beta2, eps, weight_decay, grad_scale,
                       nb_adam, mask, (size_t)n_mask, thresh, seed, offset, *fin);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* AdamW on a [rows x cols] block (leading dimension ld) of p/g/m/v -- all four share the layout. */
R3D_EXPORT int r3d_adamw_2d(float* p, const float* g, float* m, float* v, int rows, int cols, int ld, const float* lr,
                            const int64_t* step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                            void* stream) {
    R3D_REQUIRE(p && g && m && v && lr && step && rows > 0 && cols > 0 && ld >= cols);
    R3D_REQUIRE((cols % 4) == 0 && (ld % 4) == 0);
    if (!(r3d_aligned16(p) && r3d_aligned16(g) && r3d_aligned16(m) && r3d_aligned16(v))) return R3D_EALIGN;
    const size_t total = (size_t)rows * (cols / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(adamw_2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, rows, cols / 4, ld / 4,
                       lr, step, beta1, beta2, eps, weight_decay, grad_scale);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, const int64_t* offset, void* stream) {
    R3D_REQUIRE(mask && n > 0 && p >= 0.f && p < 1.f);
    const double t = (double)p * 4294967296.0;
    const uint32_t thresh = (uint32_t)(t >= 4294967295.0 ? 4294967295.0 : t);
    const size_t n4 = ((size_t)n + 3) / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, (size_t)n, thresh, seed,
                       offset);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
