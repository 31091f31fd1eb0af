// bf16x3 building blocks of the row-local chain kernels (fuser_chain_bf3.hip, decoder_chain_bf3.hip).
//
// The fp32 chain kernels (fuser_chain.hip, decoder_chain.hip) are bound by the fp32 matrix pipe: a 128-deep chunk is 32
// v_mfma_f32_16x16x4_f32 = 1024 cycles, the chip holds such MFMA-dense code at ~1.6 GHz, and 13-18 chunks per wave with two
// waves per SIMD are 17-24 us per kernel before anything else (measured with stage timelines and stubbed builds:
// profiles/r03_chain_timeline.txt, tools/chain_probe.sh).  The bf16 matrix pipe is 16x faster per FLOP; with every fp32
// operand split EXACTLY into three bf16 terms (x = h + m + l by truncation, gemm_bf3.hip) and the six leading products
// accumulated in fp32 -- the dropped terms are <= 3 * 2^-24 |a b|, an fp32 rounding -- a 128-deep chunk is 24
// v_mfma_f32_16x16x32_bf16 = 384 cycles.
//
// What makes that usable for 16-row stages, where a workgroup streams 0.6 MB of weights for 16 rows of activations:
//   * the WEIGHTS arrive pre-split: r3d_weight_planes writes, once per optimiser step, three bf16 planes of every chain
//     weight (and of its transpose, for the input-gradient products) in MFMA OPERAND ORDER -- for output tile t and k-step
//     s the 64 lanes' 16-byte operands lie back to back (1 KB per plane) -- so a wave loads its B operands with perfectly
//     coalesced 16-byte global loads straight into the registers the MFMA reads.  No LDS staging, no ds_write / ds_read per
//     chunk, no split arithmetic in the consumer;
//   * the ACTIVATIONS (16 x K per stage) are split where they are produced: an epilogue writes its values as three bf16 to
//     [plane][16][K + 8] LDS images (row pitch K + 8 bf16: 16-byte operand reads of consecutive rows fall on distinct banks),
//     and the next stage reads its A operands from them, one k-step (12 registers) ahead of the MFMAs.
#pragma once
#include "chain_common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// ---- operand-order weight planes -----------------------------------------------------------------------------------------
// Planes of a matrix B[n][k] (n = output column of the product, k = contraction index), N x K:
//   bf16 element (tile t = n / 16, k-step s = k / 32, plane p, lane = (n % 16) + 16 ((k % 32) / 8), e = k % 8)
//   at index ((((t * ksteps + s) * 3 + p) * 64 + lane) * 8 + e);   zero where n >= N or k >= K.
__host__ __device__ inline size_t bf3_plane_elems(int N, int K) {
    return (size_t)((N + 15) / 16) * (size_t)((K + 31) / 32) * 3 * 64 * 8;
}

// B operands of up to 4 consecutive k-steps of one tile: 12 x 16 bytes per lane
struct Bf3B { uint4 v[4][3]; };
template <int NS>
__device__ __forceinline__ void bf3_bload(Bf3B& r, const unsigned short* planes, int ksteps, int tile, int s0, int lane) {
    const uint4* p = reinterpret_cast<const uint4*>(planes) + ((size_t)tile * ksteps + s0) * (3 * 64) + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) r.v[s][pl] = p[(s * 3 + pl) * 64];
    }
}

// ---- activation planes in LDS --------------------------------------------------------------------------------------------
// image: [3][16][K + 8] bf16.  pitch (bf16) = K + 8; plane stride = 16 * pitch.
__device__ __forceinline__ void bf3_split1(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    h = (unsigned short)(u >> 16);
    const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);
    const unsigned ur = __builtin_bit_cast(unsigned, r);
    m = (unsigned short)(ur >> 16);
    const float q = r - __builtin_bit_cast(float, ur & 0xffff0000u);
    l = (unsigned short)(__builtin_bit_cast(unsigned, q) >> 16);
}
// one value of an epilogue (accumulator layout: any row / column) -> the three planes
__device__ __forceinline__ void bf3_store1(unsigned short* img, int pitch, int row, int col, float x) {
    unsigned short h, m, l;
    bf3_split1(x, h, m, l);
    unsigned short* p = img + row * pitch + col;
    p[0] = h;
    p[16 * pitch] = m;
    p[32 * pitch] = l;
}
// four consecutive-k values of one row (a staged global load) -> 8-byte stores into the three planes
// (each value through bf3_split1: a pair-wise v_perm_b32 form of this split measured 4.6e-5 instead of 1.6e-7 against fp64 in
//  tools/bf3_probe2.hip although it reconstructs exactly when emulated -- not pursued: staging runs once per kernel)
__device__ __forceinline__ void bf3_store4(unsigned short* img, int pitch, int row, int col, const f32x4 x) {
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bf3_split1(x[e], h[e], m[e], l[e]);
    unsigned short* p = img + row * pitch + col;
    *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(p + 16 * pitch) = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
    *reinterpret_cast<uint2*>(p + 32 * pitch) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}

// A operands of one k-step: lane (row li, k group q) reads 16 bytes of each plane at k = 32 s + 8 q
struct Bf3A { uint4 v[3]; };
__device__ __forceinline__ void bf3_aload(Bf3A& a, const unsigned short* img, int pitch, int li, int q, int kstep) {
    const unsigned short* p = img + li * pitch + 32 * kstep + 8 * q;
    a.v[0] = *reinterpret_cast<const uint4*>(p);
    a.v[1] = *reinterpret_cast<const uint4*>(p + 16 * pitch);
    a.v[2] = *reinterpret_cast<const uint4*>(p + 32 * pitch);
}

// the six leading products of one k-step, small terms first, into one accumulator chain
__device__ __forceinline__ void bf3_mfma6(const Bf3A& a, const uint4 (&b)[3], f32x4& acc) {
#define R3D_B8(x) __builtin_bit_cast(bf16x8_t, x)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R3D_B8(a.v[0]), R3D_B8(b[2]), acc, 0, 0, 0);      // h . l
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R3D_B8(a.v[2]), R3D_B8(b[0]), acc, 0, 0, 0);      // l . h
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R3D_B8(a.v[1]), R3D_B8(b[1]), acc, 0, 0, 0);      // m . m
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R3D_B8(a.v[0]), R3D_B8(b[1]), acc, 0, 0, 0);      // h . m
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R3D_B8(a.v[1]), R3D_B8(b[0]), acc, 0, 0, 0);      // m . h
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R3D_B8(a.v[0]), R3D_B8(b[0]), acc, 0, 0, 0);      // h . h
#undef R3D_B8
}

// One chunk of NS k-steps: A from the LDS image (k-steps s0 ..), B from registers.  The A operands are read one k-step
// ahead of their MFMAs; two accumulator chains alternate by k-step (a dependent MFMA waits out its predecessor's passes).
template <int NS>
__device__ __forceinline__ void bf3_chunk(const unsigned short* img, int pitch, int li, int q, int s0, const Bf3B& b, f32x4& acc0,
                                          f32x4& acc1) {
    Bf3A a0, a1;
    bf3_aload(a0, img, pitch, li, q, s0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        Bf3A& cur = (s & 1) ? a1 : a0;
        Bf3A& nxt = (s & 1) ? a0 : a1;
        if (s + 1 < NS) bf3_aload(nxt, img, pitch, li, q, s0 + s + 1);
        bf3_mfma6(cur, b.v[s], (s & 1) ? acc1 : acc0);
    }
}

// One (tile, k-step) block of one job of r3d_weight_planes, by one wave: block `blk` of `total` over all jobs (r3d_hip.h).
// Also run by rider workgroups of the embedding seam's launch (embed.hip: r3d_embed_fuse_fwd_planes).
__device__ __forceinline__ void weight_planes_block(const r3d_plane_job* __restrict__ jobs, int njobs, int total, int blk, int lane) {
    if (blk >= total) return;
    int j = 0;
    for (int t = 1; t < njobs; ++t) j += (blk >= jobs[t].first_block) ? 1 : 0;
    const r3d_plane_job J = jobs[j];
    const int ksteps = (J.K + 31) / 32;
    const int local = blk - J.first_block, t = local / ksteps, s = local - t * ksteps;
    const int n = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        const bool in = n < J.N && k < J.K;
        const size_t idx = J.transposed ? (size_t)(in ? k : 0) * J.ld + (in ? n : 0) : (size_t)(in ? n : 0) * J.ld + (in ? k : 0);
        const float x = J.src[idx];
        v[e] = in ? x : 0.f;
    }
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned short h0, m0, l0, h1, m1, l1;
        bf3_split1(v[2 * p], h0, m0, l0);
        bf3_split1(v[2 * p + 1], h1, m1, l1);
        hh[p] = (unsigned)h0 | ((unsigned)h1 << 16);
        mm[p] = (unsigned)m0 | ((unsigned)m1 << 16);
        ll[p] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    uint4* dst = reinterpret_cast<uint4*>(J.dst) + (size_t)local * (3 * 64) + lane;
    dst[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    dst[64] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    dst[128] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// rows 0..15 of a [3][16][pitch] image <- 16 rows x 128 columns of a dense fp32 matrix (thread = (row, 4 columns))
__device__ __forceinline__ void bf3_stage_tile(unsigned short* img, int pitch, int col0, const f32x4 v, int tid) {
    bf3_store4(img, pitch, tid >> 5, col0 + 4 * (tid & 31), v);
}

}  // namespace r3d
