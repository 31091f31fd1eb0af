// Building blocks of the row-local chain kernels (fuser_chain.hip, decoder_chain.hip): a workgroup of 8 waves owns 16
// complete token rows of hidden size 128; wave w computes the 16 x 16 output tiles of a stage with v_mfma_f32_16x16x4_f32
// (exact fp32), its weight chunks (16 rows x 128 k for y = x W^T products, 128 k-rows x 16 columns for dx = dy W products)
// travelling global -> registers -> a wave-private LDS region -> MFMA operand layout.
#pragma once
#include "common.h"

namespace r3d {

// profiling aid: thread 0 of the chosen workgroup stores the 100 MHz wall clock at a stage boundary
#define R3D_CHAIN_MARK(TL, WG_OK, K) do { if ((TL) && (WG_OK) && threadIdx.x == 0) (TL)[(K)] = wall_clock64(); } while (0)

// An optional dropout keep-mask as an UNCONDITIONAL load (a load under a branch -- even a uniform `if (ptr)` -- is waited
// for on the spot): an absent mask reads a valid fallback address with stride 0; the raw byte stays in a register and is
// turned into the keep factor only where it is used, so that the prologue's loads need no arithmetic and no wait.
struct FcMaskSrc {
    const uint8_t* p; size_t ld; bool on;
    __device__ __forceinline__ FcMaskSrc(const uint8_t* m, const void* fallback, size_t ld_)
        : p(m ? m : reinterpret_cast<const uint8_t*>(fallback)), ld(m ? ld_ : 0), on(m != nullptr) {}
    __device__ __forceinline__ uint8_t raw(size_t row, int col) const { return p[row * ld + (on ? col : 0)]; }
    __device__ __forceinline__ float keep(uint8_t b, float scale) const { return on ? scale * (float)b : 1.f; }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFcH = 128;                   // hidden size this file is compiled for
constexpr int kFcRows = 16;                 // token rows per workgroup
constexpr int kFcP1 = kFcH + 4;             // LDS pitch of a [16][128] activation tile (conflict-free 16-byte operand reads)
constexpr int kFcP4 = 4 * kFcH + 4;         // ... of the [16][512] MLP activation
constexpr int kFcWP = kFcH + 4;             // ... of a wave's 16 x 128 weight chunk
constexpr float kFcEps = 1e-5f;
__device__ __forceinline__ float fc_row16_sum(float v) {
    v += dpp_mov_f<0x128>(v);
    v += dpp_mov_f<0x124>(v);
    v += dpp_mov_f<0x122>(v);
    v += dpp_mov_f<0x121>(v);
    return v;
}

// One 16-row x 128-k weight chunk in flight: lane l of load i holds 16 bytes of row 2i + (l >> 5) at k = 4 (l & 31).
// Rows >= nvalid read row nvalid - 1 (heads whose row count is not a multiple of 16): every load is unconditional.
struct FcW { f32x4 v[8]; };
__device__ __forceinline__ void fc_wload(FcW& r, const float* w, int ldw, int nvalid, int lane) {
    const float* p = w + 4 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int row = 2 * i + (lane >> 5);
        row = row < nvalid ? row : nvalid - 1;
        r.v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)row * ldw);
    }
}
__device__ __forceinline__ void fc_wstore(const FcW& r, float* wl, int lane) {
    float* p = wl + (lane >> 5) * kFcWP + 4 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(p + 2 * i * kFcWP) = r.v[i];
}
// operand registers of one 128-deep chunk (lane = row li, k group q; see gemm_ln.hip: the k order inside a 16-deep step
// is permuted identically on both operands, so one 16-byte read feeds four MFMAs)
struct FcOp { f32x4 v[8]; };
__device__ __forceinline__ void fc_opload(FcOp& o, const float* rowk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = *reinterpret_cast<const f32x4*>(rowk + 16 * j);
}
__device__ __forceinline__ void fc_mfma(const FcOp& a, const FcOp& b, f32x4& acc0, f32x4& acc1) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j][0], b.v[j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j + 1][0], b.v[j + 1][0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j][1], b.v[j][1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j + 1][1], b.v[j + 1][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j][2], b.v[j][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j + 1][2], b.v[j + 1][2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j][3], b.v[j][3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j + 1][3], b.v[j + 1][3], acc1, 0, 0, 0);
    }
}
// chunk product: B operand from the wave's weight region, A operand already in registers
__device__ __forceinline__ void fc_chunk(const FcOp& a, const float* wr, f32x4& acc0, f32x4& acc1) {
    FcOp b;
    fc_opload(b, wr);
    fc_mfma(a, b, acc0, acc1);
}

// LayerNorm of 16 complete rows held in the accumulator layout (lane: column c of the wave's tile, rows 4q .. 4q + 3):
// 16 lanes by DPP, 8 waves through LDS, two-pass mean / variance as nn.LayerNorm.  Two workgroup barriers.
__device__ __forceinline__ void fc_layernorm(const float (&v)[4], float (*red)[8][kFcRows], int wave, int li, int q,
                                             float (&mean)[4], float (&rstd)[4]) {
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = fc_row16_sum(v[i]);
    if (li == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[0][wave][4 * q + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) m += red[0][w8][4 * q + i];
        mean[i] = m * (1.0f / (float)kFcH);
        const float d = v[i] - mean[i];
        s[i] = fc_row16_sum(d * d);
    }
    if (li == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[1][wave][4 * q + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m2 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) m2 += red[1][w8][4 * q + i];
        rstd[i] = 1.0f / sqrtf(m2 * (1.0f / (float)kFcH) + kFcEps);
    }
}

// ---- dx = dy . W products (W as nn.Linear stores it: [out, in] = [k, n]) -----------------------------------------------------
// A tile's weight chunk is 128 k-rows x 16 columns, staged k-major in the wave's region (pitch 20 floats: the four k groups
// of an MFMA operand read land on disjoint banks) and read back with 32 scalar LDS loads in the same (k group, k) register
// layout the NT chunks use, so the MFMA sequence is shared.
constexpr int kFbWP = 20;
// lane l of load i: 16 bytes of k-row (l >> 2) + 16 i at column 4 (l & 3); k-rows >= kvalid read row kvalid - 1 (the A
// operand is zero there)
__device__ __forceinline__ void fb_wload(FcW& r, const float* w, int ldw, int kvalid, int lane) {
    const float* p = w + 4 * (lane & 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int k = (lane >> 2) + 16 * i;
        k = k < kvalid ? k : kvalid - 1;
        r.v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)k * ldw);
    }
}
__device__ __forceinline__ void fb_wstore(const FcW& r, float* wl, int lane) {
    float* p = wl + (lane >> 2) * kFbWP + 4 * (lane & 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(p + 16 * i * kFbWP) = r.v[i];
}
__device__ __forceinline__ void fb_chunk(const FcOp& a, const float* wl, int li, int q, f32x4& acc0, f32x4& acc1) {
    FcOp b;
    const float* p = wl + (4 * q) * kFbWP + li;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int t = 0; t < 4; ++t) b.v[j][t] = p[(16 * j + t) * kFbWP];
    }
    fc_mfma(a, b, acc0, acc1);
}

// Two simultaneous sums over the 128 columns for NR row slots per lane group (slot = NR q + i): 16 lanes by DPP, the 8
// waves through LDS; one workgroup barrier.  red: [2][8][16] floats.
template <int NR>
__device__ __forceinline__ void fb_rowsum2(const float (&a)[NR], const float (&b)[NR], float* red, int wave, int li, int q,
                                           float (&sa)[NR], float (&sb)[NR]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) { sa[i] = fc_row16_sum(a[i]); sb[i] = fc_row16_sum(b[i]); }
    if (li == 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            red[(0 * 8 + wave) * kFcRows + NR * q + i] = sa[i];
            red[(1 * 8 + wave) * kFcRows + NR * q + i] = sb[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
            x += red[(0 * 8 + w8) * kFcRows + NR * q + i];
            y += red[(1 * 8 + w8) * kFcRows + NR * q + i];
        }
        sa[i] = x * (1.0f / (float)kFcH);
        sb[i] = y * (1.0f / (float)kFcH);
    }
}

}  // namespace r3d
