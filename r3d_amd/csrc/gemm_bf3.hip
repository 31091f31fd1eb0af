// bf16x3 MFMA path for the two depth-projection GEMMs (model/futr_safuser_tokenfusion.py:143-146,194-195 and the weight
// gradient autograd derives for it).  gfx950 has no TF32; its fp32 MFMA runs at the fp32 VECTOR peak (157 TFLOP/s) while
// v_mfma_f32_32x32x16_bf16 runs 16x faster.  Every fp32 operand is split EXACTLY into three bf16 terms by truncation,
//     x = h + m + l,   h = top 8 mantissa bits, m = the next 8, l = the last 8   (x - h and (x - h) - m are exact),
// and a . b is evaluated as the six leading products  h.h + h.m + m.h + h.l + l.h + m.m  on the bf16 matrix cores with
// fp32 accumulation; the three dropped terms (m.l, l.m, l.l) are <= 2^-24 |a b| each, i.e. the size of the rounding an
// fp32 fma makes anyway (SURVEY.md 8(d) "a 3-term bf16 split if proven within 1e-3": measured ~1e-6 against fp64, the
// same as the fp32 MFMA path -- tests/test_kernels_gpu.py::test_gemm_bf16x3_*).  6 MFMAs of K = 16 replace 8 of K = 2:
// 2.7x less matrix-core time, which turns both GEMMs from MFMA-bound into HBM-bound kernels.
// Selected per problem by r3d_gemm_desc::prec = 1 (the engine sets it for the depth projection only).
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 bit patterns of the three terms, each in the HIGH 16 bits of a dword (low bits zero)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
}
// two consecutive-k terms -> one dword (element k in the low half: little-endian vector order)
__device__ __forceinline__ unsigned pack2(unsigned lo_k, unsigned hi_k) { return (lo_k >> 16) | hi_k; }

__device__ __forceinline__ f32x16 mfma_bf3(const uint4 ah, const uint4 am, const uint4 al, const uint4 bh, const uint4 bm,
                                            const uint4 bl, f32x16 acc) {
#define R3D_BF(x) __builtin_bit_cast(bf16x8, x)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(ah), R3D_BF(bl), acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(al), R3D_BF(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(am), R3D_BF(bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(ah), R3D_BF(bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(am), R3D_BF(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(ah), R3D_BF(bh), acc, 0, 0, 0);
#undef R3D_BF
    return acc;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of a WIDE layer from FEW rows, C[M, N] = A^T . B with A [K, M], B [K, N], K <= 128, M <= 128, N huge
// (same contract as wgrad_panel_kernel in gemm_f32.hip: persistent workgroups over 64-column panels of B).
//
// With the matrix-core time cut 2.7x the phases of a panel (load, split, LDS write, MFMA, C store) are each ~1 us and,
// run by all waves in lockstep, simply add up (measured 24 us: skeleton 12 + MFMA 5.6 + C stores 5.2 + loads 1.5).  So
// the workgroup is split by ROLE:
//   waves 0-3  consumers: wave w owns output rows 32 w .. 32 w + 31 for both 32-column halves of the panel.  A^T for
//              its rows (all K, three bf16 planes) stays in 96 registers; per panel 8 k-steps x 2 halves x 6 products
//              = 96 MFMAs; the C stores of the previous panel are issued in between (4 per k-step) so that they
//              trickle into the memory pipe under the matrix cores instead of stalling the wave in one burst;
//   waves 4-7  producers: 256 threads fetch the next panels (three register stages: the loads of panel p + 3 are in
//              flight while p is multiplied), split them and write the [n][k] bf16 images -- their VALU / LDS work
//              shares each SIMD with one consumer wave whose time goes to the matrix core.
// The operands are M / N-contiguous in memory but the bf16 MFMA wants 8 consecutive k per lane, so the split happens on
// the way into LDS: a producer thread loads 4 (k) x 4 (n) blocks, splits them and writes, per column and plane, four
// k-consecutive bf16 as one 8-byte store into a [n][k] image of row stride 136 bf16 (= 68 dwords: 16-byte operand reads
// of 16 consecutive rows fall on distinct banks); the 8-byte chunks of a row are XOR-swizzled by (n >> 4) & 3 so that
// the producers' stores are conflict-free too.  Panels are dealt round-robin (turn i of workgroup w = panel
// i * gridDim.x + w): the resident workgroups sweep one contiguous range of columns at any moment.
// ---------------------------------------------------------------------------------------------------------
constexpr int kP3N = 64;
constexpr int kP3S = 136;                                 // bf16 elements per LDS row
constexpr int kP3Plane = kP3N * kP3S;                     // bf16 elements per plane
constexpr int kP3CS = kP3N + 4;                           // floats per row of the C tile image

__global__ __launch_bounds__(512, 1) void wgrad_panel_bf3_kernel(const r3d_gemm_desc d, const int npanels) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];     // [2 images][3 planes][64][136] bf16, C tile
    const int K = d.K, M = d.M, N = d.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nturn = (npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (nturn <= 0) return;
    auto pid = [&](int turn) { return turn * (int)gridDim.x + (int)blockIdx.x; };
    unsigned short* img0 = lds16;
    unsigned short* img1 = lds16 + 3 * kP3Plane;
    float* ctile = reinterpret_cast<float*>(lds16 + 6 * kP3Plane);           // [128][68] fp32: the finished panel's C tile

    // Barrier protocol (every wave executes the same sequence): B3 after the producers have written image 0, then one
    // barrier per turn.
    if (wave < 4) {
        // ================================= consumers =================================
        // A^T for this wave's 32 rows and every k: 64 loads per lane (128-byte row segments) issued TOGETHER -- one round
        // trip -- then split once into the three bf16 planes: 8 consecutive k = 16 s + 8 lhi .. + 7 per MFMA step s
        uint4 a_h[8], a_m[8], a_l[8];
        {
            const int m_row = wave * 32 + l31;
            const bool m_ok = m_row < M;
            float araw[64];
#pragma unroll
            for (int e = 0; e < 64; ++e) {
                const int k = 16 * (e >> 3) + 8 * lhi + (e & 7);
                const int kc = k < K ? k : K - 1;
                araw[e] = d.A[(size_t)kc * d.lda + (m_ok ? m_row : 0)];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                unsigned h[8], m[8], l[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * lhi + j;
                    split3((k < K && m_ok) ? d.alpha * araw[8 * s + j] : 0.f, h[j], m[j], l[j]);
                }
                a_h[s] = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
                a_m[s] = make_uint4(pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7]));
                a_l[s] = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
            }
        }
        __syncthreads();                                            // B3
        const int wrow = wave * 32 + 4 * lhi;
        const int x0 = (l31 >> 4) & 3, x1 = ((32 + l31) >> 4) & 3;  // octet swizzle of rows l31 and 32 + l31
        for (int p = 0; p < nturn; ++p) {
            const unsigned short* cur = (p & 1) ? img1 : img0;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const unsigned short* r0 = cur + (size_t)l31 * kP3S;
            const unsigned short* r1 = cur + (size_t)(32 + l31) * kP3S;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int o = 2 * s + lhi;
                const unsigned short* q0 = r0 + 8 * (o ^ x0);
                const unsigned short* q1 = r1 + 8 * (o ^ x1);
                const uint4 bh0 = *reinterpret_cast<const uint4*>(q0);
                const uint4 bm0 = *reinterpret_cast<const uint4*>(q0 + kP3Plane);
                const uint4 bl0 = *reinterpret_cast<const uint4*>(q0 + 2 * kP3Plane);
                const uint4 bh1 = *reinterpret_cast<const uint4*>(q1);
                const uint4 bm1 = *reinterpret_cast<const uint4*>(q1 + kP3Plane);
                const uint4 bl1 = *reinterpret_cast<const uint4*>(q1 + 2 * kP3Plane);
                // the two tiles' chains alternate: a dependent MFMA (SrcC = the previous result) issued back to back
                // waits out the 8 passes of its predecessor -- measured 60 cycles per MFMA with one chain after the other
#define R3D_BF(x) __builtin_bit_cast(bf16x8, x)
#define R3D_MM(a, b0, b1)                                                                                   \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(a), R3D_BF(b0), acc0, 0, 0, 0);      \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(a), R3D_BF(b1), acc1, 0, 0, 0);
                R3D_MM(a_h[s], bl0, bl1)                            // small terms first
                R3D_MM(a_l[s], bh0, bh1)
                R3D_MM(a_m[s], bm0, bm1)
                R3D_MM(a_h[s], bm0, bm1)
                R3D_MM(a_m[s], bh0, bh1)
                R3D_MM(a_h[s], bh0, bh1)
#undef R3D_MM
#undef R3D_BF
            }
            __syncthreads();                                        // mid-turn: the producers have drained the C tile
            // the panel's C tile goes to LDS; the producers stream it out with 16-byte row-contiguous stores next turn
            // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wrow + (r & 3) + 8 * (r >> 2);
                ctile[m * kP3CS + l31] = acc0[r];
                ctile[m * kP3CS + 32 + l31] = acc1[r];
            }
            __syncthreads();
        }
        __syncthreads();                                            // (the producers' extra half turn: last C tile out)
    } else {
        // ================================= producers =================================
        const int pt = tid - 256;                                  // 0..255
        const int kq = pt >> 4, c4 = pt & 15;                      // two blocks per thread: k0 = 4 kq and 4 kq + 64; n = 4 c4 ..
        float4 st0[8], st1[8], st2[8];
        auto load_panel = [&](float4* breg, int turn) {
            const int n0 = pid(turn) * kP3N;
            const int nc = n0 + 4 * c4 < N ? n0 + 4 * c4 : 0;   // N % 4 == 0 (validated): a float4 is all-in or all-out
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = 4 * kq + (t & 3) + 64 * (t >> 2);
                const int kc = k < K ? k : K - 1;
                breg[t] = *reinterpret_cast<const float4*>(d.B + (size_t)kc * d.ldb + nc);
            }
        };
        auto store_panel = [&](unsigned short* img, const float4* breg, int turn) {
            const bool n_ok = pid(turn) * kP3N + 4 * c4 < N;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int k0 = 4 * kq + 64 * hb;
                float v[4][4];                                      // [n][k]
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = n_ok && (k0 + t < K);
                    const float4 b = breg[4 * hb + t];
                    v[0][t] = ok ? b.x : 0.f; v[1][t] = ok ? b.y : 0.f; v[2][t] = ok ? b.z : 0.f; v[3][t] = ok ? b.w : 0.f;
                }
                const int chunk = (k0 >> 2) ^ ((c4 >> 2) << 1);     // 8-byte chunk index, swizzled by (n >> 4) & 3
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned h[4], m[4], l[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) split3(v[c][t], h[t], m[t], l[t]);
                    unsigned short* row = img + (size_t)(4 * c4 + c) * kP3S + 4 * chunk;
                    *reinterpret_cast<uint2*>(row) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
                    *reinterpret_cast<uint2*>(row + kP3Plane) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
                    *reinterpret_cast<uint2*>(row + 2 * kP3Plane) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
                }
            }
        };
        load_panel(st0, 0);                                         // the first loads go out before anything else
        if (1 < nturn) load_panel(st1, 1);
        if (2 < nturn) load_panel(st2, 2);
        store_panel(img0, st0, 0);
        __syncthreads();                                            // B3
        // C tile of the previous turn: LDS -> memory, 128 rows x 256 bytes, 8 float4 per thread
        auto drain_c = [&](int turn_done) {
            const int n0 = pid(turn_done) * kP3N;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int f = pt + 256 * t, m = f >> 4, q = (f & 15) << 2;
                const float4 v = *reinterpret_cast<const float4*>(ctile + m * kP3CS + q);
                if (m < M && n0 + q < N) *reinterpret_cast<float4*>(d.C + (size_t)m * d.ldc + n0 + q) = v;
            }
        };
        // turn p: the C tile of panel p - 1 leaves, panel p + 1 (a register stage) goes to the other image, the stage
        // panel p came from is refilled with panel p + 3.  Stage of panel q = q % 3; 6 = lcm(2 images, 3 stages).
        auto turn = [&](int p, unsigned short* oth, float4* nxt, float4* mine) {
            if (p > 0) drain_c(p - 1);
            if (p + 1 < nturn) store_panel(oth, nxt, p + 1);
            if (p + 3 < nturn) load_panel(mine, p + 3);
            __syncthreads();                                        // mid-turn
            __syncthreads();
        };
        for (int p = 0; p < nturn; p += 6) {
            turn(p, img1, st1, st0);
            if (p + 1 < nturn) turn(p + 1, img0, st2, st1);
            if (p + 2 < nturn) turn(p + 2, img1, st0, st2);
            if (p + 3 < nturn) turn(p + 3, img0, st1, st0);
            if (p + 4 < nturn) turn(p + 4, img1, st2, st1);
            if (p + 5 < nturn) turn(p + 5, img0, st0, st2);
        }
        drain_c(nturn - 1);
        __syncthreads();
    }
}

int launch_wgrad_panel_bf3(const r3d_gemm_desc& d, hipStream_t s) {
    const int npanels = r3d_cdiv(d.N, kP3N);
    const int per = r3d_cdiv(npanels, 256);                 // panels per workgroup: balanced over <= 256 workgroups
    const int wgs = r3d_cdiv(npanels, per);
    const size_t lds = (size_t)2 * 3 * kP3Plane * sizeof(unsigned short) + (size_t)128 * kP3CS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_panel_bf3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_panel_bf3_kernel, dim3(wgs), dim3(512), lds, s, d, npanels);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

}  // namespace r3d
