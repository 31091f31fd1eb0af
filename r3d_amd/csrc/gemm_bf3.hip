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
#include <type_traits>
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// register stages of global loads are NATIVE vectors: with HIP's float4 struct the members become separate scalars, and where the
// register allocator fails to coalesce them with a load's 128-bit tuple it copies the just-loaded registers at the loop's back
// edge -- s_waitcnt vmcnt(0) on the newest loads every iteration (found in gemm_bf3_nt_kernel's ISA, round 3)
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Two consecutive-k values -> the three bf16 terms of each, already packed (element k in the low half of the dword:
// little-endian vector order).  Truncation split: h = the top 16 bits of x (v_perm_b32 takes the high halves of the pair
// in ONE full-rate instruction), x - h and (x - h) - m are exact in fp32 (8 of the 24 mantissa bits leave each time), so
// x = h + m + l EXACTLY.  4 full-rate VALU instructions per pair and level (perm, 2 x and, packed subtract).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#ifdef R3D_SPLIT_PACKED
    // (the two subtractions of a level as ONE v_pk_add_f32: written on a 2-vector, scalar floats compile to two v_sub_f32)
    const f32x2_t x = {x0, x1};
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const f32x2_t hf = {__builtin_bit_cast(float, u0 & 0xffff0000u), __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const f32x2_t r = x - hf;
    const float r0 = r[0], r1 = r[1];      // (__builtin_bit_cast straight on a vector ELEMENT reads element 0 for both)
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const f32x2_t mf = {__builtin_bit_cast(float, v0 & 0xffff0000u), __builtin_bit_cast(float, v1 & 0xffff0000u)};
    const f32x2_t q = r - mf;
    const float q0 = q[0], q1 = q[1];
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
#else
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u);
    const float r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u);
    const float q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
#endif
}
// eight consecutive-k values -> one 16-byte MFMA operand per plane
__device__ __forceinline__ void split3_oct(const float* v, uint4& h, uint4& m, uint4& l) {
    split3_pair(v[0], v[1], h.x, m.x, l.x);
    split3_pair(v[2], v[3], h.y, m.y, l.y);
    split3_pair(v[4], v[5], h.z, m.z, l.z);
    split3_pair(v[6], v[7], h.w, m.w, l.w);
}

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
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * lhi + j;
                    v[j] = (k < K && m_ok) ? d.alpha * araw[8 * s + j] : 0.f;
                }
                split3_oct(v, a_h[s], a_m[s], a_l[s]);
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
        f32x4n st0[8], st1[8], st2[8];
        auto load_panel = [&](f32x4n* breg, int turn) {
            const int n0 = pid(turn) * kP3N;
            const int nc = n0 + 4 * c4 < N ? n0 + 4 * c4 : 0;   // N % 4 == 0 (validated): a float4 is all-in or all-out
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = 4 * kq + (t & 3) + 64 * (t >> 2);
                const int kc = k < K ? k : K - 1;
                breg[t] = *reinterpret_cast<const f32x4n*>(d.B + (size_t)kc * d.ldb + nc);
            }
        };
        auto store_panel = [&](unsigned short* img, const f32x4n* breg, int turn) {
            const bool n_ok = pid(turn) * kP3N + 4 * c4 < N;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int k0 = 4 * kq + 64 * hb;
                float v[4][4];                                      // [n][k]
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = n_ok && (k0 + t < K);
                    const f32x4n b = breg[4 * hb + t];
                    v[0][t] = ok ? b[0] : 0.f; v[1][t] = ok ? b[1] : 0.f; v[2][t] = ok ? b[2] : 0.f; v[3][t] = ok ? b[3] : 0.f;
                }
                const int chunk = (k0 >> 2) ^ ((c4 >> 2) << 1);     // 8-byte chunk index, swizzled by (n >> 4) & 3
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint2 h, m, l;
                    split3_pair(v[c][0], v[c][1], h.x, m.x, l.x);
                    split3_pair(v[c][2], v[c][3], h.y, m.y, l.y);
                    unsigned short* row = img + (size_t)(4 * c4 + c) * kP3S + 4 * chunk;
                    *reinterpret_cast<uint2*>(row) = h;
                    *reinterpret_cast<uint2*>(row + kP3Plane) = m;
                    *reinterpret_cast<uint2*>(row + 2 * kP3Plane) = l;
                }
            }
        };
        // (every load is unconditional, turn indices clamped: a load under a branch makes hipcc wait vmcnt(0) at the next
        //  use, which exposes the round trip of the loads issued one turn earlier)
        const int lastt = nturn - 1;
        load_panel(st0, 0);                                         // the first loads go out before anything else
        load_panel(st1, min(1, lastt));
        load_panel(st2, min(2, lastt));
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
        auto turn = [&](int p, unsigned short* oth, f32x4n* nxt, f32x4n* mine) {
            if (p > 0) drain_c(p - 1);
            if (p + 1 < nturn) store_panel(oth, nxt, p + 1);
            load_panel(mine, min(p + 3, lastt));
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

// ---------------------------------------------------------------------------------------------------------
// Split-K NT product  C_partial[split][M][N] = A[M, k-range] . B[N, k-range]^T  on the bf16 matrix cores: the forward depth
// projection (model/futr_safuser_tokenfusion.py:194-195: [B S, 50176] x [H, 50176]^T).  Both operands are K-contiguous,
// which is what the bf16 MFMA wants (8 consecutive k per lane): a producer thread loads 8 consecutive fp32 of one row
// (two float4), splits them and writes three 16-byte bf16x8 fragments into [row][k] images of row stride BK + 8 bf16
// (conflict-free for the 16-byte reads of 16 consecutive rows and for the producers' stores).
// Workgroup = 512 threads: waves 0-3 consume (a 2 x 2 arrangement over the BM x BN tile, TM x TN MFMA tiles each; with a
// single tile per wave the even and odd k-steps go to two accumulators so that two dependent MFMA chains alternate),
// waves 4-7 produce (two register stages of loads in flight, LDS images double-buffered, one barrier per k-step).
// Output: raw fp32 slabs exactly as gemm_f32_kernel leaves them for split-K (the caller's reducer applies bias / LayerNorm /
// the epilogue), placed XCD-aware (the tiles of one K-split share linear id mod 8).
// ---------------------------------------------------------------------------------------------------------
// Profiling builds only (tools/nt_stub_probe.sh; results wrong by construction): R3D_NT_PROBE bits -- 1 consumers do nothing but
// the barriers, 8 consumers read their operands but issue no MFMA, 2 producers store the raw bits (no split), 4 producers
// store nothing (the loads are still waited for); 16 timeline: threads 0 (consumer wave 0) and 256 (producer wave 4) of workgroup 16
// store wall_clock64() (100 MHz) marks behind the slabs (the caller's workspace must have room: tools/nt_timeline.py).
#ifndef R3D_NT_PROBE
#define R3D_NT_PROBE 0
#endif
#if R3D_NT_PROBE & 16
#define R3D_NT_MARK(K) do { if (blockIdx.x == 16 && (threadIdx.x & 255) == 0)                                                       \
        reinterpret_cast<unsigned long long*>(d.partial + (size_t)NG * d.M * d.N)[(threadIdx.x >> 8) * 64 + (K)] = wall_clock64(); } while (0)
#define R3D_NT_CYC(K) do { if (blockIdx.x == 16 && (threadIdx.x & 255) == 0)                                                        \
        reinterpret_cast<unsigned long long*>(d.partial + (size_t)NG * d.M * d.N)[(threadIdx.x >> 8) * 64 + (K)] = __builtin_readcyclecounter(); } while (0)
// (the weight-gradient kernel has no workspace: its marks go behind the C matrix, whose allocation the probe tool oversizes)
#define R3D_TN_MARK(K) do { if (blockIdx.x == 16 && (threadIdx.x & 255) == 0)                                                       \
        reinterpret_cast<unsigned long long*>(d.C + (size_t)d.M * d.ldc)[(threadIdx.x >> 8) * 64 + (K)] = wall_clock64(); } while (0)
#else
#define R3D_NT_MARK(K) do { } while (0)
#define R3D_NT_CYC(K) do { } while (0)
#define R3D_TN_MARK(K) do { } while (0)
#endif
// One MFMA, then three VALU instructions, 24 times; the six LDS stores and four loads of the step spread between
// (sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x200 DS write, 0x020 VMEM read).  R3D_NTU_NOSCHED: leave it to hipcc.
#ifdef R3D_NTU_NOSCHED
#define R3D_NTU_SCHED
#else
#ifndef R3D_NTU_V
#define R3D_NTU_V 3
#endif
#define R3D_NTU_SCHED                                                                                    \
    _Pragma("unroll") for (int g_ = 0; g_ < 24; ++g_) {                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
        __builtin_amdgcn_sched_group_barrier(0x002, R3D_NTU_V, 0);                                        \
    }
#endif
// An optional SECOND product with the same M x N (r3d_gemm_bf3_nt_pair: the RGB embedding beside the depth projection): its
// K-splits take the workgroups after the first product's -- at the headline shape exactly the 12 of the 256 placed workgroups
// that the first product's 61 splits x 4 tiles leave without work.
struct NtSecond { const float* A; const float* B; float* partial; int lda, ldb, K, k_per_split, NG2; };
// R3D_NT_PROD producer threads (256: four producer waves, one per SIMD beside a consumer wave; 512: eight, two per SIMD, each with
// half the octets of a stage -- more independent split chains for the SIMD's issue port to pick from)
#ifndef R3D_NT_PROD
#define R3D_NT_PROD 512
#endif
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256 + R3D_NT_PROD, 1) void gemm_bf3_nt_kernel(const r3d_gemm_desc d, const int G, const int NG, const NtSecond s2) {
    constexpr int S = BK + 8;                                  // bf16 per image row
    constexpr int OPR = BK / 8;                                // octets per row
    constexpr int PLANE_A = BM * S, PLANE_B = BN * S;
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);             // bf16 elements per LDS stage
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int NKS = BK / 16;
    constexpr bool TWO = (TM * TN == 1);                       // one tile per wave: split the k-steps over two chains
    constexpr int NPROD = R3D_NT_PROD;
    constexpr int NOCT = (BM + BN) * OPR / NPROD;              // octets per producer thread and stage
    static_assert((BM + BN) * OPR % NPROD == 0, "producer mapping");
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];     // [2 stages][A planes | B planes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware placement (see gemm_f32_kernel): group g = one K-split goes to XCD g % 8, its G tiles sit at stride 8
    const int p = blockIdx.x, idx = p >> 3;
    int split = (idx / G) * 8 + (p & 7);
    const int tile = idx % G;
    const float* pA = d.A;
    const float* pB = d.B;
    float* ppart = d.partial;
    int plda = d.lda, pldb = d.ldb, pK = d.K, pkps = d.k_per_split;
    if (split >= NG) {                                              // (uniform) the second product's splits, or nothing
        split -= NG;
        if (split >= s2.NG2) return;
        pA = s2.A; pB = s2.B; ppart = s2.partial; plda = s2.lda; pldb = s2.ldb; pK = s2.K; pkps = s2.k_per_split;
    }
    const int tiles_n = (d.N + BN - 1) / BN;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int k_begin = split * pkps;
    const int k_end = min(pK, k_begin + pkps);
    const int nk = (k_end - k_begin + BK - 1) / BK;
#ifdef R3D_NT_PRIO
    if (wave < 4) __builtin_amdgcn_s_setprio((R3D_NT_PRIO) & 3); else __builtin_amdgcn_s_setprio(((R3D_NT_PRIO) >> 2) & 3);
#endif
    if (wave < 4) {
        // ================================= consumers =================================
        const int wm = wave >> 1, wn = wave & 1;
        f32x16 acc[TM][TN][TWO ? 2 : 1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int c = 0; c < (TWO ? 2 : 1); ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0.f;
        // Operands in two register sets, one k16 ahead: the reads of k16 g + 1 are issued BEFORE the MFMAs of k16 g -- also across
        // the stage boundary, where the barrier comes first and the next stage's first operands are then read under the current
        // stage's last MFMAs (the matrix pipe no longer drains at every barrier while the first ds_read of a stage is in flight).
        // The read at the very end fetches a stale stage and is dropped.
        const int oa = (wm * (BM / 2) + l31) * S + 8 * lhi, ob = 3 * PLANE_A + (wn * (BN / 2) + l31) * S + 8 * lhi;
        uint4 oah[2][TM], oam[2][TM], oal[2][TM], obh[2][TN], obm[2][TN], obl[2][TN];
        auto rd = [&](int set, const unsigned short* img, int ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned short* q = img + oa + (size_t)i * 32 * S + 16 * ks;
                oah[set][i] = *reinterpret_cast<const uint4*>(q);
                oam[set][i] = *reinterpret_cast<const uint4*>(q + PLANE_A);
                oal[set][i] = *reinterpret_cast<const uint4*>(q + 2 * PLANE_A);
            }
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                const unsigned short* q = img + ob + (size_t)jj * 32 * S + 16 * ks;
                obh[set][jj] = *reinterpret_cast<const uint4*>(q);
                obm[set][jj] = *reinterpret_cast<const uint4*>(q + PLANE_B);
                obl[set][jj] = *reinterpret_cast<const uint4*>(q + 2 * PLANE_B);
            }
        };
        R3D_NT_MARK(0);
        __syncthreads();                                            // stage 0 written
        rd(0, lds16, 0);
        for (int kt = 0; kt < nk; ++kt) {
            R3D_NT_MARK(1 + 2 * kt);
            const unsigned short* img = lds16 + (kt & 1) * STAGE;
            const unsigned short* nimg = lds16 + ((kt + 1) & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < ((R3D_NT_PROBE & 1) ? 0 : NKS); ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;              // (NKS is even: the parity carries over the stage boundary)
                if (ks + 1 < NKS) {
                    rd(nxt, img, ks + 1);
                } else {
                    R3D_NT_MARK(2 + 2 * kt);
                    __syncthreads();                                // every wave has read stage kt; stage kt + 1 is written
                    rd(nxt, nimg, 0);
                }
                const int c = TWO ? (ks & 1) : 0;
#define R3D_BF(x) __builtin_bit_cast(bf16x8, x)
#define R3D_TERM(A_, B_)                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jj = 0; jj < TN; ++jj)             \
                    if (R3D_NT_PROBE & 8) { acc[i][jj][c][0] += __builtin_bit_cast(float, A_[cur][i].x ^ B_[cur][jj].x); } else \
                    acc[i][jj][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(A_[cur][i]), R3D_BF(B_[cur][jj]), acc[i][jj][c], 0, 0, 0);
                R3D_TERM(oah, obl)                                  // small terms first; tiles alternate inside a term
                R3D_TERM(oal, obh)
                R3D_TERM(oam, obm)
                R3D_TERM(oah, obm)
                R3D_TERM(oam, obh)
                R3D_TERM(oah, obh)
#undef R3D_TERM
#undef R3D_BF
            }
            if (R3D_NT_PROBE & 1) __syncthreads();
        }
        R3D_NT_MARK(1 + 2 * nk);
        if (nk & 1) __syncthreads();                                // (the producers' loop runs whole pairs of k-steps)
        // raw partial sums -> slab `split` (C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
        float* slab = ppart + (size_t)split * d.M * d.N;
        const bool whole = m0 + BM <= d.M && n0 + BN <= d.N;        // (uniform: a whole tile stores without per-element branches)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 32 + l31;
                const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lhi;
                if (whole) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][0][r];
                        if (TWO) v += acc[i][j][TWO ? 1 : 0][r];
                        slab[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * d.N + n] = v;
                    }
                    continue;
                }
                if (n >= d.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][0][r];
                    if (TWO) v += acc[i][j][TWO ? 1 : 0][r];
                    if (m < d.M) slab[(size_t)m * d.N + n] = v;
                }
            }
        R3D_NT_MARK(2 + 2 * nk);
    } else {
        // ================================= producers =================================
        const int pt = tid - 256;
        // Per-thread row pointers, fixed for the whole k-loop (the descriptor fields are read ONCE here: selecting
        // between d.A and d.B per lane inside the loop turned them into vector-memory loads of the kernel arguments,
        // whose vmcnt waits serialised with the tile prefetches -- seen in the ISA).
        const float* rowp[NOCT];
        bool rok[NOCT];
#pragma unroll
        for (int t = 0; t < NOCT; ++t) {
            const int e = pt + NPROD * t, row = e / OPR;
            const bool isa = row < BM;                               // (compile-time per t when 256 * t is a multiple of BM * OPR)
            const int gr = isa ? m0 + row : n0 + (row - BM);
            const int lim = isa ? d.M : d.N;
            rok[t] = gr < lim;
            const size_t off = (size_t)(rok[t] ? gr : 0) * (size_t)(isa ? plda : pldb);
            rowp[t] = (isa ? pA : pB) + off + 8 * (e % OPR);
        }
        const int Kt = pK;
        // (native 4-vectors: with HIP's float4 STRUCT the members become separate scalars and the register allocator, failing to
        //  coalesce one octet's members with the load's 128-bit tuple, copied the JUST-LOADED registers at the loop's back edge --
        //  s_waitcnt vmcnt(0) on the newest loads every iteration, the whole prefetch distance lost: seen in the ISA, round 3)
        f32x4n s0[2 * NOCT], s1[2 * NOCT];
        // CHECK = false (whole tile, whole slice): the k offset of a step is UNIFORM -- no per-lane clamp, so it stays a scalar and
        // the per-thread row pointers are the only vector address registers that live across the loop
        auto load_stage = [&](f32x4n* reg, int kt, auto check) {
            constexpr bool CHECK = decltype(check)::value;
            const int k0 = k_begin + kt * BK;
#pragma unroll
            for (int t = 0; t < NOCT; ++t) {
                const float* src;
                if (CHECK) {
                    const int o8 = 8 * ((pt + NPROD * t) % OPR);
                    const int k = k0 + o8;
                    const int kc = k + 8 <= Kt ? k : Kt - 8;        // K % 8 == 0 (validated): an octet is all-in or all-out
                    src = rowp[t] + (kc - o8);
                } else {
                    src = rowp[t] + k0;
                }
                reg[2 * t] = *reinterpret_cast<const f32x4n*>(src);
                reg[2 * t + 1] = *reinterpret_cast<const f32x4n*>(src + 4);
            }
        };
        // CHECK = false: every row of the tile and every k of the slice exist (the headline shape): no zero-fill selects
        auto store_stage = [&](unsigned short* img, const f32x4n* reg, int kt, auto check) {
            constexpr bool CHECK = decltype(check)::value;
            const int k0 = k_begin + kt * BK;
#ifdef R3D_SPLIT_WIDE
            if (!CHECK) {
                // all NOCT octets level by level, a scheduling barrier between the levels: every instruction of a level is
                // independent of the others (ILP = 8 NOCT instead of the two dependent chains per pair hipcc emits by itself)
                constexpr int NE = 8 * NOCT;
                float x[NE], r[NE], q[NE];
                unsigned u[NE], w[NE], hp[NE / 2], mp[NE / 2], lp[NE / 2];
#pragma unroll
                for (int t = 0; t < NOCT; ++t) {
                    const f32x4n a4 = reg[2 * t], b4 = reg[2 * t + 1];
                    x[8 * t + 0] = a4[0]; x[8 * t + 1] = a4[1]; x[8 * t + 2] = a4[2]; x[8 * t + 3] = a4[3];
                    x[8 * t + 4] = b4[0]; x[8 * t + 5] = b4[1]; x[8 * t + 6] = b4[2]; x[8 * t + 7] = b4[3];
                }
#pragma unroll
                for (int i = 0; i < NE; ++i) u[i] = __builtin_bit_cast(unsigned, x[i]) & 0xffff0000u;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NE; ++i) r[i] = x[i] - __builtin_bit_cast(float, u[i]);
#pragma unroll
                for (int i = 0; i < NE / 2; ++i) hp[i] = __builtin_amdgcn_perm(u[2 * i + 1], u[2 * i], 0x07060302u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NE; ++i) w[i] = __builtin_bit_cast(unsigned, r[i]) & 0xffff0000u;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NE; ++i) q[i] = r[i] - __builtin_bit_cast(float, w[i]);
#pragma unroll
                for (int i = 0; i < NE / 2; ++i) mp[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NE / 2; ++i)
                    lp[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q[2 * i + 1]), __builtin_bit_cast(unsigned, q[2 * i]), 0x07060302u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NOCT; ++t) {
                    const int e = pt + NPROD * t, row = e / OPR, o = e % OPR;
                    const bool isa = row < BM;
                    unsigned short* dst = img + (isa ? (size_t)row * S : (size_t)3 * PLANE_A + (size_t)(row - BM) * S) + 8 * o;
                    const int plane = isa ? PLANE_A : PLANE_B;
                    *reinterpret_cast<uint4*>(dst) = make_uint4(hp[4 * t], hp[4 * t + 1], hp[4 * t + 2], hp[4 * t + 3]);
                    *reinterpret_cast<uint4*>(dst + plane) = make_uint4(mp[4 * t], mp[4 * t + 1], mp[4 * t + 2], mp[4 * t + 3]);
                    *reinterpret_cast<uint4*>(dst + 2 * plane) = make_uint4(lp[4 * t], lp[4 * t + 1], lp[4 * t + 2], lp[4 * t + 3]);
                }
                return;
            }
#endif
#pragma unroll
            for (int t = 0; t < NOCT; ++t) {
                const int e = pt + NPROD * t, row = e / OPR, o = e % OPR;
                const bool isa = row < BM;
                const f32x4n x = reg[2 * t], y = reg[2 * t + 1];
                float v[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                if (CHECK) {
                    const bool ok = rok[t] && (k0 + 8 * o < k_end);
                    if (!ok) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = 0.f;
                    }
                }
                uint4 h, m, l;
                if (R3D_NT_PROBE & 2) {
                    h = __builtin_bit_cast(uint4, x); m = __builtin_bit_cast(uint4, y); l = h;
                } else {
                    split3_oct(v, h, m, l);
                }
                unsigned short* dst = img + (isa ? (size_t)row * S : (size_t)3 * PLANE_A + (size_t)(row - BM) * S) + 8 * o;
                const int plane = isa ? PLANE_A : PLANE_B;
                if (R3D_NT_PROBE & 4) {
                    if ((h.x ^ m.y ^ l.z) == 0x12345678u) *reinterpret_cast<uint4*>(dst) = h;
                } else {
                    *reinterpret_cast<uint4*>(dst) = h;
                    *reinterpret_cast<uint4*>(dst + plane) = m;
                    *reinterpret_cast<uint4*>(dst + 2 * plane) = l;
                }
            }
        };
        // Every load below is UNCONDITIONAL (tile indices are clamped: a surplus load re-reads the last tile and its image
        // is never consumed): a load under a branch makes hipcc wait vmcnt(0) at the next use, i.e. for the loads issued
        // one step earlier as well -- measured 1.8 us per k-step (one HBM round trip) instead of 0.7.
        const int last = nk - 1;
        auto run = [&](auto check) {
            R3D_NT_MARK(0);
            load_stage(s0, 0, check);
            load_stage(s1, min(1, last), check);
            store_stage(lds16, s0, 0, check);
            load_stage(s0, min(2, last), check);
            R3D_NT_MARK(1);
            __syncthreads();                                        // stage 0 written
            // step kt: tile kt + 1 (register stage) -> the other image; refill that register stage with tile kt + 3
            // (straight-line body: with an exit between the two halves the compiler re-rolls the loop, rotates the two
            //  register stages through copies and has to wait for the NEWEST loads at every copy -- s_waitcnt vmcnt(0) once
            //  per k-step, the whole prefetch distance lost.  An odd nk runs one surplus half whose image nobody reads; the
            //  consumers match its barrier.)
            for (int kt = 0; kt < nk; kt += 2) {
                R3D_NT_MARK(2 + 2 * kt);
#if R3D_NT_PROBE & 16
                __builtin_amdgcn_s_waitcnt(0xF78);                  // vmcnt(8): this stage's loads have landed
                R3D_NT_MARK(32 + 2 * kt);
#endif
                store_stage(lds16 + STAGE, s1, min(kt + 1, last), check);
#if R3D_NT_PROBE & 16
                __builtin_amdgcn_sched_barrier(0);
                R3D_NT_MARK(33 + 2 * kt);
#endif
                load_stage(s1, min(kt + 3, last), check);
                __builtin_amdgcn_sched_barrier(0);                  // (keeps the other stage's split below these loads)
                R3D_NT_MARK(3 + 2 * kt);
                __syncthreads();
                R3D_NT_MARK(4 + 2 * kt);
                store_stage(lds16, s0, min(kt + 2, last), check);
                load_stage(s0, min(kt + 4, last), check);
                __builtin_amdgcn_sched_barrier(0);
                R3D_NT_MARK(5 + 2 * kt);
                __syncthreads();
            }
        };
        const bool full = m0 + BM <= d.M && n0 + BN <= d.N && k_begin + nk * BK <= k_end;
        if (full) run(std::false_type{}); else run(std::true_type{});
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same split-K NT product with UNIFORM waves (tile 11: 128 x 128 x 32).  In the role-split kernel above the producers'
// split (5.5 VALU instructions per operand element) is issued from OTHER waves than the MFMAs it has to hide under, and
// the SIMD's arbiter does not interleave the two streams well: measured per k-step (tools/nt_timeline.sh) the consumers'
// 48 MFMAs take 1.0 us, the producers' split + store 1.3-1.45 us beside them, and the consumers wait out the difference at
// the barrier (matrix cores 70 % busy inside the loop; wave priorities change nothing, v_pk_add_f32 for the subtractions
// is slower than two v_sub_f32).  Here every wave does both: wave w owns a 32 x 64 slice of the output (2 MFMA tiles, 24
// MFMAs per k-step) and a 1/8 share of the operand stream (two 8-element octets per lane and k-step: ~70 VALU, 6
// ds_write_b128, 4 loads), and the split of tile kt + 1 sits in the SAME instruction stream as the MFMAs of tile kt --
// 3 fillers per MFMA gap, which one wave's stream hides (MI355X_MICROARCH.md: <= 5 single-issue instructions per
// v_mfma_f32_32x32x16_bf16 gap).  Two register stages of loads in flight, LDS images double-buffered, one barrier per
// k-step, slabs and placement as above.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void gemm_bf3_nt_u_kernel(const r3d_gemm_desc d, const int G, const int NG) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int S = BK + 8, OPR = BK / 8;
    constexpr int PLANE = BM * S;                              // bf16 elements per plane (A and B alike)
    constexpr int STAGE = 6 * PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];     // [2 stages][A planes | B planes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int p = blockIdx.x, idx = p >> 3;
    const int split = (idx / G) * 8 + (p & 7), tile = idx % G;
    if (split >= NG) return;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int k_begin = split * d.k_per_split;
    const int k_end = min(d.K, k_begin + d.k_per_split);
    const int nk = (k_end - k_begin + BK - 1) / BK;
    const int wm = wave & 3, wn = wave >> 2;                   // rows 32 wm .., columns 64 wn ..
    // ---- this lane's two octets of the operand stream: e = tid, tid + 512 -> (row e / 4 of A | B, octet e % 4)
    const float* rowp[2];
    bool rok[2];
    int dsto[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = tid + 512 * t, row = e / OPR, o = e % OPR;
        const bool isa = row < BM;                              // (t == 0: A, t == 1: B)
        const int gr = isa ? m0 + row : n0 + (row - BM);
        rok[t] = gr < (isa ? d.M : d.N);
        rowp[t] = (isa ? d.A : d.B) + (size_t)(rok[t] ? gr : 0) * (size_t)(isa ? d.lda : d.ldb) + 8 * o;
        dsto[t] = (isa ? row * S : 3 * PLANE + (row - BM) * S) + 8 * o;
    }
    const int Kt = d.K;
    f32x4n s0[4], s1[4];
    auto load_stage = [&](f32x4n* reg, int kt) {
        const int k0 = k_begin + kt * BK;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int o8 = 8 * ((tid + 512 * t) % OPR);
            const int k = k0 + o8;
            const int kc = k + 8 <= Kt ? k : Kt - 8;            // K % 8 == 0 (validated)
            const float* src = rowp[t] + (kc - o8);
            reg[2 * t] = *reinterpret_cast<const f32x4n*>(src);
            reg[2 * t + 1] = *reinterpret_cast<const f32x4n*>(src + 4);
        }
    };
    auto store_stage = [&](unsigned short* img, const f32x4n* reg, int kt, auto check) {
        constexpr bool CHECK = decltype(check)::value;
        const int k0 = k_begin + kt * BK;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4n x = reg[2 * t], y = reg[2 * t + 1];
            float v[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            if (CHECK) {
                const bool ok = rok[t] && (k0 + 8 * ((tid + 512 * t) % OPR) < k_end);
                if (!ok) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = 0.f;
                }
            }
            uint4 h, m, l;
            split3_oct(v, h, m, l);
            unsigned short* dst = img + dsto[t];
            *reinterpret_cast<uint4*>(dst) = h;
            *reinterpret_cast<uint4*>(dst + PLANE) = m;
            *reinterpret_cast<uint4*>(dst + 2 * PLANE) = l;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // the MFMAs of one k-step on image `img` (2 k16 x 2 column tiles x 6 products)
    auto mma = [&](const unsigned short* img) {
        const unsigned short* ia = img + (size_t)(wm * 32 + l31) * S + 8 * lhi;
        const unsigned short* ib = img + 3 * PLANE + (size_t)(wn * 64 + l31) * S + 8 * lhi;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 a[3], b[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = *reinterpret_cast<const uint4*>(ia + pl * PLANE + 16 * ks);
                b[0][pl] = *reinterpret_cast<const uint4*>(ib + pl * PLANE + 16 * ks);
                b[1][pl] = *reinterpret_cast<const uint4*>(ib + pl * PLANE + 32 * S + 16 * ks);
            }
#define R3D_BF(x) __builtin_bit_cast(bf16x8, x)
#define R3D_T2(PA, PB)                                                                                              \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(a[PA]), R3D_BF(b[0][PB]), acc[0], 0, 0, 0);     \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(a[PA]), R3D_BF(b[1][PB]), acc[1], 0, 0, 0);
            R3D_T2(0, 2)                                            // h . l   (small terms first)
            R3D_T2(2, 0)                                            // l . h
            R3D_T2(1, 1)                                            // m . m
            R3D_T2(0, 1)                                            // h . m
            R3D_T2(1, 0)                                            // m . h
            R3D_T2(0, 0)                                            // h . h
#undef R3D_T2
#undef R3D_BF
        }
    };
    const int last = nk - 1;
    auto run = [&](auto check) {
        R3D_NT_MARK(0);
        load_stage(s0, 0);
        load_stage(s1, min(1, last));
        store_stage(lds16, s0, 0, check);
        load_stage(s0, min(2, last));
        R3D_NT_MARK(1);
        __syncthreads();                                            // image 0 written
        R3D_NT_CYC(60);
        // step kt: MFMAs on image kt & 1; tile kt + 1 (a register stage) is split into the other image in the same
        // instruction stream; that register stage is refilled with tile kt + 3.  (Straight-line pairs of steps, surplus
        // half for an odd nk, unconditional clamped loads: see the role-split kernel.)
        for (int kt = 0; kt < nk; kt += 2) {
            R3D_NT_MARK(2 + 2 * kt);
            mma(lds16);
            store_stage(lds16 + STAGE, s1, min(kt + 1, last), check);
            load_stage(s1, min(kt + 3, last));
            R3D_NTU_SCHED
            R3D_NT_MARK(3 + 2 * kt);
            __syncthreads();
            R3D_NT_MARK(4 + 2 * kt);
            if (kt + 1 < nk) mma(lds16 + STAGE);
            store_stage(lds16, s0, min(kt + 2, last), check);
            load_stage(s0, min(kt + 4, last));
            R3D_NTU_SCHED
            R3D_NT_MARK(5 + 2 * kt);
            __syncthreads();
        }
        R3D_NT_MARK(2 + 2 * nk);
        R3D_NT_CYC(61);
    };
    const bool full = m0 + BM <= d.M && n0 + BN <= d.N && k_begin + nk * BK <= k_end;
    if (full) run(std::false_type{}); else run(std::true_type{});
    // raw partial sums -> slab `split` (C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    float* slab = d.partial + (size_t)split * d.M * d.N;
    const bool whole = m0 + BM <= d.M && n0 + BN <= d.N;
    const int mb = m0 + wm * 32 + 4 * lhi;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (whole) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * d.N + n] = acc[j][r];
            continue;
        }
        if (n >= d.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < d.M) slab[(size_t)m * d.N + n] = acc[j][r];
        }
    }
    R3D_NT_MARK(3 + 2 * nk);
}

bool gemm_bf3_nt_ok(const r3d_gemm_desc& d) {
    if (d.layout != R3D_GEMM_NT || d.splitk <= 1 || !d.partial) return false;
    if ((d.K & 7) || (d.lda & 3) || (d.ldb & 3) || (d.k_per_split & (d.tile == 9 || d.tile == 11 ? 31 : 63)) || d.K < 64) return false;
    if (d.a_add || d.a_row_xor || d.adam_m || d.alpha != 1.0f) return false;
    return r3d_aligned16(d.A) && r3d_aligned16(d.B);
}

template <int BM, int BN, int BK>
static int launch_bf3_nt_cfg(const r3d_gemm_desc& d, hipStream_t s, const NtSecond s2 = NtSecond{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}) {
    const int tiles = r3d_cdiv(d.M, BM) * r3d_cdiv(d.N, BN);
    const int ns = r3d_cdiv(d.K, d.k_per_split);
    const size_t lds = (size_t)2 * 3 * (BM + BN) * (BK + 8) * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf3_nt_kernel<BM, BN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf3_nt_kernel<BM, BN, BK>), dim3(8 * tiles * r3d_cdiv(ns + s2.NG2, 8)), dim3(256 + R3D_NT_PROD), lds, s, d, tiles, ns, s2);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

// two planned split-K NT products with the same M x N in one launch of the 64 x 64 x 64 tile (see NtSecond)
int launch_gemm_bf3_nt_pair(const r3d_gemm_desc& d, const r3d_gemm_desc& e, hipStream_t s) {
    if (d.tile != 8 || e.tile != 8 || !gemm_bf3_nt_ok(d) || !gemm_bf3_nt_ok(e)) return R3D_EINVAL;
    if (d.M != e.M || d.N != e.N || d.partial == e.partial) return R3D_EINVAL;
    const NtSecond s2{e.A, e.B, e.partial, e.lda, e.ldb, e.K, e.k_per_split, r3d_cdiv(e.K, e.k_per_split)};
    return launch_bf3_nt_cfg<64, 64, 64>(d, s, s2);
}

int launch_gemm_bf3_nt(const r3d_gemm_desc& d, hipStream_t s) {
    if (!gemm_bf3_nt_ok(d)) return R3D_EINVAL;
    if (d.tile == 8) return launch_bf3_nt_cfg<64, 64, 64>(d, s);
    if (d.tile == 9) return launch_bf3_nt_cfg<128, 128, 32>(d, s);
    if (d.tile == 11) {
        const int tiles = r3d_cdiv(d.M, 128) * r3d_cdiv(d.N, 128);
        const int ns = r3d_cdiv(d.K, d.k_per_split);
        const size_t lds = (size_t)2 * 6 * 128 * 40 * sizeof(unsigned short);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf3_nt_u_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_bf3_nt_u_kernel, dim3(8 * tiles * r3d_cdiv(ns, 8)), dim3(512), lds, s, d, tiles, ns);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    return R3D_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------
// TN product  C[M, N] = alpha A[K, M]^T . B[K, N] (+ C)  on the bf16 matrix cores for the weight gradient of the depth
// projection when there are more rows than the persistent panel kernel holds in registers (K > 128 tokens or M > 128
// hidden units: BASELINE configs[2..4]).  128 x 128 output tiles, 32 k per step, the same consumer / producer split as
// gemm_bf3_nt_kernel; both operands are M / N-contiguous, so a producer thread loads an 8 (k) x 4 (m or n) block, splits
// it and writes four k-octets (one per column, three planes each) into [m or n][k] images of row stride 40 bf16, the
// octet position XOR-swizzled by (row >> 4) & 3 (conflict-free stores and operand reads).
// Placement: the tiles of one N panel (they share the B operand) get equal linear id mod 8, i.e. one XCD's L2.
// ---------------------------------------------------------------------------------------------------------
// ONE = true (tile 12): a SINGLE LDS stage (61 KB; 68 KB with the AdamW epilogue's tile image) and at most 128 registers, so TWO
// workgroups share a CU.  Inside a workgroup the producers then store a stage only after the consumers have read the previous
// one (two barriers per k-step, no overlap of the LDS stores with the MFMAs) -- the overlap comes from the OTHER workgroup of the
// CU, which is not in lockstep with this one: its MFMAs run under this one's split and stores, and its epilogue (the AdamW
// streaming) under this one's k-loop.
template <bool ONE>
__global__ __launch_bounds__(512, ONE ? 2 : 1) void gemm_bf3_tn_kernel(const r3d_gemm_desc d, const int G, const int NG) {
    constexpr int BM = 128, BN = 128, BK = 32, S = BK + 8;
    constexpr int kTnCS = BN + 4;                               // floats per row of the AdamW epilogue's tile image
    constexpr int PLANE = BM * S;                               // BM == BN: same plane size for both operands
    constexpr int STAGE = 6 * PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];     // [2 stages][A planes | B planes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int p = blockIdx.x, idx = p >> 3;
    const int tn = (idx / G) * 8 + (p & 7), tm = idx % G;
    if (tn >= NG) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = d.K, M = d.M, N = d.N;
    const int nk = (K + BK - 1) / BK;
    if (wave < 4) {
        // ================================= consumers: wave quadrant 64 x 64 = 2 x 2 MFMA tiles, four chains =============
        const int wm = wave >> 1, wn = wave & 1;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int ra = wm * 64 + l31, rb = wn * 64 + l31;       // rows of tile i = 0 / j = 0; i, j = 1: + 32
        const int xa0 = (ra >> 4) & 3, xa1 = ((ra + 32) >> 4) & 3, xb0 = (rb >> 4) & 3, xb1 = ((rb + 32) >> 4) & 3;
        if (!ONE) __syncthreads();                                  // stage 0 written
        for (int kt = 0; kt < nk; ++kt) {
            if (ONE) __syncthreads();                               // (A) the stage holds tile kt
            R3D_TN_MARK(1 + 2 * kt);
            const unsigned short* ia = lds16 + (ONE ? 0 : (kt & 1) * STAGE);
            const unsigned short* ib = ia + 3 * PLANE;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int o = 2 * ks + lhi;
                uint4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
                const unsigned short* qa0 = ia + (size_t)ra * S + 8 * (o ^ xa0);
                const unsigned short* qa1 = ia + (size_t)(ra + 32) * S + 8 * (o ^ xa1);
                const unsigned short* qb0 = ib + (size_t)rb * S + 8 * (o ^ xb0);
                const unsigned short* qb1 = ib + (size_t)(rb + 32) * S + 8 * (o ^ xb1);
                ah[0] = *reinterpret_cast<const uint4*>(qa0); am[0] = *reinterpret_cast<const uint4*>(qa0 + PLANE);
                al[0] = *reinterpret_cast<const uint4*>(qa0 + 2 * PLANE);
                ah[1] = *reinterpret_cast<const uint4*>(qa1); am[1] = *reinterpret_cast<const uint4*>(qa1 + PLANE);
                al[1] = *reinterpret_cast<const uint4*>(qa1 + 2 * PLANE);
                bh[0] = *reinterpret_cast<const uint4*>(qb0); bm[0] = *reinterpret_cast<const uint4*>(qb0 + PLANE);
                bl[0] = *reinterpret_cast<const uint4*>(qb0 + 2 * PLANE);
                bh[1] = *reinterpret_cast<const uint4*>(qb1); bm[1] = *reinterpret_cast<const uint4*>(qb1 + PLANE);
                bl[1] = *reinterpret_cast<const uint4*>(qb1 + 2 * PLANE);
#define R3D_BF(x) __builtin_bit_cast(bf16x8, x)
#define R3D_TERM(A_, B_)                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R3D_BF(A_[i]), R3D_BF(B_[j]), acc[i][j], 0, 0, 0);
                R3D_TERM(ah, bl)                                    // small terms first; the four tiles alternate inside a term
                R3D_TERM(al, bh)
                R3D_TERM(am, bm)
                R3D_TERM(ah, bm)
                R3D_TERM(am, bh)
                R3D_TERM(ah, bh)
#undef R3D_TERM
#undef R3D_BF
            }
            R3D_TN_MARK(2 + 2 * kt);
            __syncthreads();
        }
        R3D_TN_MARK(1 + 2 * nk);
        if (!ONE && (nk & 1)) __syncthreads();                      // (the producers' loop runs whole pairs of k-steps)
        // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const float alpha = d.alpha;
        const bool accum = d.accumulate != 0;
        if (d.adam_m) {                                             // AdamW epilogue (below): the raw tile goes to LDS
            float* ct = reinterpret_cast<float*>(lds16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ct[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * kTnCS + wn * 64 + j * 32 + l31] = acc[i][j][r];
        } else
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (n >= N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (m < M) {
                        float* c = d.C + (size_t)m * d.ldc + n;
                        float v = alpha * acc[i][j][r];
                        if (accum) v += *c;
                        *c = v;
                    }
                }
            }
    } else {
        // ================================= producers: one 8 (k) x 4 (m | n) block per thread and k-step ===================
        const int pt = tid - 256;
        const bool isa = pt < 128;
        const int b = isa ? pt : pt - 128;
        const int ko = b >> 5, x4 = b & 31;                        // k-octet 0..3, column group
        const int xg = (isa ? m0 : n0) + 4 * x4;                   // global column of the block
        const bool x_ok = xg < (isa ? M : N);                      // M % 4 == 0, N % 4 == 0 (validated)
        const float* colp = (isa ? d.A : d.B) + (x_ok ? xg : 0);
        const size_t ld = (size_t)(isa ? d.lda : d.ldb);
        unsigned short* imgrow = lds16 + (isa ? 0 : 3 * PLANE) + (size_t)(4 * x4) * S;
        const int sw = (x4 >> 2) & 3;                              // (row >> 4) & 3 for rows 4 x4 .. 4 x4 + 3
        f32x4n s0[8], s1[8];
        // (unconditional loads from clamped rows: see gemm_bf3_nt_kernel.  CHECK = false -- a whole tile and K a multiple of 32, the
        //  training shapes: no row clamps, no zero-fill selects, the row offsets of a step are scalars)
        auto load_stage = [&](f32x4n* reg, int kt, auto check) {
            constexpr bool CHECK = decltype(check)::value;
            const int kb = kt * BK + 8 * ko;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = CHECK ? (kb + j < K ? kb + j : K - 1) : kb + j;
                reg[j] = *reinterpret_cast<const f32x4n*>(colp + (size_t)k * ld);
            }
        };
        auto store_stage = [&](unsigned short* img, const f32x4n* reg, int kt, auto check) {
            constexpr bool CHECK = decltype(check)::value;
            const int kb = kt * BK + 8 * ko;
            float v[4][8];                                          // [column][k]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = !CHECK || (x_ok && (kb + j < K));
                v[0][j] = ok ? reg[j][0] : 0.f; v[1][j] = ok ? reg[j][1] : 0.f;
                v[2][j] = ok ? reg[j][2] : 0.f; v[3][j] = ok ? reg[j][3] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint4 h, m, l;
#ifdef R3D_TN_STUB            /* profiling builds: 1 = producers store the raw bits (no split), 2 = producers store nothing */
                h = make_uint4(__builtin_bit_cast(unsigned, v[c][0]), __builtin_bit_cast(unsigned, v[c][1]),
                               __builtin_bit_cast(unsigned, v[c][2]), __builtin_bit_cast(unsigned, v[c][3]));
                m = h; l = h;
                if (R3D_TN_STUB == 2) { if (h.x == 0x12345678u) *reinterpret_cast<uint4*>(img) = h; continue; }
#else
                split3_oct(v[c], h, m, l);
#endif
                unsigned short* dst = img + (imgrow - lds16) + (size_t)c * S + 8 * (ko ^ sw);
                *reinterpret_cast<uint4*>(dst) = h;
                *reinterpret_cast<uint4*>(dst + PLANE) = m;
                *reinterpret_cast<uint4*>(dst + 2 * PLANE) = l;
            }
        };
        const int last = nk - 1;
        auto run = [&](auto check) {
            R3D_TN_MARK(0);
            if (ONE) {
                load_stage(s0, 0, check);
                load_stage(s1, min(1, last), check);
                for (int kt = 0; kt < nk; kt += 2) {                // straight-line pairs (register stages s0 / s1 alternate)
                    store_stage(lds16, s0, kt, check);
                    load_stage(s0, min(kt + 2, last), check);
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();                                // (A)
                    __syncthreads();                                // (B) the consumers have read the stage
                    if (kt + 1 < nk) {
                        store_stage(lds16, s1, kt + 1, check);
                        load_stage(s1, min(kt + 3, last), check);
                        __builtin_amdgcn_sched_barrier(0);
                        __syncthreads();
                        __syncthreads();
                    }
                }
                return;
            }
            load_stage(s0, 0, check);
            load_stage(s1, min(1, last), check);
            store_stage(lds16, s0, 0, check);
            load_stage(s0, min(2, last), check);
            R3D_TN_MARK(1);
            __syncthreads();                                        // stage 0 written
            for (int kt = 0; kt < nk; kt += 2) {                    // (straight-line pairs: see gemm_bf3_nt_kernel)
                R3D_TN_MARK(2 + 2 * kt);
                store_stage(lds16 + STAGE, s1, min(kt + 1, last), check);
                load_stage(s1, min(kt + 3, last), check);
                __builtin_amdgcn_sched_barrier(0);                  // (keeps the other stage's split below these loads)
                R3D_TN_MARK(3 + 2 * kt);
                __syncthreads();
                R3D_TN_MARK(4 + 2 * kt);
                store_stage(lds16, s0, min(kt + 2, last), check);
                load_stage(s0, min(kt + 4, last), check);
                __builtin_amdgcn_sched_barrier(0);
                R3D_TN_MARK(5 + 2 * kt);
                __syncthreads();
            }
        };
        const bool full = m0 + BM <= M && n0 + BN <= N && (K % BK) == 0;
        if (full) run(std::false_type{}); else run(std::true_type{});
    }
    if (!d.adam_m) return;
    // ---- AdamW in the epilogue (r3d_gemm_desc::adam_*: C is the PARAMETER, the product its gradient, which is never stored).
    // All 512 threads stream parameter and both moments row-contiguously with 16-byte accesses while the other workgroups of
    // the chip multiply: at hidden >= 512 this kernel is bound by the matrix cores / the issue port and leaves HBM idle, and
    // the flat AdamW launch no longer reads and re-writes 86 % of the model.  Same arithmetic and order as adamw_kernel
    // (optim.hip) and as gemm_f32_kernel's epilogue.
    __syncthreads();                                                // the tile is in LDS (every wave is past its last stage)
    {
        const float* ct = reinterpret_cast<const float*>(lds16);
        const float lr = *d.adam_lr;
        const double stepd = (double)*d.adam_step;
        const float bc1 = (float)(1.0 - pow((double)d.adam_beta1, stepd));
        const float bc2_sqrt = (float)sqrt(1.0 - pow((double)d.adam_beta2, stepd));
        const float decay = 1.0f - lr * d.adam_wd;
        const float step_size = lr / bc1;
        const float b2 = d.adam_beta2, eps = d.adam_eps, gs = d.alpha * d.adam_gscale;
        const float omb1 = 1.0f - d.adam_beta1, omb2 = 1.0f - b2;
        // 8 float4 per thread and array in R3D_TN_ADAM_BATCH batches: all loads of a batch are in flight before the first
        // update (unconditional, from clamped addresses; the stores of out-of-range elements are skipped)
#define R3D_ADAM_E(c)                                              \
            {                                                      \
                const float gr = gg.c * gs;                        \
                pp[u].c *= decay;                                  \
                mm[u].c = mm[u].c + (gr - mm[u].c) * omb1;         \
                vv[u].c = vv[u].c * b2 + gr * gr * omb2;           \
                const float den = sqrtf(vv[u].c) / bc2_sqrt + eps; \
                pp[u].c -= step_size * (mm[u].c / den);            \
            }
#ifndef R3D_TN_ADAM_BATCH
#define R3D_TN_ADAM_BATCH 2
#endif
        constexpr int NB = 8 / R3D_TN_ADAM_BATCH;
#pragma unroll
        for (int half = 0; half < R3D_TN_ADAM_BATCH; ++half) {
            float4 pp[NB], mm[NB], vv[NB];
            size_t off[NB];
            bool in[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int f = tid + 512 * (NB * half + u);
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                const int gm = m0 + row, gn = n0 + 4 * c4;
                in[u] = gm < M && gn < N;
                off[u] = in[u] ? (size_t)gm * d.ldc + gn : 0;
                pp[u] = *reinterpret_cast<const float4*>(d.C + off[u]);
                mm[u] = *reinterpret_cast<const float4*>(d.adam_m + off[u]);
                vv[u] = *reinterpret_cast<const float4*>(d.adam_v + off[u]);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int f = tid + 512 * (NB * half + u);
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                const float4 gg = *reinterpret_cast<const float4*>(ct + row * kTnCS + 4 * c4);
                R3D_ADAM_E(x) R3D_ADAM_E(y) R3D_ADAM_E(z) R3D_ADAM_E(w)
                if (in[u]) {
                    *reinterpret_cast<float4*>(d.C + off[u]) = pp[u];
                    *reinterpret_cast<float4*>(d.adam_m + off[u]) = mm[u];
                    *reinterpret_cast<float4*>(d.adam_v + off[u]) = vv[u];
                }
            }
        }
#undef R3D_ADAM_E
    }
}

bool gemm_bf3_tn_ok(const r3d_gemm_desc& d) {
    if (d.layout != R3D_GEMM_TN || d.splitk > 1 || d.K < 16) return false;
    if ((d.M & 3) || (d.N & 3) || (d.lda & 3) || (d.ldb & 3)) return false;
    if (d.b_add || d.bias || d.pre_out || d.act || d.drop_mask || d.mul || d.res1 || d.res2) return false;
    if (d.bias_grad || d.c_row_xor) return false;
    if (d.adam_m && (d.accumulate || (d.ldc & 3) || (d.N & 3) || !r3d_aligned16(d.C))) return false;
    return r3d_aligned16(d.A) && r3d_aligned16(d.B);
}

int launch_gemm_bf3_tn(const r3d_gemm_desc& d, hipStream_t s) {
    if (!gemm_bf3_tn_ok(d)) return R3D_EINVAL;
    const int tm = r3d_cdiv(d.M, 128), tn = r3d_cdiv(d.N, 128);
    if (d.tile == 12) {                      // single LDS stage, two workgroups per CU
        const size_t lds1 = (size_t)128 * 132 * sizeof(float);      // >= one stage (61 440 B): the AdamW epilogue's tile image
        static bool attr1 = false;
        if (!attr1) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf3_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
            if (e != hipSuccess) return (int)e;
            attr1 = true;
        }
        hipLaunchKernelGGL(gemm_bf3_tn_kernel<true>, dim3(8 * tm * r3d_cdiv(tn, 8)), dim3(512), lds1, s, d, tm, tn);
        R3D_LAUNCH_CHECK();
        return R3D_OK;
    }
    const size_t lds = (size_t)2 * 6 * 128 * 40 * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf3_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_bf3_tn_kernel<false>, dim3(8 * tm * r3d_cdiv(tn, 8)), dim3(512), lds, s, d, tm, tn);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
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
