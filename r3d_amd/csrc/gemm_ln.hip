// nn.Linear -> (dropout, residuals) -> LayerNorm in ONE launch with row-complete workgroups (hidden = 128).
//
// Every LayerNorm site of the step follows a GEMM whose N is the hidden size.  As two launches the pair costs a kernel
// boundary plus a memory round trip of the pre-norm rows (the LayerNorm launch alone is ~4.5 us of pure latency: its
// input was just written through another XCD's L2).  Here a workgroup owns 16 COMPLETE rows: 8 waves, wave w computes
// the 16 x 16 tile of columns [16w, 16w + 16) with v_mfma_f32_16x16x4_f32 (exact fp32).  Both operands are
// K-contiguous: coalesced 16-byte global loads -> LDS (the 16 A rows once for the whole K behind the kernel's only
// product-phase barrier; each wave's 16 W rows in 128-deep chunks in a wave-private region, the next chunks' loads in
// flight under the current chunk's MFMAs) -> conflict-free 16-byte LDS reads in the MFMA operand layout.  The epilogue applies bias / dropout mask / two residuals, stores the pre-norm rows (the backward reads them),
// reduces each row over the 16 lanes of a DPP row and over the 8 waves through 2 x 128 floats of LDS (two-pass mean /
// variance, as nn.LayerNorm), and stores the normalised rows, the row statistics and -- optionally -- the mean over
// each (2n, 2n+1) row pair (torch.mean over the two modality tokens, model/futr_safuser_tokenfusion.py:94).
// Up to 4 independent jobs share a launch (blockIdx -> job through a prefix table in the kernel arguments); a job with
// K == 0 is a plain LayerNorm of rows that are already in pre_out.
//
// Replaces (reference): transformerblock.py:131-134 (attn.proj + residual -> norm2; mlp fc2 + residual),
// futr_safuser_tokenfusion.py:92-94 (x + x_res -> norm -> mean), transformer.py:292-293,304-306 (out_proj -> dropout ->
// residual -> norm1 / norm2).
#include "common.h"
#include "mha_small.h"
#include "../../include/r3d_hip.h"

namespace r3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef r3d_gemm_ln_job GemmLnJob;
constexpr int kGlnH = 128;
constexpr int kGlnRows = 16;
constexpr float kGlnEps = 1e-5f;

struct GemmLnArgs {
    GemmLnJob j[4];
    int prefix[5];
    int njobs;
    // rider: an independent small attention core (dh 16, 8 queries, <= 64 keys; mha_small.h) -- workgroups past prefix[4]
    // run 8 (clip, head) units each, one per wave.  The query self-attention of decoder layer 0 depends on parameters
    // only (tgt = 0), so it needs no launch of its own: it rides beside the fuser's attn.proj + norm2.
    MhaArgs mha;
    int mha_units;
};

// sum over the 16 lanes of a DPP row; every lane of the row ends with the row's value
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov_f<0x128>(v);
    v += dpp_mov_f<0x124>(v);
    v += dpp_mov_f<0x122>(v);
    v += dpp_mov_f<0x121>(v);
    return v;
}

constexpr int kGlnWP = 132;                 // LDS row pitch (floats) of a 128-deep W chunk: conflict-free 16-byte reads
constexpr int kGlnMaxK = 512;
constexpr int kGlnAFloats = kGlnRows * (kGlnMaxK + 4);
constexpr int kGlnWFloats = 8 * 16 * kGlnWP;
constexpr int kGlnLdsBytes = (kGlnAFloats + kGlnWFloats + 2 * 8 * kGlnRows) * 4;

// One 128-deep chunk of the wave's 16 W rows, global -> registers: lane l of load i reads 16 bytes of row 2i + (l >> 5)
// at k = 4 (l & 31) -- adjacent lanes read adjacent addresses.  (Reading the MFMA operand layout straight from global
// memory -- lane = row -- was built first: the texture path serves 4 different rows per quad at a quarter of its rate,
// 3.5 us per chunk.)
struct GlnW { f32x4 v[8]; };   // (native vectors: HIP's float4 is a struct whose copies are memcpys that pin the array in scratch)
__device__ __forceinline__ void gln_wload(GlnW& r, const float* wrow, int ldw, int lane) {
    const float* p = wrow + (size_t)(lane >> 5) * ldw + 4 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(2 * i) * ldw);
}
__device__ __forceinline__ void gln_wstore(const GlnW& r, float* ws, int lane) {
    float* p = ws + (lane >> 5) * kGlnWP + 4 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(p + 2 * i * kGlnWP) = r.v[i];
}

// 32 MFMAs of one chunk: operands from LDS in the MFMA's lane layout (lane = row li, k group q); the k order inside a
// 16-deep step is permuted identically on both operands, so one 16-byte read per operand feeds four MFMAs.
__device__ __forceinline__ void gln_chunk(const float* as, const float* ws, f32x4& acc0, f32x4& acc1) {
    f32x4 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = *reinterpret_cast<const f32x4*>(as + 16 * j);
        b[j] = *reinterpret_cast<const f32x4*>(ws + 16 * j);
    }
#pragma unroll
    for (int j = 0; j < 8; j += 2) {                         // two accumulator chains alternate
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], b[j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1][0], b[j + 1][0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], b[j][1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1][1], b[j + 1][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], b[j][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1][2], b[j + 1][2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], b[j][3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1][3], b[j + 1][3], acc1, 0, 0, 0);
    }
}

// The 16 x 16 tile of wave `wave` over NCH chunks of 128 k.  A (16 rows x K, shared by the 8 waves) is staged once for
// the whole K behind one barrier; every wave stages its own 16 W rows chunk by chunk in a private LDS region (LDS
// operations of one wave execute in order, so no barrier is needed there), the next chunk's global loads in flight
// under the current chunk's MFMAs.
template <int NCH>
__device__ __forceinline__ void gln_product(const GemmLnJob& J, int row0, float* lds, f32x4& acc0, f32x4& acc1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    constexpr int AP = 128 * NCH + 4;
    float* as = lds;
    float* ws = lds + kGlnAFloats + wave * (16 * kGlnWP);
    const float* ag = J.A + (size_t)(row0 + (tid >> 5)) * J.lda + 4 * (tid & 31);
    const float* wg = J.W + (size_t)(wave * 16) * J.ldw;
    f32x4 av[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) av[ch] = *reinterpret_cast<const f32x4*>(ag + 128 * ch);
    GlnW w0, w1;
    gln_wload(w0, wg, J.ldw, lane);
    if (NCH > 1) gln_wload(w1, wg + 128, J.ldw, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
        *reinterpret_cast<f32x4*>(as + (tid >> 5) * AP + 128 * ch + 4 * (tid & 31)) = av[ch];
    gln_wstore(w0, ws, lane);
    if (NCH > 2) gln_wload(w0, wg + 256, J.ldw, lane);          // chunk c lives in w[c & 1]; stored -> free -> chunk c + 2
    __syncthreads();
    const float* ar = as + li * AP + 4 * q;
    const float* wr = ws + li * kGlnWP + 4 * q;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        gln_chunk(ar + 128 * i, wr, acc0, acc1);
        if (i + 1 < NCH) {
            __builtin_amdgcn_sched_barrier(0);
            if (i & 1) {
                gln_wstore(w0, ws, lane);
                if (i + 3 < NCH) gln_wload(w0, wg + 128 * (i + 3), J.ldw, lane);
            } else {
                gln_wstore(w1, ws, lane);
                if (i + 3 < NCH) gln_wload(w1, wg + 128 * (i + 3), J.ldw, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(512) void gemm_ln_fwd_kernel(const GemmLnArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float (*red)[8][kGlnRows] = reinterpret_cast<float (*)[8][kGlnRows]>(lds + kGlnAFloats + kGlnWFloats);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, q = lane >> 4;
    if ((int)blockIdx.x >= args.prefix[4]) {                    // rider role (see GemmLnArgs)
        const int unit = ((int)blockIdx.x - args.prefix[4]) * 8 + wave;
        static_assert(8 * mha_small_lds_floats(16, 8) <= kGlnAFloats + kGlnWFloats, "rider LDS");
        if (unit < args.mha_units) mha_fwd_small_unit<16, 8, false>(args.mha, unit, lds + wave * mha_small_lds_floats(16, 8));
        return;
    }
    int jb = 0;
#pragma unroll
    for (int t = 1; t < 4; ++t) jb += (t < args.njobs && (int)blockIdx.x >= args.prefix[t]) ? 1 : 0;
    const GemmLnJob& J = args.j[jb];
    const int row0 = ((int)blockIdx.x - args.prefix[jb]) * kGlnRows;
    const int c = wave * 16 + li;                       // this lane's output column
    const int K = J.K;

    // ---- epilogue operands first: they are as cold as the tile operands.  Every load is unconditional (an absent
    // operand reads gamma instead and is discarded): a load under a branch is waited for on the spot.
    // A K == 0 job (plain LayerNorm) takes the same path with its rows as the only residual and no product.
    const bool prod = K > 0;
    const float gam = J.gamma[c], bet = J.beta[c];
    const bool has_b = prod && J.bias != nullptr;
    const bool has_d = prod && J.drop_mask != nullptr, has_r1 = !prod || J.res1 != nullptr, has_r2 = prod && J.res2 != nullptr;
    const float bia = (has_b ? J.bias : J.gamma)[c];
    const uint8_t* dmp = has_d ? J.drop_mask : reinterpret_cast<const uint8_t*>(J.gamma);
    const float* r1p = has_r1 ? (prod ? J.res1 : J.pre_out) : J.gamma;
    const float* r2p = has_r2 ? J.res2 : J.gamma;
    const int ldd = has_d ? J.lddrop : 0, ld1 = has_r1 ? (prod ? J.ldr1 : J.ldpre) : 0, ld2 = has_r2 ? J.ldr2 : 0;
    uint8_t dmb[4];
    float r1[4], r2[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t r = (size_t)(row0 + 4 * q + i);
        dmb[i] = dmp[r * ldd + c];
        r1[i] = r1p[r * ld1 + c];
        r2[i] = r2p[r * ld2 + c];
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (prod) {
        if (K == 128) gln_product<1>(J, row0, lds, acc0, acc1);
        else if (K == 256) gln_product<2>(J, row0, lds, acc0, acc1);
        else if (K == 384) gln_product<3>(J, row0, lds, acc0, acc1);
        else gln_product<4>(J, row0, lds, acc0, acc1);
    }
    // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float t = (acc0[i] + acc1[i]) + (has_b ? bia : 0.f);
        if (has_d) t *= J.drop_scale * (float)dmb[i];
        t += (has_r1 ? r1[i] : 0.f) + (has_r2 ? r2[i] : 0.f);
        v[i] = t;
        if (prod) J.pre_out[(size_t)(row0 + 4 * q + i) * J.ldpre + c] = t;
    }

    // ---- LayerNorm over the complete rows: 16 lanes by DPP, 8 waves through LDS
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = row16_sum(v[i]);
    if (li == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[0][wave][4 * q + i] = s[i];
    }
    __syncthreads();
    float mean[4], d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) m += red[0][w8][4 * q + i];
        mean[i] = m * (1.0f / (float)kGlnH);
        d[i] = v[i] - mean[i];
        s[i] = row16_sum(d[i] * d[i]);
    }
    if (li == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[1][wave][4 * q + i] = s[i];
    }
    __syncthreads();
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m2 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) m2 += red[1][w8][4 * q + i];
        const float rstd = 1.0f / sqrtf(m2 * (1.0f / (float)kGlnH) + kGlnEps);
        const int r = row0 + 4 * q + i;
        y[i] = d[i] * rstd * gam + bet;
        J.y[(size_t)r * J.ldy + c] = y[i];
        if (wave == 0 && li == 0) { J.mean[r] = mean[i]; J.rstd[r] = rstd; }
    }
    if (J.pair_out) {
        const int p0 = (row0 + 4 * q) >> 1;
        J.pair_out[(size_t)p0 * kGlnH + c] = (y[0] + y[1]) * 0.5f;
        J.pair_out[(size_t)(p0 + 1) * kGlnH + c] = (y[2] + y[3]) * 0.5f;
    }
}

}  // namespace r3d

R3D_EXPORT int r3d_gemm_ln_mha_supported(int heads, int Lq, int Lk, int dh);

R3D_EXPORT int r3d_gemm_ln_supported(int M, int K, int H) {
    return (H == r3d::kGlnH && M > 0 && (M % r3d::kGlnRows) == 0 && K >= 0 && K <= 512 && (K % 128) == 0) ? 1 : 0;
}

static int gemm_ln_launch(const r3d_gemm_ln_job* jobs, int njobs, int H, const r3d_mha_job* mha, void* stream) {
    R3D_REQUIRE(jobs && njobs >= 1 && njobs <= 4);
    r3d::GemmLnArgs a{};
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        const r3d_gemm_ln_job& j = jobs[i];
        R3D_REQUIRE(r3d_gemm_ln_supported(j.M, j.K, H));
        R3D_REQUIRE(j.pre_out && j.gamma && j.beta && j.y && j.mean && j.rstd);
        R3D_REQUIRE(j.ldpre >= H && j.ldy >= H);
        if (j.K > 0) {
            R3D_REQUIRE(j.A && j.W && j.lda >= j.K && j.ldw >= j.K);
            if (!r3d_aligned16(j.A) || !r3d_aligned16(j.W) || (j.lda % 4) != 0 || (j.ldw % 4) != 0) return R3D_EALIGN;
            R3D_REQUIRE(!j.drop_mask || j.lddrop >= H);
            R3D_REQUIRE(!j.res1 || j.ldr1 >= H);
            R3D_REQUIRE(!j.res2 || j.ldr2 >= H);
        }
        a.j[i] = j;
        a.prefix[i] = total;
        total += j.M / r3d::kGlnRows;
    }
    for (int i = njobs; i <= 4; ++i) a.prefix[i] = total;
    a.njobs = njobs;
    int extra = 0;
    if (mha) {
        R3D_REQUIRE(r3d_gemm_ln_mha_supported(mha->heads, mha->Lq, mha->Lk, mha->dh));
        R3D_REQUIRE(mha->q && mha->k && mha->v && mha->probs && mha->o && mha->B > 0);
        const int Hm = mha->heads * mha->dh;
        R3D_REQUIRE(mha->ldq >= Hm && mha->ldk >= Hm && mha->ldv >= Hm && mha->ldo >= Hm);
        if (((mha->ldk | mha->ldv) & 3) || !r3d_aligned16(mha->k) || !r3d_aligned16(mha->v)) return R3D_EALIGN;
        r3d::MhaArgs& m = a.mha;
        m.q = mha->q; m.ldq = mha->ldq; m.k = mha->k; m.ldk = mha->ldk; m.v = mha->v; m.ldv = mha->ldv;
        m.kpm = mha->key_padding_mask; m.key_label = mha->key_label; m.pad_idx = mha->pad_idx; m.probs = mha->probs;
        m.drop = mha->drop_mask; m.drop_scale = mha->drop_scale; m.o = mha->o; m.ldo = mha->ldo;
        m.B = mha->B; m.heads = mha->heads; m.Lq = mha->Lq; m.Lk = mha->Lk; m.dh = mha->dh;
        m.scale = 1.0f / sqrtf((float)mha->dh);
        a.mha_units = mha->B * mha->heads;
        extra = r3d_cdiv(a.mha_units, 8);
    }
    hipError_t e = hipFuncSetAttribute((const void*)r3d::gemm_ln_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       r3d::kGlnLdsBytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(r3d::gemm_ln_fwd_kernel, dim3(total + extra), dim3(512), (size_t)r3d::kGlnLdsBytes,
                       (hipStream_t)stream, a);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_gemm_ln_mha_supported(int heads, int Lq, int Lk, int dh) {
    return (heads > 0 && Lq == 8 && dh == 16 && Lk > 0 && Lk <= 64) ? 1 : 0;
}

R3D_EXPORT int r3d_gemm_ln_fwd(const r3d_gemm_ln_job* jobs, int njobs, int H, void* stream) {
    return gemm_ln_launch(jobs, njobs, H, nullptr, stream);
}

R3D_EXPORT int r3d_gemm_ln_mha_fwd(const r3d_gemm_ln_job* jobs, int njobs, int H, const r3d_mha_job* mha, void* stream) {
    R3D_REQUIRE(mha);
    return gemm_ln_launch(jobs, njobs, H, mha, stream);
}
