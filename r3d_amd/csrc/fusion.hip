// Token selection and channel exchange of the Rank-enhancing Token Fuser
// (CMFuser.token_fusion, model/futr_safuser_tokenfusion.py:33-66).
//   scores  : wave-per-column-block reduction of |x| over all (clip, frame) rows, fp64 accumulation (:47-50)
//   select  : k = C/4 smallest scores, bit-exact index SET of torch.topk(..., largest=False) on CPU (:52-54)
//   exchange: hard channel swap between the two modalities on clones + stack to [N,2,C] (:56-62), coalesced float4
#include "common.h"
#include "../../include/r3d_hip.h"

namespace r3d {

// ---- per-channel sum of |x| ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colabssum_kernel(const float* x, int ld, int rows, int cols, double* out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (c < cols)
        for (int r = wave; r < rows; r += 4) s += (double)fabsf(x[(size_t)r * ld + c]);
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ---- selection --------------------------------------------------------------------------------------------
// value-only comparator of ATen's CPU top-k (NaN sorts last): aten/src/ATen/native/cpu/TopKImpl.h
__device__ __forceinline__ bool tk_less(float x, float y) { return ((x == x) && (y != y)) || (x < y); }
__device__ __forceinline__ bool tk_equiv(float x, float y) { return !tk_less(x, y) && !tk_less(y, x); }

struct Elem { float v; int i; };

// Serial restatement of libstdc++ std::nth_element (introselect) on an LDS array -- only reached when the k-th
// boundary falls inside a group of equal scores (always in train mode, where every score is equal: SURVEY F5a).
// Same algorithm as oracle/topk_introselect.c; kept in one lane because tie resolution is defined by the data
// movement of the sequential algorithm itself.
__device__ void tk_swap(Elem* a, Elem* b) { Elem t = *a; *a = *b; *b = t; }
__device__ bool tk_lt(const Elem* a, const Elem* b) { return tk_less(a->v, b->v); }

__device__ void tk_adjust_heap(Elem* first, int hole, int len, Elem value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (tk_lt(first + child, first + (child - 1))) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && tk_lt(first + parent, &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ void tk_heap_select(Elem* first, Elem* middle, Elem* last) {
    const int len = (int)(middle - first);
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            Elem value = first[parent];
            tk_adjust_heap(first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    for (Elem* i = middle; i < last; ++i)
        if (tk_lt(i, first)) {
            Elem value = *i;
            *i = *first;
            tk_adjust_heap(first, 0, len, value);
        }
}

__device__ void tk_insertion_sort(Elem* first, Elem* last) {
    if (first == last) return;
    for (Elem* i = first + 1; i != last; ++i) {
        Elem val = *i;
        if (tk_lt(i, first)) {
            for (Elem* j = i; j != first; --j) *j = *(j - 1);
            *first = val;
        } else {
            Elem* cur = i;
            Elem* next = i - 1;
            while (tk_lt(&val, next)) { *cur = *next; cur = next; --next; }
            *cur = val;
        }
    }
}

__device__ void tk_introselect(Elem* first, Elem* nth, Elem* last, int depth_limit) {
    while (last - first > 3) {
        if (depth_limit == 0) {
            tk_heap_select(first, nth + 1, last);
            tk_swap(first, nth);
            return;
        }
        --depth_limit;
        Elem* mid = first + (last - first) / 2;
        Elem *a = first + 1, *b = mid, *c = last - 1;
        if (tk_lt(a, b)) {
            if (tk_lt(b, c)) tk_swap(first, b);
            else if (tk_lt(a, c)) tk_swap(first, c);
            else tk_swap(first, a);
        } else if (tk_lt(a, c)) tk_swap(first, a);
        else if (tk_lt(b, c)) tk_swap(first, c);
        else tk_swap(first, b);
        Elem* lo = first + 1;
        Elem* hi = last;
        for (;;) {
            while (tk_lt(lo, first)) ++lo;
            --hi;
            while (tk_lt(first, hi)) --hi;
            if (!(lo < hi)) break;
            tk_swap(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    tk_insertion_sort(first, last);
}

constexpr int kMaxC = 4096;

// One workgroup per score vector.  score_f (float) or score_sum (double sums / count) -> sel mask + sorted indices.
__global__ __launch_bounds__(256) void token_select_kernel(const float* score_f, const double* score_sum, double count,
                                                           int C, int k, int64_t* idx_out, float* mask_out,
                                                           int stride_in, int stride_idx, int stride_mask,
                                                           int* used_serial) {
    __shared__ Elem q[kMaxC];
    __shared__ int sel[kMaxC];
    __shared__ int boundary;
    __shared__ int wave_cnt[4];
    const int v = blockIdx.x;                 // which vector (0 = rgb, 1 = depth)
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) {
        float s = score_f ? score_f[(size_t)v * stride_in + c] : (float)(score_sum[(size_t)v * stride_in + c] / count);
        q[c].v = s;
        q[c].i = c;
    }
    if (tid == 0) boundary = 0;
    __syncthreads();
    // rank counting: exact whenever the k-th boundary does not cut a group of equal scores
    for (int c = tid; c < C; c += 256) {
        const float s = q[c].v;
        int less = 0, eq = 0;
        for (int j = 0; j < C; ++j) {
            const float t = q[j].v;
            less += tk_less(t, s) ? 1 : 0;
            eq += tk_equiv(t, s) ? 1 : 0;
        }
        int sflag;
        if (less + eq <= k) sflag = 1;
        else if (less >= k) sflag = 0;
        else { sflag = 0; boundary = 1; }
        sel[c] = sflag;
    }
    __syncthreads();
    if (boundary) {
        if (tid == 0) {
            int lg = 0;
            for (int n = C; n > 1; n >>= 1) ++lg;
            tk_introselect(q, q + (k - 1), q + C, 2 * lg);
            if (used_serial) used_serial[v] = 1;
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) sel[c] = 0;
        __syncthreads();
        for (int j = tid; j < k; j += 256) sel[q[j].i] = 1;
        __syncthreads();
    } else if (tid == 0 && used_serial) {
        used_serial[v] = 0;
    }
    // mask + ascending index list (ordered compaction: per-thread contiguous chunk, then exclusive scan of counts)
    const int chunk = (C + 255) / 256;
    const int c0 = tid * chunk, c1 = min(C, c0 + chunk);
    int mine = 0;
    for (int c = c0; c < c1; ++c) mine += sel[c];
    // exclusive scan over 256 threads: wave scan + wave totals
    const int lane = tid & 63, wave = tid >> 6;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wave; ++w) base += wave_cnt[w];
    for (int c = c0; c < c1; ++c) {
        if (mask_out) mask_out[(size_t)v * stride_mask + c] = sel[c] ? 1.0f : 0.0f;
        if (sel[c]) idx_out[(size_t)v * stride_idx + base++] = c;
    }
}

// ---- exchange ---------------------------------------------------------------------------------------------
// x0[n,0,:] = m_rgb ? dep : rgb ;  x0[n,1,:] = m_dep ? rgb : dep ;  then embd_drop (futr_safuser_tokenfusion.py:83)
__global__ __launch_bounds__(256) void exchange_fwd_kernel(const float4* rgb, const float4* dep, const float4* m_rgb,
                                                           const float4* m_dep, float4* x0, const uint8_t* drop,
                                                           float drop_scale, int N, int H4) {
    const size_t total = (size_t)N * H4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int n = (int)(e / H4), c4 = (int)(e % H4);
        const float4 r = rgb[e], d = dep[e], mr = m_rgb[c4], md = m_dep[c4];
        float4 o0, o1;
        o0.x = mr.x != 0.f ? d.x : r.x; o0.y = mr.y != 0.f ? d.y : r.y;
        o0.z = mr.z != 0.f ? d.z : r.z; o0.w = mr.w != 0.f ? d.w : r.w;
        o1.x = md.x != 0.f ? r.x : d.x; o1.y = md.y != 0.f ? r.y : d.y;
        o1.z = md.z != 0.f ? r.z : d.z; o1.w = md.w != 0.f ? r.w : d.w;
        const size_t p0 = ((size_t)2 * n) * H4 + c4, p1 = p0 + H4;
        if (drop) {
            const uint8_t* k0 = drop + p0 * 4;
            const uint8_t* k1 = drop + p1 * 4;
            o0.x *= drop_scale * k0[0]; o0.y *= drop_scale * k0[1]; o0.z *= drop_scale * k0[2]; o0.w *= drop_scale * k0[3];
            o1.x *= drop_scale * k1[0]; o1.y *= drop_scale * k1[1]; o1.z *= drop_scale * k1[2]; o1.w *= drop_scale * k1[3];
        }
        x0[p0] = o0;
        x0[p1] = o1;
    }
}

// d_rgb_pre = ((1-m_rgb) g0 + m_dep g1) * [rgb > 0]   (index_put + clone backward, then relu backward :183)
// d_dep     =  m_rgb g0 + (1-m_dep) g1                 (relu/LN backward of the depth branch happen downstream)
__global__ __launch_bounds__(256) void exchange_bwd_kernel(const float4* dx0, const float4* rgb, const float4* m_rgb,
                                                           const float4* m_dep, const uint8_t* drop, float drop_scale,
                                                           float4* d_rgb_pre, float4* d_dep, int N, int H4) {
    const size_t total = (size_t)N * H4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int n = (int)(e / H4), c4 = (int)(e % H4);
        const size_t p0 = ((size_t)2 * n) * H4 + c4, p1 = p0 + H4;
        float4 g0 = dx0[p0], g1 = dx0[p1];
        if (drop) {
            const uint8_t* k0 = drop + p0 * 4;
            const uint8_t* k1 = drop + p1 * 4;
            g0.x *= drop_scale * k0[0]; g0.y *= drop_scale * k0[1]; g0.z *= drop_scale * k0[2]; g0.w *= drop_scale * k0[3];
            g1.x *= drop_scale * k1[0]; g1.y *= drop_scale * k1[1]; g1.z *= drop_scale * k1[2]; g1.w *= drop_scale * k1[3];
        }
        const float4 mr = m_rgb[c4], md = m_dep[c4], r = rgb[e];
        float4 a, b;
        a.x = ((mr.x != 0.f ? 0.f : g0.x) + (md.x != 0.f ? g1.x : 0.f)) * (r.x > 0.f ? 1.f : 0.f);
        a.y = ((mr.y != 0.f ? 0.f : g0.y) + (md.y != 0.f ? g1.y : 0.f)) * (r.y > 0.f ? 1.f : 0.f);
        a.z = ((mr.z != 0.f ? 0.f : g0.z) + (md.z != 0.f ? g1.z : 0.f)) * (r.z > 0.f ? 1.f : 0.f);
        a.w = ((mr.w != 0.f ? 0.f : g0.w) + (md.w != 0.f ? g1.w : 0.f)) * (r.w > 0.f ? 1.f : 0.f);
        b.x = (mr.x != 0.f ? g0.x : 0.f) + (md.x != 0.f ? 0.f : g1.x);
        b.y = (mr.y != 0.f ? g0.y : 0.f) + (md.y != 0.f ? 0.f : g1.y);
        b.z = (mr.z != 0.f ? g0.z : 0.f) + (md.z != 0.f ? 0.f : g1.z);
        b.w = (mr.w != 0.f ? g0.w : 0.f) + (md.w != 0.f ? 0.f : g1.w);
        d_rgb_pre[e] = a;
        d_dep[e] = b;
    }
}

}  // namespace r3d

using namespace r3d;

/* out[v][c] = sum over rows of |x_v[r, c]| for the two modality embeddings (eval / val / test scores,
 * futr_safuser_tokenfusion.py:49-50 before the division by B*T).  fp64 sums so a multi-GPU caller can all-reduce them
 * exactly before r3d_token_select divides by the global row count. */
R3D_EXPORT int r3d_colabssum(const float* x, int ld, int rows, int cols, double* out, void* stream) {
    R3D_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols);
    hipLaunchKernelGGL(colabssum_kernel, dim3(r3d_cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, out);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

/* nvec score vectors of length C -> for each: mask[C] (1.0 = selected) and the k selected indices, ascending.
 * Exactly one of score_f (float scores) / score_sum (fp64 sums, divided by `count` and rounded to fp32) is given.
 * used_serial[v] (optional) reports whether the tie path (introselect emulation) was taken. */
R3D_EXPORT int r3d_token_select(const float* score_f, const double* score_sum, double count, int nvec, int C, int k,
                                int64_t* idx_out, float* mask_out, int* used_serial, void* stream) {
    R3D_REQUIRE((score_f != nullptr) != (score_sum != nullptr));
    R3D_REQUIRE(idx_out && nvec > 0 && C > 0 && C <= kMaxC && k > 0 && k <= C);
    R3D_REQUIRE((long)k * 64 > C);            // the nth_element branch of ATen's top-k (always true for k = C/4)
    R3D_REQUIRE(score_f || count > 0);
    hipLaunchKernelGGL(token_select_kernel, dim3(nvec), dim3(256), 0, (hipStream_t)stream, score_f, score_sum, count, C,
                       k, idx_out, mask_out, C, k, C, used_serial);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_token_exchange_fwd(const float* rgb, const float* dep, const float* mask_rgb, const float* mask_dep,
                                      float* x0, const uint8_t* drop_mask, float drop_scale, int N, int H,
                                      void* stream) {
    R3D_REQUIRE(rgb && dep && mask_rgb && mask_dep && x0 && N > 0 && H > 0);
    R3D_REQUIRE((H % 4) == 0);
    if (!(r3d_aligned16(rgb) && r3d_aligned16(dep) && r3d_aligned16(mask_rgb) && r3d_aligned16(mask_dep) &&
          r3d_aligned16(x0))) return R3D_EALIGN;
    const size_t total = (size_t)N * (H / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(exchange_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)rgb,
                       (const float4*)dep, (const float4*)mask_rgb, (const float4*)mask_dep, (float4*)x0, drop_mask,
                       drop_scale, N, H / 4);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}

R3D_EXPORT int r3d_token_exchange_bwd(const float* dx0, const float* rgb, const float* mask_rgb, const float* mask_dep,
                                      const uint8_t* drop_mask, float drop_scale, float* d_rgb_pre, float* d_dep,
                                      int N, int H, void* stream) {
    R3D_REQUIRE(dx0 && rgb && mask_rgb && mask_dep && d_rgb_pre && d_dep && N > 0 && H > 0);
    R3D_REQUIRE((H % 4) == 0);
    if (!(r3d_aligned16(dx0) && r3d_aligned16(rgb) && r3d_aligned16(mask_rgb) && r3d_aligned16(mask_dep) &&
          r3d_aligned16(d_rgb_pre) && r3d_aligned16(d_dep))) return R3D_EALIGN;
    const size_t total = (size_t)N * (H / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(exchange_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)dx0,
                       (const float4*)rgb, (const float4*)mask_rgb, (const float4*)mask_dep, drop_mask, drop_scale,
                       (float4*)d_rgb_pre, (float4*)d_dep, N, H / 4);
    R3D_LAUNCH_CHECK();
    return R3D_OK;
}
