/* libr3d_hip.so -- C ABI of the MI355X (gfx950) kernels behind R3D's RGB+Depth token-fusion training step.
 *
 * The reference (olivesgatech/R3D) is pure Python on stock PyTorch ops and has NO FFI / plugin API of its own
 * (SURVEY.md F3, 8(b)); every entry point below therefore replaces an ATen op *sequence* of the reference and cites
 * the reference lines it stands in for (paths relative to the reference root).  The host side that calls them
 * (r3d_amd/engine.py via ctypes) mirrors the reference's Python surface (FUTR / train / opts); INTEGRATION.md shows
 * the binding a maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no torch types; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in practice), fp32 unless stated, row-major with explicit leading dimensions;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); work is only ENQUEUED on it:
 *     no device synchronisation, no device allocation, no host<->device copies, no mutable globals
 *     -> safe under hipGraph capture and from autograd's backward thread (one process per GPU);
 *   - return 0 on success, a negative R3D_E* code for a rejected argument, or the positive hipError_t of a failed launch;
 *   - rows of activations are ordered (clip b, frame s) b-major: row = b*S + s.  The reference's seq-first [S,B,H]
 *     tensors (model/futr_safuser_tokenfusion.py:201-204) are the same data indexed the other way round.
 */
#ifndef R3D_HIP_H
#define R3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3D_ABI_VERSION 2

/* ---- error codes ---------------------------------------------------------------------------------------- */
#define R3D_OK 0
#define R3D_EINVAL (-1)
#define R3D_EALIGN (-2)
#define R3D_ENORCCL (-3)      /* r3d_allreduce_flat: no RCCL is mapped into the process */
#define R3D_ERCCL_BASE (-100) /* r3d_allreduce_flat: ncclResult_t r is returned as R3D_ERCCL_BASE - r */

int r3d_abi_version(void);
/* Writes up to `cap` bytes "gfx950;<build info>" -- used by the host to fail loudly on a stale library. */
int r3d_build_info(char* buf, int cap);

/* ---- gradient exchange (data parallel; replaces the per-step nn.DataParallel gather through device 0, main_darai.py:129-133)
 * In-place sum of buf[0..count) (fp32) over the ranks of `comm` (an ncclComm_t the caller created, e.g. r3d_amd.rccl.RcclComm):
 * ncclAllReduce of the RCCL copy already loaded in the process (PyTorch's), enqueued on `stream` like every other entry
 * point -- one more kernel in stream order, capturable into the step's hipGraph.  1/world is folded into
 * r3d_adamw_flat's grad_scale.  The library itself does not link RCCL. */
int r3d_allreduce_flat(float* buf, int64_t count, void* comm, void* stream);

/* ---- GEMM family ------------------------------------------------------------------------------------------
 * One fp32-exact MFMA GEMM (v_mfma_f32_32x32x2_f32, LDS-staged k-major tiles) carries every dense contraction of
 * the step: nn.Linear forward  (layout NT: model/futr_safuser_tokenfusion.py:179,195,222-231;
 * model/extras/transformerblock.py:21,34,84-88; nn.MultiheadAttention in/out projections and FFN of
 * model/extras/transformer.py:289-327), its input gradient (NN) and its weight gradient (TN) -- the ops autograd
 * derives for the reference at train/train_proposed_depth.py:214.
 *
 *   layout R3D_GEMM_NT: C[M,N] = A[M,K] . B[N,K]^T      (A rows lda, B rows ldb; both K-contiguous)
 *   layout R3D_GEMM_NN: C[M,N] = A[M,K] . B[K,N]
 *   layout R3D_GEMM_TN: C[M,N] = A[K,M]^T . B[K,N]
 *
 * Prologue on A (NT/NN only): row m is read from row (m ^ a_row_xor) and, if a_add != NULL,
 *   A'[m,k] = A[m^xor,k] + a_add[(m % a_add_mod), k]   (broadcast add of positional / query embeddings,
 *   model/extras/transformer.py:289,300-302).
 * Epilogue, in this order:  v = alpha*acc (+ bias[n]);  pre_out[m,n] = v (if given);
 *   act: 0 none | 1 relu | 2 exact-erf gelu;  v *= drop_scale*drop_mask[m,n] (if given);
 *   mul: 0 none | 1 v *= (aux[m,n] > 0) | 2 v *= gelu'(aux[m,n])         (backward of relu / gelu);
 *   v += res1[m,n] + res2[m,n] (if given);  if (accumulate) v += C[m,n];  C[m,n] = v.
 *   c_row_xor = 1 applies the whole epilogue at row m^1 instead of m: the modality swap that the fuser's masked
 *   2-token attention reduces to (softmax([[-inf,s],[s,-inf]]) = [[0,1],[1,0]] exactly; SURVEY.md F5b,
 *   model/extras/transformerblock.py:24-33 under the mask of model/futr_safuser_tokenfusion.py:68-72).
 * splitk > 1: the K range is cut into `splitk` slabs of k_per_split (multiple of 16); raw partial sums go to
 *   `partial` ([splitk][M][N] floats, caller-owned) and r3d_splitk_reduce*() applies the epilogue.
 */
#define R3D_GEMM_NT 0
#define R3D_GEMM_NN 1
#define R3D_GEMM_TN 2

typedef struct r3d_gemm_desc {
    const float* A; const float* B; float* C;
    int32_t layout, M, N, K;
    int32_t lda, ldb, ldc;
    const float* a_add; int32_t a_add_mod, a_add_ld, a_row_xor;
    const float* b_add; int32_t b_add_mod, b_add_ld;   /* NN/TN only: B'[k,:] = B[k,:] + b_add[k % mod,:] */
    const float* bias;
    float* pre_out; int32_t ldpre;
    int32_t act;
    const uint8_t* drop_mask; int32_t lddrop; float drop_scale;
    const float* aux; int32_t ldaux, mul;
    const float* res1; int32_t ldr1;
    const float* res2; int32_t ldr2;
    float alpha; int32_t accumulate;
    float* bias_grad;        /* TN only, splitk == 1: bias_grad[m] = sum_k A[k,m] (the nn.Linear bias gradient, free
                                 with the weight-gradient GEMM that already streams dY through LDS) */
    int32_t c_row_xor;       /* output (and pre_out/aux/res/drop operand) row index = m ^ c_row_xor: pair swap at store */
    int32_t splitk, k_per_split; float* partial;
    int32_t tile;            /* 0 = auto; workgroup tile: 1 = 32x32 (4 k-split waves), 2 = 64x64, 3 = 128x128,
                                4 = 64x64 with 2 k-split wave groups, 5 = 128x128 with 2 k-split wave groups,
                                10 = TN product with 128 x 128 tiles on the bf16 matrix cores (prec == 1; M, N, lda, ldb % 4 == 0,
                                    alpha / accumulate epilogue or adam_*: the depth weight gradient beyond tile 7's limits),
                                12 = tile 10's kernel with ONE LDS stage and <= 128 registers: two workgroups per CU (the planner's pick),
                                11 = tile 9 with uniform waves (kept for measurement),
                                8 / 9 = split-K NT product with 64 x 64 / 128 x 128 tiles on the bf16 matrix cores (prec == 1,
                                    K % 8 == 0, 16-byte aligned K-contiguous operands, no prologue: the forward depth projection),
                                7 = tile 6 on the bf16 matrix cores (prec == 1),
                                6 = persistent 64-column panels with A^T resident in registers (TN, K <= 128, M <= 128,
                                    plain epilogue: the weight gradient of a wide layer from few rows) */
    int32_t vec;             /* filled by the library: operands allow 16-byte loads */
    /* AdamW in the epilogue (adam_m != NULL; weight-gradient GEMMs with splitk == 1, tile 2, 3 or -- TN, prec == 1 -- 10 / 12,
     * N % 4 == 0, 16-byte aligned C / moments, no other epilogue operand): the
     * product alpha * A.B is the GRADIENT and is not stored; C is the PARAMETER, adam_m / adam_v its moments (same
     * layout and ldc), all three updated in place exactly as r3d_adamw_flat would (lr, step: device scalars).  Saves the
     * gradient's write and re-read and one pass over the parameter: depth_projection.weight is 86 % of the model. */
    float* adam_m; float* adam_v; const float* adam_lr; const int64_t* adam_step;
    float adam_beta1, adam_beta2, adam_eps, adam_wd, adam_gscale;
    int32_t prec;            /* 0: v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain).  1: the kernels that have one may take the
                                 bf16x3 path: every fp32 operand split EXACTLY into three bf16 terms, the six leading
                                 products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error per product <= 3 * 2^-24,
                                 the size of an fp32 rounding; 2.7x less matrix-core time).  Today: tile 7 = tile 6's
                                 persistent weight-gradient panels, chosen by r3d_gemm_plan when prec == 1 */
} r3d_gemm_desc;

int r3d_gemm_f32(const r3d_gemm_desc* d, void* stream);
/* Sums the split-K slabs of a previous r3d_gemm_f32(d with splitk>1) and applies d's epilogue. */
int r3d_splitk_reduce(const r3d_gemm_desc* d, void* stream);
/* Workspace floats r3d_gemm_f32 needs in d->partial for this (M,N,splitk). */
int64_t r3d_gemm_partial_floats(int32_t M, int32_t N, int32_t splitk);
/* Two split-K NT products with the same M x N, both on tile 8 (prec == 1; K % 8 == 0, k_per_split % 64 == 0, partial set and
 * distinct), in ONE launch: the second product's K-splits take the workgroups after the first's; the raw slabs of both are left
 * for the caller's reducer (r3d_embed_fuse_fwd: the depth projection and the RGB embedding of
 * model/futr_safuser_tokenfusion.py:179,194-195 side by side). */
int r3d_gemm_bf3_nt_pair(const r3d_gemm_desc* first, const r3d_gemm_desc* second, void* stream);
/* Heuristic the host uses to pick (tile, splitk, k_per_split) for a shape on a 256-CU part; fills the desc. */
int r3d_gemm_plan(r3d_gemm_desc* d);
/* Grouped launch: n independent problems of one layout in ONE kernel (all weight gradients of a step; each is
 * latency-bound, so the ~5 us per-launch floor dominates, not the FLOPs).  prepare() (host only) validates the
 * descriptors, fills tile/vec and prefix[0..n] (first workgroup of each problem; prefix[n] = grid size); the caller
 * copies both arrays to device memory once per shape and calls launch() every step. */
int r3d_gemm_grouped_prepare(r3d_gemm_desc* descs, int n, int tile, int32_t* prefix);
/* host_prefix (optional): the HOST copy of prefix[0..n]; with it, groups of <= 32 problems carry the table in the kernel
 * arguments and a workgroup finds its problem without touching memory (the device-side search is a chain of dependent
 * scalar loads in front of every workgroup's first operand load). */
int r3d_gemm_grouped_launch(const r3d_gemm_desc* dev_descs, const int32_t* dev_prefix, const int32_t* host_prefix, int n,
                            int total_tiles, int layout, int tile, void* stream);

/* ---- row-wise kernels ---------------------------------------------------------------------------------------
 * LayerNorm (eps 1e-5, biased variance, affine): depth_layernorm (model/futr_safuser_tokenfusion.py:147,196),
 * Block.norm1/norm2 (model/extras/transformerblock.py:122,127,131-134), fuser.norm (:25,93), decoder norm1/2/3 and
 * decoder.norm (model/extras/transformer.py:265-267,292-329,182-183).
 *
 * r3d_layernorm_fwd: y = act(LN(x)), saves mean/rstd per row.
 *   nsplit > 0: x is the [nsplit][rows][H] split-K slab buffer of r3d_gemm_f32; the slabs are summed, `bias` added and
 *     the sum (the nn.Linear output) stored to pre_out before normalising -- fuses depth_projection's reduction with
 *     depth_layernorm + ReLU (:195-197).
 *   pair_out != NULL: rows come in (token, modality) pairs and pair_out[n] = (y[2n] + y[2n+1]) / 2, the
 *     torch.mean(x, dim=1) over the two modality tokens (:94).
 * r3d_layernorm_bwd: dx = LN'(dy) [+ add1 + add2]; dgamma/dbeta written (not accumulated).
 *   dy2 (optional, same shape as x) is added to dy first (a second consumer of the LN output);
 *   pair_in: dy holds one row per pair and each row of the pair receives dy/2 (backward of the mean above);
 *   relu: dy is first masked by [LN(x) > 0] (recomputed);  dx2 (optional) = dx * drop_scale * drop_mask;
 *   ws: r3d_layernorm_bwd_ws_floats(rows, H) floats of scratch (0 when rows <= 64).
 */
int r3d_layernorm_fwd(const float* x, int ldx, int nsplit, const float* bias, float* pre_out, const float* gamma,
                      const float* beta, float* y, int ldy, float* mean, float* rstd, float* pair_out, int rows, int H,
                      int relu, void* stream);
int64_t r3d_layernorm_bwd_ws_floats(int rows, int H);
int r3d_layernorm_bwd(const float* dy, int lddy, int pair_in, const float* dy2, int lddy2, const float* x, int ldx,
                      const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, const float* add1, int ldadd1,
                      const float* add2, int ldadd2, float* dx, int lddx, float* dx2, int lddx2,
                      const uint8_t* drop_mask, int lddrop, float drop_scale, float* dgamma, float* dbeta, float* ws,
                      int rows, int H, int defer_finalize, void* stream);
/* With defer_finalize != 0 the reduction of the per-block partial dgamma/dbeta left in ws is done later by this call
 * (so the host can take it off the critical path). */
int r3d_layernorm_bwd_finalize(const float* ws, int rows, int H, float* dgamma, float* dbeta, void* stream);
/* Up to 4 independent LayerNorm sites of the same width in ONE launch (what a step saves is the dependent launch, ~5 us).
 * The job structs carry exactly the arguments of r3d_layernorm_fwd / r3d_layernorm_bwd (backward: ws must be given when
 * the site needs it -- parameter reductions are always deferred to r3d_layernorm_bwd_finalize*; rows_per_block and
 * nblocks are filled by the library).  Host arrays; copied into the kernel argument. */
typedef struct r3d_ln_fwd_job {
    const float* x; int32_t ldx, nsplit; const float* bias; float* pre_out;
    const float* gamma; const float* beta; float* y; int32_t ldy; float* mean; float* rstd;
    float* pair_out; int32_t rows, H, relu;
} r3d_ln_fwd_job;
typedef struct r3d_ln_bwd_job {
    const float* dy; int32_t lddy, pair_in; const float* dy2; int32_t lddy2;
    const float* x; int32_t ldx; const float* mean; const float* rstd; const float* gamma; const float* beta; int32_t relu;
    const float* add1; int32_t ldadd1; const float* add2; int32_t ldadd2;
    float* dx; int32_t lddx;
    float* dx2; int32_t lddx2; const uint8_t* drop_mask; int32_t lddrop; float drop_scale;
    float* dgamma; float* dbeta; float* ws;
    int32_t rows, H, rows_per_block, nblocks;
} r3d_ln_bwd_job;
int r3d_layernorm_fwd_multi(const r3d_ln_fwd_job* jobs, int njobs, void* stream);
int r3d_layernorm_bwd_multi(r3d_ln_bwd_job* jobs, int njobs, void* stream);
/* r3d_layernorm_bwd_multi (hidden <= 128) carrying an INDEPENDENT small attention backward (the arguments of
 * r3d_mha_core_bwd; 8 queries, dh 16, <= 64 keys) as extra workgroups: the backward of decoder layer 0's query
 * self-attention feeds parameter gradients only and rides beside the fuser's norm2 backward. */
typedef struct r3d_mha_bwd_job {
    const float* q; int32_t ldq; const float* k; int32_t ldk; const float* v; int32_t ldv;
    const float* probs; const uint8_t* drop_mask; float drop_scale;
    const float* d_o; int32_t lddo; float* dq; int32_t lddq; float* dk; int32_t lddk; float* dv; int32_t lddv;
    int32_t B, heads, Lq, Lk, dh;
} r3d_mha_bwd_job;
int r3d_layernorm_bwd_multi_mha(r3d_ln_bwd_job* jobs, int njobs, const r3d_mha_bwd_job* mha, void* stream);
/* The same for many LayerNorm sites in one launch; jobs live in device memory. */
/* rows > 0: ws holds the partials r3d_layernorm_bwd left for that many rows; rows < 0: exactly -rows (dgamma, dbeta)
 * pairs [-rows][2][H] written by another producer (r3d_embed_fuse_bwd). */
typedef struct r3d_ln_finalize_job { const float* ws; float* dgamma; float* dbeta; int32_t rows, H; } r3d_ln_finalize_job;
int r3d_layernorm_bwd_finalize_batched(const r3d_ln_finalize_job* dev_jobs, int njobs, int max_H, void* stream);
/* out[c] (+)= sum_r x[r,c]: bias gradients.  ws: r3d_colsum_ws_floats(rows, cols) floats. */
int64_t r3d_colsum_ws_floats(int rows, int cols);
int r3d_colsum(const float* x, int ld, int rows, int cols, float* out, float* ws, int accumulate, void* stream);
/* out[r,c] (+)= sum over rows with row % mod == r: gradients of query_embed / pos_embedding, which the forward
 * broadcasts over clips (model/futr_safuser_tokenfusion.py:190,205-209). */
int r3d_rowmod_sum(const float* x, int ld, int rows, int cols, int mod, float* out, int ldo, int accumulate,
                   void* stream);

/* Batched form: job j writes dst[r,:] = sum over rows with row % mod == r of (src1 [+ src2]); mod == 1 is a bias gradient. */
typedef struct r3d_rowsum_job { const float* src1; const float* src2; float* dst; int32_t ld1, ld2, ldd, rows, cols, mod; } r3d_rowsum_job;
int r3d_rowmod_sum_batched(const r3d_rowsum_job* dev_jobs, int njobs, int max_cols, int max_mod, void* stream);
/* ++*a, ++*b (either may be NULL): step counter and dropout offset in one launch. */
int r3d_tick(int64_t* a, int64_t* b, void* stream);
/* out[r,:] = x[r,:] + add[r % mod,:]: with_pos_embed (model/extras/transformer.py:278-279,289,300-302); x may be NULL. */
int r3d_add_rowbcast(const float* x, int ldx, const float* add, int ldadd, int mod, float* out, int ldo, int rows, int cols,
                     void* stream);

/* ---- nn.Linear -> dropout -> residuals -> LayerNorm in ONE launch (gemm_ln.hip; hidden size 128) ----------------
 * y = LN(pre), pre = drop_scale * drop_mask * (A . W^T + bias) + res1 + res2      (A [M,K], W [H,K], both K-contiguous)
 * -- the sub-layer epilogues of model/extras/transformerblock.py:131-134 (attn.proj + x -> norm2; mlp fc2 + x),
 * model/futr_safuser_tokenfusion.py:92-94 (+ x_res -> norm -> mean over the two modality tokens: pair_out[n] =
 * (y[2n] + y[2n+1]) / 2) and model/extras/transformer.py:292-293,304-306 (out_proj -> dropout -> + tgt -> norm1 / norm2).
 * A workgroup owns 16 complete rows, so the LayerNorm is the GEMM's epilogue: one dependent launch and one memory round
 * trip of the pre-norm rows less per site than r3d_gemm_f32 + r3d_layernorm_fwd.  pre_out (the LayerNorm's input, read
 * by the backward), mean and rstd are written exactly as those two entry points write them.  K == 0: no product, the
 * rows already in pre_out are normalised (a plain LayerNorm job riding in the same launch).  Up to 4 jobs per launch.
 * Requirements (r3d_gemm_ln_supported): H == 128, M % 16 == 0, K in {0, 128, 256, 384, 512}, A / W 16-byte aligned with lda, ldw % 4 == 0.
 */
typedef struct r3d_gemm_ln_job {
    const float* A; int32_t lda;
    const float* W; int32_t ldw;
    const float* bias;
    const uint8_t* drop_mask; int32_t lddrop; float drop_scale;
    const float* res1; int32_t ldr1;
    const float* res2; int32_t ldr2;
    float* pre_out; int32_t ldpre;
    const float* gamma; const float* beta;
    float* y; int32_t ldy; float* mean; float* rstd;
    float* pair_out;
    int32_t M, K;
} r3d_gemm_ln_job;
int r3d_gemm_ln_supported(int M, int K, int H);
int r3d_gemm_ln_fwd(const r3d_gemm_ln_job* jobs, int njobs, int H, void* stream);
/* The same launch carrying an INDEPENDENT small attention core (the arguments of r3d_mha_core_fwd; heads * 8 queries, dh 16,
 * <= 64 keys: r3d_gemm_ln_mha_supported) as extra workgroups: the query self-attention of decoder layer 0
 * (model/extras/transformer.py:289-291 with tgt = 0, model/futr_safuser_tokenfusion.py:205-209) depends on parameters only,
 * so it needs no launch of its own and rides beside the fuser's attn.proj + norm2. */
typedef struct r3d_mha_job {
    const float* q; int32_t ldq; const float* k; int32_t ldk; const float* v; int32_t ldv;
    const uint8_t* key_padding_mask; const int64_t* key_label; int32_t pad_idx;
    float* probs; const uint8_t* drop_mask; float drop_scale; float* o; int32_t ldo;
    int32_t B, heads, Lq, Lk, dh;
} r3d_mha_job;
int r3d_gemm_ln_mha_supported(int heads, int Lq, int Lk, int dh);
int r3d_gemm_ln_mha_fwd(const r3d_gemm_ln_job* jobs, int njobs, int H, const r3d_mha_job* mha, void* stream);

/* ---- the SA-Fuser block and its row-local neighbourhood as ONE launch (csrc/fuser_chain.hip; hidden = 128) -------------
 * Fuser role, 16 token rows (8 frames) per workgroup:  V = h1 Wv^T stored pair-swapped (the closed form of the masked
 * 2-token attention, model/futr_safuser_tokenfusion.py:68-72,77) -> x1 = x0 + vsw Wproj^T + b -> h2 = norm2(x1) ->
 * u = h2 W1^T + b1, f1 = GELU(u) -> x3 = x1 (+ x0 when add_xres) + f1 W2^T + b2 -> y = fuser.norm(x3) ->
 * fused = mean of the token pair (model/extras/transformerblock.py:118-135, model/futr_safuser_tokenfusion.py:83-94) ->
 * seg = fused Wseg^T + b (:228-232), cakv = (fused + pos[frame %% S]) Wkv^T + b (model/extras/transformer.py:300-302).
 * Query role, 2 clips per workgroup (decoder layer 0 with tgt = 0, transformer.py:289-293,300):
 * sa_qkv = qpos Win^T + b -> attention core (probabilities p_sa, dropout drop_sa) -> t1_pre = dropout(sa_o Wout^T + b) ->
 * t1 = norm1(t1_pre) -> caq = (t1 + qpos) Wq^T + b.
 * All matrices row-major and dense: activations [2N, 128] (u, f1: [2N, 512]; fused [N, 128]; seg [N, K]; cakv [N, 256];
 * sa_qkv [B*Q, 384]), weights as nn.Linear stores them ([out, in]).  y may be NULL.  Requirements:
 * r3d_fuser_chain_supported. */
typedef struct r3d_fuser_chain_fwd_args {
    const float* x0; const float* h1;
    const float* wv; const float* wproj; const float* bproj; const float* g2; const float* be2;
    const float* w1; const float* b1; const float* w2; const float* b2; const float* gf; const float* bef;
    const float* pos; const float* wkv; const float* bkv; const float* wseg; const float* bseg;
    float* vsw; float* x1; float* h2; float* m2; float* r2; float* u; float* f1; float* x3; float* y; float* mf; float* rf;
    float* fused; float* seg; float* cakv;
    const float* qpos; const float* w_in; const float* b_in; const float* w_out; const float* b_out; const float* g1;
    const float* be1; const float* wq; const float* bq;
    const uint8_t* drop_sa; const uint8_t* drop_d1; float drop_scale;
    float* sa_qkv; float* p_sa; float* sa_o; float* t1_pre; float* t1; float* m1; float* r1; float* caq;
    int32_t N, S, K, H, add_xres, B, Q, heads;
    uint64_t* timeline;      /* profiling aid, normally NULL: wave 0 of the first workgroup of each role stores wall_clock64()
                                (100 MHz) at its stage boundaries -- fuser role [0..15], query role [16..31] */
    /* optional (all or none): bf16x3 operand-order planes (r3d_weight_planes) of wv, wproj, w1, w2, wkv, wseg as
     * B[n = output column][k = input column] -- the fuser role then runs on the bf16 matrix cores (three-way exact operand
     * split, six products, fp32 accumulate: the precision of r3d_gemm_desc::prec = 1) */
    const uint16_t* pl_wv; const uint16_t* pl_wproj; const uint16_t* pl_w1; const uint16_t* pl_w2; const uint16_t* pl_wkv;
    const uint16_t* pl_wseg;
} r3d_fuser_chain_fwd_args;
/* bf16x3 operand-order planes: for a matrix B[n][k] (N x K), bf16 element (tile t = n / 16, k-step s = k / 32, plane p of
 * {high, middle, low}, lane = n %% 16 + 16 ((k %% 32) / 8), e = k %% 8) at index ((((t * ceil(K/32) + s) * 3 + p) * 64 + lane) * 8
 * + e); zero where n >= N or k >= K.  transposed != 0: B[n][k] = src[k * ld + n] (the planes of the transpose: input-gradient
 * products), else src[n * ld + k]. */
typedef struct r3d_plane_job {
    const float* src; uint16_t* dst; int32_t ld, N, K, transposed, first_block, pad_;
} r3d_plane_job;
int64_t r3d_weight_plane_elems(int N, int K);
int r3d_weight_planes(const r3d_plane_job* jobs_device, int njobs, int total_blocks, void* stream);
int r3d_fuser_chain_supported(int N, int H, int K, int B, int Q, int heads);
int r3d_fuser_chain_fwd(const r3d_fuser_chain_fwd_args* a, void* stream);
/* The adjoint of r3d_fuser_chain_fwd, one launch.  Fuser role: d(memory + pos) = d_cakv Wkv (stored in d_fused: the
 * positional embedding's gradient) + d_seg Wseg (+ d_extra [N,128], optional: the effective-rank penalty's gradient) ->
 * fuser.norm backward (half of a frame's gradient to each token row) -> d_u = (d_x3 W2) GELU'(u) -> d_h2 = d_u W1 ->
 * norm2 backward + d_x3 -> d_v = unswap(d_x1 Wproj) -> d_h1 = d_v Wv -> norm1 backward + d_x1 (+ d_x3 when add_xres) ->
 * embd_drop -> token-exchange backward (masks m_rgb / m_dep [128], ReLU gate of the RGB embedding) -> depth LayerNorm +
 * ReLU backward (autograd of model/futr_safuser_tokenfusion.py:83-94,56-62,183,195-197).  Stored for the weight-gradient
 * launch: d_x3, d_u, d_x1, d_v (d_h2, d_h1 optional).  LayerNorm parameter-gradient partials: part_nf / part_n2
 * [2N/4][2][128] (one (dgamma, dbeta) pair per 4 rows), part_n1 / part_dep [N][2][128] (one per frame).
 * Query role: caqin = d_caq Wq -> decoder norm1 backward (dy = caqin + d_t1_res; partials part_d1 [B*Q/4][2][128]) ->
 * sap = dropout' -> sao = sap Wout -> attention core backward (saqkv) -> sain = saqkv Win (transformer.py:289-293,300). */
typedef struct r3d_fuser_chain_bwd_args {
    const float* d_cakv; const float* d_seg; const float* d_extra; const float* wkv; const float* wseg;
    const float* x3; const float* mf; const float* rf; const float* gf;
    const float* w2; const float* u; const float* w1; const float* x1; const float* m2; const float* r2; const float* g2;
    const float* wproj; const float* wv; const float* x0; const float* m1; const float* r1; const float* g1n;
    const uint8_t* drop_x0; const float* m_rgb; const float* m_dep; const float* rgb; const float* dep_pre;
    const float* mean_d; const float* rstd_d; const float* lnd_g; const float* lnd_b;
    float* d_fused; float* d_x3; float* d_u; float* d_h2; float* d_x1; float* d_v; float* d_h1; float* d_rgb_pre;
    float* d_dep_pre; float* part_nf; float* part_n2; float* part_n1; float* part_dep;
    const float* d_caq; const float* d_t1_res; const float* wq; const float* t1_pre; const float* m1d; const float* r1d;
    const float* g1d; const uint8_t* drop_d1; const float* w_out; const float* sa_qkv; const float* p_sa;
    const uint8_t* drop_sa; const float* w_in;
    float* caqin; float* t1pre_out; float* sap; float* sao; float* saqkv; float* sain; float* part_d1;
    float drop_scale;
    int32_t N, S, K, H, add_xres, B, Q, heads;
    uint64_t* timeline;      /* as in r3d_fuser_chain_fwd_args */
    /* optional (all or none; K <= 32): bf16x3 operand-order planes of the TRANSPOSED weights (r3d_plane_job::transposed = 1) of
     * wkv, wseg, w2, w1, wproj, wv -- the fuser role's input-gradient products then run on the bf16 matrix cores */
    const uint16_t* pl_wkv_t; const uint16_t* pl_wseg_t; const uint16_t* pl_w2_t; const uint16_t* pl_w1_t;
    const uint16_t* pl_wproj_t; const uint16_t* pl_wv_t;
} r3d_fuser_chain_bwd_args;
int r3d_fuser_chain_bwd(const r3d_fuser_chain_bwd_args* a, void* stream);

/* ---- build-defined THREE-modality fuser pieces (csrc/fuser3.hip; BASELINE configs[4]; the reference's CMFuser is
 * two-token, model/futr_safuser_tokenfusion.py:74-81 -- SURVEY.md 8(d): checked against the build's own CPU restatement) ----
 * Rows frame-major: the tokens of frame n are rows 3n, 3n + 1, 3n + 2.
 * exchange3: x0[3n + m] = (mask[m] ? x_{(m+1) %% 3}[n] : x_m[n]) * keep  (mask [3][C] 1.0 / 0.0, drop_mask optional [3N, C]);
 * attn3: attention over a frame's three tokens with the -inf diagonal (every token attends to the two others), qkv [3N, 3C] =
 *        [q | k | v] per row, C / heads <= 128, probs [N][heads][3][2]; triple_mean: torch.mean over the three tokens. */
int r3d_token_exchange3_fwd(const float* xa, const float* xb, const float* xc, const float* mask, const uint8_t* drop_mask,
                            float drop_scale, float* x0, int N, int C, void* stream);
int r3d_token_exchange3_bwd(const float* dx0, const float* mask, const uint8_t* drop_mask, float drop_scale, float* da, float* db,
                            float* dc, int N, int C, void* stream);
int r3d_attn3_fwd(const float* qkv, float* probs, float* out, int N, int C, int heads, void* stream);
int r3d_attn3_bwd(const float* qkv, const float* d_out, float* d_qkv, int N, int C, int heads, void* stream);
int r3d_triple_mean_fwd(const float* y, float* out, int N, int C, void* stream);
int r3d_triple_mean_bwd(const float* d_out, float* dy, int N, int C, void* stream);

/* ---- token selection / exchange: CMFuser.token_fusion (model/futr_safuser_tokenfusion.py:33-66) ------------- */
/* out[c] = sum_r |x[r,c]| in fp64 (the eval-mode score before the division by B*T, :49-50). */
int r3d_colabssum(const float* x, int ld, int rows, int cols, double* out, void* stream);
/* k smallest of nvec score vectors of length C (torch.topk(..., largest=False) on CPU, :52-54), bit-exact as a SET
 * including ties (libstdc++ introselect emulation when the k-th boundary cuts a group of equal scores).
 * idx_out [nvec][k] ascending; mask_out [nvec][C] 1.0/0.0 (optional); used_serial [nvec] (optional diagnostics). */
int r3d_token_select(const float* score_f, const double* score_sum, double count, int nvec, int C, int k,
                     int64_t* idx_out, float* mask_out, int* used_serial, void* stream);
/* x0[2n] = mask_rgb ? dep[n] : rgb[n];  x0[2n+1] = mask_dep ? rgb[n] : dep[n]; then embd_drop (:56-62, :83). */
int r3d_token_exchange_fwd(const float* rgb, const float* dep, const float* mask_rgb, const float* mask_dep, float* x0,
                           const uint8_t* drop_mask, float drop_scale, int N, int H, void* stream);
/* d_rgb_pre = ((1-mask_rgb) g[2n] + mask_dep g[2n+1]) * [rgb > 0];  d_dep = mask_rgb g[2n] + (1-mask_dep) g[2n+1]. */
int r3d_token_exchange_bwd(const float* dx0, const float* rgb, const float* mask_rgb, const float* mask_dep,
                           const uint8_t* drop_mask, float drop_scale, float* d_rgb_pre, float* d_dep, int N, int H,
                           void* stream);

/* ---- decoder attention core: nn.MultiheadAttention minus its projections (model/extras/transformer.py:289-304) --
 * q rows b*Lq+i, k/v rows b*Lk+j, head h in columns [h*dh, (h+1)*dh).  probs [B][heads][Lq][Lk] = softmax before
 * dropout (saved for backward).  Padded keys: key_padding_mask [B][Lk] (1 = padded) and/or key_label [B][Lk] int64
 * with key padded iff label == pad_idx (get_pad_mask, model/futr_safuser_tokenfusion.py:168,243-244); both may be NULL.
 * drop_mask like probs or NULL. */
int r3d_mha_core_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                     const uint8_t* key_padding_mask, const int64_t* key_label, int pad_idx, float* probs,
                     const uint8_t* drop_mask, float drop_scale, float* o, int ldo, int B, int heads, int Lq, int Lk,
                     int dh, void* stream);
int r3d_mha_core_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* probs,
                     const uint8_t* drop_mask, float drop_scale, const float* d_o, int lddo, float* dq, int lddq, float* dk,
                     int lddk, float* dv, int lddv, int B, int heads, int Lq, int Lk, int dh, void* stream);

/* ---- fused decoder layer, one workgroup per clip (TransformerDecoderLayer.forward_post, model/extras/transformer.py:
 * 281-330; final decoder.norm :182-183; fc|fc_len head futr_safuser_tokenfusion.py:219-226) --------------------------
 * Replaces ~17 dependent launches per layer by one.  Supported when one clip's layer fits a CU's LDS
 * (r3d_decoder_fused_supported); otherwise the host composes the layer from the GEMM / attention / LayerNorm entry points.
 * ptrs[R3D_DEC_FWD_NPTRS], in this order (N = optional, may be NULL):
 *   0-17  parameters: self_attn.in_proj_weight, .in_proj_bias, .out_proj.weight, .out_proj.bias, norm1.weight, norm1.bias,
 *         multihead_attn.in_proj_weight, .in_proj_bias, .out_proj.weight, .out_proj.bias, norm2.weight, norm2.bias,
 *         linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm3.weight, norm3.bias
 *   18 memory [B*S,H]   19 pos [S,H]   20 query_pos [Q,H]   21 tgt_in [B*Q,H] (N: layer 0, tgt = 0)
 *   22 key_label int64 [B*S] (N)       23-28 dropout keep masks uint8 (N): sa_p, ca_p, d1, d2, d3, ff
 *   29-48 saved activations (outputs): sa_qkv [BQ,3H], p_sa [B,h,Q,Q], sa_o, t1_pre, t1 [BQ,H], m1, r1 [BQ], caq [BQ,H],
 *         cakv [B*S,2H], p_ca [B,h,Q,S], ca_o, t2_pre, t2 [BQ,H], m2, r2 [BQ], ff1 [BQ,4H], t3_pre, t3 [BQ,H], m3, r3 [BQ]
 *   49-56 tail of the last layer (all N together): decoder.norm.weight, .bias, tgtF [BQ,H], mF, rF [BQ],
 *         head weight [n_head_out,H], head bias [n_head_out], actdur [BQ,n_head_out] */
#define R3D_DEC_FWD_NPTRS 57
int r3d_decoder_fused_supported(int S, int Q, int H, int heads);
int r3d_decoder_layer_fwd(const void* const* ptrs, int nptrs, int B, int S, int Q, int H, int heads, int pad_idx,
                          float drop_scale, int n_head_out, void* stream);

/* ---- the seam between the input projections and the fuser block, one launch per direction (train mode) -----------
 * forward : split-K slab sums of input_embed (+bias, ReLU :183) and depth_projection (+bias, LayerNorm, ReLU :195-197),
 *           token exchange + embd_drop (:56-62,83), fuser.blocks.0.norm1 (transformerblock.py:122).
 *           rgb_src: ns_r > 0 -> slabs [ns_r][N][H]; ns_r == 0 -> the finished embedding (rgb_out may alias it).
 *           dep_src: slabs [ns_d][N][H], ns_d >= 1 (a finished pre-LayerNorm matrix is one slab).
 * backward: norm1 backward (+ residual gradients add1, add2), exchange backward, ReLU of the RGB embedding, depth
 *           LayerNorm + ReLU backward; ws_n1 / ws_dep [N][2][H] receive one (dgamma, dbeta) partial per frame
 *           (r3d_layernorm_bwd_finalize_batched job with rows = -N).  All matrices contiguous, ld = H <= 1024. */
int r3d_embed_fuse_fwd(const float* rgb_src, int ns_r, const float* bias_r, const float* dep_src, int ns_d,
                       const float* bias_d, const float* lnd_gamma, const float* lnd_beta, const float* mask_rgb,
                       const float* mask_dep, const uint8_t* drop_mask, float drop_scale, const float* ln1_gamma,
                       const float* ln1_beta, float* rgb_out, float* dep_pre_out, float* mean_d, float* rstd_d,
                       float* dep_out, float* x0, float* h1, float* m1, float* r1, int N, int H, void* stream);
/* r3d_embed_fuse_fwd and r3d_weight_planes (below: jobs_device / njobs / total_blocks as there) in ONE launch: the re-split of
 * the chain weights rides as extra workgroups of the seam. */
int r3d_embed_fuse_fwd_planes(const float* rgb_src, int ns_r, const float* bias_r, const float* dep_src, int ns_d,
                              const float* bias_d, const float* lnd_gamma, const float* lnd_beta, const float* mask_rgb,
                              const float* mask_dep, const uint8_t* drop_mask, float drop_scale, const float* ln1_gamma,
                              const float* ln1_beta, float* rgb_out, float* dep_pre_out, float* mean_d, float* rstd_d,
                              float* dep_out, float* x0, float* h1, float* m1, float* r1, int N, int H,
                              const r3d_plane_job* jobs_device, int njobs, int total_blocks, void* stream);
int r3d_embed_fuse_bwd(const float* d_h1, const float* x0, const float* m1, const float* r1, const float* ln1_gamma,
                       const float* add1, const float* add2, const uint8_t* drop_mask, float drop_scale,
                       const float* mask_rgb, const float* mask_dep, const float* rgb, const float* dep_pre,
                       const float* mean_d, const float* rstd_d, const float* lnd_gamma, const float* lnd_beta,
                       float* d_rgb_pre, float* d_dep_pre, float* ws_n1, float* ws_dep, int N, int H, void* stream);

/* ---- the BN-blend token fuser of model/futr_safuser_batchnormalization.py:38-76 -------------------------
 * r3d_bn_stats     : BatchNorm1d statistics of both embeddings [N, C] (batch statistics + running-stat update when
 *                    training, running statistics otherwise) -> mean / rstd [2][C], absgamma [2][C] (selection score).
 * r3d_bn_blend_fwd : x0 [2N, C] = embd_drop(blend(BN(rgb), BN(dep))) with alpha on the selected channels, h1 = norm1(x0).
 * r3d_bn_blend_bwd : norm1 backward (+ add1), dropout, blend backward -> five [N, C] term matrices whose column sums
 *                    (r3d_rowmod_sum_batched, mod = 1) are d beta_rgb, d gamma_rgb, d beta_dep, d gamma_dep, d alpha;
 *                    norm1 parameter partials ws_n1 [N][2][C] (finalize job with rows = -N).
 * r3d_bn_bwd_apply : BatchNorm input gradients from those sums; d_rgb_pre includes input_embed's ReLU. */
int r3d_bn_stats(const float* x_rgb, const float* x_dep, float* run_mean_rgb, float* run_var_rgb, int64_t* nbt_rgb,
                 float* run_mean_dep, float* run_var_dep, int64_t* nbt_dep, const float* gamma_rgb, const float* gamma_dep,
                 float* mean, float* rstd, float* absgamma, int N, int C, int training, float momentum, void* stream);
int r3d_bn_blend_fwd(const float* rgb, const float* dep, const float* mean, const float* rstd, const float* gamma_rgb,
                     const float* beta_rgb, const float* gamma_dep, const float* beta_dep, const float* alpha,
                     const float* mask_rgb, const float* mask_dep, const uint8_t* drop_mask, float drop_scale,
                     const float* ln1_gamma, const float* ln1_beta, float* x0, float* h1, float* m1, float* r1, int N, int C,
                     void* stream);
int r3d_bn_blend_bwd(const float* d_h1, const float* x0, const float* m1, const float* r1, const float* ln1_gamma,
                     const float* add1, const uint8_t* drop_mask, float drop_scale, const float* rgb, const float* dep,
                     const float* mean, const float* rstd, const float* gamma_rgb, const float* beta_rgb,
                     const float* gamma_dep, const float* beta_dep, const float* alpha, const float* mask_rgb,
                     const float* mask_dep, float* t_drb, float* t_drbx, float* t_ddb, float* t_ddbx, float* t_dal,
                     float* ws_n1, int N, int C, void* stream);
int r3d_bn_bwd_apply(const float* rgb, const float* dep, const float* mean, const float* rstd, const float* gamma_rgb,
                     const float* gamma_dep, const float* t_drb, const float* t_ddb, const float* dgamma_rgb,
                     const float* dbeta_rgb, const float* dgamma_dep, const float* dbeta_dep, float* d_rgb_pre, float* d_dep,
                     int N, int C, int training, void* stream);
/* Data-parallel BatchNorm for that fuser (the statistics of the GLOBAL batch, as one process running the reference sees
 * them; replaces what torch.nn.SyncBatchNorm would do for futr_safuser_batchnormalization.py:45-46):
 * r3d_bn_sync_pack     : the local moments left by r3d_bn_stats -> out [4C + 1] = (mean [2][C], M2 = N var [2][C], N);
 * r3d_bn_sync_finalize : all [world][4C + 1] (every rank's pack, rank order) -> global mean / rstd [2][C] by Chan's
 *                        parallel-variance combination, running-statistics update from the global moments, nfrac[0] =
 *                        n_local / n_global. */
int r3d_bn_sync_pack(const float* mean, const float* rstd, int N, int C, float* out, void* stream);
int r3d_bn_sync_finalize(const float* all, int world, int n_local, int C, float* mean, float* rstd, float* run_mean_rgb,
                         float* run_var_rgb, int64_t* nbt_rgb, float* run_mean_dep, float* run_var_dep, int64_t* nbt_dep,
                         float* nfrac, float momentum, void* stream);

/* ---- the decoder's tail, one launch per direction (row-local on the B*Q query rows) ------------------------------
 * forward : last layer's norm3 (transformer.py:329) -> decoder.norm (:182-183) -> heads fc | fc_len as one [n_head, H]
 *           product (futr_safuser_tokenfusion.py:219-226).  x = pre-norm3 rows [rows, H] contiguous.
 * backward: heads' input gradient -> decoder.norm backward -> norm3 backward; dx = gradient w.r.t. x, dx2 = dx * dropout3
 *           mask (:328).  LayerNorm parameter gradients: partials wsF / ws3 in r3d_layernorm_bwd's layout for `rows` rows,
 *           or final values in dgF/dbF/dg3/db3 when one block covers all rows (rows <= 4). */
int r3d_decoder_tail_fwd(const float* x, const float* g3, const float* b3, const float* gF, const float* bF,
                         const float* w_head, const float* b_head, int n_head, float* t3, float* m3, float* r3, float* tgtF,
                         float* mF, float* rF, float* out, int ld_out, int rows, int H, void* stream);
int r3d_decoder_tail_bwd(const float* d_out, int ld_dout, const float* w_head, int n_head, const float* t3, const float* mF,
                         const float* rF, const float* gF, const float* x, const float* m3, const float* r3, const float* g3,
                         const uint8_t* drop_mask, float drop_scale, float* dx, float* dx2, float* dgF, float* dbF,
                         float* dg3, float* db3, float* wsF, float* ws3, int rows, int H, void* stream);

/* ---- losses: utils.py:325-328,358-378,410-490 as composed at train/train_proposed_depth.py:171-213 ------------- */
int r3d_losses_fwd_bwd(const float* seg_logits, int ld_seg, const float* act_logits, int ld_act, const float* dur,
                       int ld_dur, const int64_t* past_label, const int64_t* target, const float* target_dur, int B, int S,
                       int Q, int K, int pad_idx, int exclude_idx, int val_mode, const float* dur_den, float grad_scale,
                       float* d_seg, int ld_dseg, float* d_act, int ld_dact, float* d_dur, int ld_ddur, float* loss_out,
                       int64_t* counts, float* ws, int64_t* tick_a, int64_t* tick_b, void* stream);
/* ws: r3d_losses_ws_floats floats, 16-byte aligned and ZERO before the first call (it ends with the arrival counter of
 * the single-launch reduction, which every call leaves at zero).  tick_a / tick_b: optional device int64 counters
 * incremented once per call (the engine's step counter and dropout offset). */
int64_t r3d_losses_ws_floats(int B, int S, int Q);

/* ---- training step only: the decoder's tail, the three losses and the tail's backward as ONE launch -------------------
 * = r3d_decoder_tail_fwd (transformer.py:329,182-183; futr_safuser_tokenfusion.py:219-226) -> r3d_losses_fwd_bwd
 * (train_proposed_depth.py:171-213) -> r3d_decoder_tail_bwd, same arithmetic per row.  The anticipation logits live in
 * out [B*Q, ld_out]: K action columns + 1 duration column; d_out likewise.  wsF / ws3: LayerNorm parameter-gradient
 * partials in r3d_layernorm_bwd's layout for B*Q rows.  ws: as for r3d_losses_fwd_bwd.
 * r3d_decoder_tail_losses_supported: hidden <= 128, K + 1 <= 24, Q == 8, B*Q <= 1024; callers fall back to the three
 * separate entry points otherwise. */
typedef struct r3d_tail_losses_args {
    const float* x; const float* g3; const float* b3; const float* gF; const float* bF;
    const float* w_head; const float* b_head; int n_head;
    float* t3; float* m3; float* r3; float* tgtF; float* mF; float* rF; float* out; int ld_out; int H;
    const float* seg; int ld_seg; const int64_t* past_label; const int64_t* target; const float* target_dur;
    int B, S, Q, K, pad_idx, exclude_idx; const float* dur_den; float grad_scale;
    float* d_seg; int ld_dseg; float* d_out; int ld_dout; float* loss_out; int64_t* counts; int64_t* tick_a; int64_t* tick_b;
    const uint8_t* drop; float drop_scale; float* dx; float* dx2; float* wsF; float* ws3;
    int defer_finalize;      /* != 0: the launch leaves the per-unit loss partials in ws (and ticks the counters) but does NOT
                                reduce them into loss_out / counts -- no arrival atomics, no last-workgroup pass at the end of
                                the step's longest small kernel; r3d_losses_finalize or r3d_adamw_flat_dropout_fin does it */
} r3d_tail_losses_args;
int r3d_decoder_tail_losses_supported(int H, int n_head, int Q, int rows);
int r3d_decoder_tail_losses(const r3d_tail_losses_args* a, float* ws, void* stream);

/* ---- the decoder layer's query side as ONE launch (csrc/decoder_chain.hip; hidden 128, 8 queries, 8 heads, <= 64 keys) ----
 * One workgroup per clip.  phases (bit mask):
 *   1  forward : cross-attention core on (caq, cakv[:, :128] = keys, cakv[:, 128:] = values; key j of clip b padded iff
 *                key_label[b][j] == pad_idx; probabilities p_ca, dropout drop_ca) -> t2_pre = dropout(ca_o Wo^T + bo) + t1 ->
 *                t2 = norm2(t2_pre) -> ff1 = dropout(relu(t2 W1^T + b1)) -> t3_pre = dropout(ff1 W2^T + b2) + t2
 *                (model/extras/transformer.py:300-306,325-328)
 *   2  tail    : r3d_decoder_tail_losses on the clip (norm3 -> decoder.norm -> heads -> the three losses -> back to d_t3pre /
 *                d_ff2); `tail` and `ws` as for that entry point; tail->x must be t3_pre, tail->dx / dx2 must be d_t3pre / d_ff2
 *   4  backward: d_ff1 = (d_ff2 W2) relu' dropout' -> d t2 = d_ff1 W1 + d_t3pre -> norm2 backward (d_t2pre; LayerNorm parameter
 *                partials part_d2 [B*Q/4][2][128], one pair per 4 rows) -> d_cap = dropout' -> d_cao = d_cap Wo -> attention core
 *                backward (d_caq [B*Q,128], d_cakv [B*S,256])
 * 7 = the training step's whole decoder after the key/value projection in one launch; 1 and 4 stand alone.  Matrices dense
 * row-major; dropout masks optional (NULL). */
typedef struct r3d_decoder_chain_args {
    const float* caq; const float* cakv; const int64_t* key_label; float* p_ca; const uint8_t* drop_ca; float* ca_o;
    const float* wo; const float* bo; const uint8_t* drop_d2; const float* t1; float* t2_pre; const float* g2; const float* be2;
    float* t2; float* m2; float* r2; const float* w1; const float* b1; const uint8_t* drop_ff; float* ff1;
    const float* w2; const float* b2; const uint8_t* drop_d3; float* t3_pre;
    const float* d_t3pre; const float* d_ff2; float* d_ff1; float* d_t2pre; float* d_cap; float* d_cao; float* d_caq;
    float* d_cakv; float* part_d2;
    float drop_scale;
    int32_t pad_idx, B, S, H, Q, heads, phases;
    uint64_t* timeline;      /* profiling aid, normally NULL: wave 0 of clip 0 stores wall_clock64() (100 MHz) at its stage
                                boundaries [0..31] */
    /* optional (all or none): bf16x3 operand-order planes (r3d_weight_planes) of wo, w1, w2 and of their transposes -- the
     * products then run on the bf16 matrix cores */
    const uint16_t* pl_wo; const uint16_t* pl_w1; const uint16_t* pl_w2; const uint16_t* pl_w2_t; const uint16_t* pl_w1_t;
    const uint16_t* pl_wo_t;
} r3d_decoder_chain_args;
int r3d_decoder_chain_supported(int H, int Q, int heads, int S);
int r3d_decoder_chain(const r3d_decoder_chain_args* d, const r3d_tail_losses_args* tail, float* ws, void* stream);

/* ---- optimiser / dropout masks: main_darai.py:135, train/train_proposed_depth.py:215 -------------------------- */
/* torch.optim.AdamW semantics over flat arenas of n floats (n % 4 == 0, 16-byte aligned); g is multiplied by
 * grad_scale first (1/world after a sum all-reduce).  lr and the 1-based step are read from device memory. */
int r3d_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, const int64_t* step,
                   float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* r3d_adamw_flat and r3d_dropout_mask (the NEXT step's masks: the pool is free once the backward has run) in one launch. */
int r3d_adamw_flat_dropout(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, const int64_t* step,
                           float beta1, float beta2, float eps, float weight_decay, float grad_scale, uint8_t* mask,
                           int64_t n_mask, float p_drop, uint64_t seed, const int64_t* offset, void* stream);
/* The reduction r3d_decoder_tail_losses(defer_finalize) left out: loss_out[4] / counts[4] from the 4-float partials of the
 * B*S + B*Q + B units in `part` (the ws of that call), exactly as the undeferred launch computes them (fp64 sums, fixed
 * order).  Stand-alone launch, or -- r3d_adamw_flat_dropout_fin -- one extra workgroup of the AdamW launch, where it costs
 * nothing: the losses are host-visible statistics (utils.py:375-376 reads them per step with .item(); here once per
 * epoch), nothing on the device depends on them. */
typedef struct r3d_loss_finalize_job {
    const float* part; int32_t B, S, Q, has_seg;
    const float* dur_den;            /* optional device scalar, as in r3d_losses_fwd_bwd */
    float* loss_out; int64_t* counts;
    double* acc_loss; int64_t* acc_counts;   /* optional running sums [4] each: acc += this step's values (the epoch statistics
                                                of train/train_proposed_depth.py:216-228, kept on the device) */
} r3d_loss_finalize_job;
int r3d_losses_finalize(const r3d_loss_finalize_job* job, void* stream);
/* r3d_adamw_flat carrying the reduction as its workgroup 0 (fin != NULL), as r3d_adamw_flat_dropout_fin does. */
int r3d_adamw_flat_fin(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, const int64_t* step,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                       const r3d_loss_finalize_job* fin, void* stream);
int r3d_adamw_flat_dropout_fin(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, const int64_t* step,
                               float beta1, float beta2, float eps, float weight_decay, float grad_scale, uint8_t* mask,
                               int64_t n_mask, float p_drop, uint64_t seed, const int64_t* offset,
                               const r3d_loss_finalize_job* fin, void* stream);
/* The same update on a [rows x cols] block (leading dimension ld) of p/g/m/v: a pixel shard of depth_projection.weight. */
int r3d_adamw_2d(float* p, const float* g, float* m, float* v, int rows, int cols, int ld, const float* lr,
                 const int64_t* step, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* mask[i] = 1 with probability 1-p (Philox4x32-10 keyed by seed, counter (i/4, *offset)). */
int r3d_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, const int64_t* offset, void* stream);

/* ---- effective rank (build-side) ------------------------------------------------- */
int64_t r3d_erank_lds_bytes(int R, int C);
int r3d_erank_jacobi(const float* x, int ld, int64_t batch_stride, int batch, int R, int C, int gram, float* sigma,
                     float* af_t, float* stats, int max_sweeps, void* stream);
/* Any size: two-level (block) one-sided Jacobi with the columns in HBM; one matrix X[R, C].  r3d_erank_blocked_sizes
 * fills out[0] = floats of af_t ([Cpad][Rp], the rotated columns (X V)^T; Rp = R rounded up to 4 is the row stride,
 * rows past C and the row tails R..Rp-1 are zero), out[1] = ints of ctrl (scratch), out[2] = Rp, out[3] = columns per
 * block.  af_t must be 16-byte aligned.  sigma [C] unsorted, stats [4] = {erank, entropy, sum sigma, sweeps}.
 * max_sweeps <= 0 selects 16.  Enqueues its launches, no sync. */
int r3d_erank_blocked_sizes(int R, int C, int max_sweeps, int64_t* out);
int r3d_erank_blocked(const float* x, int ld, int R, int C, float* sigma, float* af_t, int* ctrl, float* stats,
                      int max_sweeps, void* stream);
/* The same with the input optionally given transposed (x_transposed != 0: x is [C][ld] = X^T of the [R, C] matrix that
 * is decomposed): the rank-regularised training step decomposes fused^T without a transposition pass when the token
 * matrix has fewer rows than columns. */
int r3d_erank_blocked_t(const float* x, int ld, int x_transposed, int R, int C, float* sigma, float* af_t, int* ctrl,
                        float* stats, int max_sweeps, void* stream);
/* coef[k] = gout * d erank / d sigma_k / sigma_k^3; zero for negligible sigma and, with max_rank = min(R, C) > 0, for
 * everything but the max_rank largest (a rank-deficient X has no defined singular vectors beyond its rank). */
int r3d_erank_bwd_coef(const float* sigma, const float* stats, const float* gout, float* coef, int C, int max_rank,
                       void* stream);
int r3d_scale_rows(float* x, int ld, int rows, int cols, const float* coef, void* stream);
/* The backward in its well-conditioned form: U = (X V) Sigma^-1, V^T = Sigma^-1 (2 I - U^T U) U^T X (one Neumann term of
 * (U^T U)^-1: the plain V^T = Sigma^-1 U^T X amplifies the residual coupling of a small column with a large one by
 * sigma_j / sigma_i), dX = U diag(g) V^T.  r3d_erank_bwd_coef2: cg[k] = gout * (d erank / d sigma_k) / sigma_k,
 * inv[k] = 1 / sigma_k (zero where r3d_erank_bwd_coef's coefficient is).  r3d_erank_bwd_fix: w[r,:] = cg[r] (2 w[r,:] - p[r,:])
 * on dense [rows, cols] matrices.  The host composes the four GEMMs (r3d_amd/erank.py: erank_backward). */
int r3d_erank_bwd_coef2(const float* sigma, const float* stats, const float* gout, float* cg, float* inv, int C, int max_rank,
                        void* stream);
int r3d_erank_bwd_fix(float* w, const float* p, const float* cg, int rows, int cols, void* stream);
/* r3d_erank_jacobi that also carries the right singular basis (batch == 1): every rotation is applied to the rows of
 * V^T as well, vt_out [C][C] = (V0 V')^T.  vt_in = V0^T from an earlier decomposition of a nearby matrix and x = X V0
 * (NULL: identity): a warm start, 3-5 sweeps instead of 10-11.  Matrix and basis must fit the LDS together
 * (r3d_erank_lds_bytes_v).  r3d_erank_vt_polish: one Newton-Schulz step vt <- 1.5 vt_raw - 0.5 gv with
 * gv = (vt_raw vt_raw^T) vt_raw, which keeps the carried basis orthogonal to rounding over many steps. */
int64_t r3d_erank_lds_bytes_v(int R, int C);
int r3d_erank_jacobi_warm(const float* x, int ld, int64_t batch_stride, int batch, int R, int C, int gram, float* sigma,
                          float* af_t, float* stats, int max_sweeps, const float* vt_in, float* vt_out, void* stream);
int r3d_erank_vt_polish(const float* vt_raw, const float* gv, float* vt, int64_t n, void* stream);

/* ---- depth-as-query model (reference model/futr_unsupervised_depth.py) ------------------------------------------------
 * r3d_posenc_fwd / bwd: PositionalEncoding.forward (model/extras/position.py:29-35) on b-major rows,
 *   y[r, :] = dropout(x[r, :] + table[r % S, :]); backward dx = dy * keep * scale, zero where gate <= 0 (gate: NULL or
 *   the post-ReLU embedding of futr_unsupervised_depth.py:97).  drop: NULL or rows*H contiguous keep-bytes.
 * r3d_avgpool_rows_fwd / bwd: F.adaptive_avg_pool1d over the S decoder rows of each clip down to Q rows (:134). */
int r3d_posenc_fwd(const float* x, int ldx, const float* table, int ldt, int S, const uint8_t* drop, float drop_scale,
                   float* y, int ldy, int rows, int H, void* stream);
int r3d_posenc_bwd(const float* dy, int lddy, const uint8_t* drop, float drop_scale, const float* gate, int ldg, float* dx,
                   int lddx, int rows, int H, void* stream);
int r3d_avgpool_rows_fwd(const float* x, int ldx, float* y, int ldy, int B, int S, int Q, int H, void* stream);
int r3d_avgpool_rows_bwd(const float* dy, int lddy, float* dx, int lddx, int B, int S, int Q, int H, void* stream);
/* Label-index queries of model/futr_proposed.py:103-106: out[r, :] = weight[idx[r], :] + table[r % S, :] (nn.Embedding +
 * the sinusoidal table) and the lookup's adjoint d_weight[e, :] = sum_{r : idx[r] == e} d_out[r, :] (deterministic). */
int r3d_embed_gather_fwd(const float* weight, int n_embed, const int64_t* idx, const float* table, int ldt, int S, float* out,
                         int ldo, int rows, int H, void* stream);
int r3d_embed_gather_bwd(const float* d_out, int ldd, const int64_t* idx, float* d_weight, int n_embed, int rows, int H,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* R3D_HIP_H */
