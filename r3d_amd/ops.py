"""Tensor-level wrappers over the C ABI (include/r3d_hip.h).  PyTorch is only plumbing here: it owns the device
buffers and the stream; every call enqueues hand-written HIP kernels on torch's current stream."""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc, GEMM_NT, GEMM_NN, GEMM_TN, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pv(t):
    return None if t is None else t.data_ptr()


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "r3d_amd kernels take device tensors only"
    return C.c_void_p(t.data_ptr())


def _f32(t, name="tensor"):
    assert t.dtype == torch.float32 and t.is_cuda, f"{name} must be a float32 device tensor"
    return t


def _ld(t):
    """leading dimension of a 2-D row-major (possibly column-sliced) tensor"""
    assert t.dim() == 2 and t.stride(1) == 1, "expected a 2-D tensor with unit column stride"
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ----------------------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------------------
class GemmWorkspace:
    """Grow-only split-K slab buffer shared by all GEMMs of a stream-ordered step."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nfloats):
        if nfloats <= 0:
            return None
        if self.buf is None or self.buf.numel() < nfloats:
            self.buf = torch.empty(int(nfloats), dtype=torch.float32, device=self.device)
        return self.buf


def _fill_desc(d, layout, a, b, c, *, bias=None, act=0, pre_out=None, a_add=None, a_add_mod=0, a_row_xor=0, b_add=None,
               b_add_mod=0, drop_mask=None, drop_scale=1.0, aux=None, mul=0, res1=None, res2=None, alpha=1.0,
               accumulate=False, c_row_xor=0, bias_grad=None, prec=0):
    """Fills the problem and epilogue fields of a GemmDesc (not tile / split).  Returns (M, N, K)."""
    _f32(a, "A"), _f32(b, "B"), _f32(c, "C")
    if layout == GEMM_NT:
        M, K = a.shape
        N = b.shape[0]
        assert b.shape[1] == K
    elif layout == GEMM_NN:
        M, K = a.shape
        N = b.shape[1]
        assert b.shape[0] == K
    else:
        K, M = a.shape
        N = b.shape[1]
        assert b.shape[0] == K
    assert tuple(c.shape) == (M, N), f"C is {tuple(c.shape)}, expected {(M, N)}"
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
    d.layout, d.M, d.N, d.K = layout, M, N, K
    d.lda, d.ldb, d.ldc = _ld(a), _ld(b), _ld(c)
    if a_add is not None:
        d.a_add, d.a_add_mod, d.a_add_ld = a_add.data_ptr(), a_add_mod, _ld(a_add)
    d.a_row_xor = a_row_xor
    if b_add is not None:
        d.b_add, d.b_add_mod, d.b_add_ld = b_add.data_ptr(), b_add_mod, _ld(b_add)
    if bias is not None:
        assert bias.numel() == N
        d.bias = bias.data_ptr()
    if pre_out is not None:
        d.pre_out, d.ldpre = pre_out.data_ptr(), _ld(pre_out)
    d.act = act
    if drop_mask is not None:
        assert drop_mask.dtype == torch.uint8
        d.drop_mask, d.lddrop, d.drop_scale = drop_mask.data_ptr(), _ld(drop_mask), drop_scale
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), _ld(aux)
    d.mul = mul
    if res1 is not None:
        d.res1, d.ldr1 = res1.data_ptr(), _ld(res1)
    if res2 is not None:
        d.res2, d.ldr2 = res2.data_ptr(), _ld(res2)
    d.alpha, d.accumulate = alpha, 1 if accumulate else 0
    d.prec = prec
    d.c_row_xor = c_row_xor
    if bias_grad is not None:
        assert layout == GEMM_TN and bias_grad.numel() == M
        d.bias_grad = bias_grad.data_ptr()
    return M, N, K


def gemm(layout, a, b, c, *, ws=None, tile=0, splitk=0, defer_reduce=False, adam=None, **epi):
    """C = epilogue(A op B), see r3d_gemm_desc.  Shapes: NT a[M,K] b[N,K]; NN a[M,K] b[K,N]; TN a[K,M] b[K,N].
    Epilogue keywords: see _fill_desc.  Returns the (filled) descriptor; with defer_reduce=True and split-K the caller
    reduces the slabs itself (layernorm_fwd(nsplit=...), embed_fuse_fwd).
    adam: dict(m, v, lr_t, step_t, beta1, beta2, eps, weight_decay, grad_scale) -> the product is a weight gradient
    and C is the PARAMETER: AdamW is applied in the epilogue (C, m, v updated in place; the gradient is not stored)."""
    lib = _lib.load()
    d = GemmDesc()
    M, N, K = _fill_desc(d, layout, a, b, c, **epi)
    check(lib.r3d_gemm_plan(C.byref(d)), "r3d_gemm_plan")
    if adam is not None:
        assert _ld(adam["m"]) == d.ldc and _ld(adam["v"]) == d.ldc and adam["m"].shape == c.shape
        d.adam_m, d.adam_v = adam["m"].data_ptr(), adam["v"].data_ptr()
        d.adam_lr, d.adam_step = adam["lr_t"].data_ptr(), adam["step_t"].data_ptr()
        d.adam_beta1, d.adam_beta2, d.adam_eps = adam["beta1"], adam["beta2"], adam["eps"]
        d.adam_wd, d.adam_gscale = adam["weight_decay"], adam["grad_scale"]
        d.splitk, d.k_per_split = 1, K
        if d.tile not in (10, 12) and tile not in (10, 12):     # (the bf16x3 TN tiles have the epilogue themselves)
            d.tile = 3 if d.tile in (3, 5) else 2       # the AdamW epilogue needs a tile without k-split waves
    if tile:
        d.tile = tile
    if splitk:
        if splitk == 1:
            d.splitk, d.k_per_split = 1, K
        else:
            q = 64 if d.tile == 8 else (32 if d.tile in (9, 11) else 16)     # the bf16x3 NT kernels step K by 64 / 32
            kps = ((K + splitk - 1) // splitk + q - 1) // q * q
            d.splitk, d.k_per_split = (K + kps - 1) // kps, kps
            if d.splitk < 2:
                d.splitk, d.k_per_split = 1, K
    if epi.get("bias_grad") is not None and d.splitk > 1:   # the fused column sum needs the whole K range in one block
        d.splitk, d.k_per_split = 1, K
    if d.splitk > 1:
        assert ws is not None, "split-K needs a GemmWorkspace"
        d.partial = ws.get(lib.r3d_gemm_partial_floats(M, N, d.splitk)).data_ptr()
    s = _stream()
    check(lib.r3d_gemm_f32(C.byref(d), s), "r3d_gemm_f32")
    if d.splitk > 1 and not defer_reduce:
        check(lib.r3d_splitk_reduce(C.byref(d), s), "r3d_splitk_reduce")
    return d


def gemm_planned_tile(layout, a, b, c, **epi):
    """The tile r3d_gemm_plan picks for this product (no launch)."""
    d = GemmDesc()
    _fill_desc(d, layout, a, b, c, **epi)
    check(_lib.load().r3d_gemm_plan(C.byref(d)), "r3d_gemm_plan")
    return int(d.tile)


def gemm_bf3_nt_pair(a1, b1, c1, ws1, a2, b2, c2, ws2):
    """The long-K product C1 = A1 B1^T (planned onto the bf16x3 split-K tile 8) and a second NT product with the same M x N in
    ONE launch (r3d_gemm_bf3_nt_pair): the second product's K-splits fill the workgroup slots the first leaves empty in its last
    group of 8 (61 splits x 4 tiles = 244 of 256 at the headline shape).  Both leave raw slabs (ws1.buf, ws2.buf) for the
    caller's reducer.  Returns (d1, d2), or None when the shapes do not allow it (the caller launches them separately)."""
    lib = _lib.load()
    d1, d2 = GemmDesc(), GemmDesc()
    M, N, K1 = _fill_desc(d1, GEMM_NT, a1, b1, c1, prec=1)
    M2, N2, K2 = _fill_desc(d2, GEMM_NT, a2, b2, c2, prec=1)
    check(lib.r3d_gemm_plan(C.byref(d1)), "r3d_gemm_plan")
    if d1.tile != 8 or d1.splitk < 2 or (M2, N2) != (M, N) or K2 % 8 or K2 < 64:
        return None
    if any(t.data_ptr() % 16 or t.stride(0) % 4 for t in (a2, b2)):      # (the bf16x3 kernel loads 16-byte row segments)
        return None
    spare = -d1.splitk % 8
    if spare == 0:
        return None
    kps2 = ((K2 + spare - 1) // spare + 63) // 64 * 64
    if kps2 > d1.k_per_split:                   # (its workgroups would outlast the first product's)
        return None
    d2.tile, d2.k_per_split, d2.splitk = 8, kps2, (K2 + kps2 - 1) // kps2
    if d2.splitk < 2:
        return None
    d1.partial = ws1.get(lib.r3d_gemm_partial_floats(M, N, d1.splitk)).data_ptr()
    d2.partial = ws2.get(lib.r3d_gemm_partial_floats(M, N, d2.splitk)).data_ptr()
    check(lib.r3d_gemm_bf3_nt_pair(C.byref(d1), C.byref(d2), _stream()), "r3d_gemm_bf3_nt_pair")
    return d1, d2


class GemmGroup:
    """A set of independent GEMMs of one layout that run as ONE launch (r3d_gemm_grouped_*).  Built once per shape:
    descriptors and the workgroup prefix table are uploaded to device memory; launch() replays them.  The problems may
    use every epilogue of gemm() (each is latency-bound: what a group saves is the ~5 us of a dependent launch)."""

    def __init__(self, layout, problems, tile=1):
        """problems: list of dicts with the keyword arguments of gemm() (a, b, c required; no split-K)."""
        lib = _lib.load()
        n = len(problems)
        arr = (GemmDesc * n)()
        self._keep = list(problems)
        for i, pr in enumerate(problems):
            kw = {k: v for k, v in pr.items() if k not in ("a", "b", "c")}
            _fill_desc(arr[i], layout, pr["a"], pr["b"], pr["c"], **kw)
        prefix = (C.c_int32 * (n + 1))()
        check(lib.r3d_gemm_grouped_prepare(arr, n, tile, prefix), "r3d_gemm_grouped_prepare")
        dev = problems[0]["a"].device
        self.descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.prefix = torch.tensor(list(prefix), dtype=torch.int32, device=dev)
        self._prefix_host = prefix                       # (kernel-argument copy of the table: see r3d_gemm_grouped_launch)
        self.n, self.total, self.layout, self.tile = n, int(prefix[n]), layout, tile
        self._b_ptr = [pr["b"].data_ptr() for pr in problems]

    def set_b(self, i, b):
        """Re-point operand B of problem i (same shape and leading dimension), e.g. at this step's input batch.  A no-op
        while the pointer is unchanged (always, under hipGraph replay with static inputs); otherwise one 8-byte upload."""
        p = b.data_ptr()
        if p == self._b_ptr[i]:
            return
        assert _ld(b) == _ld(self._keep[i]["b"]) and b.shape == self._keep[i]["b"].shape
        off = i * C.sizeof(GemmDesc) + GemmDesc.B.offset
        self.descs[off:off + 8].copy_(torch.frombuffer(bytearray(C.c_uint64(p)), dtype=torch.uint8), non_blocking=False)
        self._b_ptr[i] = p
        self._keep[i] = dict(self._keep[i], b=b)

    def launch(self):
        check(_lib.load().r3d_gemm_grouped_launch(_p(self.descs), _p(self.prefix), self._prefix_host, self.n, self.total,
                                                  self.layout, self.tile, _stream()), "r3d_gemm_grouped_launch")


class RowsumGroup:
    """Bias / broadcast-parameter gradient sums in one launch (r3d_rowmod_sum_batched)."""

    def __init__(self, jobs):
        """jobs: list of (src1, src2 or None, mod, dst) 2-D tensors; dst is [mod, cols]."""
        from ._lib import RowsumJob
        n = len(jobs)
        arr = (RowsumJob * n)()
        self._keep = jobs
        for i, (s1, s2, mod, dst) in enumerate(jobs):
            rows, cols = s1.shape
            assert tuple(dst.shape) == (mod, cols) and (s2 is None or s2.shape == s1.shape)
            arr[i].src1, arr[i].src2, arr[i].dst = s1.data_ptr(), (s2.data_ptr() if s2 is not None else None), dst.data_ptr()
            arr[i].ld1, arr[i].ld2, arr[i].ldd = _ld(s1), (_ld(s2) if s2 is not None else 0), _ld(dst)
            arr[i].rows, arr[i].cols, arr[i].mod = rows, cols, mod
        self.jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(jobs[0][0].device)
        self.n = n
        self.max_cols = max(j[0].shape[1] for j in jobs)
        self.max_mod = max(j[2] for j in jobs)

    def launch(self):
        check(_lib.load().r3d_rowmod_sum_batched(_p(self.jobs), self.n, self.max_cols, self.max_mod, _stream()),
              "r3d_rowmod_sum_batched")


def tick(a, b=None):
    check(_lib.load().r3d_tick(_p(a), _p(b), _stream()), "r3d_tick")


class LnFinalizeGroup:
    """All deferred LayerNorm parameter-gradient reductions of a step in one launch."""

    def __init__(self, jobs):
        """jobs: list of (partial_ws, rows, H, dgamma, dbeta) tensors."""
        from ._lib import LnFinalizeJob
        n = len(jobs)
        arr = (LnFinalizeJob * n)()
        self._keep = jobs
        for i, (wsb, rows, H, dg, db) in enumerate(jobs):
            arr[i].ws, arr[i].dgamma, arr[i].dbeta, arr[i].rows, arr[i].H = wsb.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, H
        self.jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(jobs[0][0].device)
        self.n, self.max_h = n, max(j[2] for j in jobs)

    def launch(self):
        check(_lib.load().r3d_layernorm_bwd_finalize_batched(_p(self.jobs), self.n, self.max_h, _stream()),
              "r3d_layernorm_bwd_finalize_batched")


# ----------------------------------------------------------------------------------------------------------
# row-wise
# ----------------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, y, mean, rstd, *, relu=False, pair_out=None, nsplit=0, bias=None, pre_out=None,
                  rows=None, H=None):
    lib = _lib.load()
    if nsplit > 0:
        assert rows is not None and H is not None
        ldx = H
    else:
        rows, H = x.shape
        ldx = _ld(x)
    check(lib.r3d_layernorm_fwd(_p(x), ldx, nsplit, _p(bias), _p(pre_out), _p(gamma), _p(beta), _p(y), _ld(y), _p(mean),
                                _p(rstd), _p(pair_out), rows, H, 1 if relu else 0, _stream()), "r3d_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, *, pair_in=False, relu=False, dy2=None, add1=None, add2=None,
                  dx2=None, drop_mask=None, drop_scale=1.0, ws=None, partial=None):
    """partial: a dedicated [ws_floats] tensor -> the parameter-gradient reduction is deferred to
    layernorm_bwd_finalize(partial, ...) (which may run on another stream)."""
    lib = _lib.load()
    rows, H = x.shape
    need = lib.r3d_layernorm_bwd_ws_floats(rows, H)
    if partial is not None:
        assert partial.numel() >= need
        wsb = partial
    else:
        wsb = ws.get(need) if (need > 0 and dgamma is not None) else None
    check(lib.r3d_layernorm_bwd(_p(dy), _ld(dy), 1 if pair_in else 0, _p(dy2), _ld(dy2) if dy2 is not None else 0, _p(x), _ld(x), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                1 if relu else 0, _p(add1), _ld(add1) if add1 is not None else 0, _p(add2),
                                _ld(add2) if add2 is not None else 0, _p(dx), _ld(dx), _p(dx2),
                                _ld(dx2) if dx2 is not None else 0, _p(drop_mask),
                                _ld(drop_mask) if drop_mask is not None else 0, drop_scale, _p(dgamma), _p(dbeta),
                                _p(wsb), rows, H, 1 if partial is not None else 0, _stream()), "r3d_layernorm_bwd")


def layernorm_fwd_multi(jobs):
    """jobs: up to 4 dicts (x, gamma, beta, y, mean, rstd[, relu, pair_out]) of independent sites, equal width."""
    from ._lib import LnFwdJob
    arr = (LnFwdJob * len(jobs))()
    for i, j in enumerate(jobs):
        rows, H = j["x"].shape
        a = arr[i]
        a.x, a.ldx, a.nsplit = j["x"].data_ptr(), _ld(j["x"]), 0
        a.gamma, a.beta, a.y, a.ldy = j["gamma"].data_ptr(), j["beta"].data_ptr(), j["y"].data_ptr(), _ld(j["y"])
        a.mean, a.rstd = j["mean"].data_ptr(), j["rstd"].data_ptr()
        a.pair_out = j["pair_out"].data_ptr() if j.get("pair_out") is not None else None
        a.rows, a.H, a.relu = rows, H, 1 if j.get("relu") else 0
    check(_lib.load().r3d_layernorm_fwd_multi(arr, len(jobs), _stream()), "r3d_layernorm_fwd_multi")


def gemm_ln_supported(M, K, H):
    return bool(_lib.load().r3d_gemm_ln_supported(int(M), int(K), int(H)))


def gemm_ln_mha_supported(heads, Lq, Lk, dh):
    return bool(_lib.load().r3d_gemm_ln_mha_supported(int(heads), int(Lq), int(Lk), int(dh)))


def gemm_ln_fwd(jobs, mha=None):
    """nn.Linear -> dropout -> residuals -> LayerNorm as ONE launch (gemm_ln.hip), up to 4 independent jobs.
    mha: dict with the arguments of mha_core_fwd (q, k, v, probs, o, B, heads, Lq, Lk, dh[, drop_mask, drop_scale]) -- an
    independent small attention core riding in the same launch (r3d_gemm_ln_mha_fwd).
    job: dict(a [M,K] | None, w [H,K], bias, drop_mask, drop_scale, res1, res2, pre [M,H], gamma, beta, y, mean, rstd,
    pair_out); a is None: the rows already in `pre` are normalised (plain LayerNorm job)."""
    from ._lib import GemmLnJob
    arr = (GemmLnJob * len(jobs))()
    H = jobs[0]["pre"].shape[1]
    for i, j in enumerate(jobs):
        g = lambda k: j.get(k)                    # noqa: E731
        a, t = arr[i], g("a")
        M = j["pre"].shape[0]
        assert j["pre"].shape[1] == H
        if t is not None:
            _f32(t, "A"), _f32(j["w"], "W")
            assert j["w"].shape == (H, t.shape[1]) and t.shape[0] == M
            a.A, a.lda, a.W, a.ldw, a.K = t.data_ptr(), _ld(t), j["w"].data_ptr(), _ld(j["w"]), t.shape[1]
            a.bias = _pv(g("bias"))
            a.drop_mask, a.lddrop = _pv(g("drop_mask")), _ld(g("drop_mask")) if g("drop_mask") is not None else 0
            a.drop_scale = g("drop_scale") or 1.0
            a.res1, a.ldr1 = _pv(g("res1")), _ld(g("res1")) if g("res1") is not None else 0
            a.res2, a.ldr2 = _pv(g("res2")), _ld(g("res2")) if g("res2") is not None else 0
        else:
            a.K = 0
        a.pre_out, a.ldpre = j["pre"].data_ptr(), _ld(j["pre"])
        a.gamma, a.beta, a.y, a.ldy = j["gamma"].data_ptr(), j["beta"].data_ptr(), j["y"].data_ptr(), _ld(j["y"])
        a.mean, a.rstd = j["mean"].data_ptr(), j["rstd"].data_ptr()
        a.pair_out = _pv(g("pair_out"))
        a.M = M
    if mha is not None:
        from ._lib import MhaJob
        m, g = MhaJob(), mha.get
        m.q, m.ldq, m.k, m.ldk, m.v, m.ldv = (mha["q"].data_ptr(), _ld(mha["q"]), mha["k"].data_ptr(), _ld(mha["k"]),
                                              mha["v"].data_ptr(), _ld(mha["v"]))
        m.key_padding_mask, m.key_label, m.pad_idx = _pv(g("kpm")), _pv(g("key_labels")), g("pad_idx") or 0
        m.probs, m.drop_mask, m.drop_scale = mha["probs"].data_ptr(), _pv(g("drop_mask")), g("drop_scale") or 1.0
        m.o, m.ldo = mha["o"].data_ptr(), _ld(mha["o"])
        m.B, m.heads, m.Lq, m.Lk, m.dh = mha["B"], mha["heads"], mha["Lq"], mha["Lk"], mha["dh"]
        check(_lib.load().r3d_gemm_ln_mha_fwd(arr, len(jobs), H, C.byref(m), _stream()), "r3d_gemm_ln_mha_fwd")
        return
    check(_lib.load().r3d_gemm_ln_fwd(arr, len(jobs), H, _stream()), "r3d_gemm_ln_fwd")


def fuser_chain_supported(N, H, K, B, Q, heads):
    return bool(_lib.load().r3d_fuser_chain_supported(N, H, K, B, Q, heads))


class WeightPlanes:
    """bf16x3 operand-order planes (csrc/chain_bf3.h) of a set of weight matrices, rebuilt by ONE launch (refresh()).
    entries: {key: (tensor [rows, cols] float32 view of the parameter arena, transposed)} -- transposed = False gives the planes
    of B[n][k] = W[n][k] (y = x W^T products), True those of B[n][k] = W[k][n] (dx = dy W products)."""

    def __init__(self, entries, device):
        from ._lib import PlaneJob
        lib = _lib.load()
        self.keys = list(entries)
        sizes, offs, total_el, blocks = [], [], 0, 0
        jobs = (PlaneJob * len(self.keys))()
        for i, k in enumerate(self.keys):
            w, tr = entries[k]
            assert w.dim() == 2 and w.stride(1) == 1 and w.dtype == torch.float32 and w.is_cuda
            N, K = (w.shape[1], w.shape[0]) if tr else (w.shape[0], w.shape[1])
            el = int(lib.r3d_weight_plane_elems(N, K))
            offs.append(total_el)
            sizes.append((N, K))
            j = jobs[i]
            j.src, j.ld, j.N, j.K, j.transposed, j.first_block = w.data_ptr(), w.stride(0), N, K, 1 if tr else 0, blocks
            total_el += el
            blocks += ((N + 15) // 16) * ((K + 31) // 32)
        self.buf = torch.zeros(total_el, dtype=torch.int16, device=device)
        for i in range(len(self.keys)):
            jobs[i].dst = self.buf.data_ptr() + 2 * offs[i]
        raw = bytes(jobs)
        self.jobs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.njobs, self.blocks = len(self.keys), blocks
        self._ptr = {k: self.buf.data_ptr() + 2 * offs[i] for i, k in enumerate(self.keys)}
        self._view = {k: self.buf[offs[i]:offs[i] + int(lib.r3d_weight_plane_elems(*sizes[i]))] for i, k in enumerate(self.keys)}

    def refresh(self):
        check(_lib.load().r3d_weight_planes(_p(self.jobs_dev), self.njobs, self.blocks, _stream()), "r3d_weight_planes")

    def ptr(self, key):
        return self._ptr[key]

    def view(self, key):
        return self._view[key]


class FuserChainFwd:
    """The argument block of r3d_fuser_chain_fwd (csrc/fuser_chain.hip), built once per workspace: every operand is a
    dense row-major tensor whose address is fixed for the life of the workspace (the dropout masks are optional)."""

    def __init__(self, **t):
        from ._lib import FuserChainFwdArgs
        a = FuserChainFwdArgs()
        dims = {k: t.pop(k) for k in ("N", "S", "K", "H", "add_xres", "B", "Q", "heads")}
        a.drop_scale = float(t.pop("drop_scale", 1.0))
        opt = {"y", "drop_sa", "drop_d1"}
        planes = t.pop("planes", None)                 # dict name -> device address of the weight's bf16x3 planes, or None
        if planes is not None:
            for name in FuserChainFwdArgs._PLANES:
                setattr(a, name, planes[name])
        for name in FuserChainFwdArgs._PTRS + FuserChainFwdArgs._PTRS2:
            v = t.pop(name, None)
            if v is None:
                assert name in opt, name
                continue
            assert v.is_cuda and v.is_contiguous(), name
            assert v.dtype == (torch.uint8 if name.startswith("drop_") else torch.float32), name
            setattr(a, name, v.data_ptr())
        assert not t, t.keys()
        for k, v in dims.items():
            setattr(a, k, int(v))
        self.args = a

    def launch(self):
        check(_lib.load().r3d_fuser_chain_fwd(C.byref(self.args), _stream()), "r3d_fuser_chain_fwd")


class FuserChainBwd:
    """The argument block of r3d_fuser_chain_bwd, built once per workspace (see FuserChainFwd)."""

    def __init__(self, **t):
        from ._lib import FuserChainBwdArgs
        a = FuserChainBwdArgs()
        dims = {k: t.pop(k) for k in ("N", "S", "K", "H", "add_xres", "B", "Q", "heads")}
        a.drop_scale = float(t.pop("drop_scale", 1.0))
        opt = {"d_extra", "drop_x0", "drop_d1", "drop_sa", "d_h2", "d_h1", "t1pre_out"}
        planes = t.pop("planes", None)
        if planes is not None:
            for name in FuserChainBwdArgs._PLANES:
                setattr(a, name, planes[name])
        for name in FuserChainBwdArgs._PTRS:
            v = t.pop(name, None)
            if v is None:
                assert name in opt, name
                continue
            assert v.is_cuda and v.is_contiguous(), name
            assert v.dtype == (torch.uint8 if name.startswith("drop_") else torch.float32), name
            setattr(a, name, v.data_ptr())
        assert not t, t.keys()
        for k, v in dims.items():
            setattr(a, k, int(v))
        self.args = a

    def launch(self):
        check(_lib.load().r3d_fuser_chain_bwd(C.byref(self.args), _stream()), "r3d_fuser_chain_bwd")


def decoder_chain_supported(H, Q, heads, S):
    return bool(_lib.load().r3d_decoder_chain_supported(H, Q, heads, S))


class DecoderChain:
    """The argument block of r3d_decoder_chain (csrc/decoder_chain.hip), built once per workspace; key_label is re-pointed
    per call (the step's label tensor)."""

    def __init__(self, **t):
        from ._lib import DecoderChainArgs
        a = DecoderChainArgs()
        dims = {k: t.pop(k) for k in ("pad_idx", "B", "S", "H", "Q", "heads")}
        a.drop_scale = float(t.pop("drop_scale", 1.0))
        opt = {"key_label", "drop_ca", "drop_d2", "drop_ff", "drop_d3", "d_t3pre", "d_ff2", "d_ff1", "d_t2pre", "d_cap", "d_cao",
               "d_caq", "d_cakv", "part_d2"}
        planes = t.pop("planes", None)
        if planes is not None:
            for name in ("pl_wo", "pl_w1", "pl_w2", "pl_w2_t", "pl_w1_t", "pl_wo_t"):
                setattr(a, name, planes[name])
        for name in DecoderChainArgs._PTRS:
            v = t.pop(name, None)
            if v is None:
                assert name in opt, name
                continue
            assert v.is_cuda and v.is_contiguous(), name
            want = torch.uint8 if name.startswith("drop_") else (torch.int64 if name == "key_label" else torch.float32)
            assert v.dtype == want, name
            setattr(a, name, v.data_ptr())
        assert not t, t.keys()
        for k, v in dims.items():
            setattr(a, k, int(v))
        self.args = a

    def launch(self, phases, key_label=None, tail=None, ws=None):
        """tail: a filled TailLossesArgs (phases & 2), ws: the loss scratch."""
        a = self.args
        a.phases = int(phases)
        a.key_label = None if key_label is None else key_label.data_ptr()
        check(_lib.load().r3d_decoder_chain(C.byref(a), C.byref(tail) if tail is not None else None, _p(ws), _stream()),
              "r3d_decoder_chain")


def layernorm_bwd_multi(jobs, mha=None):
    """jobs: up to 4 dicts with the arguments of layernorm_bwd (dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta,
    partial required; pair_in, relu, dy2, add1, add2, dx2, drop_mask, drop_scale optional).
    mha: dict with the arguments of mha_core_bwd (q, k, v, probs, d_o, dq, dk, dv, B, heads, Lq, Lk, dh[, drop_mask,
    drop_scale]) -- an independent small attention backward riding in the same launch (r3d_layernorm_bwd_multi_mha)."""
    from ._lib import LnBwdJob
    arr = (LnBwdJob * len(jobs))()
    for i, j in enumerate(jobs):
        rows, H = j["x"].shape
        a = arr[i]
        g = lambda k: j.get(k)                    # noqa: E731
        a.dy, a.lddy, a.pair_in = j["dy"].data_ptr(), _ld(j["dy"]), 1 if g("pair_in") else 0
        a.dy2, a.lddy2 = _pv(g("dy2")), _ld(g("dy2")) if g("dy2") is not None else 0
        a.x, a.ldx, a.mean, a.rstd = j["x"].data_ptr(), _ld(j["x"]), j["mean"].data_ptr(), j["rstd"].data_ptr()
        a.gamma, a.beta, a.relu = j["gamma"].data_ptr(), j["beta"].data_ptr(), 1 if g("relu") else 0
        a.add1, a.ldadd1 = _pv(g("add1")), _ld(g("add1")) if g("add1") is not None else 0
        a.add2, a.ldadd2 = _pv(g("add2")), _ld(g("add2")) if g("add2") is not None else 0
        a.dx, a.lddx = j["dx"].data_ptr(), _ld(j["dx"])
        a.dx2, a.lddx2 = _pv(g("dx2")), _ld(g("dx2")) if g("dx2") is not None else 0
        a.drop_mask, a.lddrop = _pv(g("drop_mask")), _ld(g("drop_mask")) if g("drop_mask") is not None else 0
        a.drop_scale = g("drop_scale") or 1.0
        a.dgamma, a.dbeta, a.ws = j["dgamma"].data_ptr(), j["dbeta"].data_ptr(), j["partial"].data_ptr()
        a.rows, a.H = rows, H
        assert j["partial"].numel() >= _lib.load().r3d_layernorm_bwd_ws_floats(rows, H)
    if mha is not None:
        from ._lib import MhaBwdJob
        m, gg = MhaBwdJob(), mha.get
        for n in ("q", "k", "v"):
            setattr(m, n, mha[n].data_ptr())
            setattr(m, "ld" + n, _ld(mha[n]))
        m.probs, m.drop_mask, m.drop_scale = mha["probs"].data_ptr(), _pv(gg("drop_mask")), gg("drop_scale") or 1.0
        m.d_o, m.lddo = mha["d_o"].data_ptr(), _ld(mha["d_o"])
        for n in ("dq", "dk", "dv"):
            setattr(m, n, mha[n].data_ptr())
            setattr(m, "ld" + n, _ld(mha[n]))
        m.B, m.heads, m.Lq, m.Lk, m.dh = mha["B"], mha["heads"], mha["Lq"], mha["Lk"], mha["dh"]
        check(_lib.load().r3d_layernorm_bwd_multi_mha(arr, len(jobs), C.byref(m), _stream()), "r3d_layernorm_bwd_multi_mha")
        return
    check(_lib.load().r3d_layernorm_bwd_multi(arr, len(jobs), _stream()), "r3d_layernorm_bwd_multi")


def layernorm_bwd_finalize(partial, rows, H, dgamma, dbeta):
    lib = _lib.load()
    check(lib.r3d_layernorm_bwd_finalize(_p(partial), rows, H, _p(dgamma), _p(dbeta), _stream()), "r3d_layernorm_bwd_finalize")


def layernorm_bwd_ws_floats(rows, H):
    return int(_lib.load().r3d_layernorm_bwd_ws_floats(rows, H))


def add_rowbcast(x, add, mod, out):
    lib = _lib.load()
    rows, cols = out.shape
    check(lib.r3d_add_rowbcast(_p(x), _ld(x) if x is not None else 0, _p(add), _ld(add), mod, _p(out), _ld(out), rows, cols,
                               _stream()), "r3d_add_rowbcast")


def colsum(x, out, *, accumulate=False, ws=None):
    lib = _lib.load()
    rows, cols = x.shape
    need = lib.r3d_colsum_ws_floats(rows, cols)
    wsb = ws.get(need) if need > 0 else None
    check(lib.r3d_colsum(_p(x), _ld(x), rows, cols, _p(out), _p(wsb), 1 if accumulate else 0, _stream()), "r3d_colsum")


def rowmod_sum(x, mod, out, *, accumulate=False):
    lib = _lib.load()
    rows, cols = x.shape
    check(lib.r3d_rowmod_sum(_p(x), _ld(x), rows, cols, mod, _p(out), _ld(out), 1 if accumulate else 0, _stream()),
          "r3d_rowmod_sum")


# ----------------------------------------------------------------------------------------------------------
# token selection / exchange
# ----------------------------------------------------------------------------------------------------------
def colabssum(x, out):
    lib = _lib.load()
    rows, cols = x.shape
    assert out.dtype == torch.float64
    check(lib.r3d_colabssum(_p(x), _ld(x), rows, cols, _p(out), _stream()), "r3d_colabssum")


def token_select(k, idx_out, mask_out, *, score_f=None, score_sum=None, count=0.0, used_serial=None):
    lib = _lib.load()
    src = score_f if score_f is not None else score_sum
    nvec, Cc = src.shape
    assert src.is_contiguous() and idx_out.dtype == torch.int64 and tuple(idx_out.shape) == (nvec, k)
    check(lib.r3d_token_select(_p(score_f), _p(score_sum), float(count), nvec, Cc, k, _p(idx_out), _p(mask_out),
                               _p(used_serial), _stream()), "r3d_token_select")


def token_exchange_fwd(rgb, dep, mask_rgb, mask_dep, x0, *, drop_mask=None, drop_scale=1.0):
    lib = _lib.load()
    N, H = rgb.shape
    assert rgb.is_contiguous() and dep.is_contiguous() and x0.is_contiguous()
    check(lib.r3d_token_exchange_fwd(_p(rgb), _p(dep), _p(mask_rgb), _p(mask_dep), _p(x0), _p(drop_mask), drop_scale, N, H,
                                     _stream()), "r3d_token_exchange_fwd")


def token_exchange_bwd(dx0, rgb, mask_rgb, mask_dep, d_rgb_pre, d_dep, *, drop_mask=None, drop_scale=1.0):
    lib = _lib.load()
    N, H = rgb.shape
    check(lib.r3d_token_exchange_bwd(_p(dx0), _p(rgb), _p(mask_rgb), _p(mask_dep), _p(drop_mask), drop_scale, _p(d_rgb_pre),
                                     _p(d_dep), N, H, _stream()), "r3d_token_exchange_bwd")


# ----------------------------------------------------------------------------------------------------------
# attention core
# ----------------------------------------------------------------------------------------------------------
def mha_core_fwd(q, k, v, probs, o, B, heads, Lq, Lk, dh, *, kpm=None, key_labels=None, pad_idx=0, drop_mask=None,
                 drop_scale=1.0):
    lib = _lib.load()
    check(lib.r3d_mha_core_fwd(_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(kpm), _p(key_labels), pad_idx, _p(probs), _p(drop_mask),
                               drop_scale, _p(o), _ld(o), B, heads, Lq, Lk, dh, _stream()), "r3d_mha_core_fwd")


def mha_core_bwd(q, k, v, probs, d_o, dq, dk, dv, B, heads, Lq, Lk, dh, *, drop_mask=None, drop_scale=1.0):
    lib = _lib.load()
    check(lib.r3d_mha_core_bwd(_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(probs), _p(drop_mask), drop_scale, _p(d_o),
                               _ld(d_o), _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv), B, heads, Lq, Lk, dh,
                               _stream()), "r3d_mha_core_bwd")


# ----------------------------------------------------------------------------------------------------------
# fused decoder layer
# ----------------------------------------------------------------------------------------------------------
def decoder_fused_supported(S, Q, H, heads):
    return bool(_lib.load().r3d_decoder_fused_supported(S, Q, H, heads))


class PtrTable:
    """A fixed array of device pointers (None -> NULL) for the entry points that take `const void* const*`."""

    def __init__(self, tensors):
        self._keep = tensors
        self.n = len(tensors)
        self.arr = (C.c_void_p * self.n)(*[(t.data_ptr() if t is not None else None) for t in tensors])


def decoder_layer_fwd(table, B, S, Q, H, heads, pad_idx, drop_scale, n_head_out):
    check(_lib.load().r3d_decoder_layer_fwd(table.arr, table.n, B, S, Q, H, heads, pad_idx, drop_scale, n_head_out, _stream()),
          "r3d_decoder_layer_fwd")


def bn_stats(x_rgb, x_dep, bn_rgb, bn_dep, mean, rstd, absg, training, momentum=0.1):
    """bn_rgb / bn_dep: the nn.BatchNorm1d parameter / buffer holders (weight, running_mean, running_var, num_batches_tracked)."""
    N, Cc = x_rgb.shape
    check(_lib.load().r3d_bn_stats(_p(x_rgb), _p(x_dep), _p(bn_rgb.running_mean), _p(bn_rgb.running_var),
                                   _p(bn_rgb.num_batches_tracked), _p(bn_dep.running_mean), _p(bn_dep.running_var),
                                   _p(bn_dep.num_batches_tracked), _p(bn_rgb.weight), _p(bn_dep.weight), _p(mean), _p(rstd),
                                   _p(absg), N, Cc, 1 if training else 0, momentum, _stream()), "r3d_bn_stats")


def bn_blend_fwd(rgb, dep, mean, rstd, g_r, b_r, g_d, b_d, alpha, m_rgb, m_dep, drop, drop_scale, ln1_g, ln1_b, x0, h1, m1, r1):
    N, Cc = rgb.shape
    check(_lib.load().r3d_bn_blend_fwd(_p(rgb), _p(dep), _p(mean), _p(rstd), _p(g_r), _p(b_r), _p(g_d), _p(b_d), _p(alpha),
                                       _p(m_rgb), _p(m_dep), _p(drop), drop_scale, _p(ln1_g), _p(ln1_b), _p(x0), _p(h1),
                                       _p(m1), _p(r1), N, Cc, _stream()), "r3d_bn_blend_fwd")


def bn_blend_bwd(d_h1, x0, m1, r1, ln1_g, add1, drop, drop_scale, rgb, dep, mean, rstd, g_r, b_r, g_d, b_d, alpha, m_rgb, m_dep,
                 t0, t1, t2, t3, t4, ws_n1):
    N, Cc = rgb.shape
    assert ws_n1.numel() >= 2 * N * Cc
    check(_lib.load().r3d_bn_blend_bwd(_p(d_h1), _p(x0), _p(m1), _p(r1), _p(ln1_g), _p(add1), _p(drop), drop_scale, _p(rgb),
                                       _p(dep), _p(mean), _p(rstd), _p(g_r), _p(b_r), _p(g_d), _p(b_d), _p(alpha), _p(m_rgb),
                                       _p(m_dep), _p(t0), _p(t1), _p(t2), _p(t3), _p(t4), _p(ws_n1), N, Cc, _stream()),
          "r3d_bn_blend_bwd")


def bn_bwd_apply(rgb, dep, mean, rstd, g_r, g_d, t_drb, t_ddb, dg_r, db_r, dg_d, db_d, d_rgb_pre, d_dep, training):
    N, Cc = rgb.shape
    check(_lib.load().r3d_bn_bwd_apply(_p(rgb), _p(dep), _p(mean), _p(rstd), _p(g_r), _p(g_d), _p(t_drb), _p(t_ddb), _p(dg_r),
                                       _p(db_r), _p(dg_d), _p(db_d), _p(d_rgb_pre), _p(d_dep), N, Cc, 1 if training else 0,
                                       _stream()), "r3d_bn_bwd_apply")


def bn_sync_pack(mean, rstd, N, out):
    Cc = mean.shape[-1]
    assert out.numel() == 4 * Cc + 1 and out.is_contiguous()
    check(_lib.load().r3d_bn_sync_pack(_p(mean), _p(rstd), N, Cc, _p(out), _stream()), "r3d_bn_sync_pack")


def bn_sync_finalize(allp, n_local, mean, rstd, bn_rgb, bn_dep, nfrac, momentum=0.1):
    world, Cc = allp.shape[0], mean.shape[-1]
    assert allp.is_contiguous() and allp.shape[1] == 4 * Cc + 1
    check(_lib.load().r3d_bn_sync_finalize(_p(allp), world, n_local, Cc, _p(mean), _p(rstd), _p(bn_rgb.running_mean),
                                           _p(bn_rgb.running_var), _p(bn_rgb.num_batches_tracked), _p(bn_dep.running_mean),
                                           _p(bn_dep.running_var), _p(bn_dep.num_batches_tracked), _p(nfrac), momentum,
                                           _stream()), "r3d_bn_sync_finalize")


def decoder_tail_fwd(x, g3, b3, gF, bF, w_head, b_head, t3, m3, r3, tgtF, mF, rF, out):
    rows, H = x.shape
    assert x.is_contiguous() and t3.is_contiguous() and tgtF.is_contiguous()
    check(_lib.load().r3d_decoder_tail_fwd(_p(x), _p(g3), _p(b3), _p(gF), _p(bF), _p(w_head), _p(b_head), w_head.shape[0],
                                           _p(t3), _p(m3), _p(r3), _p(tgtF), _p(mF), _p(rF), _p(out), _ld(out), rows, H,
                                           _stream()), "r3d_decoder_tail_fwd")


def decoder_tail_bwd(d_out, w_head, t3, mF, rF, gF, x, m3, r3, g3, drop_mask, drop_scale, dx, dx2, dgF, dbF, dg3, db3, wsF, ws3):
    rows, H = x.shape
    check(_lib.load().r3d_decoder_tail_bwd(_p(d_out), _ld(d_out), _p(w_head), w_head.shape[0], _p(t3), _p(mF), _p(rF), _p(gF),
                                           _p(x), _p(m3), _p(r3), _p(g3), _p(drop_mask), drop_scale, _p(dx), _p(dx2),
                                           _p(dgF), _p(dbF), _p(dg3), _p(db3), _p(wsF), _p(ws3), rows, H, _stream()),
          "r3d_decoder_tail_bwd")


def embed_fuse_fwd(rgb_src, ns_r, bias_r, dep_src, ns_d, bias_d, lnd_g, lnd_b, m_rgb, m_dep, drop, drop_scale, ln1_g, ln1_b,
                   rgb_out, dep_pre_out, mean_d, rstd_d, dep_out, x0, h1, m1, r1, planes=None):
    """planes (a WeightPlanes): its refresh() rides in the same launch as extra workgroups."""
    N, H = dep_out.shape
    if planes is not None:
        check(_lib.load().r3d_embed_fuse_fwd_planes(
            _p(rgb_src), ns_r, _p(bias_r), _p(dep_src), ns_d, _p(bias_d), _p(lnd_g), _p(lnd_b), _p(m_rgb), _p(m_dep), _p(drop),
            drop_scale, _p(ln1_g), _p(ln1_b), _p(rgb_out), _p(dep_pre_out), _p(mean_d), _p(rstd_d), _p(dep_out), _p(x0), _p(h1),
            _p(m1), _p(r1), N, H, _p(planes.jobs_dev), planes.njobs, planes.blocks, _stream()), "r3d_embed_fuse_fwd_planes")
        return
    check(_lib.load().r3d_embed_fuse_fwd(_p(rgb_src), ns_r, _p(bias_r), _p(dep_src), ns_d, _p(bias_d), _p(lnd_g), _p(lnd_b),
                                         _p(m_rgb), _p(m_dep), _p(drop), drop_scale, _p(ln1_g), _p(ln1_b), _p(rgb_out),
                                         _p(dep_pre_out), _p(mean_d), _p(rstd_d), _p(dep_out), _p(x0), _p(h1), _p(m1),
                                         _p(r1), N, H, _stream()), "r3d_embed_fuse_fwd")


def embed_fuse_bwd(d_h1, x0, m1, r1, ln1_g, add1, add2, drop, drop_scale, m_rgb, m_dep, rgb, dep_pre, mean_d, rstd_d, lnd_g,
                   lnd_b, d_rgb_pre, d_dep_pre, ws_n1, ws_dep):
    N, H = rgb.shape
    assert ws_n1.numel() >= 2 * N * H and ws_dep.numel() >= 2 * N * H
    check(_lib.load().r3d_embed_fuse_bwd(_p(d_h1), _p(x0), _p(m1), _p(r1), _p(ln1_g), _p(add1), _p(add2), _p(drop),
                                         drop_scale, _p(m_rgb), _p(m_dep), _p(rgb), _p(dep_pre), _p(mean_d), _p(rstd_d),
                                         _p(lnd_g), _p(lnd_b), _p(d_rgb_pre), _p(d_dep_pre), _p(ws_n1), _p(ws_dep), N, H,
                                         _stream()), "r3d_embed_fuse_bwd")


# ----------------------------------------------------------------------------------------------------------
# losses / optimiser / dropout / erank
# ----------------------------------------------------------------------------------------------------------
def losses_fwd_bwd(seg, act, dur, ld_dur, past_label, target, target_dur, B, S, Q, K, pad_idx, exclude_idx, loss_out,
                   counts, *, val_mode=False, dur_den=None, grad_scale=1.0, d_seg=None, d_act=None, d_dur=None,
                   ld_ddur=1, ws=None, tick_a=None, tick_b=None):
    """ws: zero-initialised float32 scratch of losses_ws_floats(B, S, Q) (kept zero-terminated by the kernel)."""
    lib = _lib.load()
    assert past_label.dtype == torch.int64 and target.dtype == torch.int64 and target_dur.dtype == torch.float32
    assert past_label.is_contiguous() and target.is_contiguous() and target_dur.is_contiguous()
    assert counts.dtype == torch.int64 and loss_out.dtype == torch.float32
    need = lib.r3d_losses_ws_floats(B, S, Q)
    if ws is None:
        ws = torch.zeros(need, dtype=torch.float32, device=act.device)
    assert ws.numel() >= need
    check(lib.r3d_losses_fwd_bwd(_p(seg), _ld(seg) if seg is not None else 0, _p(act), _ld(act), _p(dur), ld_dur,
                                 _p(past_label), _p(target), _p(target_dur), B, S, Q, K, pad_idx, exclude_idx,
                                 1 if val_mode else 0, _p(dur_den), grad_scale, _p(d_seg),
                                 _ld(d_seg) if d_seg is not None else 0, _p(d_act), _ld(d_act) if d_act is not None else 0,
                                 _p(d_dur), ld_ddur, _p(loss_out), _p(counts), _p(ws), _p(tick_a), _p(tick_b), _stream()),
          "r3d_losses_fwd_bwd")


def tail_losses_supported(H, n_head, Q, rows):
    return bool(_lib.load().r3d_decoder_tail_losses_supported(H, n_head, Q, rows))


def decoder_tail_losses(*, x, g3, b3, gF, bF, w_head, b_head, t3, m3, r3, tgtF, mF, rF, out, seg, past_label, target,
                        target_dur, B, S, Q, K, pad_idx, exclude_idx, dur_den, grad_scale, d_seg, d_out, loss_out, counts,
                        tick_a, tick_b, drop, drop_scale, dx, dx2, wsF, ws3, ws, defer_finalize=False, chain=None,
                        chain_key_label=None):
    """r3d_decoder_tail_fwd + r3d_losses_fwd_bwd + r3d_decoder_tail_bwd in one launch (training step).
    defer_finalize: leave the reduction of the loss partials (ws) to losses_finalize() / adamw_flat_dropout(loss_fin=...)."""
    from ._lib import TailLossesArgs
    assert past_label.dtype == torch.int64 and target.dtype == torch.int64 and target_dur.dtype == torch.float32
    assert past_label.is_contiguous() and target.is_contiguous() and target_dur.is_contiguous()
    for t_ in (x, t3, tgtF, dx, dx2):
        assert t_.is_contiguous()
    pv = lambda t_: None if t_ is None else t_.data_ptr()        # noqa: E731
    a = TailLossesArgs()
    for n, v in dict(x=x, g3=g3, b3=b3, gF=gF, bF=bF, w_head=w_head, b_head=b_head, t3=t3, m3=m3, r3=r3, tgtF=tgtF, mF=mF,
                     rF=rF, out=out, seg=seg, past_label=past_label, target=target, target_dur=target_dur, dur_den=dur_den,
                     d_seg=d_seg, d_out=d_out, loss_out=loss_out, counts=counts, tick_a=tick_a, tick_b=tick_b, drop=drop,
                     dx=dx, dx2=dx2, wsF=wsF, ws3=ws3).items():
        if v is not None:
            assert v.is_cuda
        setattr(a, n, pv(v))
    a.n_head, a.ld_out, a.H = w_head.shape[0], _ld(out), x.shape[1]
    a.ld_seg, a.ld_dseg, a.ld_dout = _ld(seg), _ld(d_seg), _ld(d_out)
    a.B, a.S, a.Q, a.K, a.pad_idx, a.exclude_idx = B, S, Q, K, pad_idx, exclude_idx
    a.grad_scale, a.drop_scale = grad_scale, drop_scale
    a.defer_finalize = 1 if defer_finalize else 0
    if chain is not None:              # the whole query side of the decoder layer around the tail: one launch
        chain.launch(7, key_label=chain_key_label, tail=a, ws=ws)
        return
    check(_lib.load().r3d_decoder_tail_losses(C.byref(a), _p(ws), _stream()), "r3d_decoder_tail_losses")


def losses_ws_floats(B, S, Q):
    return int(_lib.load().r3d_losses_ws_floats(B, S, Q))


def adamw_flat(p, g, m, v, lr_t, step_t, *, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, loss_fin=None):
    lib = _lib.load()
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    assert lr_t.dtype == torch.float32 and step_t.dtype == torch.int64
    if loss_fin is not None:               # the deferred loss reduction rides as workgroup 0 (r3d_adamw_flat_fin)
        check(lib.r3d_adamw_flat_fin(_p(p), _p(g), _p(m), _p(v), n, _p(lr_t), _p(step_t), beta1, beta2, eps, weight_decay,
                                     grad_scale, C.byref(loss_fin), _stream()), "r3d_adamw_flat_fin")
        return
    check(lib.r3d_adamw_flat(_p(p), _p(g), _p(m), _p(v), n, _p(lr_t), _p(step_t), beta1, beta2, eps, weight_decay, grad_scale,
                             _stream()), "r3d_adamw_flat")


def loss_finalize_job(part, B, S, Q, has_seg, dur_den, loss_out, counts, acc_loss=None, acc_counts=None):
    """r3d_loss_finalize_job for the partials a decoder_tail_losses(defer_finalize=True) launch left in `part`.
    acc_loss (float64[4]) / acc_counts (int64[4]): running sums the reduction also adds this step's values to."""
    from ._lib import LossFinalizeJob
    j = LossFinalizeJob()
    j.part, j.B, j.S, j.Q, j.has_seg = part.data_ptr(), B, S, Q, 1 if has_seg else 0
    j.dur_den, j.loss_out, j.counts = _pv(dur_den), loss_out.data_ptr(), counts.data_ptr()
    if acc_loss is not None:
        assert acc_loss.dtype == torch.float64 and acc_counts.dtype == torch.int64 and acc_loss.numel() == 4
        j.acc_loss, j.acc_counts = acc_loss.data_ptr(), acc_counts.data_ptr()
    return j


def losses_finalize(job):
    check(_lib.load().r3d_losses_finalize(C.byref(job), _stream()), "r3d_losses_finalize")


def adamw_flat_dropout(p, g, m, v, lr_t, step_t, mask, p_drop, seed, offset_t, *, beta1=0.9, beta2=0.999, eps=1e-8,
                       weight_decay=0.0, grad_scale=1.0, loss_fin=None):
    """adamw_flat + dropout_mask(mask, p_drop, seed, offset_t) in one launch (the masks are the next step's); loss_fin
    (a loss_finalize_job): the deferred loss reduction rides along as one more workgroup."""
    lib = _lib.load()
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and mask.dtype == torch.uint8
    if loss_fin is not None:
        check(lib.r3d_adamw_flat_dropout_fin(_p(p), _p(g), _p(m), _p(v), n, _p(lr_t), _p(step_t), beta1, beta2, eps,
                                             weight_decay, grad_scale, _p(mask), mask.numel(), p_drop, seed, _p(offset_t),
                                             C.byref(loss_fin), _stream()), "r3d_adamw_flat_dropout_fin")
        return
    check(lib.r3d_adamw_flat_dropout(_p(p), _p(g), _p(m), _p(v), n, _p(lr_t), _p(step_t), beta1, beta2, eps, weight_decay,
                                     grad_scale, _p(mask), mask.numel(), p_drop, seed, _p(offset_t), _stream()),
          "r3d_adamw_flat_dropout")


def adamw_2d(p, g, m, v, lr_t, step_t, *, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """p, g, m, v: [rows, cols] views with the same (possibly larger) row stride."""
    lib = _lib.load()
    rows, cols = p.shape
    ld = _ld(p)
    assert _ld(g) == ld and _ld(m) == ld and _ld(v) == ld and g.shape == p.shape
    check(lib.r3d_adamw_2d(_p(p), _p(g), _p(m), _p(v), rows, cols, ld, _p(lr_t), _p(step_t), beta1, beta2, eps, weight_decay,
                           grad_scale, _stream()), "r3d_adamw_2d")


def dropout_mask(mask, p, seed, offset_t=None):
    lib = _lib.load()
    assert mask.dtype == torch.uint8 and mask.is_contiguous()
    check(lib.r3d_dropout_mask(_p(mask), mask.numel(), p, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), _p(offset_t), _stream()),
          "r3d_dropout_mask")


def erank_fits(R, Cc):
    return _lib.load().r3d_erank_lds_bytes(R, Cc) <= 160 * 1024 - 256


def erank_jacobi(x, sigma, stats, *, af_t=None, gram=False, max_sweeps=30):
    """x: [batch, R, C] contiguous (or [R, C])."""
    lib = _lib.load()
    if x.dim() == 2:
        x = x.unsqueeze(0)
    batch, R, Cc = x.shape
    assert x.stride(2) == 1
    check(lib.r3d_erank_jacobi(_p(x), x.stride(1), x.stride(0), batch, R, Cc, 1 if gram else 0, _p(sigma), _p(af_t), _p(stats),
                               max_sweeps, _stream()), "r3d_erank_jacobi")


def erank_blocked(x, max_sweeps=30):
    """(max_sweeps: the sweeps ENQUEUED -- the launches past convergence are no-ops of ~3 us each.  Fused tokens converge in
    9-13; a spectrum with clusters of singular values 0.1 % apart at sigma_max / sigma_min = 1e4 needed 19 at [512, 512]
    (tests/test_kernels_gpu.py::test_erank_clustered_and_ill_conditioned_spectra), and an unconverged sweep leaves an
    effective rank that is still right to 1e-3 but a gradient that is not -- so the differentiable op enqueues 30; the
    training step, where every no-op launch sits in the replayed graph, enqueues engine.erank_max_sweeps = 16 and the
    test of the step asserts that convergence came earlier.)
    x: [R, C] (row stride >= C).  Returns (sigma [C], stats [4], af_t [Cpad, R] -- a view of the [Cpad, Rp] buffer the
    kernel sweeps, Rp = R rounded up to 4) -- any size, columns in HBM."""
    import ctypes
    lib = _lib.load()
    R, Cc = x.shape
    assert x.stride(1) == 1
    sz = (ctypes.c_int64 * 4)()
    check(lib.r3d_erank_blocked_sizes(R, Cc, max_sweeps, ctypes.cast(sz, ctypes.c_void_p)), "r3d_erank_blocked_sizes")
    Rp = int(sz[2])
    af_t = torch.empty(sz[0] // Rp, Rp, dtype=torch.float32, device=x.device)
    ctrl = torch.empty(sz[1], dtype=torch.int32, device=x.device)
    sigma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    stats = torch.empty(4, dtype=torch.float32, device=x.device)
    check(lib.r3d_erank_blocked(_p(x), x.stride(0), R, Cc, _p(sigma), _p(af_t), _p(ctrl), _p(stats), max_sweeps, _stream()),
          "r3d_erank_blocked")
    return sigma, stats, af_t[:, :R]


class ErankBlockedBufs:
    """Pre-allocated outputs / scratch of r3d_erank_blocked_t for one [R, C] matrix (the training step calls it every step,
    inside a hipGraph): sigma [C], stats [4], af_t [Cpad, Rp] (rows = rotated columns (X V)^T), ctrl."""

    def __init__(self, R, Cc, device, max_sweeps=16):
        import ctypes
        sz = (ctypes.c_int64 * 4)()
        check(_lib.load().r3d_erank_blocked_sizes(R, Cc, max_sweeps, ctypes.cast(sz, ctypes.c_void_p)), "r3d_erank_blocked_sizes")
        self.R, self.C, self.Rp, self.max_sweeps = R, Cc, int(sz[2]), max_sweeps
        self.af_full = torch.zeros(sz[0] // self.Rp, self.Rp, dtype=torch.float32, device=device)
        self.ctrl = torch.zeros(sz[1], dtype=torch.int32, device=device)
        self.sigma = torch.empty(Cc, dtype=torch.float32, device=device)
        self.stats = torch.empty(4, dtype=torch.float32, device=device)

    @property
    def af_t(self):
        """[C, R] view (row stride Rp) of the rotated columns."""
        return self.af_full[:self.C, :self.R]


def erank_blocked_into(x, bufs, transposed=False):
    """Decomposes X [R, C] = x (or x^T when transposed: x is then [C, R] row-major) into bufs; enqueue only."""
    if transposed:
        assert x.shape == (bufs.C, bufs.R) and x.stride(1) == 1
    else:
        assert x.shape == (bufs.R, bufs.C) and x.stride(1) == 1
    check(_lib.load().r3d_erank_blocked_t(_p(x), x.stride(0), 1 if transposed else 0, bufs.R, bufs.C, _p(bufs.sigma),
                                          _p(bufs.af_full), _p(bufs.ctrl), _p(bufs.stats), bufs.max_sweeps, _stream()),
          "r3d_erank_blocked_t")


def erank_bwd_coef(sigma, stats, gout, coef, max_rank=0):
    """max_rank = min(R, C) of the decomposed matrix: singular values beyond it are rounding noise."""
    lib = _lib.load()
    check(lib.r3d_erank_bwd_coef(_p(sigma), _p(stats), _p(gout), _p(coef), sigma.numel(), max_rank, _stream()),
          "r3d_erank_bwd_coef")


def erank_bwd_coef2(sigma, stats, gout, cg, inv, max_rank=0):
    check(_lib.load().r3d_erank_bwd_coef2(_p(sigma), _p(stats), _p(gout), _p(cg), _p(inv), sigma.numel(), max_rank, _stream()),
          "r3d_erank_bwd_coef2")


def erank_bwd_fix(w, p, cg):
    assert w.is_contiguous() and p.is_contiguous() and w.shape == p.shape and cg.numel() == w.shape[0]
    check(_lib.load().r3d_erank_bwd_fix(_p(w), _p(p), _p(cg), w.shape[0], w.shape[1], _stream()), "r3d_erank_bwd_fix")


def scale_rows(x, coef):
    lib = _lib.load()
    rows, cols = x.shape
    check(lib.r3d_scale_rows(_p(x), _ld(x), rows, cols, _p(coef), _stream()), "r3d_scale_rows")


def erank_fits_warm(R, Cc):
    return _lib.load().r3d_erank_lds_bytes_v(R, Cc) <= 160 * 1024 - 256


def erank_jacobi_warm(x, sigma, stats, vt_out, *, vt_in=None, af_t=None, max_sweeps=30):
    """x: [R, C] = X V0 (vt_in = V0^T, None: identity); vt_out [C, C] <- (V0 V')^T.  See r3d_erank_jacobi_warm."""
    lib = _lib.load()
    R, Cc = x.shape
    assert x.stride(1) == 1 and vt_out.is_contiguous() and tuple(vt_out.shape) == (Cc, Cc)
    assert vt_in is None or (vt_in.is_contiguous() and tuple(vt_in.shape) == (Cc, Cc))
    check(lib.r3d_erank_jacobi_warm(_p(x), x.stride(0), 0, 1, R, Cc, 0, _p(sigma), _p(af_t), _p(stats), max_sweeps, _p(vt_in),
                                    _p(vt_out), _stream()), "r3d_erank_jacobi_warm")


def erank_vt_polish(vt_raw, gv, vt):
    lib = _lib.load()
    assert vt_raw.is_contiguous() and gv.is_contiguous() and vt.is_contiguous()
    check(lib.r3d_erank_vt_polish(_p(vt_raw), _p(gv), _p(vt), vt.numel(), _stream()), "r3d_erank_vt_polish")


# ----------------------------------------------------------------------------------------------------------
# depth-as-query model (model/futr_unsupervised_depth.py)
# ----------------------------------------------------------------------------------------------------------
def posenc_fwd(x, table, S, y, *, drop_mask=None, drop_scale=1.0):
    """y = dropout(x + table[row % S]) (model/extras/position.py:29-35)."""
    rows, H = x.shape
    assert table.shape[0] >= S and table.shape[1] == H
    check(_lib.load().r3d_posenc_fwd(_p(x), _ld(x), _p(table), _ld(table), S, _p(drop_mask), drop_scale, _p(y), _ld(y), rows, H,
                                     _stream()), "r3d_posenc_fwd")


def posenc_bwd(dy, dx, *, drop_mask=None, drop_scale=1.0, gate=None):
    rows, H = dy.shape
    check(_lib.load().r3d_posenc_bwd(_p(dy), _ld(dy), _p(drop_mask), drop_scale, _p(gate), _ld(gate) if gate is not None else 0,
                                     _p(dx), _ld(dx), rows, H, _stream()), "r3d_posenc_bwd")


def avgpool_rows_fwd(x, y, B, S, Q):
    H = x.shape[1]
    assert x.shape[0] == B * S and tuple(y.shape) == (B * Q, H)
    check(_lib.load().r3d_avgpool_rows_fwd(_p(x), _ld(x), _p(y), _ld(y), B, S, Q, H, _stream()), "r3d_avgpool_rows_fwd")


def avgpool_rows_bwd(dy, dx, B, S, Q):
    H = dy.shape[1]
    assert dy.shape[0] == B * Q and tuple(dx.shape) == (B * S, H)
    check(_lib.load().r3d_avgpool_rows_bwd(_p(dy), _ld(dy), _p(dx), _ld(dx), B, S, Q, H, _stream()), "r3d_avgpool_rows_bwd")


def embed_gather_fwd(weight, idx, table, S, out):
    """out[r] = weight[idx[r]] + table[r % S] (model/futr_proposed.py:103-106)."""
    rows, H = out.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == rows and weight.is_contiguous()
    check(_lib.load().r3d_embed_gather_fwd(_p(weight), weight.shape[0], _p(idx), _p(table), _ld(table), S, _p(out), _ld(out), rows,
                                           H, _stream()), "r3d_embed_gather_fwd")


def embed_gather_bwd(d_out, idx, d_weight):
    rows, H = d_out.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous() and d_weight.is_contiguous() and d_weight.shape[1] == H
    check(_lib.load().r3d_embed_gather_bwd(_p(d_out), _ld(d_out), _p(idx), _p(d_weight), d_weight.shape[0], rows, H, _stream()),
          "r3d_embed_gather_bwd")


# ----------------------------------------------------------------------------------------------------------
# build-defined three-modality fuser pieces (csrc/fuser3.hip)
# ----------------------------------------------------------------------------------------------------------
def token_exchange3_fwd(xa, xb, xc, mask, x0, *, drop_mask=None, drop_scale=1.0):
    N, Cc = xa.shape
    assert xb.shape == xa.shape == xc.shape and tuple(mask.shape) == (3, Cc) and tuple(x0.shape) == (3 * N, Cc)
    for t in (xa, xb, xc, mask, x0):
        assert t.is_contiguous()
    check(_lib.load().r3d_token_exchange3_fwd(_p(xa), _p(xb), _p(xc), _p(mask), _p(drop_mask), drop_scale, _p(x0), N, Cc, _stream()),
          "r3d_token_exchange3_fwd")


def token_exchange3_bwd(dx0, mask, da, db, dc, *, drop_mask=None, drop_scale=1.0):
    N, Cc = da.shape
    assert tuple(dx0.shape) == (3 * N, Cc) and dx0.is_contiguous() and da.is_contiguous() and db.is_contiguous() and dc.is_contiguous()
    check(_lib.load().r3d_token_exchange3_bwd(_p(dx0), _p(mask), _p(drop_mask), drop_scale, _p(da), _p(db), _p(dc), N, Cc, _stream()),
          "r3d_token_exchange3_bwd")


def attn3_fwd(qkv, probs, out, heads):
    rows, C3 = qkv.shape
    N, Cc = rows // 3, C3 // 3
    assert qkv.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (rows, Cc) and probs.numel() == N * heads * 6
    check(_lib.load().r3d_attn3_fwd(_p(qkv), _p(probs), _p(out), N, Cc, heads, _stream()), "r3d_attn3_fwd")


def attn3_bwd(qkv, d_out, d_qkv, heads):
    rows, C3 = qkv.shape
    N, Cc = rows // 3, C3 // 3
    assert qkv.is_contiguous() and d_out.is_contiguous() and d_qkv.is_contiguous() and d_qkv.shape == qkv.shape
    check(_lib.load().r3d_attn3_bwd(_p(qkv), _p(d_out), _p(d_qkv), N, Cc, heads, _stream()), "r3d_attn3_bwd")


def triple_mean_fwd(y, out):
    N, Cc = out.shape
    assert tuple(y.shape) == (3 * N, Cc) and y.is_contiguous() and out.is_contiguous()
    check(_lib.load().r3d_triple_mean_fwd(_p(y), _p(out), N, Cc, _stream()), "r3d_triple_mean_fwd")


def triple_mean_bwd(d_out, dy):
    N, Cc = d_out.shape
    assert tuple(dy.shape) == (3 * N, Cc) and dy.is_contiguous() and d_out.is_contiguous()
    check(_lib.load().r3d_triple_mean_bwd(_p(d_out), _p(dy), N, Cc, _stream()), "r3d_triple_mean_bwd")
