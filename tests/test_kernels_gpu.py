"""Per-kernel parity of libr3d_hip.so (called through the C ABI via r3d_amd.ops) against plain PyTorch fp32/fp64
CPU references of the same op, and against the oracle where the op is index / integer work.  Needs an MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.helpers import assert_close  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from r3d_amd import ops as o
    return o


def dev(t):
    return t.to("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


# ----------------------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------------------
def _gemm_ref(layout, a, b):
    a, b = a.double(), b.double()
    if layout == 0:
        return a @ b.t()
    if layout == 1:
        return a @ b
    return a.t() @ b


def _mk_ab(layout, M, N, K, pad, seed):
    """operands with leading dimension = cols + pad (pad not multiple of 4 -> scalar-load twin)"""
    sa = (M, K) if layout in (0, 1) else (K, M)
    sb = (N, K) if layout == 0 else (K, N)
    A = rnd(sa[0], sa[1] + pad, seed=seed)
    B = rnd(sb[0], sb[1] + pad, seed=seed + 1)
    return A, B, A[:, :sa[1]], B[:, :sb[1]]


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("tile", [1, 2, 3])
@pytest.mark.parametrize("shape", [(128, 128, 256), (50, 17, 33), (64, 1, 128), (130, 260, 70), (8, 384, 128)])
@pytest.mark.parametrize("pad", [0, 3])
def test_gemm_layouts_tiles(ops, layout, tile, shape, pad):
    M, N, K = shape
    A, B, Av, Bv = _mk_ab(layout, M, N, K, pad, seed=M + N + K + layout)
    Ad, Bd = dev(A), dev(B)
    sa1 = K if layout in (0, 1) else M
    sb1 = K if layout == 0 else N
    Cd = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(layout, Ad[:, :sa1], Bd[:, :sb1], Cd, tile=tile, splitk=1)
    torch.cuda.synchronize()
    assert_close(Cd.cpu(), _gemm_ref(layout, Av, Bv), 1e-4, 1e-4 * math.sqrt(K), f"gemm L{layout} t{tile} {shape}")


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_splitk_and_auto_plan(ops, layout):
    ws = ops.GemmWorkspace("cuda")
    for (M, N, K) in [(128, 128, 2048), (16, 32, 50176), (96, 40, 1000)]:
        A, B, Av, Bv = _mk_ab(layout, M, N, K, 0, seed=K)
        Cd = torch.full((M, N), float("nan"), device="cuda")
        d = ops.gemm(layout, dev(A), dev(B), Cd, ws=ws)            # auto plan
        torch.cuda.synchronize()
        assert_close(Cd.cpu(), _gemm_ref(layout, Av, Bv), 2e-4, 2e-4 * math.sqrt(K), f"auto L{layout} {M,N,K} splitk={d.splitk}")
        Cd.fill_(float("nan"))
        ops.gemm(layout, dev(A), dev(B), Cd, ws=ws, splitk=7, tile=2)
        torch.cuda.synchronize()
        assert_close(Cd.cpu(), _gemm_ref(layout, Av, Bv), 2e-4, 2e-4 * math.sqrt(K), f"splitk7 L{layout} {M,N,K}")


@pytest.mark.parametrize("K,M,N", [(128, 128, 50176), (96, 64, 4100), (128, 128, 2048), (56, 96, 8192)])
def test_gemm_persistent_panel_weight_gradient(ops, K, M, N):
    """tile 6: the few-row weight gradient of a wide layer (A^T in registers, B streamed in 64-column panels); the
    planner picks it for depth_projection.weight's gradient.  Ragged K, M and a last partial panel included."""
    A, B = rnd(K, M, seed=K + M), rnd(K, N, seed=N)
    Cd = torch.full((M, N), float("nan"), device="cuda")
    d = ops.gemm(2, dev(A), dev(B), Cd)
    torch.cuda.synchronize()
    assert d.tile == 6 and d.splitk == 1
    assert_close(Cd.cpu(), A.double().t() @ B.double(), 1e-4, 1e-4 * math.sqrt(K), f"panel wgrad {K, M, N}")
    # strided operands (views of wider matrices) and alpha
    Aw, Bw, Cw = dev(rnd(K, M + 8, seed=1)), dev(rnd(K, N + 12, seed=2)), torch.zeros(M, N + 4, device="cuda")
    ops.gemm(2, Aw[:, :M], Bw[:, :N], Cw[:, :N], alpha=0.5, tile=6)
    torch.cuda.synchronize()
    assert_close(Cw[:, :N].cpu(), 0.5 * (Aw[:, :M].cpu().double().t() @ Bw[:, :N].cpu().double()), 1e-4, 1e-4 * math.sqrt(K),
                 "panel wgrad strided")
    assert float(Cw[:, N:].abs().max()) == 0.0


def test_gemm_epilogue_and_prologue(ops):
    ws = ops.GemmWorkspace("cuda")
    M, N, K = 96, 72, 64
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias, res1, res2, aux = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5), rnd(M, N, seed=6)
    add = rnd(8, K, seed=7)
    mask = (torch.rand(M, N, generator=torch.Generator().manual_seed(8)) > 0.3).to(torch.uint8)
    c0 = rnd(M, N, seed=9)
    for splitk in (1, 2):
        for act in (0, 1, 2):
            for mul in (0, 1, 2):
                Cd = dev(c0.clone())
                pre = torch.empty(M, N, device="cuda")
                ops.gemm(0, dev(A), dev(B), Cd, bias=dev(bias), act=act, pre_out=pre, a_add=dev(add), a_add_mod=8,
                         a_row_xor=1, drop_mask=dev(mask), drop_scale=1.25, aux=dev(aux), mul=mul, res1=dev(res1),
                         res2=dev(res2), alpha=0.5, accumulate=True, ws=ws, splitk=splitk, tile=1)
                torch.cuda.synchronize()
                rows = torch.arange(M)
                Ap = A[rows ^ 1].double() + add[rows % 8].double()
                v = 0.5 * (Ap @ B.double().t()) + bias.double()
                ref_pre = v.clone()
                if act == 1:
                    v = v.relu()
                elif act == 2:
                    v = F.gelu(v)
                v = v * 1.25 * mask.double()
                if mul == 1:
                    v = v * (aux > 0).double()
                elif mul == 2:
                    x = aux.double()
                    v = v * (0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi))
                v = v + res1.double() + res2.double() + c0.double()
                assert_close(pre.cpu(), ref_pre, 1e-4, 1e-4, f"pre_out sk{splitk}")
                assert_close(Cd.cpu(), v, 1e-4, 1e-4, f"epilogue sk{splitk} act{act} mul{mul}")


def test_gemm_c_row_xor_is_pair_swap(ops):
    M, N, K = 64, 48, 32
    A, B, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(M, N, seed=3)
    Cd = torch.empty(M, N, device="cuda")
    ops.gemm(0, dev(A), dev(B), Cd, res1=dev(res), c_row_xor=1, tile=1, splitk=1)
    torch.cuda.synchronize()
    rows = torch.arange(M) ^ 1
    ref = (A.double() @ B.double().t())[rows] + res.double()       # row m of the product lands in row m^1
    assert_close(Cd.cpu(), ref, 1e-4, 1e-4, "c_row_xor")


def test_gemm_tn_fused_bias_grad(ops):
    for tile in (1, 2, 3):
        Kc, M, N = 200, 70, 96                       # dW[M,N] = dY[Kc,M]^T . X[Kc,N];  db[M] = column sums of dY
        dY, X = rnd(Kc, M, seed=1), rnd(Kc, N, seed=2)
        dW, db = torch.empty(M, N, device="cuda"), torch.full((M,), float("nan"), device="cuda")
        ops.gemm(2, dev(dY), dev(X), dW, bias_grad=db, tile=tile)
        torch.cuda.synchronize()
        assert_close(dW.cpu(), dY.double().t() @ X.double(), 1e-4, 1e-3, "dW")
        assert_close(db.cpu(), dY.double().sum(0), 1e-4, 1e-4, "fused bias grad")
        add = rnd(8, N, seed=3)                      # B'[k,:] = X[k,:] + add[k % 8,:]  (with_pos_embed on the wgrad operand)
        ops.gemm(2, dev(dY), dev(X), dW, b_add=dev(add), b_add_mod=8, tile=tile)
        torch.cuda.synchronize()
        Xp = X.double() + add.double()[torch.arange(Kc) % 8]
        assert_close(dW.cpu(), dY.double().t() @ Xp, 1e-4, 1e-3, "dW with b_add")


def test_gemm_grouped_matches_individual(ops):
    shapes = [(200, 70, 96), (64, 18, 128), (256, 512, 128), (33, 40, 50)]          # (Kc, M, N) of TN problems
    probs, refs = [], []
    for i, (Kc, M, N) in enumerate(shapes):
        dY, X = rnd(Kc, M, seed=10 + i), rnd(Kc, N, seed=20 + i)
        add = rnd(8, N, seed=30 + i) if i % 2 == 0 else None
        pr = dict(a=dev(dY), b=dev(X), c=torch.full((M, N), float("nan"), device="cuda"),
                  bias_grad=torch.full((M,), float("nan"), device="cuda") if i != 3 else None,
                  b_add=dev(add) if add is not None else None, b_add_mod=8)
        probs.append(pr)
        Xp = X.double() + (add.double()[torch.arange(Kc) % 8] if add is not None else 0)
        refs.append((dY.double().t() @ Xp, dY.double().sum(0)))
    for tile in (1, 2):
        for pr in probs:
            pr["c"].fill_(float("nan"))
        grp = ops.GemmGroup(2, probs, tile=tile)
        grp.launch()
        torch.cuda.synchronize()
        for pr, (rw, rb) in zip(probs, refs):
            assert_close(pr["c"].cpu(), rw, 1e-4, 1e-3, f"grouped dW tile{tile}")
            if pr["bias_grad"] is not None:
                assert_close(pr["bias_grad"].cpu(), rb, 1e-4, 1e-4, "grouped db")


@pytest.mark.parametrize("nprob", [1, 32, 33, 45])
def test_gemm_grouped_problem_lookup(ops, nprob):
    """A workgroup finds its problem from the prefix table in the kernel arguments (<= 32 problems) or by the search in
    device memory (more): every problem of groups on both sides of the limit, ragged tile counts."""
    probs, refs = [], []
    for i in range(nprob):
        M, N, K = 20 + 13 * (i % 5), 30 + 17 * (i % 4), 24 + 8 * (i % 3)
        a, b = rnd(M, K, seed=100 + i), rnd(N, K, seed=300 + i)
        probs.append(dict(a=dev(a), b=dev(b), c=torch.full((M, N), float("nan"), device="cuda")))
        refs.append(a.double() @ b.double().t())
    grp = ops.GemmGroup(0, probs, tile=1)                       # NT
    grp.launch()
    torch.cuda.synchronize()
    for i, (pr, r) in enumerate(zip(probs, refs)):
        assert_close(pr["c"].cpu(), r, 1e-4, 1e-3, f"problem {i} of {nprob}")


def test_adamw_flat_is_bitwise_independent_of_how_the_range_is_cut(ops):
    """One launch over [0, n) and two launches over [0, k) + [k, n) must give the same bits for every element (the
    grid-stride loop's main body and tail used to be contracted into different FMA patterns)."""
    n, k = 4 * 1234567, 4 * 246793
    g = torch.Generator().manual_seed(3)
    p0, gr = torch.randn(n, generator=g).cuda(), (torch.randn(n, generator=g) * 1e-2).cuda()
    m0, v0 = (torch.randn(n, generator=g) * 1e-3).cuda(), (torch.rand(n, generator=g) * 1e-4).cuda()
    lr, st = torch.tensor([1e-3], device="cuda"), torch.tensor([7], dtype=torch.int64, device="cuda")
    a = [p0.clone(), m0.clone(), v0.clone()]
    ops.adamw_flat(a[0], gr, a[1], a[2], lr, st, weight_decay=5e-3)
    b = [p0.clone(), m0.clone(), v0.clone()]
    ops.adamw_flat(b[0][:k], gr[:k], b[1][:k], b[2][:k], lr, st, weight_decay=5e-3)
    ops.adamw_flat(b[0][k:], gr[k:], b[1][k:], b[2][k:], lr, st, weight_decay=5e-3)
    torch.cuda.synchronize()
    for x, y, name in zip(a, b, ("p", "m", "v")):
        assert torch.equal(x, y), name


def test_layernorm_deferred_and_batched_finalize(ops):
    jobs, refs = [], []
    for i, (rows, H) in enumerate([(256, 128), (64, 128), (3, 128)]):
        x, g, b, dy = rnd(rows, H, seed=i), 1 + 0.2 * rnd(H, seed=i + 5), 0.1 * rnd(H, seed=i + 9), rnd(rows, H, seed=i + 13)
        xr = x.double().requires_grad_(True)
        gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
        F.layer_norm(xr, (H,), gr, br, 1e-5).backward(dy.double())
        yd, mean, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        ops.layernorm_fwd(dev(x), dev(g), dev(b), yd, mean, rstd)
        part = torch.empty(max(ops.layernorm_bwd_ws_floats(rows, H), 4), device="cuda")
        dx, dg, db = torch.empty(rows, H, device="cuda"), torch.full((H,), float("nan"), device="cuda"), torch.full((H,), float("nan"), device="cuda")
        ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dev(b), dx, dg, db, partial=part)
        jobs.append((part, rows, H, dg, db))
        refs.append((xr.grad, gr.grad, br.grad, dx))
    ops.LnFinalizeGroup(jobs).launch()
    torch.cuda.synchronize()
    for (part, rows, H, dg, db), (rx, rg, rb, dx) in zip(jobs, refs):
        assert_close(dx.cpu(), rx, 1e-3, 1e-4, "dx")
        assert_close(dg.cpu(), rg, 1e-3, 1e-3, f"batched dgamma rows={rows}")
        assert_close(db.cpu(), rb, 1e-3, 1e-3, f"batched dbeta rows={rows}")


@pytest.mark.parametrize("M,K", [(256, 128), (256, 512), (64, 128), (16, 256), (48, 384)])
@pytest.mark.parametrize("full", [False, True])
def test_gemm_ln_fwd_is_linear_dropout_residual_layernorm(ops, M, K, full):
    """r3d_gemm_ln_fwd (row-complete GEMM with the LayerNorm as its epilogue) against fp64 nn.Linear -> mask -> residuals
    -> F.layer_norm -> pair mean, alone and with a plain-LayerNorm job (K = 0) riding in the same launch."""
    H = 128
    a, w, bias = rnd(M, K + 4, seed=M + K)[:, :K], rnd(H, K, seed=K + 1, scale=K ** -0.5), 0.1 * rnd(H, seed=3)
    g, b = 1 + 0.2 * rnd(H, seed=4), 0.1 * rnd(H, seed=5)
    r1, r2 = rnd(M, H, seed=6), rnd(M, H, seed=7)
    mask = (torch.rand(M, H, generator=torch.Generator().manual_seed(8)) > 0.1).to(torch.uint8)
    dsc = 1.0 / 0.9
    pre = a.double() @ w.double().t() + bias.double()
    if full:
        pre = pre * dsc * mask.double() + r1.double() + r2.double()
    y = F.layer_norm(pre, (H,), g.double(), b.double(), 1e-5)
    x2 = rnd(32, H, seed=9)
    y2 = F.layer_norm(x2.double(), (H,), g.double(), b.double(), 1e-5)
    f = lambda *sh: torch.full(sh, float("nan"), device="cuda")      # noqa: E731
    pre_d, yd, mean, rstd, pair = f(M, H), f(M, H), f(M), f(M), f(M // 2, H)
    x2d, y2d, m2, s2 = dev(x2), f(32, H), f(32), f(32)
    ad = dev(rnd(M, K + 4, seed=M + K))[:, :K]
    job = dict(a=ad, w=dev(w), bias=dev(bias), pre=pre_d, gamma=dev(g), beta=dev(b), y=yd, mean=mean, rstd=rstd, pair_out=pair)
    if full:
        job.update(drop_mask=dev(mask), drop_scale=dsc, res1=dev(r1), res2=dev(r2))
    assert ops.gemm_ln_supported(M, K, H) and not ops.gemm_ln_supported(M + 1, K, H) and not ops.gemm_ln_supported(M, K, 256)
    ops.gemm_ln_fwd([job, dict(a=None, pre=x2d, gamma=dev(g), beta=dev(b), y=y2d, mean=m2, rstd=s2)])
    torch.cuda.synchronize()
    assert_close(pre_d.cpu(), pre, 1e-5, 1e-5, "pre-norm rows")
    assert_close(yd.cpu(), y, 1e-4, 1e-5, "y")
    assert_close(mean.cpu(), pre.mean(1), 1e-5, 1e-5, "mean")
    assert_close(rstd.cpu(), 1.0 / torch.sqrt(pre.var(1, unbiased=False) + 1e-5), 1e-4, 1e-5, "rstd")
    assert_close(pair.cpu(), y.view(M // 2, 2, H).mean(1), 1e-4, 1e-5, "pair mean")
    assert_close(y2d.cpu(), y2, 1e-4, 1e-5, "plain LayerNorm job")
    assert torch.equal(x2d.cpu(), x2), "a K = 0 job must leave its rows untouched"
    # the two-launch path writes the same buffers
    pre_t, y_t, m_t, s_t = f(M, H), f(M, H), f(M), f(M)
    kw = dict(drop_mask=dev(mask), drop_scale=dsc, res1=dev(r1), res2=dev(r2)) if full else {}
    ops.gemm(0, ad, dev(w), pre_t, bias=dev(bias), ws=ops.GemmWorkspace("cuda"), **kw)
    ops.layernorm_fwd(pre_t, dev(g), dev(b), y_t, m_t, s_t)
    torch.cuda.synchronize()
    assert_close(yd.cpu(), y_t.cpu().double(), 2e-5, 2e-5, "fused vs gemm + layernorm")


def test_add_rowbcast(ops):
    x, add = rnd(24, 40, seed=1), rnd(8, 40, seed=2)
    out = torch.empty(24, 40, device="cuda")
    ops.add_rowbcast(dev(x), dev(add), 8, out)
    out2 = torch.empty(24, 40, device="cuda")
    ops.add_rowbcast(None, dev(add), 8, out2)
    torch.cuda.synchronize()
    assert_close(out.cpu(), x + add.repeat(3, 1), 1e-7, 1e-7, "add_rowbcast")
    assert_close(out2.cpu(), add.repeat(3, 1), 0, 0, "bcast")


def test_gemm_rejects_bad_arguments(ops):
    from r3d_amd._lib import R3DHipError
    a, b, c = torch.zeros(4, 8, device="cuda"), torch.zeros(4, 8, device="cuda"), torch.zeros(4, 4, device="cuda")
    with pytest.raises(R3DHipError):
        ops.gemm(0, a, b, c, mul=1)                      # mul without aux
    with pytest.raises(R3DHipError):
        ops.gemm(0, torch.zeros(3, 8, device="cuda"), b, torch.zeros(3, 4, device="cuda"), a_row_xor=1)  # odd M


# ----------------------------------------------------------------------------------------------------------
# LayerNorm / reductions
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,H", [(7, 32), (128, 128), (256, 512), (70, 1024), (3, 1536)])
@pytest.mark.parametrize("relu", [False, True])
def test_layernorm_fwd_bwd(ops, rows, H, relu):
    ws = ops.GemmWorkspace("cuda")
    x, g, b = rnd(rows, H, seed=1), 1 + 0.2 * rnd(H, seed=2), 0.1 * rnd(H, seed=3)
    dy, add1 = rnd(rows, H, seed=4), rnd(rows, H, seed=5)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.layer_norm(xr, (H,), gr, br, 1e-5)
    if relu:
        y = y.relu()
    y.backward(dy.double())
    yd, mean, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(dev(x), dev(g), dev(b), yd, mean, rstd, relu=relu)
    dx, dg, db = torch.empty(rows, H, device="cuda"), torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
    ops.layernorm_bwd(dev(0.25 * dy), dev(x), mean, rstd, dev(g), dev(b), dx, dg, db, relu=relu, dy2=dev(0.75 * dy),
                      add1=dev(add1), ws=ws)
    torch.cuda.synchronize()
    assert_close(yd.cpu(), y.detach(), 1e-4, 1e-5, "ln y")
    assert_close(dx.cpu(), xr.grad + add1.double(), 1e-3, 1e-4, "ln dx")
    assert_close(dg.cpu(), gr.grad, 1e-3, 1e-3, "ln dgamma")
    assert_close(db.cpu(), br.grad, 1e-3, 1e-3, "ln dbeta")


def test_layernorm_pair_mean_and_splitk_input(ops):
    ws = ops.GemmWorkspace("cuda")
    rows, H, ns = 64, 128, 5
    part, bias = rnd(ns, rows, H, seed=1), rnd(H, seed=2)
    g, b = 1 + 0.2 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    pre = part.double().sum(0) + bias.double()
    y = F.layer_norm(pre, (H,), g.double(), b.double(), 1e-5).relu()
    yd, mean, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    pre_d, pair = torch.empty(rows, H, device="cuda"), torch.empty(rows // 2, H, device="cuda")
    ops.layernorm_fwd(dev(part), dev(g), dev(b), yd, mean, rstd, relu=True, pair_out=pair, nsplit=ns, bias=dev(bias),
                      pre_out=pre_d, rows=rows, H=H)
    torch.cuda.synchronize()
    assert_close(pre_d.cpu(), pre, 1e-5, 1e-5, "pre")
    assert_close(yd.cpu(), y, 1e-4, 1e-5, "y")
    assert_close(pair.cpu(), y.view(rows // 2, 2, H).mean(1), 1e-4, 1e-5, "pair mean")
    # pair_in backward
    x = rnd(rows, H, seed=5)
    dyp = rnd(rows // 2, H, seed=6)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    F.layer_norm(xr, (H,), gr, br, 1e-5).view(rows // 2, 2, H).mean(1).backward(dyp.double())
    ops.layernorm_fwd(dev(x), dev(g), dev(b), yd, mean, rstd)
    dx, dg, db = torch.empty(rows, H, device="cuda"), torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
    ops.layernorm_bwd(dev(dyp), dev(x), mean, rstd, dev(g), dev(b), dx, dg, db, pair_in=True, ws=ws)
    torch.cuda.synchronize()
    assert_close(dx.cpu(), xr.grad, 1e-3, 1e-5, "pair dx")
    assert_close(dg.cpu(), gr.grad, 1e-3, 1e-4, "pair dgamma")
    assert_close(db.cpu(), br.grad, 1e-3, 1e-4, "pair dbeta")


@pytest.mark.parametrize("rows,cols", [(5, 17), (128, 128), (1000, 513)])
def test_colsum_rowmod(ops, rows, cols):
    ws = ops.GemmWorkspace("cuda")
    x = rnd(rows, cols, seed=rows)
    out = dev(torch.ones(cols))
    ops.colsum(dev(x), out, accumulate=True, ws=ws)
    mod = 5
    o2 = torch.zeros(mod, cols, device="cuda")
    ops.rowmod_sum(dev(x), mod, o2)
    torch.cuda.synchronize()
    assert_close(out.cpu(), x.double().sum(0) + 1, 1e-4, 1e-3, "colsum")
    ref = torch.stack([x[r::mod].double().sum(0) for r in range(mod)])
    assert_close(o2.cpu(), ref, 1e-4, 1e-3, "rowmod")


# ----------------------------------------------------------------------------------------------------------
# token selection (bit-exact vs the oracle / torch.topk) and exchange
# ----------------------------------------------------------------------------------------------------------
def test_token_select_matches_cpu_topk(ops, oracle_lib):
    from oracle import futr_oracle as O
    rng = np.random.default_rng(0)
    for Cc in (32, 128, 512, 1024):
        k = Cc // 4
        vecs = [np.full(Cc, 1.0 / (8 * 16 * Cc), np.float32), rng.random(Cc).astype(np.float32),
                rng.integers(0, 3, Cc).astype(np.float32), rng.integers(0, Cc // 2 + 1, Cc).astype(np.float32)]
        s = torch.from_numpy(np.stack(vecs))
        idx = torch.empty(len(vecs), k, dtype=torch.int64, device="cuda")
        mask = torch.empty(len(vecs), Cc, device="cuda")
        used = torch.zeros(len(vecs), dtype=torch.int32, device="cuda")
        ops.token_select(k, idx, mask, score_f=dev(s), used_serial=used)
        torch.cuda.synchronize()
        for v in range(len(vecs)):
            want = O.select_smallest(vecs[v], k)
            ref = np.sort(torch.topk(torch.from_numpy(vecs[v]).view(1, 1, Cc), k, dim=-1, largest=False)[1].view(-1).numpy())
            assert np.array_equal(want, ref)
            assert np.array_equal(idx[v].cpu().numpy(), want), (Cc, v)
            m = np.zeros(Cc, np.float32)
            m[want] = 1
            assert np.array_equal(mask[v].cpu().numpy(), m)
        assert used.cpu().tolist()[0] == 1 and used.cpu().tolist()[1] == 0      # all-equal -> tie path; distinct -> parallel


def test_colabssum_and_select_from_sums(ops, oracle_lib):
    from oracle import futr_oracle as O
    rows, Cc = 176, 128
    x = rnd(2, rows, Cc, seed=3)
    sums = torch.empty(2, Cc, dtype=torch.float64, device="cuda")
    ops.colabssum(dev(x[0]), sums[0])
    ops.colabssum(dev(x[1]), sums[1])
    idx = torch.empty(2, Cc // 4, dtype=torch.int64, device="cuda")
    mask = torch.empty(2, Cc, device="cuda")
    ops.token_select(Cc // 4, idx, mask, score_sum=sums, count=float(rows))
    torch.cuda.synchronize()
    assert_close(sums.cpu(), x.double().abs().sum(1), 1e-12, 1e-9, "abs sums")
    for v in range(2):
        want = O.select_smallest(x[v].abs().mean(0).numpy(), Cc // 4)
        assert np.array_equal(idx[v].cpu().numpy(), want)


def test_token_exchange_fwd_bwd(ops):
    N, H = 37, 64
    rgb, dep = rnd(N, H, seed=1).relu(), rnd(N, H, seed=2).relu()
    g = torch.Generator().manual_seed(3)
    mr = torch.zeros(H)
    md = torch.zeros(H)
    mr[torch.randperm(H, generator=g)[:H // 4]] = 1
    md[torch.randperm(H, generator=g)[:H // 4]] = 1
    keep = (torch.rand(2 * N, H, generator=g) > 0.1).to(torch.uint8)
    x0 = torch.empty(2 * N, H, device="cuda")
    ops.token_exchange_fwd(dev(rgb), dev(dep), dev(mr), dev(md), x0, drop_mask=dev(keep), drop_scale=1 / 0.9)
    r, d = rgb.double().requires_grad_(True), dep.double().requires_grad_(True)
    ex_r = torch.where(mr.bool(), d, r)
    ex_d = torch.where(md.bool(), r, d)
    st = torch.stack([ex_r, ex_d], 1).reshape(2 * N, H) * keep.double() / 0.9
    dx0 = rnd(2 * N, H, seed=4)
    st.backward(dx0.double())
    drp, ddp = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
    ops.token_exchange_bwd(dev(dx0), dev(rgb), dev(mr), dev(md), drp, ddp, drop_mask=dev(keep), drop_scale=1 / 0.9)
    torch.cuda.synchronize()
    assert_close(x0.cpu(), st.detach(), 1e-6, 1e-6, "exchange fwd")
    assert_close(drp.cpu(), r.grad * (rgb > 0).double(), 1e-6, 1e-6, "d_rgb_pre")
    assert_close(ddp.cpu(), d.grad, 1e-6, 1e-6, "d_dep")


# ----------------------------------------------------------------------------------------------------------
# attention core
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,heads,Lq,Lk,dh,masked,drop", [(3, 8, 8, 8, 16, False, False), (2, 8, 8, 21, 16, True, False),
                                                           (2, 4, 8, 130, 64, True, True), (1, 8, 8, 700, 4, False, False),
                                                           (2, 8, 8, 16, 128, True, True), (3, 8, 8, 8, 128, False, False),
                                                           (2, 8, 8, 64, 64, True, True), (2, 8, 8, 32, 32, False, True)])
def test_mha_core(ops, B, heads, Lq, Lk, dh, masked, drop):
    H = heads * dh
    q, kv = rnd(B * Lq, H, seed=1), rnd(B * Lk, 2 * H, seed=2)
    kpm = torch.zeros(B, Lk, dtype=torch.uint8)
    if masked:
        kpm[:, Lk - 3:] = 1
        kpm[0, 1] = 1
    g = torch.Generator().manual_seed(5)
    keep = (torch.rand(B, heads, Lq, Lk, generator=g) > 0.1).to(torch.uint8) if drop else None
    dsc = 1 / 0.9 if drop else 1.0
    d_o = rnd(B * Lq, H, seed=3)
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    qh = qr.view(B, Lq, heads, dh).transpose(1, 2)
    kh = kvr[:, :H].reshape(B, Lk, heads, dh).transpose(1, 2)
    vh = kvr[:, H:].reshape(B, Lk, heads, dh).transpose(1, 2)
    att = (qh @ kh.transpose(-2, -1)) / math.sqrt(dh)
    att = att.masked_fill(kpm.bool()[:, None, None, :], float("-inf")).softmax(-1)
    attd = att * keep.double() * dsc if drop else att
    o = (attd @ vh).transpose(1, 2).reshape(B * Lq, H)
    o.backward(d_o.double())
    qd, kvd = dev(q), dev(kv)
    probs = torch.empty(B, heads, Lq, Lk, device="cuda")
    od = torch.empty(B * Lq, H, device="cuda")
    kd = dev(keep) if drop else None
    ops.mha_core_fwd(qd, kvd[:, :H], kvd[:, H:], probs, od, B, heads, Lq, Lk, dh, kpm=dev(kpm) if masked else None,
                     drop_mask=kd, drop_scale=dsc)
    dq = torch.empty(B * Lq, H, device="cuda")
    dkv = torch.empty(B * Lk, 2 * H, device="cuda")
    ops.mha_core_bwd(qd, kvd[:, :H], kvd[:, H:], probs, dev(d_o), dq, dkv[:, :H], dkv[:, H:], B, heads, Lq, Lk, dh,
                     drop_mask=kd, drop_scale=dsc)
    torch.cuda.synchronize()
    assert_close(probs.cpu(), att.detach(), 1e-4, 1e-6, "probs")
    assert_close(od.cpu(), o.detach(), 1e-4, 1e-5, "attn out")
    assert_close(dq.cpu(), qr.grad, 1e-3, 1e-5, "dq")
    assert_close(dkv.cpu(), kvr.grad, 1e-3, 1e-5, "dkv")


# ----------------------------------------------------------------------------------------------------------
# losses / AdamW / dropout masks
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,S,Q,K", [(8, 16, 8, 17), (3, 7, 8, 122), (1, 40, 8, 17)])
def test_losses_match_oracle(ops, B, S, Q, K):
    from oracle import futr_oracle as O, synth
    pad = K + 1
    _, _, lab, dur, tgt = [torch.from_numpy(x) for x in synth.make_batch(B, S, K, pad, 7, depth_hw=(2, 2))]
    seg, act, du = rnd(B, S, K, seed=1), rnd(B, Q, K, seed=2), 0.5 * rnd(B, Q, seed=3)
    segr, actr, dur_r = seg.clone().requires_grad_(True), act.clone().requires_grad_(True), du.clone().requires_grad_(True)
    res = O.losses(dict(seg=segr, action=actr, duration=dur_r), lab, dur, tgt, pad)
    res["loss"].backward()
    comb = torch.cat([act.view(B * Q, K), du.view(B * Q, 1)], 1).contiguous()     # [BQ, K+1] as the engine lays it out
    combd = dev(comb)
    dcomb = torch.zeros_like(combd)
    dseg = torch.empty(B * S, K, device="cuda")
    loss, counts = torch.empty(4, device="cuda"), torch.empty(4, dtype=torch.int64, device="cuda")
    ops.losses_fwd_bwd(dev(seg.view(B * S, K)), combd[:, :K], combd[:, K:], K + 1, dev(lab), dev(tgt), dev(dur), B, S, Q, K, pad,
                       47, loss, counts, d_seg=dseg, d_act=dcomb[:, :K], d_dur=dcomb[:, K:], ld_ddur=K + 1)
    torch.cuda.synchronize()
    want = [float(res[k]) for k in ("loss_seg", "loss_action", "loss_dur", "loss")]
    assert_close(loss.cpu(), want, 1e-5, 1e-6, "losses")
    assert counts.cpu().tolist() == [res[k] for k in ("seg_correct", "seg_total", "act_correct", "act_total")]
    assert_close(dseg.cpu().view(B, S, K), segr.grad, 1e-4, 1e-7, "d seg")
    assert_close(dcomb[:, :K].cpu().reshape(B, Q, K), actr.grad, 1e-4, 1e-7, "d act")
    assert_close(dcomb[:, K].cpu().reshape(B, Q), dur_r.grad, 1e-4, 1e-7, "d dur")


def test_adamw_flat_matches_oracle(ops):
    from oracle import futr_oracle as O
    n = 4096 + 64
    p, g = rnd(n, seed=1), rnd(n, seed=2) * 1e-2
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = dev(p.clone()), dev(m.clone()), dev(v.clone())
    lr_t = torch.tensor([1e-3], device="cuda")
    step_t = torch.zeros(1, dtype=torch.int64, device="cuda")
    for t in range(1, 4):
        gt = g * t
        O.adamw_step(p, gt, m, v, t, 1e-3, 5e-3)
        step_t.add_(1)
        ops.adamw_flat(pd, dev(gt * 2.0), md, vd, lr_t, step_t, weight_decay=5e-3, grad_scale=0.5)
    torch.cuda.synchronize()
    assert_close(pd.cpu(), p, 1e-5, 1e-6, "adamw p")
    assert_close(md.cpu(), m, 1e-5, 1e-8, "adamw m")
    assert_close(vd.cpu(), v, 1e-5, 1e-10, "adamw v")


def test_dropout_mask_statistics(ops):
    n = 1 << 20
    mk = torch.empty(n, dtype=torch.uint8, device="cuda")
    off = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.dropout_mask(mk, 0.1, 1234, off)
    a = mk.cpu().float()
    off.add_(1)
    ops.dropout_mask(mk, 0.1, 1234, off)
    b = mk.cpu().float()
    assert abs(float(a.mean()) - 0.9) < 2e-3 and abs(float(b.mean()) - 0.9) < 2e-3
    assert float((a != b).float().mean()) > 0.1          # a new offset gives a new mask
    assert set(a.unique().tolist()) <= {0.0, 1.0}


# ----------------------------------------------------------------------------------------------------------
# bf16x3 matrix-core path (r3d_gemm_desc::prec = 1)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,M,N", [(128, 128, 50176), (128, 128, 19200), (100, 96, 4100), (37, 128, 2048), (128, 40, 6272)])
def test_gemm_bf16x3_weight_gradient_panels(ops, K, M, N):
    """TN weight-gradient panels on the bf16 matrix cores: every operand split exactly into three bf16 terms, six products.
    Against an fp64 product: the error must be of fp32-rounding size (the fp32 MFMA path's own error is measured beside
    it), far inside the 1e-3 budget of BASELINE.json."""
    from r3d_amd._lib import GEMM_TN
    a = rnd(K, M, seed=K + M) * 0.05
    b = torch.rand(K, N, generator=torch.Generator().manual_seed(N))          # depth-like inputs in [0, 1)
    want = a.double().t() @ b.double()
    scale = float(want.abs().max())
    c1, c0 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ws = ops.GemmWorkspace("cuda")
    d1 = ops.gemm(GEMM_TN, dev(a), dev(b), c1, ws=ws, prec=1)
    d0 = ops.gemm(GEMM_TN, dev(a), dev(b), c0, ws=ws, prec=0)
    torch.cuda.synchronize()
    assert d1.tile == 7 and d0.tile == 6
    e1 = float((c1.cpu().double() - want).abs().max()) / scale
    e0 = float((c0.cpu().double() - want).abs().max()) / scale
    assert e1 < 2e-6 and e1 < 4 * e0 + 2e-7, (e1, e0)
    # alpha and a strided output
    big = torch.zeros(M, N + 8, device="cuda")
    ops.gemm(GEMM_TN, dev(a), dev(b), big[:, :N], ws=ws, prec=1, alpha=0.5)
    torch.cuda.synchronize()
    assert float((big[:, :N].cpu().double() - 0.5 * want).abs().max()) / scale < 2e-6
    assert float(big[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("tile", [0, 10])
@pytest.mark.parametrize("K,M,N", [(512, 512, 50176), (256, 128, 19200), (2048, 1024, 4096), (100, 384, 4100), (130, 132, 2052)])
def test_gemm_bf16x3_weight_gradient_tiled(ops, K, M, N, tile):
    """TN product beyond the panel kernel's 128 x 128 limit (tile 10), with alpha and accumulate, against fp64."""
    from r3d_amd._lib import GEMM_TN
    a = rnd(K, M, seed=K + M) * 0.05
    b = torch.rand(K, N, generator=torch.Generator().manual_seed(N))
    c_init = rnd(M, N, seed=9)
    want = 0.5 * (a.double().t() @ b.double()) + c_init.double()
    scale = float(want.abs().max())
    c1, c0 = dev(c_init.clone()), dev(c_init.clone())
    ws = ops.GemmWorkspace("cuda")
    d1 = ops.gemm(GEMM_TN, dev(a), dev(b), c1, ws=ws, prec=1, alpha=0.5, accumulate=True, tile=tile)   # 0: the planner's (12)
    d0 = ops.gemm(GEMM_TN, dev(a), dev(b), c0, ws=ws, prec=0, alpha=0.5, accumulate=True)
    torch.cuda.synchronize()
    assert d1.tile in (10, 12) and d0.tile <= 5, (d1.tile, d0.tile)
    e1 = float((c1.cpu().double() - want).abs().max()) / scale
    e0 = float((c0.cpu().double() - want).abs().max()) / scale
    assert e1 < 3e-6 and e1 < 4 * e0 + 3e-7, (e1, e0)


@pytest.mark.parametrize("K,M,N", [(256, 256, 8192), (512, 512, 6272), (130, 132, 2052)])
def test_gemm_bf16x3_weight_gradient_tiled_with_adamw_epilogue(ops, K, M, N):
    """Tile 10 with r3d_gemm_desc::adam_*: the parameter and both moments are updated where the gradient tile is finished (the
    gradient is never stored) -- against the same kernel's stored gradient followed by the flat AdamW launch, two steps."""
    from r3d_amd._lib import GEMM_TN
    a = dev(rnd(K, M, seed=K + M) * 0.05)
    b = dev(torch.rand(K, N, generator=torch.Generator().manual_seed(N)))
    p0 = rnd(M, N, seed=5) * 0.1
    lr_t = torch.full((1,), 1e-3, dtype=torch.float32, device="cuda")
    step_t = torch.zeros(1, dtype=torch.int64, device="cuda")
    ws = ops.GemmWorkspace("cuda")
    pf, mf, vf = dev(p0.clone()), torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    pr, mr, vr = dev(p0.clone()), torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    g = torch.empty(M, N, device="cuda")
    for step in range(2):
        ops.tick(step_t, None)
        d1 = ops.gemm(GEMM_TN, a, b, pf, ws=ws, prec=1, alpha=0.5,
                      adam=dict(m=mf, v=vf, lr_t=lr_t, step_t=step_t, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-3,
                                grad_scale=0.25))
        d0 = ops.gemm(GEMM_TN, a, b, g, ws=ws, prec=1, alpha=0.5)
        ops.adamw_flat(pr.view(-1), g.view(-1), mr.view(-1), vr.view(-1), lr_t, step_t, weight_decay=5e-3, grad_scale=0.25)
        torch.cuda.synchronize()
        assert d1.tile in (10, 12) and d0.tile in (10, 12)
        assert_close(mf.cpu(), mr.cpu(), 1e-6, 1e-9, f"step {step} exp_avg")
        assert_close(vf.cpu(), vr.cpu(), 1e-6, 1e-12, f"step {step} exp_avg_sq")
        assert_close(pf.cpu(), pr.cpu(), 1e-6, 1e-7, f"step {step} parameter")


@pytest.mark.parametrize("M,N,K", [(128, 128, 50176), (128, 128, 19200), (512, 512, 50176), (100, 96, 8200), (256, 1024, 16384),
                                   (8, 40, 8192)])
def test_gemm_bf16x3_long_k_projection(ops, M, N, K):
    """NT split-K product on the bf16 matrix cores (tiles 8 / 9: the forward depth projection) with the reducer's epilogue
    (bias + ReLU) against an fp64 product; the fp32 MFMA path's own error is measured beside it."""
    from r3d_amd._lib import GEMM_NT
    a = torch.rand(M, K, generator=torch.Generator().manual_seed(M))             # depth-like inputs in [0, 1)
    b = rnd(N, K, seed=N) * (3.0 / K) ** 0.5
    bias = rnd(N, seed=3) * 0.1
    want = torch.relu(a.double() @ b.double().t() + bias.double())
    scale = float(want.abs().max())
    c1, c0 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ws = ops.GemmWorkspace("cuda")
    d1 = ops.gemm(GEMM_NT, dev(a), dev(b), c1, ws=ws, prec=1, bias=dev(bias), act=1)
    d0 = ops.gemm(GEMM_NT, dev(a), dev(b), c0, ws=ws, prec=0, bias=dev(bias), act=1)
    torch.cuda.synchronize()
    assert d1.tile in (8, 9) and d0.tile <= 5, (d1.tile, d0.tile)
    e1 = float((c1.cpu().double() - want).abs().max()) / scale
    e0 = float((c0.cpu().double() - want).abs().max()) / scale
    assert e1 < 3e-6 and e1 < 4 * e0 + 3e-7, (e1, e0)


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 50176, 196), (128, 128, 50176, 224), (100, 96, 8200, 9), (512, 512, 12544, 8),
                                          (130, 257, 8192, 5)])
def test_gemm_bf16x3_uniform_wave_tile(ops, M, N, K, splits):
    """Tile 11 (gemm_bf3_nt_u_kernel: every wave splits its share of the operand stream between its own MFMAs; kept for
    measurement, the planner does not select it) against an fp64 product, whole and ragged tiles, K tails."""
    from r3d_amd._lib import GEMM_NT
    a = torch.rand(M, K, generator=torch.Generator().manual_seed(M))
    b = rnd(N, K, seed=N) * (3.0 / K) ** 0.5
    want = a.double() @ b.double().t()
    scale = float(want.abs().max())
    c1 = torch.empty(M, N, device="cuda")
    ws = ops.GemmWorkspace("cuda")
    d1 = ops.gemm(GEMM_NT, dev(a), dev(b), c1, ws=ws, prec=1, tile=11, splitk=splits)
    torch.cuda.synchronize()
    assert d1.tile == 11 and d1.splitk > 1
    e1 = float((c1.cpu().double() - want).abs().max()) / scale
    assert e1 < 3e-6, e1


# ----------------------------------------------------------------------------------------------------------
# effective rank
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,Cc", [(128, 128), (40, 24), (256, 128), (33, 64), (64, 512), (1024, 32), (16, 16), (127, 129),
                                  (64, 64), (64, 32), (128, 32), (512, 64), (192, 128)])   # level order: 8 / 16 lanes, C = 32..128
def test_erank_jacobi_vs_svdvals(ops, R, Cc):
    from oracle import futr_oracle as O
    x = rnd(R, Cc, seed=R) @ torch.diag(torch.linspace(0.05, 2.0, Cc)) + 0.3
    sig, st = torch.empty(1, Cc, device="cuda"), torch.empty(1, 4, device="cuda")
    aft = torch.empty(1, Cc, R, device="cuda")
    ops.erank_jacobi(dev(x), sig, st, af_t=aft)
    torch.cuda.synchronize()
    sv = torch.linalg.svdvals(x.double())
    assert_close(torch.sort(sig[0].cpu(), descending=True)[0][:min(R, Cc)], sv, 1e-4, 1e-4 * float(sv[0]), "sigma")
    assert abs(float(st[0, 0]) - O.effective_rank(x)) < 5e-3, (float(st[0, 0]), O.effective_rank(x))
    assert float(st[0, 3]) < 30
    # rotated columns are orthogonal with norms sigma
    G = aft[0].cpu().double() @ aft[0].cpu().double().t()
    off = G - torch.diag(torch.diag(G))
    assert float(off.abs().max()) < 1e-3 * float(sv[0]) ** 2


def test_erank_jacobi_level_order_batched_and_rank_deficient(ops):
    """The level-order kernel (power-of-two C <= 128) on a batch of matrices, one of them rank-deficient (R < C: the surplus
    columns must report sigma = 0, not rotation noise) and one with two exactly equal columns."""
    from oracle import futr_oracle as O
    R, Cc = 64, 128
    xs = [rnd(R, Cc, seed=5), rnd(R, Cc, seed=6) @ torch.diag(torch.linspace(1e-3, 1.0, Cc)), rnd(R, Cc, seed=7)]
    xs[2][:, 17] = xs[2][:, 3]
    x = torch.stack(xs)
    sig, st = torch.empty(3, Cc, device="cuda"), torch.empty(3, 4, device="cuda")
    ops.erank_jacobi(dev(x), sig, st)
    torch.cuda.synchronize()
    for b in range(3):
        sv = torch.linalg.svdvals(x[b].double())
        got = torch.sort(sig[b].cpu(), descending=True)[0]
        assert_close(got[:R], sv, 1e-4, 1e-4 * float(sv[0]), f"sigma[{b}]")
        assert float(got[R:].abs().max()) <= 1e-5 * float(sv[0]), f"matrix {b}: surplus columns"
        assert abs(float(st[b, 0]) - O.effective_rank(x[b])) < 5e-3


@pytest.mark.parametrize("R,Cc", [(128, 128), (96, 64), (70, 30)])
def test_erank_jacobi_warm_start(ops, R, Cc):
    """r3d_erank_jacobi_warm: the right singular basis rides along (X V = rotated columns, V orthogonal), and a nearby
    matrix decomposed from that basis needs fewer sweeps for the same singular values (vs svdvals)."""
    from oracle import futr_oracle as O
    k = min(R, Cc)
    q1 = torch.linalg.qr(rnd(R, k, seed=R).double())[0]
    q2 = torch.linalg.qr(rnd(Cc, k, seed=R + 1).double())[0]
    sv0 = torch.cat([torch.tensor([6.0]), torch.exp(-torch.arange(k - 1, dtype=torch.float64) / (0.3 * k)) + 1e-3])
    x = (q1 @ torch.diag(sv0) @ q2.t()).float()
    xd = dev(x)
    sig, st = torch.empty(1, Cc, device="cuda"), torch.empty(1, 4, device="cuda")
    aft, vt = torch.empty(1, Cc, R, device="cuda"), torch.empty(Cc, Cc, device="cuda")
    ops.erank_jacobi_warm(xd, sig, st, vt, af_t=aft)                      # cold: V0 = identity
    torch.cuda.synchronize()
    cold = float(st[0, 3])
    V = vt.double().cpu().t()
    assert float((V.t() @ V - torch.eye(Cc, dtype=torch.float64)).abs().max()) < 5e-5
    assert float((x.double() @ V - aft[0].double().cpu().t()).abs().max()) < 2e-4 * float(sv0[0])
    assert abs(float(st[0, 0]) - O.effective_rank(x)) < 5e-3
    x2 = x + 2e-4 * float(x.abs().max()) * rnd(R, Cc, seed=7)
    xw = (dev(x2) @ vt.t()).contiguous()
    vt2 = torch.empty_like(vt)
    ops.erank_jacobi_warm(xw, sig, st, vt2, vt_in=vt, af_t=aft)
    torch.cuda.synchronize()
    assert float(st[0, 3]) < cold, (float(st[0, 3]), cold)
    sv = torch.linalg.svdvals(x2.double())
    assert_close(torch.sort(sig[0].cpu(), descending=True)[0][:k], sv, 1e-4, 2e-4 * float(sv[0]), "sigma (warm)")
    V2 = vt2.double().cpu().t()
    assert float((V2.t() @ V2 - torch.eye(Cc, dtype=torch.float64)).abs().max()) < 1e-4
    assert float((x2.double() @ V2 - aft[0].double().cpu().t()).abs().max()) < 4e-4 * float(sv[0])


@pytest.mark.parametrize("R,Cc", [(512, 512), (128, 512), (2048, 256), (300, 200), (64, 40), (2048, 1024), (256, 1024), (130, 70)])
def test_erank_blocked_vs_svdvals(ops, R, Cc):
    """The two-level Jacobi (columns in HBM) for matrices that exceed one CU's LDS -- and small ones for coverage."""
    from oracle import futr_oracle as O
    from r3d_amd.erank import effective_rank
    # a prescribed, well separated spectrum: in fp32 the singular VECTORS of sigma_k (and with them the gradient) carry
    # a relative error ~ eps * sigma_max / gap, so the gradient check needs a matrix whose small singular values are
    # not crowded (a raw square random matrix has sigma_min ~ 1e-3 sigma_max)
    k = min(R, Cc)
    q1 = torch.linalg.qr(rnd(R, k, seed=R + Cc).double())[0]
    q2 = torch.linalg.qr(rnd(Cc, k, seed=R + Cc + 1).double())[0]
    x = (q1 @ torch.diag(torch.linspace(0.2, 2.0, k).double()) @ q2.t()).float()
    xd = dev(x)
    if R >= Cc:
        sig, st, aft = ops.erank_blocked(xd)
        torch.cuda.synchronize()
        sv = torch.linalg.svdvals(x.double())
        assert_close(torch.sort(sig.cpu(), descending=True)[0], sv, 1e-4, 1e-4 * float(sv[0]), "sigma")
        assert float(st[3]) < 20, "did not converge"
        A = aft[:Cc].cpu().double()
        G = A @ A.t()
        off = G - torch.diag(torch.diag(G))
        assert float(off.abs().max()) < 1e-3 * float(sv[0]) ** 2
        assert float(aft[Cc:].abs().max()) == 0.0 if aft.shape[0] > Cc else True
    # differentiable wrapper (handles R < C by working on the transpose); gradient against autograd through svdvals
    xg = xd.clone().requires_grad_(True)
    er = effective_rank(xg, route="blocked")
    er.backward()
    torch.cuda.synchronize()
    xr = x.double().clone().requires_grad_(True)
    s = torch.linalg.svdvals(xr)
    p = s / s.sum()
    ref = torch.exp(-(p * torch.log(p)).sum())
    ref.backward()
    assert abs(float(er.detach()) - float(ref.detach())) < 5e-3 * max(1.0, float(ref.detach()) / 50)
    assert abs(float(er.detach()) - O.effective_rank(x)) < 0.5
    sc = float(xr.grad.abs().max())
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 2e-3 * sc


@pytest.mark.parametrize("R,Cc,route", [(128, 128, "lds"), (256, 128, "lds"), (512, 512, "blocked"), (256, 1024, "blocked")])
def test_erank_clustered_and_ill_conditioned_spectra(ops, R, Cc, route):
    """the sweeps stop after the first sweep that starts with every coupling below 1e-2 -- quadratic convergence
    leaves ~1e-4, but only for SEPARATED singular values.  A spectrum with CLUSTERS (groups of singular values 0.1 % apart:
    neither duplicate nor rank-deficient, so tan 2 theta = 2 gamma / (alpha - beta) is not small although gamma is) and a
    conditioning of 1e4 -- the regime of the fused tokens -- pins sigma, the effective rank and the gradient at 3e-3 (the
    tolerance before the early stop; the Neumann-corrected backward of round 3 is what makes the ill-conditioned half pass)."""
    from r3d_amd.erank import effective_rank
    k = min(R, Cc)
    q1 = torch.linalg.qr(rnd(R, k, seed=R + 3 * Cc).double())[0]
    q2 = torch.linalg.qr(rnd(Cc, k, seed=R + 3 * Cc + 1).double())[0]
    base = torch.logspace(0, -4, k // 4, dtype=torch.float64)                   # sigma_max / sigma_min = 1e4
    sv_true = (base[:, None] * (1.0 + 1e-3 * torch.arange(4, dtype=torch.float64))[None, :]).reshape(-1)[:k]
    sv_true = torch.sort(sv_true, descending=True)[0] * 50.0
    if sv_true.numel() < k:
        sv_true = torch.cat([sv_true, sv_true[-1:].repeat(k - sv_true.numel()) * 0.5])
    x = (q1 @ torch.diag(sv_true) @ q2.t()).float()
    xg = dev(x).clone().requires_grad_(True)
    er = effective_rank(xg, route=route)
    er.backward()
    torch.cuda.synchronize()
    if route == "blocked":
        xx = dev(x.t().contiguous() if R < Cc else x)
        _, st_, _ = ops.erank_blocked(xx, max_sweeps=30)
        torch.cuda.synchronize()
        print(f"[erank clustered {R}x{Cc}] two-level sweeps to convergence: {float(st_[3]):.0f}")
    xr = x.double().clone().requires_grad_(True)
    s = torch.linalg.svdvals(xr)
    p = s / s.sum()
    ref = torch.exp(-(p * torch.log(p)).sum())
    ref.backward()
    assert abs(float(er.detach()) - float(ref.detach())) < 3e-3 * max(1.0, float(ref.detach()) / 50), (float(er), float(ref))
    sc = float(xr.grad.abs().max())
    err = float((xg.grad.cpu().double() - xr.grad).abs().max()) / sc
    print(f"[erank clustered {R}x{Cc} {route}] erank {float(er):.4f} vs {float(ref):.4f}, gradient error / scale {err:.2e}")
    assert err < 3e-3, err


@pytest.mark.parametrize("N,K,tr", [(128, 128, False), (17, 128, False), (128, 17, True), (512, 128, True), (128, 512, False)])
def test_weight_planes_are_an_exact_three_way_split_in_operand_order(ops, N, K, tr):
    """r3d_weight_planes (csrc/chain_bf3.h): for every element h + m + l == w EXACTLY (truncation split), the element sits where
    the MFMA's lane reads it, and the padding of partial tiles / k-steps is zero."""
    rows, cols = (K, N) if tr else (N, K)
    wt = rnd(rows, cols, seed=N + 3 * K)
    wp = ops.WeightPlanes({"w": (dev(wt), tr)}, torch.device("cuda"))
    wp.refresh()
    torch.cuda.synchronize()
    raw = wp.view("w").cpu().view(torch.int16)
    tiles, ksteps = (N + 15) // 16, (K + 31) // 32
    pl = raw.view(tiles, ksteps, 3, 64, 8).to(torch.int32)
    vals = ((pl & 0xFFFF) << 16).contiguous().view(torch.float32).double()      # bf16 bits -> fp32 values [t, s, plane, lane, e]
    B = wt.t().contiguous() if tr else wt                                       # B[n][k]
    want = torch.zeros(tiles * 16, ksteps * 32, dtype=torch.float64)
    want[:N, :K] = B.double()
    lane = torch.arange(64)
    n_idx = (torch.arange(tiles)[:, None, None, None] * 16 + (lane % 16)[None, None, :, None]).expand(tiles, ksteps, 64, 8)
    k_idx = (torch.arange(ksteps)[None, :, None, None] * 32 + (8 * (lane // 16))[None, None, :, None] +
             torch.arange(8)[None, None, None, :]).expand(tiles, ksteps, 64, 8)
    got = vals.sum(dim=2)                                                       # h + m + l in fp64: exact
    assert torch.equal(got, want[n_idx, k_idx]), float((got - want[n_idx, k_idx]).abs().max())
    # the planes are ordered by magnitude: |m| <= 2^-8 |h|, |l| <= 2^-16 |h| (where h != 0)
    h, m, l = vals[:, :, 0].abs(), vals[:, :, 1].abs(), vals[:, :, 2].abs()
    assert bool((m <= h * 2.0 ** -7).all()) and bool((l <= h * 2.0 ** -15).all())


def test_attention_cores_as_riders_equal_their_own_launches(ops):
    """The small attention core carried as extra workgroups of r3d_gemm_ln_mha_fwd / r3d_layernorm_bwd_multi_mha
    (mha_small.h: one (clip, head) unit per wave) writes bit for bit what r3d_mha_core_fwd / _bwd write on their own, and
    leaves the host launch's results untouched."""
    B, heads, Q, dh, H = 8, 8, 8, 16, 128
    BQ = B * Q
    qkv = dev(rnd(BQ, 3 * H, seed=1))
    mask = dev((torch.rand(B * heads * Q * Q, generator=torch.Generator().manual_seed(2)) > 0.1).to(torch.uint8))
    f = lambda *sh: torch.full(sh, float("nan"), device="cuda")      # noqa: E731
    # ---- forward: alone ...
    p0, o0 = f(B * heads * Q * Q), f(BQ, H)
    ops.mha_core_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], p0, o0, B, heads, Q, Q, dh, drop_mask=mask, drop_scale=1 / 0.9)
    # ... and riding beside a linear + LayerNorm job
    M, K = 64, 128
    a, wgt, g, b = dev(rnd(M, K, seed=3)), dev(rnd(H, K, seed=4, scale=K ** -0.5)), dev(1 + 0.2 * rnd(H, seed=5)), dev(0.1 * rnd(H, seed=6))
    outs = []
    for ride in (False, True):
        pre, y, mean, rstd = f(M, H), f(M, H), f(M), f(M)
        p1, o1 = f(B * heads * Q * Q), f(BQ, H)
        job = dict(a=a, w=wgt, pre=pre, gamma=g, beta=b, y=y, mean=mean, rstd=rstd)
        assert ops.gemm_ln_mha_supported(heads, Q, Q, dh) and not ops.gemm_ln_mha_supported(heads, Q, Q, 32)
        ops.gemm_ln_fwd([job], mha=dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], probs=p1, o=o1, B=B, heads=heads,
                                        Lq=Q, Lk=Q, dh=dh, drop_mask=mask, drop_scale=1 / 0.9) if ride else None)
        outs.append((pre, y, mean, rstd, p1, o1))
    torch.cuda.synchronize()
    for t0, t1 in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(t0, t1)
    assert torch.equal(outs[1][4], p0) and torch.equal(outs[1][5], o0)
    # ---- backward
    d_o = dev(rnd(BQ, H, seed=7))
    dq0 = f(BQ, 3 * H)
    ops.mha_core_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], p0, d_o, dq0[:, :H], dq0[:, H:2 * H], dq0[:, 2 * H:], B, heads,
                     Q, Q, dh, drop_mask=mask, drop_scale=1 / 0.9)
    rows = 256
    x, dy = dev(rnd(rows, H, seed=8)), dev(rnd(rows, H, seed=9))
    yl, ml, rl = f(rows, H), f(rows), f(rows)
    ops.layernorm_fwd(x, g, b, yl, ml, rl)
    res = []
    for ride in (False, True):
        dx, dg, db = f(rows, H), f(H), f(H)
        part = torch.zeros(max(ops.layernorm_bwd_ws_floats(rows, H), 4), device="cuda")
        dq1 = f(BQ, 3 * H)
        ops.layernorm_bwd_multi([dict(dy=dy, x=x, mean=ml, rstd=rl, gamma=g, beta=b, dx=dx, dgamma=dg, dbeta=db, partial=part)],
                                mha=dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], probs=p0, d_o=d_o, dq=dq1[:, :H],
                                         dk=dq1[:, H:2 * H], dv=dq1[:, 2 * H:], B=B, heads=heads, Lq=Q, Lk=Q, dh=dh,
                                         drop_mask=mask, drop_scale=1 / 0.9) if ride else None)
        res.append((dx, part, dq1))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[1][2], dq0)
