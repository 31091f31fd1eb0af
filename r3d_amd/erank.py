"""Effective rank of the fused token matrix as a differentiable op (build-side; the reference only describes the
quantity, README.md:8-14 -- SURVEY.md F1).  erank(X) = exp(-sum p log p), p = sigma / sum(sigma).

forward : batched one-sided Jacobi SVD in HIP (r3d_erank_jacobi), columns resident in one CU's LDS;
backward: dX = U diag(d erank/d sigma) V^T = Af diag(g / sigma^3) (Af^T X) with Af = X V the rotated columns the
          sweep leaves behind -- two MFMA GEMMs and a row scale, no V accumulation.
Matrices that do not fit one CU's LDS (H = 512 / 1024, large batches) take the two-level block-Jacobi kernel with the
columns in HBM (r3d_erank_blocked), same outputs, same backward.  A Gram route (X^T X through the LDS kernel) is kept
as a measurement-only cross-check."""
import torch

from . import ops
from ._lib import GEMM_NN, GEMM_NT, GEMM_TN


class ErankBackward:
    """dX = d(erank)/dX * gout from what the sweep left behind: A = (X V)^T, the rotated columns as rows ([C, R] for the
    decomposed orientation), sigma, stats.  With U^T = diag(1/sigma) A:

        W = U^T X  ->  W <- diag(g / sigma) (2 W - (U^T U) W)  ->  dX = U W

    i.e. U diag(g) V^T with V^T = Sigma^-1 (2 I - U^T U) U^T X: the plain V^T = Sigma^-1 U^T X lets the residual coupling
    of a small column with a large one through amplified by sigma_j / sigma_i; one Neumann term of (U^T U)^-1 removes it to
    first order (measured against fp64 autograd through svdvals: 6-8x closer at sigma_max / sigma_min = 1e4; what remains
    is the fp32 rounding of the rotated columns themselves, ~eps * sigma_max / sigma_i in direction i).  Four GEMMs, three
    row-wise kernels, all enqueued on the current stream; buffers are allocated once (the training step replays it in a
    hipGraph).  flip: the sweep ran on X^T (A is [R, C] for X [R, C]); the same products in the transposed orientation."""

    def __init__(self, R, C, flip, device):
        self.flip, self.R, self.C = bool(flip), R, C
        k = R if flip else C                       # singular values / rows of A
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)     # noqa: E731
        self.cg, self.inv, self.w, self.g, self.p = f(k), f(k), f(k, k), f(k, k), f(k, k)

    def run(self, x, a_rows, sigma, stats, gout, out, accumulate, ws):
        """x [R, C]; a_rows [k, len] view (row stride may exceed len), scaled IN PLACE to U^T; out [R, C]."""
        k = self.R if self.flip else self.C
        ops.erank_bwd_coef2(sigma, stats, gout, self.cg, self.inv, max_rank=min(self.R, self.C))
        ops.scale_rows(a_rows, self.inv)                                   # A -> U^T
        if self.flip:
            ops.gemm(GEMM_NT, a_rows, x, self.w, ws=ws)                    # U^T X^T            [R, R]
        else:
            ops.gemm(GEMM_NN, a_rows, x, self.w, ws=ws)                    # U^T X              [C, C]
        ops.gemm(GEMM_NT, a_rows, a_rows, self.g, ws=ws)                   # U^T U
        ops.gemm(GEMM_NN, self.g, self.w, self.p, ws=ws)
        ops.erank_bwd_fix(self.w, self.p, self.cg)                         # diag(g / sigma) (2 W - (U^T U) W)
        if self.flip:
            ops.gemm(GEMM_TN, self.w, a_rows, out, accumulate=accumulate, ws=ws)       # (U W)^T = W^T U^T
        else:
            ops.gemm(GEMM_TN, a_rows, self.w, out, accumulate=accumulate, ws=ws)       # U W
        assert k == self.w.shape[0]


class _ERank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        R, C = x.shape
        sigma = torch.empty(1, C, dtype=torch.float32, device=x.device)
        stats = torch.empty(1, 4, dtype=torch.float32, device=x.device)
        af_t = torch.empty(1, C, R, dtype=torch.float32, device=x.device)
        ops.erank_jacobi(x, sigma, stats, af_t=af_t)
        ctx.save_for_backward(x, sigma, stats, af_t)
        return stats[0, 0].clone()

    @staticmethod
    def backward(ctx, gout):
        x, sigma, stats, af_t = ctx.saved_tensors
        R, C = x.shape
        dx = torch.empty_like(x)
        ErankBackward(R, C, False, x.device).run(x, af_t[0], sigma[0], stats[0], gout.contiguous().reshape(1).float(), dx, False,
                                                 ops.GemmWorkspace(x.device))
        return dx


class _ERankBlocked(torch.autograd.Function):
    """Any size: the two-level Jacobi with the columns in HBM (r3d_erank_blocked).  Works on the orientation with the
    fewer columns (the singular values of X and X^T are the same; the gradient is transposed back)."""

    @staticmethod
    def forward(ctx, x):
        R, C = x.shape
        ctx.flip = R < C
        xx = x.t().contiguous() if ctx.flip else x
        sigma, stats, af_t = ops.erank_blocked(xx)
        ctx.save_for_backward(xx, sigma, stats, af_t)
        return stats[0].clone()

    @staticmethod
    def backward(ctx, gout):
        xx, sigma, stats, af_t = ctx.saved_tensors
        R, C = xx.shape
        dx = torch.empty_like(xx)
        ErankBackward(R, C, False, xx.device).run(xx, af_t[:C], sigma, stats, gout.contiguous().reshape(1).float(), dx, False,
                                                  ops.GemmWorkspace(xx.device))
        return dx.t().contiguous() if ctx.flip else dx


def effective_rank(x, route="auto"):
    """x: [R, C] or [B, T, C] (flattened to [B*T, C]) float32 device tensor -> 0-dim tensor (differentiable).
    route: "auto" (LDS-resident kernel when the matrix fits one CU, else the blocked one), "lds", "blocked", or
    "gram" (X^T X through the LDS kernel: measurement only, squares the condition number)."""
    if x.dim() == 3:
        x = x.reshape(-1, x.shape[-1])
    x = x.contiguous().float()
    R, C = x.shape
    if route == "auto":
        route = "lds" if ops.erank_fits(R, C) else "blocked"
    if route == "lds":
        return _ERank.apply(x)
    if route == "blocked":
        return _ERankBlocked.apply(x)
    if route == "gram":
        assert ops.erank_fits(C, C) and not (x.requires_grad and torch.is_grad_enabled())
        g = torch.empty(C, C, dtype=torch.float32, device=x.device)
        ops.gemm(GEMM_TN, x.detach(), x.detach(), g, ws=ops.GemmWorkspace(x.device))
        sigma = torch.empty(1, C, dtype=torch.float32, device=x.device)
        stats = torch.empty(1, 4, dtype=torch.float32, device=x.device)
        ops.erank_jacobi(g, sigma, stats, gram=True)
        return stats[0, 0].clone()
    raise ValueError(route)
