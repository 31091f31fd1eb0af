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
from ._lib import GEMM_NN, GEMM_TN


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
        coef = torch.empty(C, dtype=torch.float32, device=x.device)
        ops.erank_bwd_coef(sigma[0], stats[0], gout.contiguous().reshape(1).float(), coef, max_rank=min(R, C))
        ws = ops.GemmWorkspace(x.device)
        t1 = torch.empty(C, C, dtype=torch.float32, device=x.device)
        ops.gemm(GEMM_NN, af_t[0], x, t1, ws=ws)            # Af^T X = Sigma^2 V^T
        ops.scale_rows(t1, coef)
        dx = torch.empty_like(x)
        ops.gemm(GEMM_TN, af_t[0], t1, dx, ws=ws)           # Af diag(coef) Af^T X
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
        af = af_t[:C]
        coef = torch.empty(C, dtype=torch.float32, device=xx.device)
        ops.erank_bwd_coef(sigma, stats, gout.contiguous().reshape(1).float(), coef, max_rank=min(R, C))
        ws = ops.GemmWorkspace(xx.device)
        t1 = torch.empty(C, C, dtype=torch.float32, device=xx.device)
        ops.gemm(GEMM_NN, af, xx, t1, ws=ws)
        ops.scale_rows(t1, coef)
        dx = torch.empty_like(xx)
        ops.gemm(GEMM_TN, af, t1, dx, ws=ws)
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
