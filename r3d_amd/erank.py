"""Effective rank of the fused token matrix as a differentiable op (build-side; the reference only describes the
quantity, README.md:8-14 -- SURVEY.md F1).  erank(X) = exp(-sum p log p), p = sigma / sum(sigma).

forward : batched one-sided Jacobi SVD in HIP (r3d_erank_jacobi), columns resident in one CU's LDS;
backward: dX = U diag(d erank/d sigma) V^T = Af diag(g / sigma^3) (Af^T X) with Af = X V the rotated columns the
          sweep leaves behind -- two MFMA GEMMs and a row scale, no V accumulation.
Tall matrices that do not fit LDS are reduced through their Gram matrix X^T X (MFMA GEMM) when C*C fits; that route
is measurement-only (no backward)."""
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
        ops.erank_bwd_coef(sigma[0], stats[0], gout.contiguous().reshape(1).float(), coef)
        ws = ops.GemmWorkspace(x.device)
        t1 = torch.empty(C, C, dtype=torch.float32, device=x.device)
        ops.gemm(GEMM_NN, af_t[0], x, t1, ws=ws)            # Af^T X = Sigma^2 V^T
        ops.scale_rows(t1, coef)
        dx = torch.empty_like(x)
        ops.gemm(GEMM_TN, af_t[0], t1, dx, ws=ws)           # Af diag(coef) Af^T X
        return dx


def effective_rank(x):
    """x: [R, C] or [B, T, C] (flattened to [B*T, C]) float32 device tensor -> 0-dim tensor."""
    if x.dim() == 3:
        x = x.reshape(-1, x.shape[-1])
    x = x.contiguous().float()
    R, C = x.shape
    if ops.erank_fits(R, C):
        return _ERank.apply(x)
    if ops.erank_fits(C, C):
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("effective_rank backward needs the [N,H] matrix to fit one CU's LDS "
                                      f"(got {R}x{C}); the Gram route is measurement-only")
        g = torch.empty(C, C, dtype=torch.float32, device=x.device)
        ops.gemm(GEMM_TN, x.detach(), x.detach(), g, ws=ops.GemmWorkspace(x.device))
        sigma = torch.empty(1, C, dtype=torch.float32, device=x.device)
        stats = torch.empty(1, 4, dtype=torch.float32, device=x.device)
        ops.erank_jacobi(g, sigma, stats, gram=True)
        return stats[0, 0].clone()
    raise NotImplementedError(f"effective_rank: {R}x{C} exceeds the LDS-resident Jacobi kernel (C <= ~200); the "
                              "multi-workgroup block-Jacobi variant is not built yet")
