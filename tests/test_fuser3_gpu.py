"""The build-defined THREE-modality fuser (r3d_amd/model/cmfuser3.py, csrc/fuser3.hip; BASELINE.json configs[4]) against the
build's own CPU restatement oracle.cm_fuser_m -- **parity unpinned**: the reference's CMFuser is two-token
(model/futr_safuser_tokenfusion.py:74-81), SURVEY.md 8(d) allows exactly this check.  What pins the restatement is that for
M = 2 it is the reference-pinned cm_fuser bit for bit (tests/test_oracle_golden.py).  Forward, selected indices (bit-exact as
sets), and every gradient -- three input streams and all 13 live parameters (Q / K now receive gradient) -- at B = 16,
hidden 1024 (configs[4]'s per-GPU shape) and at a small shape; tolerance 1e-3 of each tensor's scale (north star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import futr_oracle as O, synth  # noqa: E402


def _close(a, b, what, rtol=1e-3):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    sc = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max())
    assert err <= rtol * sc, f"{what}: max abs err {err:.3e} vs scale {sc:.3e} (rel {err / sc:.2e})"


@pytest.mark.parametrize("B,S,H,heads,mode", [(2, 5, 64, 4, "train"), (16, 16, 1024, 8, "train"), (4, 8, 256, 8, "val")])
def test_three_modality_fuser_matches_the_cpu_restatement(B, S, H, heads, mode):
    from r3d_amd.model.cmfuser3 import CMFuser3
    fz = CMFuser3(H, depth=1, num_heads=heads).to("cuda").eval()          # eval(): embd_drop off; `mode` picks the selection
    names = [("fuser." + n, tuple(p.shape)) for n, p in fz.named_parameters()]
    vals = synth.fill_state(names)
    with torch.no_grad():
        for n, p in fz.named_parameters():
            p.copy_(torch.from_numpy(vals["fuser." + n]))
    pc = {n: torch.from_numpy(v).clone().requires_grad_(True) for n, v in vals.items()}
    xs = [torch.from_numpy(synth.symmetric(B * S * H, 0xF3000 + 17 * m).reshape(B, S, H)).clone() for m in range(3)]
    xs[0] = xs[0].relu()                                                    # (an embedding after its ReLU, as :183,197)
    xs[1] = xs[1].relu() * 0.7
    cot = torch.from_numpy(synth.symmetric(B * S * H, 0xF3900).reshape(B, S, H))
    # CPU restatement
    xc = [x.clone().requires_grad_(True) for x in xs]
    want, aux = O.cm_fuser_m(pc, xc, mode, heads)
    (want * cot).sum().backward()
    # HIP
    xg = [x.cuda().requires_grad_(True) for x in xs]
    got = fz({"rgb": xg[0], "depth": xg[1], "gaze": xg[2]}, mode)
    (got * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    _close(got.detach(), want.detach(), f"H{H}/{mode} fused")
    for m in range(3):
        assert np.array_equal(np.sort(fz.last_idx[m].cpu().numpy()), np.sort(aux["idx"][m].numpy())), f"selection of modality {m}"
        _close(xg[m].grad, xc[m].grad, f"H{H}/{mode} d input {m}")
    live = 0
    for n, p in fz.named_parameters():
        ref = pc["fuser." + n].grad
        if ref is None:
            assert p.grad is None, n
            continue
        live += 1
        _close(p.grad, ref, f"H{H}/{mode} grad {n}", rtol=2e-3)
    assert live == 13
    qk = fz.blocks[0].attn.qkv.weight.grad[:2 * H]
    assert float(qk.abs().max()) > 0.0, "with three tokens the softmax is real: Q / K must receive gradient"
