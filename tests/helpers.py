"""Shared helpers for the parity tests (fixtures -> parameters / batches)."""
import json
import os

import numpy as np
import torch

from oracle import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(tag):
    z = np.load(os.path.join(GOLDEN, f"{tag}.npz"), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    fx["meta"] = json.loads(str(fx["meta"]))
    for k in ("param_names", "param_shapes", "live_names", "ckpt_files", "ckpt_keys"):
        if k in fx:
            fx[k] = json.loads(str(fx[k]))
    return fx


def fixture_params(fx):
    """dict name -> torch tensor, regenerated bit-identically from the hash fill."""
    return {n: torch.from_numpy(synth.fill_value(n, tuple(s), j))
            for j, (n, s) in enumerate(zip(fx["param_names"], fx["param_shapes"]))}


def fixture_batch(fx, seed=None, **kw):
    m = fx["meta"]
    b = synth.make_batch(m["B"], m["S"], m["n_class"], m["pad_idx"], m["seed"] if seed is None else seed,
                         zero_mean_depth=m.get("zero_mean_depth", False), **kw)
    return [torch.from_numpy(x) for x in b]


def stats(t):
    a = t.detach().double().reshape(-1).cpu()
    n = a.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, 16)).long()
    return np.concatenate([[float(a.norm()), float(a.sum()), float(a.abs().sum())],
                           a[:16].numpy() if n >= 16 else np.pad(a.numpy(), (0, 16 - n)),
                           a[idx].numpy() if n >= 16 else np.pad(a[idx].numpy(), (0, 16 - idx.numel()))])


def assert_close(a, b, rtol=1e-3, atol=1e-5, what=""):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if a.numel() == 0:
        return
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = err > lim
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{a.numel()} outside tol; max abs err "
                                 f"{float(err.max()):.3e}, ref scale {float(b.abs().max()):.3e}")
