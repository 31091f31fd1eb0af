"""The introselect restatement (C and python) against torch.topk itself -- the op the reference calls at
model/futr_safuser_tokenfusion.py:53-54.  CPU only."""
import numpy as np
import torch

from oracle import futr_oracle as O


def _ref(s, k):
    C = s.size
    return np.sort(torch.topk(torch.from_numpy(s).view(1, 1, C), k, dim=-1, largest=False)[1].view(-1).numpy())


def _cases():
    rng = np.random.default_rng(0)
    for C in [8, 12, 16, 32, 64, 100, 128, 256, 512, 1024]:
        for trial in range(24):
            kind = trial % 6
            if kind == 0:
                s = np.full(C, 1.0 / (8 * 16 * C), np.float32)          # train mode: every score equal
            elif kind == 1:
                s = rng.random(C).astype(np.float32)
            elif kind == 2:
                s = rng.integers(0, 3, C).astype(np.float32)
            elif kind == 3:
                s = rng.integers(0, C // 2 + 1, C).astype(np.float32)
            elif kind == 4:
                s = np.sort(rng.random(C).astype(np.float32))
            else:
                s = rng.integers(0, 4, C).astype(np.float32)
                s[rng.integers(0, C, 2)] = np.nan
            yield C, s


def test_c_restatement_matches_torch_topk(oracle_lib):
    for C, s in _cases():
        k = C // 4
        assert np.array_equal(O.select_smallest(s, k, use_c=True), _ref(s, k)), (C, s[:8])


def test_python_restatement_matches_torch_topk():
    for C, s in _cases():
        if C > 256:
            continue
        k = C // 4
        assert np.array_equal(O.select_smallest(s, k, use_c=False), _ref(s, k)), (C, s[:8])


def test_train_mode_tie_set_is_upper_middle_quarter(oracle_lib):
    # all-equal scores -> {C/2+1 .. 3C/4}
    for C in (128, 512, 1024):
        got = O.select_smallest(np.full(C, 0.25, np.float32), C // 4)
        assert np.array_equal(got, np.arange(C // 2 + 1, 3 * C // 4 + 1))
