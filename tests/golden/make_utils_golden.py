#!/usr/bin/env python3
"""Generates tests/golden/utils_golden.npz by CALLING the reference's own utils.py (this container only; the reference
does not travel): normalize_duration (utils.py:325-328) and eval_file (:341-356) on seeded synthetic inputs."""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import utils as R  # noqa: E402  (the reference's utils.py)

g = torch.Generator().manual_seed(5)
x = torch.randn(6, 8, generator=g)
m = (torch.rand(6, 8, generator=g) > 0.3).float()
m[0] = 1.0
out = R.normalize_duration(x, m)
classes = {"a": 0, "b": 1, "c": 2, "d": 3}
rng = np.random.RandomState(3)
names = list(classes)
gt = [names[i] for i in rng.randint(0, 4, 60)]
recog = [g_ if rng.rand() < 0.6 else names[rng.randint(0, 4)] for g_ in gt][:55]
n_t, n_f = R.eval_file(list(gt), list(recog), 0.3, classes)
tot = n_t + n_f
moc = float(np.mean([n_t[j] / tot[j] for j in range(4) if tot[j] != 0]))
here = os.path.dirname(os.path.abspath(__file__))
np.savez(os.path.join(here, "utils_golden.npz"), dur_in=x.numpy(), dur_mask=m.numpy(), dur_out=out.numpy(), n_t=n_t, n_f=n_f,
         moc=np.float64(moc), meta=json.dumps(dict(gt=gt, recog=recog, obs=0.3, classes=classes)))
print("wrote utils_golden.npz", n_t, n_f, moc)
