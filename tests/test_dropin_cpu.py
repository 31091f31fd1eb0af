"""dropin/: the reference's import lines (main_darai.py:12-47) resolve to r3d_amd without an edit.  Run in a fresh
interpreter (the names `model`, `utils`, `opts` must not collide with this test process) with PYTHONPATH=dropin:repo; the
state_dict keys / shapes of every model reached through those names equal the ones recorded from the imported reference
(tests/golden/*.npz: param_names / param_shapes)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import argparse, json, sys, torch
from opts import parser                                              # main_darai.py:12
from pl_bolts.optimizers.lr_scheduler import LinearWarmupCosineAnnealingLR   # :13
from utils import read_mapping_dict                                  # :16
from train_proposed_depth import train, validate                     # :39
from predict_utkinects import predict                                # :42-47
import model.futr_safuser_tokenfusion as M1                          # :25
import model.futr_safuser_batchnormalization as M2                   # :29
import model.futr_unsupervised_depth as M3                           # :31
import model.futr_proposed as M4                                     # :22
import r3d_amd.model.futr_safuser_tokenfusion as R1
assert M1.FUTR is R1.FUTR and M1.CMFuser is R1.CMFuser
args = parser.parse_args([])
assert args.predict == 'predict' and args.hidden_dim == 128           # opts.py:13, defaults
out = {}
a = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
for name, M, kw in (("tokenfusion", M1, {}), ("bn", M2, {})):
    m = M.FUTR(17, 128, 18, torch.device("cpu"), a, n_query=8, n_head=8, num_encoder_layers=2, num_decoder_layers=1, **kw)
    out[name] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
print(json.dumps(out))
'''


def test_reference_import_lines_resolve_through_dropin():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT]))
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().split("\n")[-1])
    import numpy as np
    for tag, fixture in (("tokenfusion", "step_cfg2.npz"), ("bn", "bn_cfg2.npz")):
        fx = np.load(os.path.join(ROOT, "tests", "golden", fixture), allow_pickle=False)
        meta = json.loads(str(fx["meta"]))
        names = json.loads(str(fx["param_names"]))
        shapes = {n: tuple(sh) for n, sh in zip(names, json.loads(str(fx["param_shapes"])))}
        have = {k: tuple(s) for k, s in got[tag]}
        assert [n for n in names if n not in have] == [], tag
        for n in names:
            assert have[n] == shapes[n], (tag, n, have[n], shapes[n])
        # (buffers -- pos_enc tables, BatchNorm statistics -- are in state_dict() but not in named_parameters())
        assert all(("pos_table" in k or "running_" in k or "num_batches" in k) for k in have if k not in shapes), tag
        assert meta["H"] == 128
