"""CPU checks of the label-query model (reference model/futr_proposed.py): the oracle restatement (forward_proposed) against
the fixtures tests/golden/make_golden_proposed.py generated from the imported reference, and the drop-in module's
state_dict contract."""
import argparse
import json

import numpy as np
import pytest
import torch

from oracle import futr_oracle as O, synth
from tests.helpers import load_fixture, fixture_params, fixture_batch, stats, assert_close


def probe_loss(out):
    return (out["seg"] ** 2).mean() + 0.01 * out["action"].sum() + (0.1 * out["duration"]).exp().mean()


def proposed_inputs(fx):
    m = fx["meta"]
    feats, _, lab, _, _ = fixture_batch(fx, depth_hw=(2, 2))
    query = torch.from_numpy(synth.randint(m["B"] * m["S"], m["query_num"], (m["seed"] << 8) + 77).reshape(m["B"], m["S"]))
    assert np.array_equal(query.numpy(), fx["query"])
    return feats, lab, query


@pytest.mark.parametrize("tag", ["proposed_tiny", "proposed_h128"])
def test_oracle_matches_reference_fixture(tag, oracle_lib):
    fx = load_fixture(tag)
    m = fx["meta"]
    feats, lab, query = proposed_inputs(fx)
    live = fx["live_names"]
    p = {n: v.requires_grad_(n in live) for n, v in fixture_params(fx).items()}
    out, _ = O.forward_proposed(p, (feats, lab), query, "train", m["pad_idx"], m["n_head"], m["n_dec"], m["n_query"])
    for k, fk in (("action", "out_action"), ("duration", "out_duration"), ("seg", "out_seg")):
        assert_close(out[k].detach(), fx[fk], 2e-5, 2e-5, f"{tag}/{k}")
    loss = probe_loss(out)
    assert abs(float(loss) - float(fx["probe_loss"][0])) < 1e-5
    loss.backward()
    gs = np.stack([stats(p[n].grad) for n in live])
    assert_close(gs[:, [0, 2]], fx["grad_stats"][:, [0, 2]], 1e-4, 1e-6, f"{tag}/grad norms")
    for k in ("query_embed.weight", "fc_seg.weight", "input_embed.bias"):
        assert_close(p[k].grad, fx["grad::" + k], 1e-4, 1e-6, k)


def test_state_dict_contract():
    from r3d_amd.model.futr_proposed import FUTR
    fx = load_fixture("proposed_h128")
    m = fx["meta"]
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(m["n_class"], m["H"], m["pad_idx"], torch.device("cpu"), args, n_query=8, n_head=8, num_encoder_layers=2,
                 num_decoder_layers=m["n_dec"], query_num=m["query_num"])
    sd = model.state_dict()
    assert list(sd.keys()) == json.loads(str(fx["state_keys"]))
    assert [list(v.shape) for v in sd.values()] == json.loads(str(fx["state_shapes"]))
    assert [n for n, _ in model.named_parameters()] == fx["param_names"]
