"""SURVEY 8(f) rows beside the hot path, on the GPU: the inference decode (predict_clip) against the oracle's eval-mode
forward, and the device-resident input pipeline (InputPrefetcher) feeding train() the same batches as a plain list."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import futr_oracle as O  # noqa: E402
from tests.helpers import load_fixture, fixture_params, fixture_batch  # noqa: E402
from tests.test_engine_gpu import build_model  # noqa: E402


def test_predict_clip_matches_oracle_decode(oracle_lib):
    from r3d_amd.predict import predict_clip
    from r3d_amd import utils as U
    fx = load_fixture("val_h128")
    m = fx["meta"]
    feats, depth, lab, dur, tgt = fixture_batch(fx, pad_tail=False)
    model = build_model(fx).train()                       # predict_clip must switch to eval and restore
    res = predict_clip(model, feats[0].cuda(), depth[0].cuda(), future_len=37)
    assert model.training
    oout, oaux = O.forward(fixture_params(fx), (feats[:1], lab[:1]), depth[:1], "val", m["pad_idx"], m["n_head"], m["n_dec"])
    assert torch.equal(res["action_labels"].cpu(), oout["action"][0].argmax(-1))
    assert torch.equal(res["seg_labels"].cpu(), oout["seg"][0].argmax(-1))
    want = U.expand_durations(oout["action"][0].detach(), oout["duration"][0].detach(), 37, m["n_class"] - 1)
    assert torch.equal(res["frames"], want)


def test_input_prefetcher_feeds_the_same_training(tmp_path):
    from r3d_amd.train_proposed_depth import train
    from r3d_amd.optim import FlatAdamW, LinearWarmupCosineAnnealingLR
    from r3d_amd.utils import InputPrefetcher
    fx = load_fixture("train_loop")
    m = fx["meta"]
    batches = [[t for t in fixture_batch(fx, seed=200 + i)] for i in range(3)] + [None]
    val = [[t[:1] for t in fixture_batch(fx, seed=300)]]
    finals = []
    for wrap in (False, True):
        model = build_model(fx)
        args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                                  min_batch=1)
        opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
        sch = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=2, max_epochs=4)
        sch.step()                                        # epoch 0 runs at lr 0 in the reference's schedule
        loader = InputPrefetcher(batches, "cuda") if wrap else batches
        vloader = InputPrefetcher(val, "cuda") if wrap else val
        model.eval()                                      # dropout off: the two runs must agree exactly
        train(args, model, loader, opt, sch, None, str(tmp_path), m["pad_idx"], torch.device("cuda"), vloader, seed=1)
        torch.cuda.synchronize()
        finals.append(model.engine().arena.params.clone())
    assert torch.equal(finals[0], finals[1])


def test_npy_files_through_pinned_reader_and_prefetcher_train_identically(tmp_path):
    """SURVEY.md 8(f).2 end to end: the batches' feature / depth rows written as per-video `.npy` files, read back by
    NpyClipReader into pinned staging buffers and moved by InputPrefetcher's side stream, train to the same parameters
    (bitwise) as the tensors handed over directly."""
    import numpy as np
    from r3d_amd.train_proposed_depth import train
    from r3d_amd.optim import FlatAdamW
    from r3d_amd.utils import InputPrefetcher, NpyClipReader
    fx = load_fixture("train_loop")
    m = fx["meta"]
    batches = [[t for t in fixture_batch(fx, seed=400 + i)] for i in range(3)]
    val = [[t[:1] for t in fixture_batch(fx, seed=300)]]
    specs = []
    for i, b in enumerate(batches):                                   # one "video" per clip, with frames before and after
        clips = []
        for c in range(b[0].shape[0]):
            S = b[0].shape[1]
            pre, post = 3 + c, 2
            f = np.concatenate([np.full((pre,) + tuple(b[0].shape[2:]), 7.0, np.float32), b[0][c].numpy(),
                                np.full((post,) + tuple(b[0].shape[2:]), 9.0, np.float32)])
            d = np.concatenate([np.full((pre,) + tuple(b[1].shape[2:]), 7.0, np.float32), b[1][c].numpy(),
                                np.full((post,) + tuple(b[1].shape[2:]), 9.0, np.float32)])
            fp, dp = tmp_path / f"b{i}c{c}.npy", tmp_path / f"b{i}c{c}_1.npy"
            np.save(fp, f)
            np.save(dp, d)
            clips.append((str(fp), str(dp), pre, pre + S, 1))
        specs.append(clips)
    rd = NpyClipReader()
    assert rd.pin

    def from_files():
        for clips, b in zip(specs, batches):
            f, d = rd.batch(clips)
            assert f.is_pinned() and d.is_pinned()
            yield [f, d, b[2], b[3], b[4]]

    class Loader:                                                      # (train() wants len() for its epoch print)
        def __iter__(self):
            return from_files()

        def __len__(self):
            return len(batches)
    finals = []
    for files in (False, True):
        model = build_model(fx)
        args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long", min_batch=1)
        opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)

        class NoSched:
            def step(self):
                pass
        model.eval()
        loader = InputPrefetcher(Loader(), "cuda") if files else batches
        train(args, model, loader, opt, NoSched(), None, str(tmp_path), m["pad_idx"], torch.device("cuda"), val, seed=1)
        torch.cuda.synchronize()
        finals.append(model.engine().arena.params.clone())
    assert torch.equal(finals[0], finals[1])


def test_train_harness_matches_reference_capture(tmp_path, capsys):
    """r3d_amd.train_proposed_depth.train() -- loop, epoch prints, validate(), checkpoint-on-improve -- against the capture
    of the reference's own train() on the same batches (tests/golden/train_loop.npz: its stdout, validate() result and
    checkpoint files; dropout probabilities 0 on both sides, AdamW lr 1e-3 / wd 5e-3, no-op scheduler)."""
    import json
    import os
    import re
    from r3d_amd.train_proposed_depth import train
    from r3d_amd.optim import FlatAdamW
    fx = load_fixture("train_loop")
    m = fx["meta"]
    batches = [fixture_batch(fx, seed=m["seed"] + i) for i in range(m["n_steps"])]
    from oracle import synth
    val = [[torch.from_numpy(x) for x in synth.make_batch(1, m["val_S"], m["n_class"], m["pad_idx"], m["seed"] + 100,
                                                          pad_tail=False)]]
    model = build_model(fx)
    model.r3d_dropout_enabled = False
    args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long")

    class NoSched:
        def step(self):
            pass
    opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
    train(args, model, batches, opt, NoSched(), None, str(tmp_path), m["pad_idx"], torch.device("cuda"), val, seed=1)
    out = capsys.readouterr().out
    ref = json.loads(str(fx["stdout"]))
    nums = lambda s: [float(x) for x in re.findall(r"-?[0-9]+[.][0-9]+", s)]          # noqa: E731
    ref_lines, got_lines = ref.strip().splitlines(), out.strip().splitlines()
    assert len(ref_lines) == len(got_lines), (ref, out)
    for rl, gl in zip(ref_lines, got_lines):
        assert re.sub(r"-?[0-9]+[.][0-9]+", "#", rl) == re.sub(r"-?[0-9]+[.][0-9]+", "#", gl), (rl, gl)     # same text
        for a, b in zip(nums(rl), nums(gl)):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a)) + 1.5e-3, (rl, gl)                      # 3-decimal prints
    assert sorted(os.listdir(tmp_path)) == fx["ckpt_files"]
    sd = torch.load(os.path.join(tmp_path, "seed_1_best.ckpt"), weights_only=True)
    assert list(sd.keys()) == fx["ckpt_keys"]


def test_predict_over_synthetic_videos_matches_fixture(tmp_path, capsys):
    """predict() with the reference's signature (evaluation/predict_utkinects.py:215, main_darai.py:164) driving the HIP model
    over three synthetic per-video .npy + ground-truth files: labels / anticipated frames equal the oracle's, the per-class
    counts equal the reference's own utils.eval_file (tests/golden/make_predict_golden.py), the MoC lines are the
    reference's text."""
    import json
    import os
    from oracle import synth
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    from tests.test_utils_cpu import check_predict_against_fixture
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predict_golden.npz"))
    m = json.loads(str(fx["meta"]))
    a = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(m["K"], m["H"], m["K"] + 1, torch.device("cuda"), a, n_query=m["Q"], n_head=m["heads"],
                 num_encoder_layers=2, num_decoder_layers=1, depth_pixels=m["pix"][0] * m["pix"][1])
    names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    params = {n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}
    model.load_state_dict(params, strict=False)
    model = model.to("cuda")
    check_predict_against_fixture(model, tmp_path, capsys, torch.device("cuda"))
