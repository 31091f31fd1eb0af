"""NpyClipReader: batches assembled from memory-mapped `.npy` files equal what the reference's
dataset + collate produce from the same files (np.load -> slice -> torch.tensor -> pad_sequence(batch_first=True,
padding_value=0): basedataset_darai_depth.py:110-130,176,199-203), and the staging buffers rotate without aliasing."""
import numpy as np
import torch

from r3d_amd.utils import NpyClipReader


def _files(tmp_path, n_videos=3, D=16, hw=(6, 8)):
    rng = np.random.default_rng(5)
    out = []
    for v in range(n_videos):
        T = 20 + 7 * v
        f = rng.standard_normal((T, D)).astype(np.float32)
        d = rng.random((T, 1) + hw).astype(np.float32)
        fp, dp = tmp_path / f"v{v}.npy", tmp_path / f"v{v}_1.npy"
        np.save(fp, f)
        np.save(dp, d)
        out.append((str(fp), str(dp), f, d))
    return out


def _reference_collate(files, clips):
    """np.load -> [depth: trim to the recording's frame range, basedataset_darai_depth.py:112-114] -> observed slice ->
    sample rate -> torch.tensor -> pad_sequence per tensor (:118-130,176,199-203)."""
    feats, depth = [], []
    for c in clips:
        f, d = np.load(c[0]), np.load(c[1])
        if len(c) > 5:
            d = d[c[5]:(c[6] if len(c) > 6 else None)]
        feats.append(torch.tensor(f[c[2]:c[3]][::c[4]], dtype=torch.float32))
        depth.append(torch.tensor(d[c[2]:c[3]][::c[4]], dtype=torch.float32))
    pad = torch.nn.utils.rnn.pad_sequence
    return pad(feats, batch_first=True, padding_value=0), pad(depth, batch_first=True, padding_value=0)


def test_npy_batches_equal_reference_collate(tmp_path):
    files = _files(tmp_path)
    rd = NpyClipReader(slots=2, pin=False)
    clips = [(files[0][0], files[0][1], 2, 14, 3), (files[1][0], files[1][1], 0, 27, 3), (files[2][0], files[2][1], 5, 9, 1)]
    f, d = rd.batch(clips)
    rf, rdp = _reference_collate(files, clips)
    assert f.shape == rf.shape and d.shape == rdp.shape
    assert torch.equal(f, rf) and torch.equal(d, rdp)
    # ragged: the longest clip shrinks -> a stale tail of the reused buffer must not leak into the padding
    clips2 = [(files[2][0], files[2][1], 0, 6, 2), (files[0][0], files[0][1], 1, 3, 1)]
    keep_f = f.clone()
    f2, d2 = rd.batch(clips2)
    assert torch.equal(f, keep_f)                                  # slot rotation: the previous batch is untouched
    rf2, rd2 = _reference_collate(files, clips2)
    assert torch.equal(f2, rf2) and torch.equal(d2, rd2)
    f3, d3 = rd.batch(clips)                                       # third batch reuses slot 0
    assert torch.equal(f3, rf) and torch.equal(d3, rdp)


def test_npy_empty_and_out_of_range_clips(tmp_path):
    files = _files(tmp_path, n_videos=1)
    rd = NpyClipReader(pin=False)
    T = files[0][2].shape[0]
    f, d = rd.batch([(files[0][0], files[0][1], T - 2, T + 50, 1), (files[0][0], files[0][1], 4, 4, 1)])
    assert f.shape[:2] == (2, 2) and torch.equal(f[0], torch.from_numpy(files[0][2][T - 2:]))
    assert float(f[1].abs().sum()) == 0.0 and float(d[1].abs().sum()) == 0.0


def test_npy_depth_trim_and_short_depth_file(tmp_path):
    """The reference trims the per-recording depth file to the sequence's frame range before the observed slice
    (basedataset_darai_depth.py:112-114) and pads features and depth independently (:199-203): a depth offset, and a depth
    file that ends before the slice does (features 30 frames, depth 18, clip 10:26:2 -> 8 feature rows, 4 depth rows)."""
    rng = np.random.default_rng(9)
    f = rng.standard_normal((30, 16)).astype(np.float32)
    d_long = rng.random((60, 1, 6, 8)).astype(np.float32)
    d_short = rng.random((18, 1, 6, 8)).astype(np.float32)
    for name, arr in (("f", f), ("dl", d_long), ("ds", d_short)):
        np.save(tmp_path / f"{name}.npy", arr)
    fp, dl, ds = (str(tmp_path / f"{n}.npy") for n in ("f", "dl", "ds"))
    rd = NpyClipReader(pin=False, workers=0)
    clips = [(fp, dl, 0, 20, 2, 25, 56), (fp, dl, 3, 12, 1, 7), (fp, ds, 10, 26, 2)]
    got_f, got_d = rd.batch(clips)
    ref_f, ref_d = _reference_collate(None, clips)
    assert got_f.shape == ref_f.shape and got_d.shape == ref_d.shape
    assert torch.equal(got_f, ref_f) and torch.equal(got_d, ref_d)
    assert got_f.shape[1] == 10 and got_d.shape[1] == 10
    only = [clips[2]]
    got_f, got_d = rd.batch(only)
    ref_f, ref_d = _reference_collate(None, only)
    assert got_f.shape[1] == 8 and got_d.shape[1] == 4
    assert torch.equal(got_f, ref_f) and torch.equal(got_d, ref_d)
