"""TEST INFRASTRUCTURE -- RNG-free, bit-portable synthetic parameters and clip batches.

Every value is produced by an integer hash (splitmix64 finaliser) evaluated in
numpy uint64 arithmetic and mapped to a 24-bit fraction, so the float32 values are
bit-identical on every machine / numpy / torch build.  This lets the golden
fixtures (made in the build container by importing the reference) stay valid on the
GPU box, where the reference is absent and inputs/parameters are regenerated.

Shapes follow the batch contract of the reference:
  data/basedataset_darai_depth.py:174-206 (item dict + my_collate) and
  train/train_proposed_depth.py:131 (the 5-tuple unpack).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def hash_bits(n, stream):
    """n 64-bit hashes for counter 0..n-1 of stream `stream` (any python int)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix(np.array([stream & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        return _splitmix(idx ^ key)


def uniform01(n, stream):
    """float32 in [0,1): 24-bit fraction, exactly representable."""
    h = hash_bits(n, stream)
    return ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def symmetric(n, stream):
    """float32 in [-1,1), exactly representable (23-bit fraction + sign)."""
    return (uniform01(n, stream).astype(np.float64) * 2.0 - 1.0).astype(np.float32)


def randint(n, hi, stream):
    """int64 in [0,hi)."""
    h = hash_bits(n, stream)
    return ((h >> np.uint64(33)) % np.uint64(hi)).astype(np.int64)


# ----------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------
def fill_value(name, shape, index):
    """Analytic, RNG-free fill for parameter number `index` (named_parameters order) of the
    reference FUTR (model/futr_safuser_tokenfusion.py:103-152).  Full-rank by construction."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = symmetric(n, 0xA5000000 + index)
    leaf = name.split(".")[-1]
    is_norm = ("norm" in name) or ("layernorm" in name)
    if is_norm and leaf == "weight":
        v = np.float32(1.0) + np.float32(0.25) * u
    elif is_norm and leaf == "bias":
        v = np.float32(0.1) * u
    elif name in ("fuser.bn_rgb.weight", "fuser.bn_depth.weight"):      # BN-blend variant: |gamma| is the selection score
        v = np.float32(1.0) + np.float32(0.5) * u
    elif name == "fuser.alpha":                                         # blend weight, torch.rand-like range
        v = np.float32(0.5) + np.float32(0.45) * u
    elif name == "query_embed.weight":
        v = u
    elif name == "pos_embedding":
        v = np.float32(0.5) * u
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if name == "fuser.modality_token":
            fan_in = 1
        # sqrt(3/fan_in): unit-variance-preserving uniform
        v = (np.float32(np.sqrt(3.0 / fan_in)) * u).astype(np.float32)
    else:
        v = np.float32(0.1) * u
    return v.astype(np.float32).reshape(shape)


def fill_state(names_shapes):
    """names_shapes: iterable of (name, shape) in named_parameters() order -> dict name -> ndarray."""
    out = {}
    for j, (name, shape) in enumerate(names_shapes):
        out[name] = fill_value(name, tuple(shape), j)
    return out


# ----------------------------------------------------------------------------------------
# batches
# ----------------------------------------------------------------------------------------
def make_batch(B, S, n_class, pad_idx, seed, n_query=8, input_dim=2048, depth_hw=(224, 224),
               pad_tail=True, zero_mean_depth=False):
    """One synthetic 5-tuple [features, depth, past_label, trans_dur_future, trans_future_target]
    (numpy), shaped like BaseDataset.my_collate's output (basedataset_darai_depth.py:185-206):
      features [B,S,input_dim] f32 ~ unit variance;  depth [B,S,1,H,W] f32 in [0,1);
      past_label [B,S] int64 in [0,n_class-1) with the last S//8 (>=1) frames of odd clips = pad_idx;
      trans_dur_future [B,Q] f32: fractions summing to 1 over the first nq_b entries, then pad_idx;
      trans_future_target [B,Q] int64: class ids, then NONE (= n_class-1), then pad_idx.
    """
    base = (seed & 0xFFFFFF) << 8
    Hh, Ww = depth_hw
    feats = (np.float32(np.sqrt(3.0)) * symmetric(B * S * input_dim, base + 1)).reshape(B, S, input_dim)
    depth = uniform01(B * S * Hh * Ww, base + 2).reshape(B, S, 1, Hh, Ww)
    if zero_mean_depth:
        depth = (depth - np.float32(0.5)).astype(np.float32)
    lab = randint(B * S, max(n_class - 1, 1), base + 3).reshape(B, S)
    if pad_tail:
        npad = max(S // 8, 1)
        for b in range(1, B, 2):
            lab[b, S - npad:] = pad_idx
    nq = 1 + randint(B, n_query, base + 4)            # 1..Q future segments per clip
    tgt = randint(B * n_query, max(n_class - 1, 1), base + 5).reshape(B, n_query)
    dur = (uniform01(B * n_query, base + 6).reshape(B, n_query) + np.float32(0.05)).astype(np.float32)
    for b in range(B):
        k = int(nq[b])
        if k < n_query:
            tgt[b, k - 1] = n_class - 1               # the appended NONE class (darai_depth.py:156)
            tgt[b, k:] = pad_idx
            dur[b, k:] = 0.0
        d = dur[b, :k].astype(np.float64)
        dur[b, :k] = (d / d.sum()).astype(np.float32)
        if k < n_query:
            dur[b, k:] = np.float32(pad_idx)
    return [feats.astype(np.float32), depth.astype(np.float32), lab.astype(np.int64),
            dur.astype(np.float32), tgt.astype(np.int64)]


# ----------------------------------------------------------------------------------------
# per-video files for the inference loop
# ----------------------------------------------------------------------------------------
def make_video(T, n_actions, seed, input_dim=2048, depth_hw=(24, 32)):
    """One synthetic video in the form the reference's predict() reads (evaluation/predict_utkinects.py:262-272):
    features [T, input_dim] f32, depth [T, 1, H, W] f32 in [0,1), and T ground-truth lines '<image>,<label>,<tag>' whose
    labels are piecewise constant (segments of 3..10 frames) over the names act00 .. act<n_actions-1>."""
    base = (seed & 0xFFFFFF) << 8
    Hh, Ww = depth_hw
    feats = (np.float32(np.sqrt(3.0)) * symmetric(T * input_dim, base + 11)).reshape(T, input_dim)
    depth = uniform01(T * Hh * Ww, base + 12).reshape(T, 1, Hh, Ww)
    seg_len = 3 + randint(T, 8, base + 13)
    seg_lab = randint(T, n_actions, base + 14)
    labels, k = [], 0
    while len(labels) < T:
        labels += [int(seg_lab[k])] * int(seg_len[k])
        k += 1
    labels = labels[:T]
    lines = [f"img_{t:05d}.png,act{labels[t]:02d},x" for t in range(T)]
    return feats.astype(np.float32), depth.astype(np.float32), lines


def write_video_files(root, name, feats, depth, lines):
    """Lays one video out as the reference's dataset directories (predict_utkinects.py:231-233)."""
    import os
    for d in ("groundTruth", "features_img", "features_depth"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    with open(os.path.join(root, "groundTruth", f"{name}.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    np.save(os.path.join(root, "features_img", f"{name}.npy"), feats)
    np.save(os.path.join(root, "features_depth", f"{name}.npy"), depth)
