"""Host-side helpers around the hot path: the inference decode of predict(), checkpoint
interchange with the reference's `.ckpt` files, and a device-resident input pipeline.  Plumbing only: nothing here is on
the timed path and nothing computes on the CPU what the HIP path computes.

Reference counterparts: utils.py:325-328 (normalize_duration), :330-339 (read_mapping_dict), :341-356 (eval_file),
evaluation/predict_utkinects.py:331-353 (duration -> frame expansion), train/train_proposed_depth.py:243-248 and
main_darai.py:133,161 (checkpoint names and the `module.` prefix of nn.DataParallel)."""
import collections

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------------
# inference decode
# ---------------------------------------------------------------------------------------------------------------------
def normalize_duration(input, mask):
    """exp(x) * mask, L1-normalised over the last dim (utils.py:325-328; eps 1e-12 as F.normalize)."""
    x = torch.exp(input) * mask
    return x / x.abs().sum(dim=-1, keepdim=True).clamp_min(1e-12)


def read_mapping_dict(file_path):
    """'<index> <name>' per line -> {name: index} (utils.py:330-339)."""
    out = {}
    with open(file_path, "r") as fh:
        for line in fh.read().split("\n")[:-1]:
            idx, name = line.split()[:2]
            out[name] = int(idx)
    return out


def eval_file(gt_content, recog_content, obs_percentage, classes):
    """Per-class true / false frame counts over the anticipated span (utils.py:341-356)."""
    last = min(len(recog_content), len(gt_content))
    start = int(obs_percentage * len(gt_content))
    n_t, n_f = np.zeros(len(classes)), np.zeros(len(classes))
    for g, r in zip(gt_content[start:last], recog_content[start:last]):
        g = g.replace(" ", "")
        if g == r:
            n_t[classes[g]] += 1
        else:
            n_f[classes[g]] += 1
    return n_t, n_f


def mean_over_classes(n_t, n_f):
    """MoC as printed by predict() (predict_utkinects.py:381-392): mean of per-class accuracy over classes that occur."""
    tot = n_t + n_f
    ok = tot != 0
    return float((n_t[ok] / tot[ok]).mean()) if ok.any() else 0.0


def expand_durations(action_logits, duration, future_len, none_idx):
    """One clip's decoder outputs -> a label per anticipated frame (predict_utkinects.py:331-353).
    action_logits [Q, K], duration [Q] (raw fc_len outputs); queries from the first NONE on carry no duration."""
    labels = action_logits.argmax(-1)
    q = labels.numel()
    mask = torch.ones(q, dtype=duration.dtype, device=duration.device)
    none_at = (labels == none_idx).nonzero()
    if none_at.numel():
        mask[int(none_at[0]):] = 0
    dur = normalize_duration(duration.reshape(1, -1), mask.reshape(1, -1)).reshape(-1)
    seg = (0.5 + future_len * dur).long().cpu().tolist()
    lab = labels.cpu().tolist()
    out = torch.ones(future_len, dtype=torch.long)          # the reference initialises with class 1 (:343)
    start = 0
    for i in range(q):
        out[start:start + seg[i]] = lab[i]
        start += seg[i]
        if i == q - 1:
            out[start - seg[i]:] = lab[i]                   # the last action runs to the end (:349-350)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------------------------------
def strip_module_prefix(state_dict):
    """nn.DataParallel checkpoints (main_darai.py:133) prefix every key with 'module.'."""
    return collections.OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())


def load_checkpoint(model, path, strict=True):
    """Loads a reference `seed_{seed}_best.ckpt` / `..._checkpoint{epoch}.ckpt` (a plain state_dict, with or without the
    'module.' prefix) into r3d_amd's FUTR.  weights_only: nothing in the file is executed."""
    sd = strip_module_prefix(torch.load(path, map_location="cpu", weights_only=True))
    target = model.module if hasattr(model, "module") else model
    return target.load_state_dict(sd, strict=strict)


def save_checkpoint(model, path, data_parallel_keys=False):
    """Writes the state_dict under the reference's key names; data_parallel_keys=True adds the 'module.' prefix so
    that the reference's DataParallel-wrapped predict scripts (main_darai.py:161) load it unchanged."""
    target = model.module if hasattr(model, "module") else model
    sd = target.state_dict()
    if data_parallel_keys:
        sd = collections.OrderedDict(("module." + k, v) for k, v in sd.items())
    torch.save(sd, path)


# ---------------------------------------------------------------------------------------------------------------------
# device-resident input pipeline
# ---------------------------------------------------------------------------------------------------------------------
class NpyClipReader:
    """Feature / depth rows of a batch of clips straight from the per-video `.npy` files the reference's preprocessing
    writes (data/nturgbd-preprocess-depth.py:118, utkinect-preprocess-depth.py:133: `[frames, 1, h, w]` float32; RGB
    features `[frames, 2048]`) into PINNED, reused staging buffers -- the host half of "depth .npy -> pinned -> async
    copy" (SURVEY.md 8(f).2).  The reference goes file -> ndarray (np.load, whole video) -> slice -> torch.tensor (copy)
    -> pad_sequence (copy) -> .to(device) (pageable, synchronous): basedataset_darai_depth.py:110-130,176,199-203 and
    train_proposed_depth.py:132-137.  Here the files are memory-mapped (only the sampled frames are read), each sampled
    frame is copied ONCE, into its place in a pinned `[B, S_max, ...]` buffer (zero padding as `pad_sequence(batch_first=
    True, padding_value=0)` leaves it), and InputPrefetcher's side stream moves that buffer while the previous step runs.
    Labels / targets are the caller's (a few hundred bytes; the transcript logic of BaseDataset stays the reference's).

    clip spec = (feature_file, depth_file, start, stop, step): rows `start:stop:step` of both files, as the dataset's
    observed-range slice and sample_rate produce them (:129-130).  `slots` buffers rotate, so a batch stays valid while
    the next `slots - 1` are being filled (InputPrefetcher holds at most two)."""

    def __init__(self, slots=3, pin=None, workers=8):
        self.slots, self._bufs, self._turn = max(2, int(slots)), {}, 0
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        self._maps = {}
        # one copy job per clip: numpy's copies release the GIL, and a single thread moves ~10 GB/s -- 2.7 ms for the
        # 27 MB batch of the bench shape, ten training steps' worth
        self._pool = None
        if workers and workers > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=int(workers))

    def _open(self, path):
        m = self._maps.get(path)
        if m is None:
            m = np.load(path, mmap_mode="r")                         # .npy only: nothing in the file is executed
            if len(self._maps) >= 64:                                # (bounded: a descriptor and a mapping per file)
                self._maps.pop(next(iter(self._maps)))
            self._maps[path] = m
        return m

    def _buf(self, key, shape):
        ring = self._bufs.setdefault(key, [])
        slot = self._turn % self.slots
        while len(ring) <= slot:
            ring.append(None)
        b = ring[slot]
        if b is None or tuple(b.shape[1:]) != tuple(shape[1:]) or b.shape[0] < shape[0]:
            b = torch.empty(shape, dtype=torch.float32, pin_memory=self.pin)
            ring[slot] = b
        return b[:shape[0]]

    @staticmethod
    def _rows(n_file, start, stop, step, offset=0, end=None):
        """Row indices of file[offset:end][start:stop:step] as a range over the FILE's rows."""
        lo, hi, _ = slice(offset, end, 1).indices(n_file)            # the trimmed view (basedataset_darai_depth.py:112-114)
        r = range(*slice(start, stop, step).indices(max(hi - lo, 0)))
        return range(lo + r.start, lo + r.stop, r.step)

    def batch(self, clips):
        """clips: (feature_file, depth_file, start, stop, step[, depth_offset[, depth_end]]) -- the features rows are
        file[start:stop:step]; the depth rows are file[depth_offset:depth_end][start:stop:step], the trim the reference applies
        to the per-recording depth file before the observed slice (basedataset_darai_depth.py:110-130).
        -> (features [B, S_f, D], depth [B, S_d, *frame_shape]) float32 in pinned memory, each zero-padded to ITS longest
        clip, as pad_sequence does per tensor (:199-203): a depth file that ends before the slice does only shortens that
        clip's depth rows."""
        maps = [(self._open(c[0]), self._open(c[1])) for c in clips]
        rows_f = [self._rows(f.shape[0], c[2], c[3], c[4]) for c, (f, _) in zip(clips, maps)]
        rows_d = [self._rows(d.shape[0], c[2], c[3], c[4], *(c[5:7])) for c, (_, d) in zip(clips, maps)]
        Sf = max((len(r) for r in rows_f), default=0)
        Sd = max((len(r) for r in rows_d), default=0)
        f0, d0 = maps[0]
        feats = self._buf("f", (len(clips), Sf) + tuple(f0.shape[1:]))
        depth = self._buf("d", (len(clips), Sd) + tuple(d0.shape[1:]))
        self._turn += 1
        fn, dn = feats.numpy(), depth.numpy()                        # (views of the pinned storage)

        def fill(b):                # one copy job per clip (jobs of 8 frames on 16 threads measured slower: 1035 vs 773 us)
            for out, src, r, S in ((fn, maps[b][0], rows_f[b], Sf), (dn, maps[b][1], rows_d[b], Sd)):
                n = len(r)
                if n:
                    out[b, :n] = src[slice(r.start, r.stop, r.step)]
                out[b, n:S] = 0
        jobs = range(len(clips))
        if self._pool is not None and len(jobs) > 1:
            list(self._pool.map(fill, jobs))
        else:
            for b in jobs:
                fill(b)
        return feats, depth


class InputPrefetcher:
    """Wraps any iterable of the 5-tuple batches of BaseDataset.my_collate (basedataset_darai_depth.py:185-206) and keeps
    the NEXT batch on the device while the current step runs: pinned staging buffers + async copies on a side stream.
    The reference moves every batch synchronously inside the step (train_proposed_depth.py:132-137): 25.7 MB of depth
    per step at the bench shape, 0.4 ms over PCIe -- longer than the whole HIP step."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device)

    def _stage(self, data):
        if data is None:
            return None
        out = []
        with torch.cuda.stream(self.stream):
            for i, t in enumerate(data):
                t = torch.as_tensor(t)
                if i in (2, 4):
                    t = t.long()                                    # labels: float for NTU / UTKinect (cast at utils.py:365)
                elif t.dtype != torch.float32:
                    t = t.float()
                if not t.is_cuda:
                    t = t.pin_memory() if not t.is_pinned() else t
                out.append(t.to(self.device, non_blocking=True).contiguous())
        return out

    def __iter__(self):
        end = object()
        it = iter(self.loader)
        raw = next(it, end)
        nxt = self._stage(raw) if raw is not end else end
        while nxt is not end:
            cur = nxt
            if cur is not None:                                     # (None items pass through: train() skips them, :128)
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for t in cur:
                    t.record_stream(torch.cuda.current_stream(self.device))
            raw = next(it, end)                                     # stage the following batch before handing this one out
            nxt = self._stage(raw) if raw is not end else end
            yield cur

    def __len__(self):
        return len(self.loader)
