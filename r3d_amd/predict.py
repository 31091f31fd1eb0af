"""Inference path of the RGB+Depth model for one observed clip: FUTR.forward in eval mode (data-dependent token
selection, futr_safuser_tokenfusion.py:47-50) on the HIP engine, then the decode of evaluation/predict_utkinects.py:300-353
(arg-max segmentation / anticipation labels, duration -> frame expansion).  File handling, logging and the MoC loop of
the reference's predict() stay with the caller; utils.eval_file / mean_over_classes provide the counters."""
import torch

from .utils import expand_durations


@torch.no_grad()
def predict_clip(model, features, depth_features, future_len, none_idx=None):
    """features [T, D] float32, depth_features [T, ...] (frames of one clip, already sub-sampled), on the model's device.
    Returns dict(seg_labels [T], action_labels [Q], frames [future_len], outputs=<the model's output dict>)."""
    was_training = model.training
    model.eval()
    try:
        # (the reference's forward unpacks `src, _ = inputs` outside train mode, futr_safuser_tokenfusion.py:171)
        out = model(inputs=(features.unsqueeze(0), None), depth_features=depth_features.unsqueeze(0), mode="test")
    finally:
        model.train(was_training)
    n_class = out["action"].shape[-1]
    none_idx = n_class - 1 if none_idx is None else none_idx          # NONE = n_class - 1 (:240)
    seg = out["seg"][0].argmax(-1)
    act = out["action"][0].argmax(-1)
    frames = expand_durations(out["action"][0], out["duration"][0], future_len, none_idx)
    return dict(seg_labels=seg, action_labels=act, frames=frames, outputs=out)
