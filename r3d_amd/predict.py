"""Inference path of the RGB+Depth model: the per-video loop of the reference's predict()
(evaluation/predict_utkinects.py:215-396, called at main_darai.py:164 and selected by default through opts.py:13) on the HIP
engine, and the single-clip decode it is built from.

predict(model, vid_list, args, obs_p, n_class, actions_dict, device) keeps the reference's signature, its stdout text and its
return value (anticipation accuracy, segmentation accuracy); the MoC lines are printed as the reference prints them.  What
the reference hard-codes and this version takes as keyword arguments (with the reference's values as defaults where they
are usable): the dataset root (`./datasets/<name>`, :222-233) and the per-prediction log (the reference writes it to an
absolute path on its author's machine, :250; here `log_dir=None` writes none).  File access goes through `reader` (default:
numpy / text files laid out as the reference's dataset directories), so the loop runs on synthetic per-video files.

The model call differs in one documented way: the reference passes the bare feature tensor (:302) to a forward that
unpacks `src, _ = inputs` outside train mode (model/futr_safuser_tokenfusion.py:171) -- an inconsistency of the snapshot
(SURVEY F4); here the tuple `(features, None)` is passed.
"""
import copy
import os

import numpy as np
import torch

from .utils import eval_file, expand_durations

EVAL_P = (0.1, 0.2, 0.3, 0.5)                 # predict_utkinects.py:235
PRED_P = 0.5                                  # :236
EXCLUDE_CLASS_IDX = 16                        # :319 (the 6th positional argument of weighted_accuracy_without_gif)
DATASET_DIRS = dict(breakfast="breakfast", darai="darai", utkinects="utkinect")        # :223-230
DATASET_DIRS["50salads"] = "50salads"


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


def weighted_accuracy_without_gif(log, pred, gold, t_n_label, actions_dict, exclude_class_idx=None, label_base=(),
                                  weight_same=1.0, weight_different=10.0):
    """predict_utkinects.py:105-137: weight 10 when the first anticipated label differs from the last observed one, else 1;
    ground-truth labels equal to exclude_class_idx are skipped.  pred: the Q = 8 anticipated class ids of one clip."""
    assert len(pred) == 8                                                           # (:111)
    correct = total = 0.0
    weight = weight_different if gold[0] != t_n_label else weight_same
    if log is not None:
        log.write("input label: \n")
        for lb in label_base:
            log.write(f"{lb}\n")
    for i in range(min(len(gold), len(pred))):
        gt = actions_dict[gold[i].replace(" ", "")]
        if exclude_class_idx is not None and gt == exclude_class_idx:
            continue
        if int(pred[i]) == gt:
            correct += weight
        total += weight
        if log is not None:
            log.write(f"\t{gold[i].replace(' ', '')}\t{int(pred[i])}\t{weight}\n")
    return correct / total if total > 0 else 0


def normal_accuracy_without_gif(pred, gold, actions_dict):
    """predict_utkinects.py:140-165: plain frame accuracy of the segmentation labels over the observed frames."""
    assert len(gold) == len(pred)
    ok = sum(1 for i in range(len(gold)) if int(pred[i]) == actions_dict[gold[i].replace(" ", "")])
    return ok / len(gold)


class DatasetFiles:
    """The reference's dataset layout under one root (:231-233): groundTruth/<video>.txt ('<image>,<L2 label>,<...>' per
    frame, :262-266), features_img/<video>.npy [T, 2048], features_depth/<video>.npy [T, ...]."""

    def __init__(self, root):
        self.gt, self.feat, self.depth = (os.path.join(root, d) for d in ("groundTruth", "features_img", "features_depth"))

    def exists(self, base):
        g, f, d = (os.path.exists(p) for p in self.paths(base))
        return not (not g or not f and d)                        # the reference's break condition, verbatim (:257)

    def paths(self, base):
        return (os.path.join(self.gt, f"{base}.txt"), os.path.join(self.feat, f"{base}.npy"),
                os.path.join(self.depth, f"{base}.npy"))

    def load(self, base):
        g, f, d = self.paths(base)
        with open(g, "r") as fh:
            valid = [ln.strip() for ln in fh.readlines() if len(ln.strip().split(",")) == 3]
        images = [ln.split(",")[0] for ln in valid]
        labels = [ln.split(",")[1] for ln in valid]
        return g, images, labels, np.load(f), np.load(d)


def predict(model, vid_list, args, obs_p, n_class, actions_dict, device, data_path=None, log_dir=None, reader=None,
            details=None):
    """The reference's predict() (evaluation/predict_utkinects.py:215-396).  Returns (anticipation accuracy, segmentation
    accuracy) averaged over the videos.  details (optional list): receives one dict per video -- labels, the anticipated
    frame sequence and the per-horizon (true, false) class counts -- for tests and callers that want more than the prints."""
    acc = seg_acc = 0.0
    idx = 0
    model.eval()
    if data_path is None:
        data_path = os.path.join("./datasets", DATASET_DIRS.get(args.dataset, args.dataset))
    files = reader if reader is not None else DatasetFiles(data_path)
    sample_rate = args.sample_rate
    NONE = n_class - 1
    T_actions = np.zeros((len(EVAL_P), len(actions_dict)))
    F_actions = np.zeros((len(EVAL_P), len(actions_dict)))
    with_none = copy.deepcopy(actions_dict)
    with_none["NONE"] = NONE
    names = list(with_none.keys())
    values = list(with_none.values())
    print(len(vid_list))
    log = None
    if log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
    with torch.no_grad():
        for vid in vid_list:
            base = vid.split("/")[-1].split(".")[0]
            if log_dir is not None:                 # (the reference re-opens one log per video in "w" mode, :250)
                log = open(os.path.join(log_dir, f"gt_pred_log_{obs_p}.txt"), "w")
                log.write("--------------------------------------\n")
                log.write("gt file\tGround Truth (GT)\tPrediction (Pred)\n")
            try:
                if not files.exists(base):
                    break
                gt_file, image_path, all_content, features, depth_features = files.load(base)
                vid_len = len(all_content)
                past_len = int(obs_p * vid_len)
                future_len = int(PRED_P * vid_len)
                past_seq = all_content[:past_len]
                inputs = torch.as_tensor(np.ascontiguousarray(features[:past_len][::sample_rate]), dtype=torch.float32)
                depth_in = torch.as_tensor(np.ascontiguousarray(depth_features[:past_len][::sample_rate]),
                                           dtype=torch.float32)
                future_content = all_content[past_len: past_len + future_len][::sample_rate]
                label_base = past_seq[::sample_rate]
                if log is not None:
                    log.write(f"\nimage base: \n{image_path[:past_len][::sample_rate]}\n")
                outputs = model(inputs=(inputs.to(device).unsqueeze(0), None), depth_features=depth_in.to(device).unsqueeze(0),
                                mode="test", epoch=idx, idx=obs_p)
                seg_label = outputs["seg"].reshape(-1, outputs["seg"].shape[-1]).max(-1)[1].cpu()
                seg_one = normal_accuracy_without_gif(seg_label, label_base, actions_dict)
                seg_acc += seg_one
                output_label = outputs["action"].max(-1)[1].cpu()                         # [1, Q]
                if log is not None:
                    log.write(f"{gt_file}\n------------------\n{len(past_seq)}\n")
                ant_one = weighted_accuracy_without_gif(log, output_label[0], future_content, past_seq[-1], actions_dict,
                                                        EXCLUDE_CLASS_IDX, label_base)
                acc += ant_one
                idx += 1
                # duration -> one label per anticipated frame (:322-353); queries from the first NONE on carry no duration
                predicted = expand_durations(outputs["action"][0], outputs["duration"][0], future_len, NONE)
                prediction = list(past_seq) + [names[values.index(int(p))] for p in predicted]
                counts = []
                for i, p in enumerate(EVAL_P):                                           # (:364-371)
                    eval_len = int((obs_p + p) * vid_len)
                    n_t, n_f = eval_file(all_content, prediction[:eval_len], obs_p, actions_dict)
                    T_actions[i] += n_t
                    F_actions[i] += n_f
                    counts.append((n_t, n_f))
                if details is not None:
                    details.append(dict(video=base, seg_labels=seg_label, action_labels=output_label[0], frames=predicted,
                                        seg_acc=seg_one, ant_acc=ant_one, counts=counts))
            finally:
                if log is not None:
                    log.close()
                    log = None
    ant = acc / idx
    seg = seg_acc / idx
    print("!!!!!!!!!!!!! ant Acc: ", ant)
    print("@!@!@!@!@!@!@ seg Acc: ", seg)
    total = T_actions + F_actions
    for i in range(len(EVAL_P)):                                                         # (:378-393)
        a_sum, n = 0.0, 0
        for j in range(len(actions_dict)):
            if total[i, j] != 0:
                a_sum += float(T_actions[i, j] / total[i, j])
                n += 1
        print(f"obs. {int(100 * obs_p)}% pred. {int(100 * EVAL_P[i])}% --> MoC: {float(a_sum) / n:.4f}")
    print("--------------------------------")
    return ant, seg
