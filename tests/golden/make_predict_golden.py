#!/usr/bin/env python3
"""Generates tests/golden/predict_golden.npz (this container only; the reference does not travel).

Three synthetic videos (oracle/synth.py: make_video) go through the ORACLE's test-mode forward and the decode of the
reference's predict() (evaluation/predict_utkinects.py:215-396); the per-class true / false frame counts come from the
REFERENCE's own utils.eval_file (utils.py:341-356) and the duration normalisation from its utils.normalize_duration
(:325-328), both imported here.  evaluation/predict_utkinects.py itself cannot be imported (imageio / cv2 are absent and
its log path is absolute on the author's machine), so the loop around those two functions -- observed / anticipated
slicing :276-291, arg-max labels :306-316, duration -> frames :322-353, the two accuracy helpers :105-165 -- is restated
here line by line: parity of that glue is unpinned, the counters and the normalisation are the reference's.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
import utils as R  # noqa: E402  (the reference's utils.py)
from oracle import futr_oracle as O, synth  # noqa: E402

H, K, Q, HEADS, PIX = 128, 17, 8, 8, (24, 32)
VIDEOS = [("vid_a", 48, 901), ("vid_b", 64, 902), ("vid_c", 40, 903)]
OBS = [0.2, 0.3]
SAMPLE_RATE = 2


def main():
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(K, H, K + 1, torch.device("cpu"), args, n_query=Q, n_head=HEADS, num_encoder_layers=2, num_decoder_layers=1,
                 depth_pixels=PIX[0] * PIX[1])
    names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    params = {n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}
    actions = {f"act{i:02d}": i for i in range(K - 1)}
    with_none = dict(actions, NONE=K - 1)
    inv = {v: k for k, v in with_none.items()}
    out = {}
    for obs_p in OBS:
        T_act = np.zeros((4, len(actions)))
        F_act = np.zeros((4, len(actions)))
        ant_sum = seg_sum = 0.0
        for name, T, seed in VIDEOS:
            feats, depth, lines = synth.make_video(T, K - 1, seed, depth_hw=PIX)
            labels = [ln.split(",")[1] for ln in lines]
            vid_len = len(labels)
            past_len, future_len = int(obs_p * vid_len), int(0.5 * vid_len)
            past_seq = labels[:past_len]
            x = torch.from_numpy(feats[:past_len][::SAMPLE_RATE]).unsqueeze(0)
            d = torch.from_numpy(depth[:past_len][::SAMPLE_RATE]).unsqueeze(0)
            with torch.no_grad():
                o, _ = O.forward(params, (x, None), d, "test", K + 1, HEADS, 1)
            seg_lab = o["seg"].reshape(-1, K).max(-1)[1]
            act_lab = o["action"].max(-1)[1]                                   # [1, Q]
            label_base = past_seq[::SAMPLE_RATE]
            seg_acc = sum(int(seg_lab[i]) == actions[label_base[i]] for i in range(len(label_base))) / len(label_base)
            future_content = labels[past_len:past_len + future_len][::SAMPLE_RATE]
            w = 10.0 if future_content[0] != past_seq[-1] else 1.0
            tc = tl = 0.0
            for i in range(min(len(future_content), Q)):
                gt = actions[future_content[i]]
                if gt == 16:
                    continue
                tc += w if int(act_lab[0, i]) == gt else 0.0
                tl += w
            ant = tc / tl if tl > 0 else 0
            # duration -> frames with the reference's normalize_duration (:322-353)
            none_idx = None
            for i in range(Q):
                if int(act_lab[0, i]) == K - 1:
                    none_idx = i
                    break
            dur = o["duration"]
            if none_idx is not None:
                mask = torch.ones(act_lab.shape, dtype=torch.bool)
                mask[0, none_idx:] = False
                dur = R.normalize_duration(dur, mask)
            else:
                dur = R.normalize_duration(dur, torch.ones_like(dur))
            pred_len = (0.5 + future_len * dur).squeeze(-1).long()
            pred_len = torch.cat((torch.zeros(1), pred_len.squeeze()), dim=0)
            predicted = torch.ones(future_len)
            action = act_lab.squeeze()
            for i in range(len(action)):
                predicted[int(pred_len[i]): int(pred_len[i] + pred_len[i + 1])] = action[i]
                pred_len[i + 1] = pred_len[i] + pred_len[i + 1]
                if i == len(action) - 1:
                    predicted[int(pred_len[i]):] = action[i]
            prediction = list(past_seq) + [inv[int(v)] for v in predicted]
            for i, p in enumerate((0.1, 0.2, 0.3, 0.5)):
                eval_len = int((obs_p + p) * vid_len)
                n_t, n_f = R.eval_file(labels, prediction[:eval_len], obs_p, actions)      # the reference's counter
                T_act[i] += n_t
                F_act[i] += n_f
            ant_sum += ant
            seg_sum += seg_acc
            key = f"{name}_{obs_p}"
            out[key + "_seg"] = seg_lab.numpy()
            out[key + "_act"] = act_lab[0].numpy()
            out[key + "_frames"] = predicted.long().numpy()
            out[key + "_margins"] = np.array([float((o["seg"].reshape(-1, K).topk(2, -1)[0].diff(dim=-1)).abs().min()),
                                              float((o["action"][0].topk(2, -1)[0].diff(dim=-1)).abs().min())])
        tot = T_act + F_act
        moc = []
        for i in range(4):
            vals = [T_act[i, j] / tot[i, j] for j in range(len(actions)) if tot[i, j] != 0]
            moc.append(float(np.mean(vals)))
        out[f"obs{obs_p}_T"] = T_act
        out[f"obs{obs_p}_F"] = F_act
        out[f"obs{obs_p}_moc"] = np.array(moc)
        out[f"obs{obs_p}_ant_seg"] = np.array([ant_sum / len(VIDEOS), seg_sum / len(VIDEOS)])
        print(obs_p, "ant/seg", out[f"obs{obs_p}_ant_seg"], "MoC", moc)
    out["meta"] = json.dumps(dict(H=H, K=K, Q=Q, heads=HEADS, pix=PIX, videos=VIDEOS, obs=OBS, sample_rate=SAMPLE_RATE))
    np.savez(os.path.join(HERE, "predict_golden.npz"), **out)
    print("wrote predict_golden.npz")


if __name__ == "__main__":
    main()
