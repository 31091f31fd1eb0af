"""Config surface of the reference (opts.py:1-110): the same module-level ``parser`` singleton with the same flag names
and defaults (including ``--predict`` defaulting to the truthy string 'predict', opts.py:13, and the utkinects
dataset paths that are un-commented there).  New flags default to the reference's behaviour."""
import argparse

parser = argparse.ArgumentParser()
parser.add_argument("--model", default="futr", help="model type")
parser.add_argument("--mode", default="train_eval", help='select action: ["train", "predict", "train_eval"]')
parser.add_argument("--dataset", type=str, default="utkinects")
parser.add_argument("--predict", "-p", action="store_true", help="predict for whole videos mode", default="predict")
# dataset (utkinects block, opts.py:30-38)
parser.add_argument("--mapping_file", default="./datasets/utkinect/mapping_l2_changed.txt")
parser.add_argument("--features_path", default="./datasets/utkinect/features_img/")
parser.add_argument("--gt_path", default="./datasets/utkinect/groundTruth/")
parser.add_argument("--split", default="1", help="split number")
parser.add_argument("--file_path", default="./datasets/utkinect/splits")
parser.add_argument("--model_save_path", default="./save_dir/models/transformer")
parser.add_argument("--results_save_path", default="./save_dir/results/transformer")
parser.add_argument("--task", type=str, help="Next Action Anticipation/long-term anticipation", default="long")
# training (opts.py:72-86)
parser.add_argument("--batch_size", type=int, default=8)
parser.add_argument("--test_batch_size", type=int, default=1)
parser.add_argument("--epochs", type=int, default=60)
parser.add_argument("--warmup_epochs", type=int, default=10)
parser.add_argument("--workers", type=int, default=8)
parser.add_argument("--lr", type=float, default=1e-3)
parser.add_argument("--lr_mul", type=float, default=2.0)
parser.add_argument("--weight_decay", type=float, default=5e-3)
parser.add_argument("-warmup", "--n_warmup_steps", type=int, default=500)
parser.add_argument("--cpu", action="store_true", help="run in cpu (rejected by r3d_amd: there is no CPU path)")
parser.add_argument("--sample_rate", type=int, default=1)
parser.add_argument("--obs_perc", default=30)
parser.add_argument("--n_query", type=int, default=8)
# FUTR (opts.py:89-96)
parser.add_argument("--n_head", type=int, default=8)
parser.add_argument("--hidden_dim", type=int, default=128)
parser.add_argument("--n_encoder_layer", type=int, default=2)
parser.add_argument("--n_decoder_layer", type=int, default=1)
parser.add_argument("--dropout", type=float, default=0.5)
parser.add_argument("--input_dim", type=int, default=2048)
# model (opts.py:99-103)
parser.add_argument("--seg", action="store_true", help="action segmentation", default=True)
parser.add_argument("--anticipate", action="store_true", help="future anticipation", default=True)
parser.add_argument("--pos_emb", action="store_true", help="positional embedding", default=True)
parser.add_argument("--max_pos_len", type=int, default=2000, help="position embedding number for linear interpolation")
parser.add_argument("--temperature", type=float, default=0.07)
parser.add_argument("--input_type", default="i3d_transcript", help='select input type: ["decoded", "gt"]')
parser.add_argument("--runs", default=0, help="save runs")
# ---- additions of this build (defaults reproduce the reference) ---------------------------------------------------
parser.add_argument("--erank_every", type=int, default=0,
                    help="measure the effective rank of the fused token matrix every N steps (0 = never)")
parser.add_argument("--erank_weight", type=float, default=0.0,
                    help="rank-enhancing penalty: total loss -= erank_weight * effective_rank(fused tokens) "
                         "(0 = the reference's loss; the reference describes the quantity, README.md:8-14, but never computes it)")
parser.add_argument("--no_graph_steps", dest="graph_steps", action="store_false", default=True,
                    help="enqueue every training step launch by launch instead of replaying it as a hipGraph (one GPU)")
parser.add_argument("--restore_train_mode", action="store_true", default=False,
                    help="call model.train() after validate(); the reference does not (train_proposed_depth.py:53,235)")
parser.add_argument("--min_batch", type=int, default=8,
                    help="batches smaller than this are skipped, as in the reference (train_proposed_depth.py:148)")
parser.add_argument("--torch_collectives", action="store_true", default=False,
                    help="multi-GPU only: gradient exchanges through torch.distributed (three graphs per step) instead of RCCL "
                         "enqueued on the launch stream inside the step's one graph (r3d_amd/rccl.py)")
parser.add_argument("--pixel_shard", action="store_true", default=False,
                    help="multi-GPU only: shard depth_projection.weight (and its AdamW state) over pixels across ranks "
                         "instead of all-reducing its gradient (r3d_amd/parallel.py); same mathematics")
