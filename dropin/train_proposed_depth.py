"""`from train_proposed_depth import train` (reference: train/train_proposed_depth.py, main_darai.py:39) -> r3d_amd."""
from r3d_amd.train_proposed_depth import train, validate, get_last_non_padding_labels, weighted_accuracy  # noqa: F401
