"""`from model.futr_unsupervised_depth import FUTR` (reference: main_darai.py:22,31) -> r3d_amd.model.futr_unsupervised_depth."""
from r3d_amd.model.futr_unsupervised_depth import FUTR  # noqa: F401
