"""`from model.futr_safuser_tokenfusion import FUTR` (reference: main_darai.py:25,29) -> r3d_amd.model.futr_safuser_tokenfusion."""
from r3d_amd.model.futr_safuser_tokenfusion import FUTR, CMFuser  # noqa: F401
