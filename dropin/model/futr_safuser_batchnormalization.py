"""`from model.futr_safuser_batchnormalization import FUTR` (reference: main_darai.py:25,29) -> r3d_amd.model.futr_safuser_batchnormalization."""
from r3d_amd.model.futr_safuser_batchnormalization import FUTR, CMFuser  # noqa: F401
