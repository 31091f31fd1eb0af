"""`from model.futr_proposed import FUTR` (reference: main_darai.py:22,31) -> r3d_amd.model.futr_proposed."""
from r3d_amd.model.futr_proposed import FUTR  # noqa: F401
