"""`from pl_bolts.optimizers.lr_scheduler import LinearWarmupCosineAnnealingLR` (reference: main_darai.py:13,137) for hosts
without pl_bolts: the restated schedule of r3d_amd.optim (pl_bolts 0.3.4 formula; parity unpinned)."""
from r3d_amd.optim import LinearWarmupCosineAnnealingLR  # noqa: F401
