"""r3d_amd -- MI355X-native (gfx950) implementation of olivesgatech/R3D's RGB+Depth token-fusion training path.

Python surface mirrors the reference (model.futr_safuser_tokenfusion.FUTR, train_proposed_depth.train, opts.parser);
all arithmetic runs in hand-written HIP kernels behind the C ABI of include/r3d_hip.h (r3d_amd/_build/libr3d_hip.so).
There is no CPU or PyTorch-op fallback: importing the compute path without the library raises.
"""
__version__ = "0.1.0"
