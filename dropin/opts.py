"""`from opts import parser` (reference: opts.py:2, main_darai.py:12) -> r3d_amd.opts."""
from r3d_amd.opts import parser  # noqa: F401
