"""`from utils import read_mapping_dict` (reference: utils.py:325-356, main_darai.py:16) -> r3d_amd.utils."""
from r3d_amd.utils import read_mapping_dict, normalize_duration, eval_file  # noqa: F401
