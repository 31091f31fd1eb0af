"""`from predict_darai import predict` (reference: evaluation/predict_darai.py, main_darai.py:42-47) -> r3d_amd.predict."""
from r3d_amd.predict import predict, predict_clip  # noqa: F401
