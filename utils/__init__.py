"""Inference-side helpers; same names as the reference's utils package (utils/__init__.py:1-4)."""
from utils.barycentric import DevicePolicy, get_barycentric_weights_and_indices, get_optimal_action  # noqa: F401
