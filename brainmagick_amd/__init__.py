"""MI355X-native SimpleConv + ClipLoss training hot path."""
from .hip_ops import set_compute_dtype, get_compute_dtype, weights_changed  # noqa: F401
