"""Mirror of ``bm/models/features.py``: models applied to the audio features before the
contrastive loss.  ``DeepMel`` is a ``ConvSequence`` on the mel spectrogram (it makes the
candidates learnable, so ClipLoss also back-propagates into them -- HIP path in
``functional.ClipLossFn``)."""
from .common import ConvSequence


class DeepMel(ConvSequence):
    """bm/models/features.py:15-35; configured by conf/feature_model/deep_mel.yaml."""

    def __init__(self, n_in_channels: int, n_hidden_channels: int, n_hidden_layers: int,
                 n_out_channels: int, **kwargs):
        channels = \
            [n_in_channels] + [n_hidden_channels] * (n_hidden_layers - 1) + [n_out_channels]
        super().__init__(channels, **kwargs)
