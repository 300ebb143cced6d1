"""Mirror of ``bm.models`` for the hot path."""
from .simpleconv import SimpleConv  # noqa: F401
from .common import ConvSequence, ChannelMerger, SubjectLayers, FourierEmb, PositionGetter  # noqa: F401
from .features import DeepMel  # noqa: F401
