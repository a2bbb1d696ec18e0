import openpvsg_amd.blocks as _blocks  # noqa: F401  (registers the backend's blocks)
from openpvsg_amd.blocks import (FFN, BaseTransformerLayer, MultiheadAttention,  # noqa: F401
                                 TransformerLayerSequence)
from openpvsg_amd.registry import (build_attention, build_feedforward_network,  # noqa: F401
                                   build_positional_encoding, build_transformer_layer,
                                   build_transformer_layer_sequence)
from .registry import (ATTENTION, FEEDFORWARD_NETWORK, POSITIONAL_ENCODING,  # noqa: F401
                       TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE)
