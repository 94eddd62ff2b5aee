"""Decoder registry (reference model/decoder/__init__.py:5-8)."""
from .crosstransformer_decoder import CrossTransformerDecoder
from .interpolation_decoder import PointInterpDecoder

decoder_dict = {
    "interp": PointInterpDecoder,
    "crossatten": CrossTransformerDecoder,
}
