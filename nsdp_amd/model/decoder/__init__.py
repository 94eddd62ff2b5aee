"""Decoder registry (reference model/decoder/__init__.py:5-8)."""
from .crosstransformer_decoder import CrossTransformerDecoder


class _OutOfScope:
    def __init__(self, name):
        self.name = name

    def __call__(self, *a, **kw):
        raise NotImplementedError(
            f"decoder '{self.name}' is a registry alternate that no shipped NSDP config selects; it is not "
            "part of the MI355X hot path (SURVEY.md section 8 a20)")


decoder_dict = {
    "interp": _OutOfScope("interp"),
    "crossatten": CrossTransformerDecoder,
}
