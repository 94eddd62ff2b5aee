"""Encoder registry (reference model/encoder/__init__.py:4-7)."""
from .pointransformer import PointTransformerEncoder


class _OutOfScope:
    def __init__(self, name):
        self.name = name

    def __call__(self, *a, **kw):
        raise NotImplementedError(
            f"encoder '{self.name}' is a registry alternate that no shipped NSDP config selects; it is not "
            "part of the MI355X hot path (SURVEY.md section 8 a20)")


encoder_dict = {
    "pointnet++": _OutOfScope("pointnet++"),
    "pointransformer": PointTransformerEncoder,
}
