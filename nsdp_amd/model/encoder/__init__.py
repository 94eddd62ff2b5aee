"""Encoder registry (reference model/encoder/__init__.py:4-7)."""
from .pointnetplusplus import PointNetPlusPlusEncoder
from .pointransformer import PointTransformerEncoder

encoder_dict = {
    "pointnet++": PointNetPlusPlusEncoder,
    "pointransformer": PointTransformerEncoder,
}
