"""``models.PointDSC`` of the reference (models/PointDSC.py:80-438), served by the HIP implementation: same class name,
constructor signature, ``state_dict`` layout and ``forward(data) -> {'final_trans', 'final_labels', 'M'}``."""
from pointdsc_amd.model import PointDSC  # noqa: F401

__all__ = ["PointDSC"]
