"""pointdsc_amd -- MI355X-native (gfx950) implementation of PointDSC's test-time outlier-rejection
hot path behind the reference's ``PointDSC(...).forward(data) -> dict`` boundary.

Only what the path needs lives here:
  csrc/        hand-written HIP kernels + the C-ABI shared library ``libpointdsc_hip.so``
  _lib.py      ctypes binding of include/pointdsc_hip.h (fails loudly when the library is missing)
  model.py     ``PointDSC`` nn.Module: reference constructor, state_dict layout and forward contract
  ops.py       stage-level tensor wrappers (``rigid_transform_3d``, ``knn`` ...) over the C-ABI
  sharding.py  one-process-per-GPU sharding of pair batches + the single RCCL pose gather
  synthetic.py seeded synthetic correspondence sets / weights (tests + bench)
"""
from .model import PointDSC  # noqa: F401

__all__ = ["PointDSC"]
