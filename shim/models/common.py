"""``models.common`` of the reference (models/common.py:7-69), the two functions its callers import, on the device."""
import torch

from pointdsc_amd import ops as _ops


def rigid_transform_3d(A, B, weights=None, weight_threshold=0):
    """models/common.py:7-45: weighted Procrustes, A, B [bs, n, 3] (GPU), weights [bs, n] -> [bs, 4, 4].  As in the reference,
    weights below the threshold are zeroed IN the caller's tensor (:20)."""
    if weights is not None:
        weights[weights < weight_threshold] = 0
    return _ops.rigid_transform_3d(A, B, weights, weight_threshold)


def knn(x, k, ignore_self=False, normalized=True):
    """models/common.py:48-69 is only called from inside the forward (models/PointDSC.py:251), where the HIP path computes the
    neighbour sets of the seeds alone (pdsc_knn_seeds); the all-rows form is not part of the drop-in surface."""
    raise NotImplementedError("models.common.knn: use pointdsc_amd.ops.knn_seeds (neighbours of the seed rows)")
