"""Deterministic synthetic correspondence sets and weights.

There is no network on the build/GPU boxes, so neither the 3DMatch/KITTI data nor the
released snapshots exist here.  Every test, the golden-vector
generator and ``bench.py`` therefore draw their inputs from this module.

Everything is drawn from ``numpy.random.RandomState`` (legacy MT19937 streams are frozen
across numpy versions), never from torch's RNG, so the build container and the GPU box
produce bit-identical inputs.

The pair recipe follows SURVEY.md Appendix A / section 8(d): random rigid motion, uniform
source points in a cube of side ``scale``, Gaussian noise on the inliers, uniformly random
targets for the outliers, ``corr_pos = concat(src, tgt) - mean`` exactly as the reference
data loaders build it (reference datasets/ThreeDMatch.py:305-308, demo_registration.py:107-108).
"""
from __future__ import annotations

import numpy as np
import torch


def random_rotation(rs: np.random.RandomState) -> np.ndarray:
    q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.astype(np.float32)


def make_pair(num_corr: int, inlier_ratio: float = 0.2, noise: float = 0.01,
              scale: float = 3.0, seed: int = 0):
    """One synthetic putative-correspondence set.

    Returns dict of float32 torch CPU tensors:
      corr_pos [1,N,6], src_keypts [1,N,3], tgt_keypts [1,N,3], gt_trans [1,4,4], gt_labels [1,N]
    """
    rs = np.random.RandomState(seed)
    R = random_rotation(rs)
    t = (rs.standard_normal(3) * 0.5).astype(np.float32)
    src = (rs.random_sample((num_corr, 3)) * scale).astype(np.float32)
    tgt = (src @ R.T + t + rs.standard_normal((num_corr, 3)).astype(np.float32) * noise).astype(np.float32)
    outlier = rs.random_sample(num_corr) > inlier_ratio
    tgt[outlier] = (rs.random_sample((int(outlier.sum()), 3)) * scale).astype(np.float32)
    corr = np.concatenate([src, tgt], axis=-1)
    corr = corr - corr.mean(axis=0, keepdims=True)
    gt = np.eye(4, dtype=np.float32)
    gt[:3, :3] = R
    gt[:3, 3] = t
    return {
        "corr_pos": torch.from_numpy(corr.astype(np.float32))[None].contiguous(),
        "src_keypts": torch.from_numpy(src)[None].contiguous(),
        "tgt_keypts": torch.from_numpy(tgt)[None].contiguous(),
        "gt_trans": torch.from_numpy(gt)[None].contiguous(),
        "gt_labels": torch.from_numpy((~outlier).astype(np.float32))[None].contiguous(),
    }


def make_batch(batch: int, num_corr: int, seed: int = 0, **kw):
    """``batch`` independent pairs stacked on dim 0 (seeds seed, seed+1, ...)."""
    pairs = [make_pair(num_corr, seed=seed + i, **kw) for i in range(batch)]
    return {k: torch.cat([p[k] for p in pairs], dim=0).contiguous() for k in pairs[0]}


DEFAULT_LOGIT_SHIFT = 0.05


def make_state_dict(template: dict, seed: int = 0, randomize_bn: bool = True,
                    logit_shift: float = DEFAULT_LOGIT_SHIFT, logit_sign: float = 1.0) -> dict:
    """Fill a PointDSC ``state_dict`` (reference key layout) with seeded values.

    Conv weights: Xavier-normal like the reference initialiser (reference models/PointDSC.py:116-121);
    conv biases: small normal (PyTorch's default uniform init is RNG-dependent, any value is a valid
    test weight).  With ``randomize_bn`` the BatchNorm affine parameters and running statistics are
    perturbed so that BN folding is actually exercised (default init makes BN the identity up to eps)
    and ``classification.4.bias`` is shifted so that logits straddle zero like a trained model's
    (second probe).  ``logit_sign=-1`` negates the last classification layer: a random head
    ranks the (attention-clustered) inliers either first or last; a trained one ranks them first, which is what
    makes the confidence-ranked seeds useful -- workloads whose seeded head happens to rank them last flip it.
    """
    rs = np.random.RandomState(10_000 + seed)
    out = {}
    for name, ref in template.items():
        shape = tuple(ref.shape)
        if name in ("sigma", "sigma_spat"):
            out[name] = ref.clone()
            continue
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=ref.dtype)
            continue
        if name.endswith("running_mean"):
            v = rs.standard_normal(shape) * 0.1 if randomize_bn else np.zeros(shape)
        elif name.endswith("running_var"):
            v = 0.5 + rs.random_sample(shape) if randomize_bn else np.ones(shape)
        elif len(shape) == 3:  # Conv1d weight [out, in, 1]
            fan_out, fan_in = shape[0], shape[1]
            v = rs.standard_normal(shape) * np.sqrt(2.0 / (fan_in + fan_out))
        elif name.endswith("weight"):  # BatchNorm gamma
            v = 0.75 + 0.5 * rs.random_sample(shape) if randomize_bn else np.ones(shape)
        else:  # biases (conv bias or BN beta)
            v = rs.standard_normal(shape) * 0.05
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape).contiguous()
    if "classification.4.bias" in out:
        if logit_sign != 1.0:
            out["classification.4.weight"] = out["classification.4.weight"] * logit_sign
            out["classification.4.bias"] = out["classification.4.bias"] * logit_sign
        out["classification.4.bias"] = out["classification.4.bias"] + logit_shift
    return out
