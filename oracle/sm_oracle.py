"""CPU oracle for the spectral-matching baseline.  TEST INFRASTRUCTURE ONLY.

Torch-CPU restatement (own code) of ``SM`` in /root/reference/baseline_scripts/baseline_3DMatch.py:19-53.  Pinned by
``oracle/check_sm_against_reference.py``, which executes the reference's own function on the seeded inputs of
tests/golden/sm_*.npz; tests/test_cpu_oracle_and_abi.py checks this restatement against those reference outputs.
"""
from __future__ import annotations

import torch

from oracle.pointdsc_oracle import rigid_transform_3d


def sm_matrix(corr: torch.Tensor, inlier_threshold: float) -> torch.Tensor:
    """corr [N,6] -> M [N,N] = max(0, 4.5 - d^2 / 2 / sigma^2), zero diagonal (:20-37)."""
    diff = corr[:, None, :] - corr[None, :, :]
    d = (diff[..., 0:3] ** 2).sum(-1) ** 0.5 - (diff[..., 3:6] ** 2).sum(-1) ** 0.5
    sigma = inlier_threshold / 3
    M = torch.clamp(4.5 - d ** 2 / 2 / sigma ** 2, min=0)
    n = M.shape[0]
    M[torch.arange(n), torch.arange(n)] = 0
    return M


def sm_baseline(corr: torch.Tensor, src_keypts: torch.Tensor, tgt_keypts: torch.Tensor, inlier_threshold: float,
                top_ratio: float = 0.1, num_iterations: int = 10):
    """corr [N,6], src/tgt [N,3] -> (pred_trans [4,4], pred_labels [N], leading_eig [N])."""
    M = sm_matrix(corr.float(), inlier_threshold)
    v = torch.ones(M.shape[0], 1)
    for _ in range(num_iterations):                              # :40-43
        v = M @ v
        v = v / (torch.norm(v, dim=0, keepdim=True) + 1e-6)
    v = v[:, 0]
    top = torch.sort(v, descending=True, stable=True).indices[: int(v.shape[0] * top_ratio)]      # :46-48
    labels = torch.zeros_like(v)
    labels[top] = 1
    trans = rigid_transform_3d(src_keypts[None].float(), tgt_keypts[None].float(), (v * labels)[None])[0]      # :51
    return trans, labels, v
