"""CPU oracle for the correspondence construction in front of the hot path.
TEST INFRASTRUCTURE ONLY -- only tests/ imports it.

Numpy restatement (own code) of what the reference computes inline in its datasets and demo:

    /root/reference/datasets/ThreeDMatch.py:283-290   distance matrix, argmin, mutual check
    /root/reference/datasets/ThreeDMatch.py:299-308   keypoint gather, corr_pos = concat - mean  (in_dim == 6)
    /root/reference/demo_registration.py:101-108      same, no mutual check

Pinning: those lines are not a callable function in the reference (they sit inside Dataset.__getitem__, which needs
the 3DMatch files), so ``oracle/check_correspondences_against_reference.py`` executes the reference's OWN source lines
(read from /root/reference at run time, nothing is copied into this repository) on seeded inputs and compares; the
outputs are committed as tests/golden/corr_*.npz.
"""
from __future__ import annotations

import numpy as np


def nn_distance_matrix(src_desc: np.ndarray, tgt_desc: np.ndarray) -> np.ndarray:
    """sqrt(2 - 2 <s_i, t_j> + 1e-6) in float32, the squared Euclidean distance of unit vectors plus a guard."""
    gram = src_desc.astype(np.float32) @ tgt_desc.astype(np.float32).T
    return np.sqrt(np.float32(2) - np.float32(2) * gram + np.float32(1e-6))


def build_correspondences(src_desc, tgt_desc, src_keypts, tgt_keypts, use_mutual=False):
    dist = nn_distance_matrix(src_desc, tgt_desc)
    s2t = np.argmin(dist, axis=1)                        # first index among equal distances
    keep = np.ones(s2t.shape[0], dtype=bool)
    if use_mutual:
        t2s = np.argmin(dist, axis=0)
        keep = t2s[s2t] == np.arange(s2t.shape[0])
    corr = np.stack([np.nonzero(keep)[0], s2t[keep]], axis=-1)
    src_sel, tgt_sel = src_keypts[corr[:, 0]], tgt_keypts[corr[:, 1]]
    corr_pos = np.concatenate([src_sel, tgt_sel], axis=-1)
    corr_pos = corr_pos - corr_pos.mean(0)
    return dict(corr=corr, src_keypts=src_sel, tgt_keypts=tgt_sel, corr_pos=corr_pos, nn_dist=dist.min(axis=1), dist=dist)


def make_descriptors(ns: int, nt: int, d: int, seed: int, overlap: float = 0.5, noise: float = 0.15):
    """Seeded unit descriptors: a fraction `overlap` of the source descriptors are noisy copies of target ones (true
    matches), the rest random; keypoints random in a 3 m cube.  float32, numpy RandomState (platform independent)."""
    rs = np.random.RandomState(seed)
    tgt = rs.randn(nt, d).astype(np.float32)
    tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    src = rs.randn(ns, d).astype(np.float32)
    m = int(ns * overlap)
    pick = rs.choice(nt, size=m, replace=nt < m)
    src[:m] = tgt[pick] + noise * rs.randn(m, d).astype(np.float32)
    src /= np.linalg.norm(src, axis=1, keepdims=True)
    skp = (rs.rand(ns, 3) * 3).astype(np.float32)
    tkp = (rs.rand(nt, 3) * 3).astype(np.float32)
    return src.astype(np.float32), tgt.astype(np.float32), skp, tkp
