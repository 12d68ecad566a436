"""The BASELINE.json configurations as seeded synthetic workloads.

One table shared by ``bench.py`` (what is timed), ``oracle/make_bench_goldens.py`` (what the unmodified reference
returns for the first pairs of each workload, committed under ``tests/golden/bench_*.npz``) and the GPU parity tests
(the timed path is compared with those reference outputs), so that the numbers in a bench line and the parity
claim next to it are about the same pairs, weights and launch plans.

Real data and released weights are absent on both boxes: the pairs follow
SURVEY.md section 8(d) (``synthetic.make_pair``), the weights are seeded (``synthetic.make_state_dict``) and their
logits are centred per workload (``logit_shift``) so that seeds are picked among local maxima with distinct positive
keys, as a trained model's are -- with every logit negative the reference takes its seeds from the tied zero keys in
backend-defined order (probe 2).
"""
from __future__ import annotations

from typing import Dict

import torch

from . import synthetic

BASE_MODEL = dict(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                  sigma_d=0.10, k=40, nms_radius=0.10)
# reference evaluation/test_KITTI.py:166-170,188: inlier_threshold 0.6 m, sigma_d 1.2 m (nms_radius = inlier_threshold)
KITTI_MODEL = dict(BASE_MODEL, inlier_threshold=0.6, sigma_d=1.2, nms_radius=0.6)

WORKLOADS: Dict[str, dict] = {
    # BASELINE.json configs[1]: 3DMatch test, N=1000, batch 1 on one GPU
    # (seed0 = 1001: on the pair seeded 1000 the REFERENCE's own fp32 and fp64 runs differ by 2.5e-4 in the pose -- six seed
    #  hypotheses tie at the maximal inlier count and kNN sets decided at the 1e-7 level pick the winner; the five pairs
    #  after it agree to < 1e-5 -- measured by oracle/make_bench_goldens.py, column reference_fp32_vs_fp64_dT)
    "n1000_b1": dict(baseline_config=1, num_corr=1000, global_batch=1, model=BASE_MODEL, wseed=6, logit_shift=0.05,
                     pair=dict(inlier_ratio=0.2, noise=0.01, scale=3.0), seed0=1001,
                     label="3DMatch-like synthetic correspondences (BASELINE.json configs[1])"),
    # BASELINE.json configs[2] (the headline metric): N=5000, 32 pairs sharded over the GPUs
    "n5000_b32": dict(baseline_config=2, num_corr=5000, global_batch=32, model=BASE_MODEL, wseed=6, logit_shift=0.05,
                      pair=dict(inlier_ratio=0.2, noise=0.01, scale=3.0), seed0=1000,
                      label="3DMatch-like synthetic correspondences (BASELINE.json configs[2])"),
    # BASELINE.json configs[3]: KITTI odometry, N=5000, sigma_d=1.2 m, 16 pairs
    # (logit_sign=-1: this seeded head ranks the inlier cluster LAST -- 0.2 % inliers among the confidence-ranked seeds and
    #  the reference itself fails on 3 of the first 4 pairs; negated it ranks them first, like a trained head)
    "kitti_n5000_b16": dict(baseline_config=3, num_corr=5000, global_batch=16, model=KITTI_MODEL, wseed=8,
                            logit_shift=None, logit_sign=-1.0, pair=dict(inlier_ratio=0.25, noise=0.1, scale=60.0), seed0=3000,
                            label="KITTI-like synthetic correspondences (BASELINE.json configs[3]: 60 m scale, "
                                  "sigma_d=1.2 m, inlier_threshold=0.6 m)"),
    # BASELINE.json configs[4]: 3DLoMatch, N=10000, 8 pairs (low overlap = low inlier ratio)
    "lomatch_n10000_b8": dict(baseline_config=4, num_corr=10000, global_batch=8, model=BASE_MODEL, wseed=7,
                              logit_shift=None, pair=dict(inlier_ratio=0.15, noise=0.01, scale=3.0), seed0=4000,
                              label="3DLoMatch-like synthetic correspondences (BASELINE.json configs[4])"),
    # the reference's real KITTI evaluation size: evaluation/test_KITTI.py:120 (num_node=12000), :166-170 (sigma_d 1.2, threshold 0.6)
    "kitti_n12000_b4": dict(baseline_config=None, num_corr=12000, global_batch=4, model=KITTI_MODEL, wseed=8,
                            logit_shift=None, logit_sign=-1.0, pair=dict(inlier_ratio=0.25, noise=0.1, scale=60.0), seed0=5000,
                            label="KITTI-like synthetic correspondences at the reference's evaluation size (evaluation/test_KITTI.py:120: "
                                  "N=12000, sigma_d=1.2 m, inlier_threshold=0.6 m)"),
    # the reference's multiway registration feeds 20000 correspondences per pair (multiway/test_multi_ate.py:245)
    "multiway_n20000_b1": dict(baseline_config=None, num_corr=20000, global_batch=1, model=BASE_MODEL, wseed=7,
                               logit_shift=None, pair=dict(inlier_ratio=0.15, noise=0.01, scale=3.0), seed0=6000,
                               label="3DMatch-like synthetic correspondences at the multiway evaluation size "
                                     "(multiway/test_multi_ate.py:245: N=20000)"),
}
# ---- trained-like weights (r05) -------------------------------------------------------------------------
# The families above use seeded random weights: a regime with a nearly collapsed feature space (logits within a few 1e-2, top-k
# boundary gaps of 5e-7) that a trained model never visits.  These use the checkpoints of oracle/make_trained_fixture.py -- the
# UNMODIFIED reference trained with its own training forward and losses on synthetic pairs until its logits separate inliers
# (tests/golden/trained_3dmatch.npz / trained_kitti.npz, 358 keys) -- and cycle the inlier ratio over 5 % .. 40 % so that the
# hard low-overlap pairs, where the learned confidence matters, are a quarter of every census.
TRAINED_CYCLE = (0.05, 0.1, 0.2, 0.4)
WORKLOADS.update({
    "trained_n1000_b1": dict(baseline_config=1, num_corr=1000, global_batch=1, model=BASE_MODEL, weights="trained_3dmatch", logit_shift=0.0,
                             pair=dict(noise=0.01, scale=3.0), inlier_cycle=TRAINED_CYCLE, seed0=11000,
                             label="3DMatch-like synthetic correspondences, TRAINED-LIKE weights (the unmodified reference trained on "
                                   "synthetic pairs, oracle/make_trained_fixture.py), inlier ratio cycling 5/10/20/40 %"),
    "trained_n5000_b32": dict(baseline_config=2, num_corr=5000, global_batch=32, model=BASE_MODEL, weights="trained_3dmatch", logit_shift=0.0,
                              pair=dict(noise=0.01, scale=3.0), inlier_cycle=TRAINED_CYCLE, seed0=12000,
                              label="3DMatch-like synthetic correspondences at N=5000, TRAINED-LIKE weights, inlier ratio cycling 5/10/20/40 %"),
    "trained_kitti_n5000_b16": dict(baseline_config=3, num_corr=5000, global_batch=16, model=KITTI_MODEL, weights="trained_kitti", logit_shift=0.0,
                                    pair=dict(noise=0.1, scale=60.0), inlier_cycle=TRAINED_CYCLE, seed0=13000,
                                    label="KITTI-like synthetic correspondences (60 m, sigma_d=1.2 m, threshold 0.6 m), TRAINED-LIKE weights, "
                                          "inlier ratio cycling 5/10/20/40 %"),
    # BASELINE.json configs[4] on trained-like weights: low overlap = the lower half of the inlier-ratio range
    "trained_lomatch_n10000_b8": dict(baseline_config=4, num_corr=10000, global_batch=8, model=BASE_MODEL, weights="trained_3dmatch", logit_shift=0.0,
                                      pair=dict(noise=0.01, scale=3.0), inlier_cycle=(0.05, 0.1, 0.15, 0.25), seed0=14000,
                                      label="3DLoMatch-like synthetic correspondences at N=10000, TRAINED-LIKE weights, inlier ratio cycling "
                                            "5/10/15/25 %"),
    # the reference's real evaluation sizes (evaluation/test_KITTI.py:120, multiway/test_multi_ate.py:245) on trained-like weights
    "trained_kitti_n12000_b4": dict(baseline_config=None, num_corr=12000, global_batch=4, model=KITTI_MODEL, weights="trained_kitti", logit_shift=0.0,
                                    pair=dict(noise=0.1, scale=60.0), inlier_cycle=TRAINED_CYCLE, seed0=15000,
                                    label="KITTI-like synthetic correspondences at the reference's evaluation size (N=12000), TRAINED-LIKE weights, "
                                          "inlier ratio cycling 5/10/20/40 %"),
    "trained_multiway_n20000_b1": dict(baseline_config=None, num_corr=20000, global_batch=1, model=BASE_MODEL, weights="trained_3dmatch", logit_shift=0.0,
                                       pair=dict(noise=0.01, scale=3.0), inlier_cycle=(0.05, 0.1, 0.15, 0.25), seed0=16000,
                                       label="3DMatch-like synthetic correspondences at the multiway evaluation size (N=20000), TRAINED-LIKE weights, "
                                             "inlier ratio cycling 5/10/15/25 %"),
})
DEFAULT = "n5000_b32"

# logit shifts of the workloads whose table entry says None: measured once by oracle/make_bench_goldens.py with the
# unmodified reference (shift = -median of the reference's logits over the first golden pairs at shift 0, 4 decimals)
# and frozen here so that every box builds the same weights.
MEASURED_LOGIT_SHIFT: Dict[str, float] = {
    "kitti_n5000_b16": -1.8915,
    "lomatch_n10000_b8": 0.4779,
    "kitti_n12000_b4": -1.8683,
    "multiway_n20000_b1": 0.4761,
}


def logit_shift(name: str) -> float:
    w = WORKLOADS[name]
    if "weights" in w:
        return 0.0                     # trained-like checkpoints are used as they are
    if w["logit_shift"] is not None:
        return float(w["logit_shift"])
    if name not in MEASURED_LOGIT_SHIFT:
        raise KeyError(f"workload {name}: logit shift not measured yet (run oracle/make_bench_goldens.py)")
    return MEASURED_LOGIT_SHIFT[name]


def trained_state_dict(which: str, template: dict) -> dict:
    """tests/golden/<which>.npz (oracle/make_trained_fixture.py) as a state_dict with the template's keys, dtypes and shapes."""
    import numpy as np
    from pathlib import Path
    fx = np.load(Path(__file__).resolve().parents[1] / "tests" / "golden" / f"{which}.npz", allow_pickle=False)
    out = {}
    for k, ref in template.items():
        v = torch.from_numpy(np.asarray(fx[k]))
        if tuple(v.shape) != tuple(ref.shape):
            raise ValueError(f"{which}: {k} has shape {tuple(v.shape)}, the module expects {tuple(ref.shape)}")
        out[k] = v.to(ref.dtype).contiguous()
    return out


def state_dict(name: str, template: dict, shift: float | None = None) -> dict:
    w = WORKLOADS[name]
    if "weights" in w:
        return trained_state_dict(w["weights"], template)
    return synthetic.make_state_dict(template, seed=w["wseed"], logit_shift=logit_shift(name) if shift is None else shift,
                                     logit_sign=w.get("logit_sign", 1.0))


def batch(name: str, first: int, count: int) -> Dict[str, torch.Tensor]:
    """Pairs [first, first+count) of the workload's global pair list (pair i is seeded seed0 + i)."""
    w = WORKLOADS[name]
    if "inlier_cycle" in w:            # pair i: inlier ratio inlier_cycle[i % len]
        cyc = w["inlier_cycle"]
        pairs = [synthetic.make_pair(w["num_corr"], seed=w["seed0"] + i, inlier_ratio=cyc[i % len(cyc)], **w["pair"])
                 for i in range(first, first + count)]
        return {k: torch.cat([p[k] for p in pairs], dim=0).contiguous() for k in pairs[0]}
    return synthetic.make_batch(count, w["num_corr"], seed=w["seed0"] + first, **w["pair"])
