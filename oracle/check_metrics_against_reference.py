#!/usr/bin/env python3
"""Runs the reference's TransformationLoss / ClassificationLoss (libs/loss.py, importable: torch + sklearn) on seeded poses
and label vectors and writes tests/golden/metrics.npz -- the fixture of the device-side evaluation row (f-4).
BUILD container only."""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import synthetic  # noqa: E402

REF = Path("/root/reference")


def main():
    warnings.filterwarnings("ignore")
    sys.path.insert(0, str(REF))
    from libs.loss import ClassificationLoss, TransformationLoss
    evaluate_metric, class_loss = TransformationLoss(re_thre=15, te_thre=30), ClassificationLoss()
    rs = np.random.RandomState(0)
    rows, T, G, P, L = [], [], [], [], []
    n = 1000
    for i in range(24):
        pair = synthetic.make_pair(n, seed=100 + i, inlier_ratio=0.05 + 0.03 * i)
        gt = pair["gt_trans"]
        # a prediction: ground truth perturbed by a rotation of 0 .. 40 degrees and a shift of 0 .. 60 cm
        ang = np.deg2rad(rs.rand() * (40 if i % 3 else 2))
        axis = rs.randn(3); axis /= np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        dR = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        pred = gt.clone()
        pred[0, :3, :3] = torch.from_numpy(dR.astype(np.float32)) @ gt[0, :3, :3]
        pred[0, :3, 3] = gt[0, :3, 3] + torch.from_numpy((rs.randn(3) * (0.3 if i % 2 else 0.01)).astype(np.float32))
        gl = pair["gt_labels"]
        flip = torch.from_numpy(rs.rand(1, n) < 0.1 * (i % 4))
        pl = torch.where(flip, 1 - gl, gl).float()
        if i == 5:
            pl = torch.zeros_like(pl)                      # nothing predicted: sklearn's zero-division case
        if i == 6:
            gl = torch.zeros_like(gl)                      # no ground-truth inlier
        cs = class_loss(pl, gl)
        loss, recall, re, te, rmse = evaluate_metric(pred, gt, pair["src_keypts"], pair["tgt_keypts"], pl)
        rows.append([float(recall / 100.0), float(re), float(te), int(gl.sum()), float(gl.float().mean()),
                     int(gl[pl > 0].sum()), cs["precision"], cs["recall"], cs["f1"]])        # test_3DMatch.py:90-98
        T.append(pred[0].numpy()); G.append(gt[0].numpy()); P.append(pl[0].numpy()); L.append(gl[0].numpy())
    np.savez_compressed(ROOT / "tests" / "golden" / "metrics.npz", trans=np.stack(T), gt_trans=np.stack(G), pred_labels=np.stack(P),
                        gt_labels=np.stack(L), ref_stats=np.array(rows, dtype=np.float64))
    print(np.array(rows)[:8].round(4))
    return 0


if __name__ == "__main__":
    sys.exit(main())
