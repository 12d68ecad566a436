#!/usr/bin/env python3
"""Runs the reference's OWN ``SM`` function (baseline_scripts/baseline_3DMatch.py:19-53) on seeded inputs and writes
tests/golden/sm_*.npz + SM_PINNING.json.  BUILD container only.

The baseline script imports open3d / tqdm / the datasets at module level (absent here), so the function's source lines
are read from /root/reference at run time and executed (``exec``) with the two names they need: ``torch`` and the
reference's ``rigid_transform_3d`` (importable: /root/reference/models/common.py).  Nothing is copied into this repo.
The fixtures are the reference outputs; the GPU tests compare the HIP path with them (this row has no separate CPU
restatement: the reference function IS the oracle, executed here, and its outputs travel as fixtures).
"""
from __future__ import annotations

import json
import sys
import textwrap
import types
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import synthetic  # noqa: E402

REF = Path("/root/reference")
GOLDEN = ROOT / "tests" / "golden"
CASES = [
    dict(name="sm_n257", N=257, seed=50, inlier_ratio=0.4, thr=0.10),
    dict(name="sm_n2000", N=2000, seed=51, inlier_ratio=0.3, thr=0.10),
    dict(name="sm_n5000", N=5000, seed=52, inlier_ratio=0.2, thr=0.10),
    dict(name="sm_kitti_n1500", N=1500, seed=53, inlier_ratio=0.3, thr=0.60, scale=60.0, noise=0.1),
]


def reference_sm():
    sys.path.insert(0, str(REF))
    from models.common import rigid_transform_3d
    lines = (REF / "baseline_scripts" / "baseline_3DMatch.py").read_text().splitlines()
    src = textwrap.dedent("\n".join(lines[18:53]))          # def SM(...): ... return pred_trans, pred_labels
    assert src.startswith("def SM(") and "return pred_trans, pred_labels" in src, "reference layout changed"
    ns = dict(torch=torch, rigid_transform_3d=rigid_transform_3d)
    exec(src, ns)
    return ns["SM"]


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(8)
    SM = reference_sm()
    report = {}
    for case in CASES:
        kw = {k: case[k] for k in ("scale", "noise") if k in case}
        pair = synthetic.make_pair(case["N"], seed=case["seed"], inlier_ratio=case["inlier_ratio"], **kw)
        args = types.SimpleNamespace(inlier_threshold=case["thr"])
        with torch.no_grad():
            trans, labels = SM(pair["corr_pos"], pair["src_keypts"], pair["tgt_keypts"], args)   # corr [1,N,6]: corr - corr.permute(1,0,2) broadcasts to [N,N,6]
        # margin of the top-10 % cut: the labels are only a meaningful bit-exact target if the cut is not a tie
        # (recompute the eigenvector here with the reference's own lines? it is not returned: use precision/recall instead)
        gt = pair["gt_labels"][0]
        tp = float((labels[0] * gt).sum())
        R, t = trans[0, :3, :3], trans[0, :3, 3]
        gR, gt_t = pair["gt_trans"][0, :3, :3], pair["gt_trans"][0, :3, 3]
        re = float(torch.acos(torch.clamp((torch.trace(R.t() @ gR) - 1) / 2, -1, 1)) * 180 / np.pi)
        rep = dict(N=case["N"], num_selected=int(labels.sum()), precision=tp / max(float(labels.sum()), 1), RE_deg=re,
                   TE_cm=float((t - gt_t).norm() * 100))
        report[case["name"]] = rep
        print(case["name"], json.dumps(rep))
        np.savez_compressed(GOLDEN / f"{case['name']}.npz", N=case["N"], seed=case["seed"], inlier_ratio=case["inlier_ratio"],
                            thr=case["thr"], scale=kw.get("scale", 3.0), noise=kw.get("noise", 0.01),
                            ref_pred_trans=trans.numpy(), ref_pred_labels=labels.numpy())
    (GOLDEN / "SM_PINNING.json").write_text(json.dumps(report, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
