#!/usr/bin/env python3
"""Golden vectors for ``cal_confidence`` from the reference's own method (build container only).

    python oracle/check_confidence_against_reference.py      # writes tests/golden/confidence.npz

For seeded pairs it builds M = the spatial-consistency matrix exactly as the reference's forward does
(models/PointDSC.py:150-153), takes the leading eigenvector from the reference's ``cal_leading_eigenvector(M, 'power')``
(:338-358) and records ``PointDSC.cal_confidence(M, v, method)`` (:366-401) for the three methods.  The GPU test rebuilds
M with pdsc_spatial_compat (bit-exact, tested separately) and compares pdsc_cal_confidence with these numbers.
Also checks a float64 restatement of the three formulas against the reference's float32 results (the formulas, not the
round-off, are what the kernel restates: B = M - lambda1 v v^T is applied as M x - lambda1 v (v . x)).
"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")
from pointdsc_amd import synthetic  # noqa: E402

warnings.filterwarnings("ignore")
from models.PointDSC import PointDSC as RefPointDSC  # noqa: E402  (the unmodified reference)

CASES = [dict(n=257, seed=11, scale=3.0, sigma=0.1), dict(n=1000, seed=12, scale=3.0, sigma=0.1),
         dict(n=2000, seed=13, scale=60.0, sigma=1.2)]
out = {"num_cases": np.int64(len(CASES))}
torch.set_num_threads(8)
for ci, c in enumerate(CASES):
    ref = RefPointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, sigma_d=c["sigma"]).eval()
    pair = synthetic.make_pair(c["n"], seed=c["seed"], inlier_ratio=0.3, scale=c["scale"], noise=c["scale"] / 300.0)
    src, tgt = pair["src_keypts"], pair["tgt_keypts"]
    with torch.no_grad():
        d = torch.norm(src[:, :, None, :] - src[:, None, :, :], dim=-1) - torch.norm(tgt[:, :, None, :] - tgt[:, None, :, :], dim=-1)
        M = torch.clamp(1.0 - d ** 2 / ref.sigma_spat ** 2, min=0)
        v = ref.cal_leading_eigenvector(M, method="power")
        conf = {m: ref.cal_confidence(M, v, method=m).reshape(-1) for m in ("eig_value", "eig_value_ratio", "xMx")}
    # float64 restatement of the formulas
    M64, v64 = M[0].double().numpy(), v[0].double().numpy()
    l1 = v64 @ M64 @ v64 / (v64 @ v64)
    x = np.ones(c["n"])
    for _ in range(10):
        x = M64 @ x - l1 * v64 * (v64 @ x)
        x = x / (np.linalg.norm(x) + 1e-6)
    Bx = M64 @ x - l1 * v64 * (v64 @ x)
    l2 = x @ Bx / (x @ x)
    want = {"eig_value": l1, "eig_value_ratio": l1 / l2, "xMx": v64 @ M64 @ v64 / c["n"]}
    for m in want:
        rel = abs(float(conf[m][0]) - want[m]) / abs(want[m])
        print(f"N={c['n']} {m:16s} reference {float(conf[m][0]):.7g}  float64 restatement {want[m]:.7g}  rel diff {rel:.2e}")
        assert rel < 2e-4, (m, rel)
    out[f"c{ci}_n"] = np.int64(c["n"]); out[f"c{ci}_seed"] = np.int64(c["seed"]); out[f"c{ci}_scale"] = np.float64(c["scale"])
    out[f"c{ci}_sigma"] = np.float64(c["sigma"]); out[f"c{ci}_leading_eig"] = v.numpy()
    out[f"c{ci}_conf"] = np.array([float(conf[m][0]) for m in ("eig_value", "eig_value_ratio", "xMx")])
    out[f"c{ci}_src_checksum"] = np.float64(float(src.double().sum()))
np.savez_compressed(ROOT / "tests" / "golden" / "confidence.npz", **out)
print("written tests/golden/confidence.npz")
