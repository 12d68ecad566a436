#!/usr/bin/env python3
"""Pins the oracle's VALIDATION forward (no 'testing' key, eval() mode -- SURVEY.md section 8 f-1) against the
reference implementation and writes the golden fixtures tests/golden/val_*.npz + VALIDATION_PINNING.json.

Run in the BUILD container only (imports the unmodified reference from /root/reference):

    python oracle/check_validation_against_reference.py

Per case: seeded inputs/weights (pointdsc_amd/synthetic.py), the reference ``PointDSC.forward(data)`` without the
'testing' key on the CPU, the oracle on the same inputs; asserts agreement (M and logits to fp32 round-off, seeds equal,
pose to 1e-5) and stores the REFERENCE outputs.  M is stored completely for the small case and as sampled rows for the
larger ones (fixtures stay small).
"""
from __future__ import annotations

import json
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from oracle import pointdsc_oracle as O  # noqa: E402
from pointdsc_amd import synthetic  # noqa: E402
from pointdsc_amd.model import PointDSC as AmdPointDSC  # noqa: E402  (state_dict template only)

GOLDEN = ROOT / "tests" / "golden"
BASE_MODEL = dict(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                  inlier_threshold=0.10, sigma_d=0.10, k=40, nms_radius=0.10)
ORACLE_KEYS = ("num_layers", "num_channels", "num_iterations", "ratio", "inlier_threshold", "k", "nms_radius")
CASES = [
    dict(name="val_n257_b1", N=257, bs=1, seed=40, inlier_ratio=0.3, wseed=0, full=True),
    dict(name="val_n1000_b3", N=1000, bs=3, seed=41, inlier_ratio=0.25, wseed=6, full=False),
    dict(name="val_n2053_b2", N=2053, bs=2, seed=42, inlier_ratio=0.3, wseed=3, full=False),
]


def maxabs(a, b):
    return float((a.double() - b.double()).abs().max())


def run_case(case):
    sys.path.insert(0, str(REF))
    from models.PointDSC import PointDSC as RefPointDSC  # the unmodified reference
    kw = dict(BASE_MODEL)
    sd = synthetic.make_state_dict(AmdPointDSC(**kw).state_dict(), seed=case["wseed"])
    ref = RefPointDSC(**kw)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    batch = synthetic.make_batch(case["bs"], case["N"], seed=case["seed"], inlier_ratio=case["inlier_ratio"])
    data = {k: batch[k] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    with torch.no_grad():
        res = ref(dict(data))                                    # no 'testing' key
        ores = O.forward_validation(sd, data["corr_pos"], data["src_keypts"], data["tgt_keypts"], return_stages=True,
                                    **{k: kw[k] for k in ORACLE_KEYS})
        S = int(case["N"] * kw["ratio"])
        r_seeds = torch.argsort(res["final_labels"], dim=1, descending=True)[:, :S]
    rep = {"N": case["N"], "bs": case["bs"]}
    rep["M_maxabs"] = maxabs(ores["M"], res["M"])
    rep["logits_maxabs"] = maxabs(ores["final_labels"], res["final_labels"])
    rep["final_trans_maxabs"] = maxabs(ores["final_trans"], res["final_trans"])
    rep["seed_sets_equal"] = all(set(ores["stages"][b]["seeds"].tolist()) == set(r_seeds[b].tolist()) for b in range(case["bs"]))
    rep["M_diag_zero"] = bool((torch.diagonal(res["M"], dim1=1, dim2=2) == 0).all())
    rep["M_range"] = [float(res["M"].min()), float(res["M"].max())]
    rep["power_iters_run"] = ores["stages"][0]["power_iters"]
    rows = torch.arange(0, case["N"], max(case["N"] // 48, 1))
    fx = dict(corr_pos=data["corr_pos"].numpy(), src_keypts=data["src_keypts"].numpy(), tgt_keypts=data["tgt_keypts"].numpy(),
              gt_trans=batch["gt_trans"].numpy(), wseed=np.int64(case["wseed"]), model_json=np.array(json.dumps(kw)),
              ref_final_trans=res["final_trans"].numpy(), ref_logits=res["final_labels"].numpy(),
              M_rows=rows.numpy(), ref_M_rows=res["M"][:, rows].numpy(),
              ref_M_checksum=res["M"].double().sum(dim=(1, 2)).numpy())
    if case["full"]:
        fx["ref_M"] = res["M"].numpy()
    GOLDEN.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLDEN / f"{case['name']}.npz", **fx)
    return rep


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(8)
    report, ok = {}, True
    for case in CASES:
        rep = run_case(case)
        report[case["name"]] = rep
        print(case["name"], json.dumps(rep))
        must = [rep["M_maxabs"] < 2e-6, rep["logits_maxabs"] < 2e-6, rep["final_trans_maxabs"] < 1e-5, rep["seed_sets_equal"],
                rep["M_diag_zero"]]
        if not all(must):
            ok = False
            print("  !! pin violated:", must)
    (GOLDEN / "VALIDATION_PINNING.json").write_text(json.dumps(report, indent=1))
    print("validation forward of the oracle pinned against the reference" if ok else "ORACLE DISAGREES WITH THE REFERENCE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
