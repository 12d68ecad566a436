#!/usr/bin/env python3
"""Pins the inner-product matching of the 3DLoMatch caller against the reference's own source lines; writes
tests/golden/corr_lomatch_*.npz.

BUILD container only.  evaluation/test_3DLoMatch.py:45-48 (inside ``get_predator_data``, which needs the Predator feature
files) computes ``dists = einsum('ac,bc->ab', src_feats, tgt_feats)``, ``source_idx = argmax(dists, -1)`` and the centred
``corr_pos``.  This script reads exactly those four lines from the reference at run time and ``exec``s them on seeded inputs
with the names they expect (``src_feats``, ``tgt_feats``, ``src_pcd``, ``tgt_pcd``); nothing is copied into this repository.
The descriptors are deliberately NOT unit length (rows scaled by 0.8 ... 2.0: inner products above 1 make the distance form NaN, and np.argmin returns the FIRST NaN), so that the arg-max of inner products differs
from the arg-min of ``sqrt(2 - 2 <s,t> + 1e-6)`` (datasets/ThreeDMatch.py:283-290) on a good share of the rows -- the two
callers' forms are only equal for exactly unit descriptors.
"""
from __future__ import annotations

import json
import sys
import textwrap
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import correspondence_oracle as CO  # noqa: E402

REF_FILE = Path("/root/reference/evaluation/test_3DLoMatch.py")
GOLDEN = ROOT / "tests" / "golden"
CASES = [dict(name="corr_lomatch_n1000_d32", ns=1000, nt=1100, d=32, seed=11),
         dict(name="corr_lomatch_n5000_d32", ns=5000, nt=5000, d=32, seed=12)]


def make_inputs(case):
    src, tgt, skp, tkp = CO.make_descriptors(case["ns"], case["nt"], case["d"], case["seed"])
    rs = np.random.RandomState(case["seed"] + 1000)
    tgt = (tgt * rs.uniform(0.8, 2.0, size=(case["nt"], 1))).astype(np.float32)
    src = (src * rs.uniform(0.8, 2.0, size=(case["ns"], 1))).astype(np.float32)
    return src, tgt, skp, tkp


def reference_lines():
    lines = REF_FILE.read_text().splitlines()
    code = textwrap.dedent("\n".join(lines[44:48]))          # :45-48
    assert "torch.argmax(dists, dim=-1)" in code and "corr_pos.mean(1, keepdims=True)" in code, "reference layout changed"
    return code


def main():
    report = {}
    code = reference_lines()
    for case in CASES:
        src, tgt, skp, tkp = make_inputs(case)
        ns = dict(torch=torch, src_feats=torch.from_numpy(src), tgt_feats=torch.from_numpy(tgt), src_pcd=torch.from_numpy(skp),
                  tgt_pcd=torch.from_numpy(tkp))
        exec(code, ns)
        idx, corr_pos, dists = ns["source_idx"].numpy(), ns["corr_pos"][0].numpy(), ns["dists"].numpy()
        l2 = np.argmin(CO.nn_distance_matrix(src, tgt), axis=1)
        top2 = np.partition(dists, -2, axis=1)[:, -2:]
        rep = dict(Ns=case["ns"], Nt=case["nt"], D=case["d"], rows_where_argmax_ip_differs_from_argmin_l2=int((idx != l2).sum()),
                   min_margin=float((top2[:, 1] - top2[:, 0]).min()))
        report[case["name"]] = rep
        print(case["name"], json.dumps(rep))
        np.savez_compressed(GOLDEN / f"{case['name']}.npz", ns=case["ns"], nt=case["nt"], d=case["d"], seed=case["seed"],
                            ref_source_idx=idx.astype(np.int32), ref_corr_pos=corr_pos.astype(np.float32),
                            ref_best_dot=dists.max(axis=1).astype(np.float32), min_margin=rep["min_margin"])
    p = GOLDEN / "CORRESPONDENCE_PINNING.json"
    full = json.loads(p.read_text()) if p.exists() else {}
    full.update(report)
    p.write_text(json.dumps(full, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
