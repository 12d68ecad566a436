#!/usr/bin/env python3
"""Pins oracle/correspondence_oracle.py against the reference's own source lines and writes tests/golden/corr_*.npz.

BUILD container only.  The correspondence construction is inline code of ``ThreeDMatchTest.__getitem__``
(/root/reference/datasets/ThreeDMatch.py:283-290 and :299-308), which cannot be called without the 3DMatch files.
This script reads exactly those lines from the reference at run time, executes them (``exec``) on seeded inputs with
the names they expect (``src_desc``, ``tgt_desc``, ``src_keypts``, ``tgt_keypts``, ``self.use_mutual``, ``self.in_dim``)
and compares every result with the oracle.  No reference source is copied into this repository.
"""
from __future__ import annotations

import json
import sys
import textwrap
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import correspondence_oracle as CO  # noqa: E402

REF_FILE = Path("/root/reference/datasets/ThreeDMatch.py")
GOLDEN = ROOT / "tests" / "golden"
CASES = [
    dict(name="corr_n300_d32", ns=300, nt=257, d=32, seed=1, mutual=False),
    dict(name="corr_n1000_d33_mutual", ns=1000, nt=1200, d=33, seed=2, mutual=True),
    dict(name="corr_n5000_d32_mutual", ns=5000, nt=4700, d=32, seed=3, mutual=True),
]


def reference_lines():
    lines = REF_FILE.read_text().splitlines()
    match = textwrap.dedent("\n".join(lines[282:290]))       # :283-290 distance, argmin, mutual check, corr
    gather = textwrap.dedent("\n".join(lines[298:308]))      # :299-308 input_src_keypts .. corr_pos (in_dim 3 / 6)
    assert "np.argmin(distance, axis=1)" in match and "corr_pos - corr_pos.mean(0)" in gather, "reference layout changed"
    return match, gather


def run_reference(src_desc, tgt_desc, src_keypts, tgt_keypts, mutual):
    match, gather = reference_lines()
    ns = dict(np=np, src_desc=src_desc, tgt_desc=tgt_desc, src_keypts=src_keypts, tgt_keypts=tgt_keypts,
              self=types.SimpleNamespace(use_mutual=mutual, in_dim=6))
    exec(match, ns)
    exec(gather, ns)
    return dict(corr=ns["corr"], corr_pos=ns["corr_pos"], src_keypts=ns["input_src_keypts"], tgt_keypts=ns["input_tgt_keypts"],
                source_idx=ns["source_idx"])


def main():
    report, ok = {}, True
    for case in CASES:
        src, tgt, skp, tkp = CO.make_descriptors(case["ns"], case["nt"], case["d"], case["seed"])
        ref = run_reference(src, tgt, skp, tkp, case["mutual"])
        ora = CO.build_correspondences(src, tgt, skp, tkp, use_mutual=case["mutual"])
        rep = dict(Ns=case["ns"], Nt=case["nt"], D=case["d"], mutual=case["mutual"], Nc=int(ref["corr"].shape[0]),
                   corr_equal=bool(np.array_equal(ref["corr"], ora["corr"])),
                   corr_pos_maxabs=float(np.abs(ref["corr_pos"] - ora["corr_pos"]).max()),
                   keypts_equal=bool(np.array_equal(ref["src_keypts"], ora["src_keypts"]) and np.array_equal(ref["tgt_keypts"], ora["tgt_keypts"])))
        # margin between the best and the second best distance of every source point: the fixture is only useful if the
        # arg-min is decided by more than fp32 round-off of the dot products
        d2 = np.partition(ora["dist"], 1, axis=1)[:, :2]
        rep["min_margin"] = float((d2[:, 1] - d2[:, 0]).min())
        report[case["name"]] = rep
        print(case["name"], json.dumps(rep))
        if not (rep["corr_equal"] and rep["keypts_equal"] and rep["corr_pos_maxabs"] < 1e-6):
            ok = False
        np.savez_compressed(GOLDEN / f"{case['name']}.npz", ns=case["ns"], nt=case["nt"], d=case["d"], seed=case["seed"],
                            mutual=case["mutual"], ref_corr=ref["corr"].astype(np.int32), ref_corr_pos=ref["corr_pos"].astype(np.float32),
                            ref_source_idx=ref["source_idx"].astype(np.int32), min_margin=rep["min_margin"])
    (GOLDEN / "CORRESPONDENCE_PINNING.json").write_text(json.dumps(report, indent=1))
    print("correspondence oracle pinned against the reference's own lines" if ok else "ORACLE DISAGREES WITH THE REFERENCE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
