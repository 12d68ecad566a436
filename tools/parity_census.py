#!/usr/bin/env python3
"""Parity census: EVERY pair of every bench workload against the unmodified reference's outputs.

    python tools/parity_census.py [--compat-format f32|u16] [--json]

The tests and `bench.py --check` gate on the first 4 pairs of each workload (tests/golden/bench_<name>.npz).  This tool reads the
census fixtures (tests/golden/bench_<name>_all.npz, written by `oracle/make_bench_goldens.py --all`: reference outputs and
the reference's own fp32-vs-fp64 stability for all 32 + 16 + 8 + 1 pairs), runs each workload's whole batch through
pdsc_forward_testing exactly as bench.py does and reports, per workload: label flips, the distribution of max|dT|, the
pairs above 1e-4 and whether the reference itself is stable on them.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--compat-format", default="f32")
ap.add_argument("--json", action="store_true")
ap.add_argument("--layer-gemm", default=None, help="f32 | h3 (default: the module's)")
ap.add_argument("--batch", type=int, default=0, help="run the workload's pairs in consecutive batches of this size (0 = the global batch)")
ap.add_argument("--only", default=None, help="one workload name")
a = ap.parse_args()
out = {}
for name, w in workloads.WORKLOADS.items():
    fxp = ROOT / "tests" / "golden" / f"bench_{name}_all.npz"
    if not fxp.exists() or (a.only and name != a.only):
        continue
    fx = np.load(fxp, allow_pickle=False)
    bs, n = w["global_batch"], w["num_corr"]
    model = PointDSC(**w["model"])
    model.load_state_dict(workloads.state_dict(name, model.state_dict()))
    model = model.eval().cuda()
    model.compat_format = a.compat_format
    if a.layer_gemm:
        model.layer_gemm = a.layer_gemm
    step = a.batch if 0 < a.batch < bs else bs
    parts = {"final_trans": [], "final_labels": []}
    for first in range(0, bs, step):
        batch = workloads.batch(name, first, min(step, bs - first))
        data = {k: batch[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        data["testing"] = True
        with torch.no_grad():
            r = model(data)
        for k in parts:
            parts[k].append(r[k].cpu())
    res = {k: torch.cat(v) for k, v in parts.items()}
    want_lab = torch.from_numpy(np.unpackbits(fx["ref_final_labels_bits"], axis=1)[:, :n].astype(np.float32))
    dT = (res["final_trans"].cpu() - torch.from_numpy(fx["ref_final_trans"])).abs().amax(dim=(1, 2)).numpy()
    flips = (res["final_labels"].cpu() != want_lab).sum(dim=1).numpy()
    stable = fx["stable"]
    rep = {"pairs": int(bs), "label_flips_total": int(flips.sum()), "pairs_with_label_flips": int((flips > 0).sum()),
           "max_dT": float(dT.max()), "median_dT": float(np.median(dT)),
           "pairs_dT_above_1e-4": [int(i) for i in np.flatnonzero(dT >= 1e-4)],
           "of_which_unstable_in_reference": [int(i) for i in np.flatnonzero((dT >= 1e-4) & ~stable)],
           "pairs_unstable_in_reference": [int(i) for i in np.flatnonzero(~stable)],
           "dT_sorted_top5": [float(x) for x in np.sort(dT)[::-1][:5]]}
    out[name] = rep
    if not a.json:
        print(f"{name} (batches of {step}, layer_gemm {model.layer_gemm}): {bs} pairs, label flips {rep['label_flips_total']} (in {rep['pairs_with_label_flips']} pairs), "
              f"max|dT| median {rep['median_dT']:.2e} max {rep['max_dT']:.2e}; >= 1e-4: {rep['pairs_dT_above_1e-4']} "
              f"(reference itself unstable on {rep['pairs_unstable_in_reference']}); top5 {['%.1e' % x for x in rep['dT_sorted_top5']]}")
if a.json:
    print(json.dumps(out))
