#!/usr/bin/env python3
"""Parity census: many seeded pairs per workload family against the unmodified reference's outputs (fp32 AND fp64 runs).

    python tools/parity_census.py [--only NAME] [--batches 1,2,4,8,16,32] [--compat-format u16|f32] [--layer-gemm h3|f32] [--json]

Fixtures: tests/golden/census_<name>.npz (oracle/make_census_goldens.py: for every pair the reference's pose and label mask as
shipped (fp32) and with the default dtype switched to fp64).  Every pair is run through pdsc_forward_testing with the module's
shipped defaults (or the overrides), in consecutive batches of each requested size -- the batch size selects the attention's
key-split plan and the layer kernel, i.e. the summation orders -- and held to BASELINE.json's contract:
    labels bit-exact and max|dT| < 1e-4 against the reference's fp32 output,
    or, where the reference's own two precisions disagree (a discrete near-tie among seed hypotheses decided by round-off,
    models/PointDSC.py:325-335), against its fp64 output.
There is no looser tolerance for any pair.  Prints one line per (workload, batch size) with the dT histogram and the list of
failing pairs; --json prints the same as one JSON object.
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

EDGES = (1e-6, 1e-5, 2e-5, 5e-5, 1e-4)


def judge(trans: torch.Tensor, labels: torch.Tensor, fx, n: int, first: int = 0):
    """Per pair: (ok, dT vs fp32 reference, dT vs the closer of the two references, label flips vs fp32, which reference matched)."""
    g = trans.shape[0]
    t32 = torch.from_numpy(fx["ref32_final_trans"][first:first + g]).double()
    t64 = torch.from_numpy(fx["ref64_final_trans"][first:first + g]).double()
    l32 = torch.from_numpy(np.unpackbits(fx["ref32_final_labels_bits"][first:first + g], axis=1)[:, :n].astype(np.float32))
    l64 = torch.from_numpy(np.unpackbits(fx["ref64_final_labels_bits"][first:first + g], axis=1)[:, :n].astype(np.float32))
    T, L = trans.cpu().double(), labels.cpu()
    d32 = (T - t32).abs().amax(dim=(1, 2))
    d64 = (T - t64).abs().amax(dim=(1, 2))
    f32 = (L != l32).sum(dim=1)
    f64 = (L != l64).sum(dim=1)
    ok32 = (d32 < 1e-4) & (f32 == 0)
    ok64 = (d64 < 1e-4) & (f64 == 0)
    return ok32 | ok64, d32, torch.where(ok32, d32, torch.minimum(d32, d64)), f32, torch.where(ok32, 0, torch.where(ok64, 1, -1))


def run_family(name: str, batches, compat_format=None, layer_gemm=None, pairs: int = 0, attention_precision=None):
    fx = np.load(ROOT / "tests" / "golden" / f"census_{name}.npz", allow_pickle=False)
    w = workloads.WORKLOADS[name]
    n = w["num_corr"]
    total = fx["ref32_final_trans"].shape[0] if pairs <= 0 else min(pairs, fx["ref32_final_trans"].shape[0])
    model = PointDSC(**w["model"])
    model.load_state_dict(workloads.state_dict(name, model.state_dict()))
    model = model.eval().cuda()
    if compat_format:
        model.compat_format = compat_format
    if layer_gemm:
        model.layer_gemm = layer_gemm
    if attention_precision:
        model.attention_precision = attention_precision
    out = {}
    for step in batches:
        T, L = [], []
        for first in range(0, total, step):
            batch = workloads.batch(name, first, min(step, total - first))
            data = {k: batch[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
            data["testing"] = True
            with torch.no_grad():
                r = model(data)
            T.append(r["final_trans"].cpu())
            L.append(r["final_labels"].cpu())
        ok, d32, dbest, f32, which = judge(torch.cat(T), torch.cat(L), fx, n)
        d = dbest.numpy()
        hist = [int(((d >= lo) & (d < hi)).sum()) for lo, hi in zip((0.0,) + EDGES, EDGES + (np.inf,))]
        t64 = torch.from_numpy(fx["ref64_final_trans"][:total]).double()
        d64 = (torch.cat(T).double() - t64).abs().amax(dim=(1, 2))
        ref_self = (torch.from_numpy(fx["ref32_final_trans"][:total]).double() - t64).abs().amax(dim=(1, 2))
        out[step] = {"pairs": int(total), "failing_pairs": [int(i) for i in np.flatnonzero(~ok.numpy())],
                     "failing_detail": [{"pair": int(i), "dT_vs_ref_fp32": float(d32[i]), "dT_vs_ref_fp64": float(d64[i]),
                                         "label_flips_vs_ref_fp32": int(f32[i]), "reference_fp32_vs_fp64_dT": float(ref_self[i])}
                                        for i in np.flatnonzero(~ok.numpy())],
                     "reference_self_disagreement_above_1e-4": [int(i) for i in np.flatnonzero(ref_self.numpy() >= 1e-4)],
                     "label_flips_vs_fp32_reference": int(f32.sum()), "pairs_matched_on_fp64_reference": [int(i) for i in np.flatnonzero(which.numpy() == 1)],
                     "median_dT": float(np.median(d)), "max_dT": float(d.max()), "max_dT_vs_fp32_reference": float(d32.max()),
                     "dT_histogram": dict(zip(["<1e-6", "<1e-5", "<2e-5", "<5e-5", "<1e-4", ">=1e-4"], hist))}
    return out, model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--batches", default="0", help="comma list of batch sizes; 0 = the workload's global batch")
    ap.add_argument("--compat-format", default=None)
    ap.add_argument("--layer-gemm", default=None)
    ap.add_argument("--attention-precision", default=None, help='"fp32" = the exact-fp32 path (with --compat-format f32 --layer-gemm f32)')
    ap.add_argument("--pairs", type=int, default=0, help="first K pairs of each family only")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    report = {}
    for name, w in workloads.WORKLOADS.items():
        if (a.only and name != a.only) or not (ROOT / "tests" / "golden" / f"census_{name}.npz").exists():
            continue
        batches = [int(x) or w["global_batch"] for x in a.batches.split(",")]
        rep, model = run_family(name, batches, a.compat_format, a.layer_gemm, a.pairs, a.attention_precision)
        report[name] = rep
        if not a.json:
            for step, r in rep.items():
                print(f"{name} (N={w['num_corr']}, attention {model.attention_precision}, compat {model.compat_format}, layer_gemm {model.layer_gemm}) "
                      f"batches of {step}: {r['pairs']} pairs, FAIL {r['failing_pairs']}, label flips vs fp32 ref {r['label_flips_vs_fp32_reference']}, "
                      f"matched on the fp64 ref {r['pairs_matched_on_fp64_reference']}, max|dT| median {r['median_dT']:.1e} max {r['max_dT']:.1e}, "
                      f"histogram {r['dT_histogram']}; the reference's own fp32 and fp64 runs differ by >= 1e-4 on {r['reference_self_disagreement_above_1e-4']}",
                      flush=True)
                for fd in r["failing_detail"]:
                    print("    ", json.dumps(fd), flush=True)
    if a.json:
        print(json.dumps(report))
    return 1 if any(r["failing_pairs"] for rep in report.values() for r in rep.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
