#!/usr/bin/env python3
"""Forwards in flight against the plain calls with the PRODUCT library, many repetitions: every forward of a pipeline (two plain streams,
two streams + tail streams, three replayed hipGraphs) must return the plain call's pose and mask bit for bit.  Run after any change to
a kernel on the forward's chain (r06: the merged tail launches and the range sentinel).   PROBE_REPS=300 python tools/inflight_soak.py"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import workloads, PointDSC  # noqa: E402
from pointdsc_amd.pipeline import InFlight  # noqa: E402

REPS = int(os.environ.get("PROBE_REPS", 300))


def run(cfg, B):
    w = workloads.WORKLOADS[cfg]
    model = PointDSC(**w["model"])
    model.load_state_dict(workloads.state_dict(cfg, model.state_dict()))
    model = model.eval().cuda()
    batches = []
    for i in range(4):
        b = workloads.batch(cfg, B * i, B)
        d = {k: b[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        d["testing"] = True
        batches.append(d)
    with torch.no_grad():
        plain = [model(d) for d in batches]
    torch.cuda.synchronize()
    line, total_bad = [], 0
    for kw in (dict(depth=2, tail_streams=False), dict(depth=2, tail_streams=True), dict(depth=3, graphs=True)):
        r = InFlight(model, **kw)
        bad, lab, worst = 0, 0, 0.0
        for _ in range(REPS):
            outs = [r(d) for d in batches]
            r.synchronize()
            for o, p in zip(outs, plain):
                te, le = torch.equal(o["final_trans"], p["final_trans"]), torch.equal(o["final_labels"], p["final_labels"])
                if not (te and le):
                    bad += 1
                    lab += int(not le)
                    worst = max(worst, float((o["final_trans"] - p["final_trans"]).abs().max()))
        line.append(f"{'graphs' if kw.get('graphs') else 'tail' if kw['tail_streams'] else 'plain'}: {bad}/{REPS * 4} differ ({lab} with label flips, worst {worst:.1e})")
        total_bad += bad
        r.close()
    print(f"{cfg} x{B}: " + "  ".join(line), flush=True)
    return total_bad


if __name__ == "__main__":
    bad = run("n5000_b32", 3) + run("n5000_b32", 1) + run("n1000_b1", 1) + run("trained_kitti_n5000_b16", 2)
    print("TOTAL differing forwards:", bad)
    sys.exit(1 if bad else 0)
