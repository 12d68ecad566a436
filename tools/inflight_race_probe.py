#!/usr/bin/env python3
"""Forwards in flight against the plain calls, many repetitions, under A/B knobs (experiments library): which stage makes a
forward depend on what else runs on the chip?"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("POINTDSC_HIP_LIB", str(ROOT / "pointdsc_amd" / "libpointdsc_hip_exp.so"))
from pointdsc_amd import workloads, PointDSC  # noqa: E402
from pointdsc_amd.pipeline import InFlight  # noqa: E402

REPS = int(os.environ.get("PROBE_REPS", 150))
cfg, B = os.environ.get("PROBE_CONFIG", "n5000_b32"), int(os.environ.get("PROBE_B", 3))
w = workloads.WORKLOADS[cfg]


def run(label, env, attrs):
    for k, v in env.items():
        os.environ[k] = v
    model = PointDSC(**w["model"])
    model.load_state_dict(workloads.state_dict(cfg, model.state_dict()))
    model = model.eval().cuda()
    for k, v in attrs.items():
        setattr(model, k, v)
    batches = []
    for i in range(4):
        b = workloads.batch(cfg, B * i, B)
        d = {k: b[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        d["testing"] = True
        batches.append(d)
    with torch.no_grad():
        plain = [model(d) for d in batches]
    line = []
    for kw in (dict(depth=2, tail_streams=False), dict(depth=2, tail_streams=True), dict(depth=3, graphs=True)):
        r = InFlight(model, **kw)
        bad, lab, worst = 0, 0, 0.0
        for rep in range(REPS):
            outs = [r(d) for d in batches]
            r.synchronize()
            for o, p in zip(outs, plain):
                te, le = torch.equal(o["final_trans"], p["final_trans"]), torch.equal(o["final_labels"], p["final_labels"])
                if not (te and le):
                    bad += 1
                    lab += int(not le)
                    worst = max(worst, float((o["final_trans"] - p["final_trans"]).abs().max()))
        line.append(f"{'graphs' if kw.get('graphs') else 'tail' if kw['tail_streams'] else 'plain'}: {bad}/{REPS * 4} ({lab} with label flips, worst {worst:.1e})")
        r.close()
    print(f"{label:44s} " + "  ".join(line), flush=True)
    for k in env:
        del os.environ[k]


run("defaults", {}, {})
run("PDSC_LAYER_H3_COOP=0", {"PDSC_LAYER_H3_COOP": "0"}, {})
run("PDSC_LAYER_PF=0 (row-order hand-offs)", {"PDSC_LAYER_PF": "0"}, {})
run("layer_gemm f32", {}, {"layer_gemm": "f32"})
run("compat f32", {}, {"compat_format": "f32"})
run("attention fp32 (exact path)", {}, {"attention_precision": "fp32", "compat_format": "f32", "layer_gemm": "f32"})
