#!/usr/bin/env python3
"""Interleaved A/B of whole-forward variants inside ONE process (the chip's power state drifts by a few per cent between
processes and boxes, so variants are compared as A B C A B C ... rounds of the same length and the medians reported).

    python tools/ab_forward.py --config n5000_b32 --variants f32 u16 f32+PDSC_ATT_WIDE=1 [--rounds 7] [--steps 25]

A variant is `<compat_format>[+<ENV>=<value>...][+@<attribute>=<value>...]`: model.compat_format, per-call environment knobs
of the library, attributes of the module (attention_precision, layer_gemm).  Environment knobs exist in the EXPERIMENTS
library only (python -m pointdsc_amd.build --experiments): variants that use one make this tool load
pointdsc_amd/libpointdsc_hip_exp.so (or pass --exp to force it); the product library ignores the environment.
"""
import argparse
import os
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
if "--exp" in sys.argv or any("+PDSC_" in x for x in sys.argv):
    os.environ.setdefault("POINTDSC_HIP_LIB", str(ROOT / "pointdsc_amd" / "libpointdsc_hip_exp.so"))
from pointdsc_amd import PointDSC, _lib, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="n5000_b32")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--variants", nargs="+", default=["f32", "u16"])
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=25)
ap.add_argument("--exp", action="store_true", help="load the experiments library")
a = ap.parse_args()
if any("+PDSC_" in v for v in a.variants) and not _lib.load().pdsc_experiments_enabled():
    raise SystemExit("environment knobs need the experiments library (python -m pointdsc_amd.build --experiments)")
w = workloads.WORKLOADS[a.config]
B = a.batch or w["global_batch"]
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(a.config, model.state_dict()))
model = model.eval().cuda()
batch = workloads.batch(a.config, 0, B)
data = {k: batch[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
data["testing"] = True


def apply(variant):
    parts = variant.split("+")
    model.compat_format = parts[0]
    env = {}
    for p in parts[1:]:
        k, v = p.split("=")
        if k.startswith("@"):            # model attribute, e.g. @attention_precision=fp16x3_all, @layer_gemm=f32
            setattr(model, k[1:], v)
            continue
        env[k] = v
    os.environ.update(env)
    return env


times = {v: [] for v in a.variants}
with torch.no_grad():
    for v in a.variants:                  # warm-up (workspace, weight packing per variant)
        apply(v)
        for _ in range(3):
            model(data)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for v in a.variants:
            env = apply(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                model(data)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / a.steps)
            for k in env:
                os.environ.pop(k, None)
base = statistics.median(times[a.variants[0]])
for v in a.variants:
    med = statistics.median(times[v])
    print(f"{a.config} B={B} {v:24s} median {med:8.4f} ms/step  ({B / med * 1e3:8.1f} pairs/s)  min {min(times[v]):.4f} max {max(times[v]):.4f}"
          f"  vs {a.variants[0]}: {med / base:.4f}")
