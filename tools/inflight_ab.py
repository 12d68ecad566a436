#!/usr/bin/env python3
"""Interleaved A/B of the ways to keep forwards in flight (pointdsc_amd/pipeline.py) inside ONE process:
one stream | two forwards in flight | ... with each forward's tail on a high-priority companion stream | depth 3.

    python tools/inflight_ab.py [--config n5000_b32] [--batch 32] [--rounds 5] [--steps 40]
"""
import argparse
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, workloads  # noqa: E402
from pointdsc_amd.pipeline import InFlight  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="n5000_b32")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
w = workloads.WORKLOADS[a.config]
B = a.batch or w["global_batch"]
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(a.config, model.state_dict()))
model = model.eval().cuda()
batch = workloads.batch(a.config, 0, B)
data = {k: batch[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
data["testing"] = True
runners = {"one stream": InFlight(model, depth=1),
           "2 in flight": InFlight(model, depth=2, tail_streams=False),
           "2 in flight + high-priority tail streams": InFlight(model, depth=2, tail_streams=True),
           "3 in flight": InFlight(model, depth=3, tail_streams=False),
           "3 in flight + high-priority tail streams": InFlight(model, depth=3, tail_streams=True),
           "4 in flight": InFlight(model, depth=4, tail_streams=False)}
ref = None
times = {k: [] for k in runners}
for name, r in runners.items():
    for _ in range(4):
        out = r(data)
    torch.cuda.synchronize()
    ref = out["final_trans"].clone() if ref is None else ref
    assert torch.equal(out["final_trans"], ref), name
for _ in range(a.rounds):
    for name, r in runners.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r(data)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / a.steps * 1e3)
base = statistics.median(times["one stream"])
for name, ts in times.items():
    med = statistics.median(ts)
    print(f"{a.config} B={B} {name:44s} median {med:8.4f} ms/step ({B / med * 1e3:8.1f} pairs/s)  min {min(ts):.4f}  vs one stream {med / base:.4f}")
