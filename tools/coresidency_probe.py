#!/usr/bin/env python3
"""Do small kernels run BESIDE an attention launch of another stream, or do they wait for it?

Stream A loops the 32-pair forward (its attention launches occupy every CU: one 8-wave workgroup per CU, 96-132 KiB of LDS,
424 of 512 registers per SIMD lane).  Stream B issues one small kernel at a time and measures its latency, alone and while A is
running: a kernel that fits in what an attention workgroup leaves free on a CU should barely slow down; one that does not has to
wait for a whole CU.   python tools/coresidency_probe.py
"""
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, ops, workloads  # noqa: E402

name = "n5000_b32"
w = workloads.WORKLOADS[name]
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(name, model.state_dict()))
model = model.eval().cuda()
bs, n = 32, 5000
batch = workloads.batch(name, 0, bs)
data = {k: batch[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
data["testing"] = True
gen = torch.Generator().manual_seed(0)
feat = torch.randn(bs * n, 128, generator=gen).cuda()
w32 = (torch.randn(32, 128, generator=gen) / 11).cuda()
b32 = torch.randn(32, generator=gen).cuda()
h2 = torch.randn(bs * n, 32, generator=gen).cuda()
w3 = torch.randn(32, generator=gen).cuda()
b3 = torch.randn(1, generator=gen).cuda()
seed_trans = torch.eye(4).repeat(bs, 500, 1, 1).cuda()
conf = torch.randn(bs, n, generator=gen).cuda()
kernels = {
    "normalize_conf (16 VGPR, no LDS)": lambda: ops.normalize_confidence(feat, h2, w3, b3),
    "linear 128->32 (24 VGPR, 67.5 KiB LDS)": lambda: ops.linear(feat, w32, b32, relu=True),
    "score + select_best (76 VGPR / 1024 threads)": lambda: ops.score_hypotheses(seed_trans, data["src_keypts"], data["tgt_keypts"], 0.1),
    "nms keys grid (1024-thread workgroups, 17 KiB LDS)": lambda: ops.nms_keys_grid(data["src_keypts"], conf, 0.1) if hasattr(ops, "nms_keys_grid") else None,
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def latency(fn, reps=30):
    ts = []
    with torch.cuda.stream(sb):
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sb)
            fn()
            e1.record(sb)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


with torch.no_grad():
    for _ in range(3):
        model(data)
    torch.cuda.synchronize()
    for label, fn in kernels.items():
        if fn() is None and "nms" in label:
            continue
        torch.cuda.synchronize()
        alone = latency(fn)
        with torch.cuda.stream(sa):
            for _ in range(40):                      # ~0.6 s of attention-dominated work on stream A
                model(data)
        time.sleep(0.02)
        busy = latency(fn)
        torch.cuda.synchronize()
        print(f"{label:52s} alone {alone:8.1f} us   while another stream runs the 32-pair forward {busy:8.1f} us   x{busy / alone:.1f}")
