#!/usr/bin/env python3
"""Does a ragged forward depend on what the workspace held before?  Runs the same ragged batch over (a) a workspace left by a
uniform batch, (b) a NaN-filled workspace, (c) a workspace left by the same ragged batch, and compares the valid regions of the
stage buffers to find the first stage that differs.   python tools/ragged_stale_probe.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, synthetic, workloads  # noqa: E402

name = "n5000_b32"
w = workloads.WORKLOADS[name]
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(name, model.state_dict()))
model = model.eval().cuda()
sizes = (2100, 2600, 2222)
pairs = [synthetic.make_pair(n, seed=31 + i, inlier_ratio=0.3) for i, n in enumerate(sizes)]
lists = {"corr_pos": [p["corr_pos"][0].cuda() for p in pairs], "src_keypts": [p["src_keypts"][0].cuda() for p in pairs],
         "tgt_keypts": [p["tgt_keypts"][0].cuda() for p in pairs], "testing": True}
uniform = workloads.batch(name, 0, 3)
udata = {k: uniform[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
udata["testing"] = True
bs, n = len(sizes), max(sizes)
S = int(n * 0.1)
sb = [int(x * 0.1) for x in sizes]


def snapshot():
    v = lambda name_, dt=torch.float32: model.workspace_view(name_, bs, n, dt)   # noqa: E731
    out = {}
    featA = v("featA")[: bs * n * 128].reshape(bs, n, 128)
    normed = v("normed")[: bs * n * 128].reshape(bs, n, 128)
    conf = v("conf")[: bs * n].reshape(bs, n)
    keys = v("keys")[: bs * n].reshape(bs, n)
    seeds = v("seeds", torch.int32)[: bs * S].reshape(bs, S)
    knn = v("knn_idx", torch.int32)[: bs * S * 40].reshape(bs, S, 40)
    counts = v("counts", torch.int32)[: bs * S].reshape(bs, S)
    strans = v("seed_trans")[: bs * S * 16].reshape(bs, S, 16)
    for b in range(bs):
        out[f"featA[{b}]"] = featA[b, : sizes[b]].clone()
        out[f"normed[{b}]"] = normed[b, : sizes[b]].clone()
        out[f"conf[{b}]"] = conf[b, : sizes[b]].clone()
        out[f"keys[{b}]"] = keys[b, : sizes[b]].clone()
        out[f"seeds[{b}]"] = seeds[b].clone()
        out[f"knn_idx[{b}]"] = knn[b, : sb[b]].sort(dim=1).values.clone()
        out[f"seed_trans[{b}]"] = strans[b, : sb[b]].clone()
        out[f"counts[{b}]"] = counts[b, : sb[b]].clone()
    out["best"] = v("best", torch.int32)[:bs].clone()
    out["initial_trans"] = v("initial_trans")[: bs * 16].clone()
    return out


runs = {}
with torch.no_grad():
    model(lists); model(lists)
    torch.cuda.synchronize()
    r = model(lists); torch.cuda.synchronize()
    runs["after the same ragged batch"] = (snapshot(), r["final_trans"].clone())
    model(udata); torch.cuda.synchronize()
    r = model(lists); torch.cuda.synchronize()
    runs["after a uniform batch"] = (snapshot(), r["final_trans"].clone())
    for ws in model._workspaces.values():
        ws.view(torch.int32).fill_(-1)
    r = model(lists); torch.cuda.synchronize()
    runs["after a NaN fill"] = (snapshot(), r["final_trans"].clone())
ref = runs["after the same ragged batch"]
for name_, (snap, T) in runs.items():
    print(f"== {name_}: final_trans max diff vs reference run {float((T - ref[1]).abs().max()):.3e}, finite {bool(torch.isfinite(T).all())}")
    for k, val in snap.items():
        a, b = val, ref[0][k]
        same = torch.equal(a, b)
        if not same:
            d = (a.float() - b.float()).abs()
            print(f"   {k:18s} differs: {int((a != b).sum())} entries, max |d| {float(torch.nan_to_num(d, nan=float('inf')).max()):.3e}, nan {int(torch.isnan(a.float()).sum())}")
