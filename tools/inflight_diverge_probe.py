#!/usr/bin/env python3
"""When a forward in flight differs from the plain call on the same batch: which workspace field is the first to differ?"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
if os.environ.get("PDSC_SCORE_DEBUG"):
    os.environ.setdefault("POINTDSC_HIP_LIB", str(ROOT / "pointdsc_amd" / "libpointdsc_hip_exp.so"))
from pointdsc_amd import workloads, PointDSC, _lib  # noqa: E402
from pointdsc_amd.pipeline import InFlight  # noqa: E402

REPS = int(os.environ.get("PROBE_REPS", 60))
cfg, B = os.environ.get("PROBE_CONFIG", "n5000_b32"), int(os.environ.get("PROBE_B", 3))
w = workloads.WORKLOADS[cfg]
n = w["num_corr"]
lib = _lib.load()
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(cfg, model.state_dict()))
model = model.eval().cuda()
for kv in os.environ.get("PROBE_ATTRS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        setattr(model, k, v)
FIELDS = ["featA", "normed", "h2", "conf", "keys", "seeds", "knn_dist", "knn_idx", "eig_iters", "conv_mask", "seed_trans", "seed_w", "counts", "best",
          "initial_trans", "solves"]
ORDER = ["compat", "featA", "featB", "featC", "qkv", "msg", "t64a", "t64b", "att_scratch", "q_split", "kv_tiles", "normed", "h1", "h2", "conf", "keys",
         "nms_ws", "seeds", "knn_dist", "knn_idx", "eig_iters", "conv_mask", "seed_trans", "seed_w", "counts", "best", "initial_trans", "solves"]
DBG = bool(os.environ.get("PDSC_SCORE_DEBUG"))
if DBG:
    ORDER.append("score_dbg")
c = model._config()
S = int(n * model.ratio)
offs = {name: int(lib.pdsc_workspace_offset(C.byref(c), B, n, S, name.encode())) for name in ORDER}
total = int(lib.pdsc_workspace_bytes(C.byref(c), B, n, S))


def field(ws, name):
    i = ORDER.index(name)
    end = total
    for later in ORDER[i + 1:]:
        if offs[later] > offs[name]:
            end = offs[later]
            break
    return ws[offs[name]:end]


batches = []
for i in range(4):
    b = workloads.batch(cfg, B * i, B)
    d = {k: b[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    d["testing"] = True
    batches.append(d)
plain, snaps = [], []
with torch.no_grad():
    for d in batches:
        plain.append(model(d))
        torch.cuda.synchronize()
        snaps.append({f: field(model._workspaces[0], f).clone() for f in FIELDS})
mode = os.environ.get("PROBE_MODE", "graphs")
r = InFlight(model, depth=3, graphs=True) if mode == "graphs" else InFlight(model, depth=2, tail_streams=(mode == "tail"))
depth = r.depth
seen = {}
bad = 0
for rep in range(REPS):
    outs = [r(d) for d in batches]
    r.synchronize()
    for j in range(len(batches) - depth, len(batches)):          # the last forward on each slot: its workspace is intact
        o, p = outs[j], plain[j]
        if torch.equal(o["final_trans"], p["final_trans"]) and torch.equal(o["final_labels"], p["final_labels"]):
            continue
        bad += 1
        ws = model._workspaces[r._slots[(rep * len(batches) + j) % depth]]
        diff = []
        for f in FIELDS:
            a, b_ = field(ws, f), snaps[j][f]
            if not torch.equal(a, b_):
                nd = int((a != b_).sum())
                diff.append(f"{f}({nd}B)")
                if f == "counts" and bad <= 3:
                    ai, bi = a.view(torch.int32), b_.view(torch.int32)
                    idx = torch.nonzero(ai != bi).flatten()[:6].tolist()
                    print(f"    counts differ at {idx}: in flight {ai[idx].tolist()} plain {bi[idx].tolist()}", flush=True)
        if DBG:
            d = field(ws, "score_dbg").view(torch.float32)[: B * S * 16].reshape(B * S, 16)
            tfin = field(ws, "seed_trans").view(torch.float32)[: B * S * 16].reshape(B * S, 16)
            cnt = field(ws, "counts").view(torch.int32)[: B * S]
            pc = snaps[j]["counts"].view(torch.int32)[: B * S]
            t_stale = (d[:, :12] != tfin[:, :12]).any(dim=1)
            short = cnt != pc
            print(f"    rep {rep} fwd {j}: seeds whose transform AS READ by the scoring kernel differs from the final seed_trans: {int(t_stale.sum())}"
                  f" (of them short-counted: {int((t_stale & short).sum())}); short-counted seeds {int(short.sum())}; "
                  f"recount with system-scope point loads equals the plain count on {int(((d[:, 13].int() == pc) & short).sum())} of them, "
                  f"equals the in-flight count on {int(((d[:, 13].int() == cnt) & short).sum())}", flush=True)
            idx = torch.nonzero(short).flatten()
            even = int((idx % 2 == 0).sum())
            third_plain = int(((d[:, 14].int() == pc) & short).sum())
            third_bad = int(((d[:, 14].int() == cnt) & short).sum())
            # seeds (any parity) whose transform equals that of a short-counted seed bit for bit but whose count is right
            twins = 0
            for i in idx[:8].tolist():
                same = (tfin[:, :12] == tfin[i, :12]).all(dim=1) & (torch.arange(B * S, device=tfin.device) // S == i // S)
                twins += int((same & ~short).sum())
            print(f"      short-counted seeds with even index {even} / {len(idx)}; third count (ordinary loads, later) equals plain on {third_plain}, "
                  f"equals the in-flight count on {third_bad}; seeds with the SAME transform as one of the first 8 short seeds but a right count: {twins}", flush=True)
            if bad <= 2 and int(t_stale.sum()):
                i = int(torch.nonzero(t_stale).flatten()[0])
                print("      as read:", [round(x, 6) for x in d[i, :12].tolist()], "\n      final:  ", [round(x, 6) for x in tfin[i, :12].tolist()], flush=True)
        key = " ".join(diff) or "(no workspace field differs)"
        seen[key] = seen.get(key, 0) + 1
print(f"{cfg} x {B}, {mode}, attrs [{os.environ.get('PROBE_ATTRS', '')}]: {bad} mismatching forwards of {REPS * depth} inspected")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  {v:4d} x  differing fields: {k}")
