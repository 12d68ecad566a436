#!/usr/bin/env python3
"""Times pdsc_layer_fused_frag_fmt alone (merge of the attention partials + tail + head, the forward's output set) at the
bench's size, interleaving variants given as ENV=value lists inside one process.

    python tools/layer_bench.py --bs 32 --variants PDSC_LAYER_GEMM=0 PDSC_LAYER_GEMM=1 PDSC_LAYER_GEMM=1,PDSC_LAYER_H3_EXP=2
"""
import argparse
import ctypes as C
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os as _os
_os.environ.setdefault("POINTDSC_HIP_LIB", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "pointdsc_amd" / "libpointdsc_hip_exp.so"))   # PDSC_* knobs / traces: experiments library (python -m pointdsc_amd.build --experiments)
from pointdsc_amd import _lib, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=5000)
ap.add_argument("--bs", type=int, default=32)
ap.add_argument("--variants", nargs="+", default=["PDSC_LAYER_GEMM=0", "PDSC_LAYER_GEMM=1"])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--calls", type=int, default=20)
ap.add_argument("--pf", action="store_true", help="hand-offs in point-fragment order (pdsc_layer_fused_frag_io, H3 only)")
a = ap.parse_args()
lib = _lib.load()
n, bs, dev = a.n, a.bs, "cuda:0"
gen = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)  # noqa: E731
m = n * bs
batch = synthetic.make_batch(bs, n, seed=1)
compat = ops.spatial_compat(batch["src_keypts"].to(dev), batch["tgt_keypts"].to(dev), torch.tensor([0.1], device=dev))
qs, kv = ops.pack_qkv_split(rnd(m, 384) * 0.3, bs, n)
scratch, nsplit = ops.sc_attention_split(qs, kv, compat, bs, n, merge=False, layout="pf" if a.pf else "rows")
del compat
npad = (n + 255) // 256 * 256
res = rnd(m, 128)
tail_w = [rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128)]
head_w = [rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384)]
streams = {g: (ops.frag_weights_tail(tail_w, g), ops.frag_weights_head(head_w, g)) for g in ("f32", "h3")}
featB = torch.empty(bs * ops.pf_rows(n), 128, device=dev)
if a.pf:
    res = ops.rows_to_pf(res, bs, n)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
part_o = C.c_void_p(scratch.data_ptr())
part_ml = C.c_void_p(scratch.data_ptr() + bs * nsplit * npad * 128 * 4)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call():
    gemm = int(os.environ.get("PDSC_LAYER_GEMM", "0"))
    wt, wh = streams["h3" if gemm else "f32"]
    if a.pf:
        _lib.check(lib.pdsc_layer_fused_frag_io(None, part_o, part_ml, nsplit, npad, p(res), None, None, p(featB), p(qs), p(kv),
                                                p(wt), p(wh), gemm, 7, bs, n, stream), "pdsc_layer_fused_frag_io")
        return
    _lib.check(lib.pdsc_layer_fused_frag_fmt(None, part_o, part_ml, nsplit, npad, p(res), None, None, p(featB), None, p(qs), p(kv),
                                             p(wt), p(wh), gemm, bs, n, stream), "pdsc_layer_fused_frag_fmt")


def apply(v):
    for kvp in v.split(","):
        k, val = kvp.split("=")
        os.environ[k] = val


times = {v: [] for v in a.variants}
for v in a.variants:
    apply(v)
    for _ in range(3):
        call()
torch.cuda.synchronize()
for _ in range(a.rounds):
    for v in a.variants:
        apply(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.calls):
            call()
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / a.calls * 1e3)
for v in a.variants:
    print(f"N={n} bs={bs} nsplit={nsplit}  {v:60s} median {statistics.median(times[v]):7.1f} us  min {min(times[v]):7.1f}")
