#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of sc_attention_split_kernel<8> (instrumented build, pdsc_attention_trace).

    python tools/attention_trace.py [--n 5000] [--bs 4]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os as _os
_os.environ.setdefault("POINTDSC_HIP_LIB", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "pointdsc_amd" / "libpointdsc_hip_exp.so"))   # PDSC_* knobs / traces: experiments library (python -m pointdsc_amd.build --experiments)
import torch  # noqa: E402

from pointdsc_amd import _lib, ops, synthetic  # noqa: E402

NAMES = ["prologue", "first tile", "own-DMA wait", "barrier", "DMA issue", "phase A (QK | exp,split)", "phase B (PV | logits)",
         "decision + epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--zeros", action="store_true", help="all-zero q/k/v: same instruction stream, minimal data toggling (DVFS probe)")
    args = ap.parse_args()
    lib = _lib.load()
    n, bs = args.n, args.bs
    dev = "cuda:0"
    batch = synthetic.make_batch(bs, n, seed=1)
    compat = ops.spatial_compat(batch["src_keypts"].to(dev), batch["tgt_keypts"].to(dev), torch.tensor([0.1], device=dev))
    gen = torch.Generator().manual_seed(0)
    qkv = (torch.randn(bs * n, 384, generator=gen) * (0.0 if args.zeros else 0.3)).to(dev)
    qs, kv = ops.pack_qkv_split(qkv, bs, n)
    for _ in range(3):
        ops.sc_attention_split(qs, kv, compat, bs, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.sc_attention_split(qs, kv, compat, bs, n)
    e1.record()
    torch.cuda.synchronize()
    print(f"plain kernel + combine: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call, nsplit={lib.pdsc_attention_split_default_split(bs, n)}")
    trace = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.pdsc_attention_trace(C.c_void_p(trace.data_ptr())), "trace")
    e0.record()
    ops.sc_attention_split(qs, kv, compat, bs, n)
    e1.record()
    torch.cuda.synchronize()
    _lib.check(lib.pdsc_attention_trace(None), "trace off")
    t = trace.cpu().reshape(-1, 8, 8)
    used = t.sum(dim=(1, 2)) > 0
    t = t[used].double()
    print(f"instrumented call: {e0.elapsed_time(e1) * 1e3:.1f} us; {int(used.sum())} workgroups")
    tot = t.sum(-1)
    print(f"per-wave total cycles: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f}")
    for k, name in enumerate(NAMES):
        v = t[..., k]
        print(f"  {k} {name:28s} mean {v.mean():9.0f} ({100 * v.mean() / tot.mean():5.1f} %)   min {v.min():9.0f}  max {v.max():9.0f}")
    tiles = (n + 31) // 32 / lib.pdsc_attention_split_default_split(bs, n)
    print(f"tiles per workgroup ~{tiles:.1f}: per tile: wait {t[..., 2].mean() / tiles:.0f} barrier {t[..., 3].mean() / tiles:.0f} "
          f"issue {t[..., 4].mean() / tiles:.0f} A {t[..., 5].mean() / tiles:.0f} B {t[..., 6].mean() / tiles:.0f} cycles")


if __name__ == "__main__":
    main()
