#!/usr/bin/env python3
"""Launch time of the spatial-consistency matrix build (a-1), unorm16 storage: GB/s of algorithmic bytes (2 N^2 + 24 N per pair).

    python tools/compat_bench.py [--exp]        # --exp: experiments library, also times PDSC_COMPAT16_VARIANT = 0 (r02 kernel) / 2
"""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
if "--exp" in sys.argv:
    os.environ.setdefault("POINTDSC_HIP_LIB", str(ROOT / "pointdsc_amd" / "libpointdsc_hip_exp.so"))
import torch  # noqa: E402
from pointdsc_amd import _lib, ops, synthetic  # noqa: E402

exp = bool(_lib.load().pdsc_experiments_enabled())
variants = [("shipped", None)] + ([("r02 exact-rounded", "0"), ("fast, no explicit clamp", "2")] if exp else [])
for n, bs, scale, sigma in ((5000, 32, 3.0, 0.1), (5000, 4, 3.0, 0.1), (5000, 16, 60.0, 1.2), (10000, 8, 3.0, 0.1), (1000, 1, 3.0, 0.1)):
    batch = synthetic.make_batch(bs, n, seed=3, scale=scale, noise=scale / 300.0)
    src, tgt, sig = batch["src_keypts"].cuda(), batch["tgt_keypts"].cuda(), torch.tensor([sigma]).cuda()
    out = {}
    for name, v in variants:
        if v is None:
            os.environ.pop("PDSC_COMPAT16_VARIANT", None)
        else:
            os.environ["PDSC_COMPAT16_VARIANT"] = v
        c = ops.spatial_compat_u16(src, tgt, sig)
        out[name] = c.clone()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.spatial_compat_u16(src, tgt, sig, out=c)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        ms = statistics.median(ts)
        gb = (2.0 * n * n + 24.0 * n) * bs / 1e9
        extra = ""
        if name != "shipped":
            d = (out[name].to(torch.int32) & 0xFFFF) - (out["shipped"].to(torch.int32) & 0xFFFF)
            extra = f"  vs shipped: max |du| {int(d.abs().max())}, entries differing {float((d != 0).float().mean()):.2e}"
        print(f"N={n} bs={bs} scale={scale}: {name:24s} {ms * 1e3:8.1f} us  {gb / ms * 1e3:7.0f} GB/s = {gb / ms * 1e3 / 8000:.3f} of 8 TB/s{extra}", flush=True)
