#!/usr/bin/env python3
"""Stage timeline of layer_fused_kernel<tail, head, split qkv> (pdsc_layer_trace stamps), in shader clocks."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os as _os
_os.environ.setdefault("POINTDSC_HIP_LIB", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "pointdsc_amd" / "libpointdsc_hip_exp.so"))   # PDSC_* knobs / traces: experiments library (python -m pointdsc_amd.build --experiments)
import torch  # noqa: E402

from pointdsc_amd import _lib, ops, synthetic  # noqa: E402

NAMES = ["start", "partials merged+stored", "barrier", "fc1 done", "fc2 done", "fc3 done", "barrier", "pcn mma done", "pcn stored",
         "featB written", "qkv0 mma done", "qkv0 staged", "qkv0 streams written", "end"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--gemm", default="f32", help="f32 | h3 (enum pdsc_layer_gemm; fragment streams only)")
    args = ap.parse_args()
    lib = _lib.load()
    n, bs, dev = args.n, args.bs, "cuda:0"
    gen = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)  # noqa: E731
    m = n * bs
    batch = synthetic.make_batch(bs, n, seed=1)
    compat = ops.spatial_compat(batch["src_keypts"].to(dev), batch["tgt_keypts"].to(dev), torch.tensor([0.1], device=dev))
    qs, kv = ops.pack_qkv_split(rnd(m, 384) * 0.3, bs, n)
    partials = ops.sc_attention_split(qs, kv, compat, bs, n, merge=False)
    res = rnd(m, 128)
    tail_w = [rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128)]
    head_w = [rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384)]
    frag = os.environ.get("PDSC_LAYER_FRAG", "1") != "0" and not os.environ.get("PDSC_LAYER_VARIANT", "w").startswith("b")
    run = lambda: ops.layer_fused_split(None, res, None, tail_w, head_w, bs, n, partials=partials, qkv_split=True, frag=frag,  # noqa: E731
                                        gemm=args.gemm if frag else "f32")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"layer_fused (merge of {partials[1]} splits + tail + head, split qkv): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call (incl. weight split + allocs)")
    nwg = ((n + 31) // 32) * bs
    trace = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
    _lib.check(lib.pdsc_layer_trace(C.c_void_p(trace.data_ptr())), "trace")
    run()
    torch.cuda.synchronize()
    _lib.check(lib.pdsc_layer_trace(None), "trace off")
    if os.environ.get("PDSC_LAYER_VARIANT", "w").startswith("b"):
        t = trace.cpu().reshape(nwg, 4, 16)[:, :, :14].double()
        t0 = t[:, :, 0].min()
        print(f"{nwg} workgroups; kernel span {float(t[:, :, 13].max() - t0):.0f} clocks; workgroup start spread {float(t[:, 0, 0].max() - t0):.0f}")
        for w in (0, 3):
            d = t[:, w, 1:] - t[:, w, :-1]
            life = t[:, w, 13] - t[:, w, 0]
            print(f"wave {w}: lifetime mean {float(life.mean()):.0f}  min {float(life.min()):.0f}  max {float(life.max()):.0f}")
            for k in range(13):
                print(f"   {NAMES[k]:26s} -> {NAMES[k + 1]:26s} mean {float(d[:, k].mean()):8.0f}  max {float(d[:, k].max()):8.0f}")
    else:       # wavefront-resident kernel: one wave per tile; stamps 0 start, 1 input in registers, 2+i chunk i done, 63 end
        t = trace.cpu()[: nwg * 64].reshape(nwg, 64).double()
        t0 = t[:, 0].min()
        life = t[:, 63] - t[:, 0]
        print(f"{nwg} waves; kernel span {float(t[:, 63].max() - t0):.0f} clocks; start spread {float(t[:, 0].max() - t0):.0f}")
        print(f"wave lifetime mean {float(life.mean()):.0f}  min {float(life.min()):.0f}  max {float(life.max()):.0f}")
        names = ["input"] + ["fc1"] * 4 + ["fc2"] * 2 + ["fc3"] * 4 + ["pcn"] * 8 + ["qkv"] * 24
        mma = [0] + [384 if args.gemm == "h3" else 2048] * 18 + [384] * 24
        d = (t[:, 1:44] - t[:, 0:43])
        # waves of the first round (resident from the start) vs the later ones (start when a slot frees up)
        first = t[:, 0] < t0 + 2000
        print(f"first-round waves {int(first.sum())}: lifetime mean {float(life[first].mean()):.0f};  later waves {int((~first).sum())}: "
              f"lifetime mean {float(life[~first].mean()) if (~first).any() else 0:.0f}; last start {float(t[:, 0].max() - t0):.0f}")
        for k in range(43):
            print(f"   {k:2d} {names[k]:6s} mean {float(d[:, k].mean()):7.0f}  min {float(d[:, k].min()):7.0f}  max {float(d[:, k].max()):7.0f}   "
                  f"first-round mean {float(d[first, k].mean()):7.0f}   (mfma {mma[k]})")
        print(f"   tail (pads, drain)  mean {float((t[:, 63] - t[:, 43]).mean()):7.0f}")
        grp = {"input": [0], "fc1": range(1, 5), "fc2": range(5, 7), "fc3": range(7, 11), "pcn": range(11, 19), "qkv": range(19, 43)}
        print("   per stage (mean clocks per wave): " + "  ".join(f"{k} {float(d[:, list(v)].sum(1).mean()):.0f}" for k, v in grp.items()))

if __name__ == "__main__":
    main()
