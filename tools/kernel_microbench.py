#!/usr/bin/env python3
"""Within-process A/B microbenchmarks of the individual HIP stages (run on the GPU box).

    python tools/kernel_microbench.py [--n 5000] [--bs 4] [--rounds 5] [--only attention,compat,...]

Every stage is timed with torch.cuda events on the stream the kernels are launched on, `--iters` launches
per round, interleaved over variants for `--rounds` rounds; median / min per launch are reported together
with the roofline figure of the stage.  Variants are selected through the library's tuning knobs
(PDSC_ATT_VARIANT is read once per process, so attention variants run in child processes).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import os as _os
_os.environ.setdefault("POINTDSC_HIP_LIB", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "pointdsc_amd" / "libpointdsc_hip_exp.so"))   # PDSC_* knobs / traces: experiments library (python -m pointdsc_amd.build --experiments)


def timed(fn, iters):
    import torch
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters * 1e3   # microseconds per launch


def bench_stage(name, fn, iters, rounds, work=None, unit=None):
    ts = [timed(fn, iters) for _ in range(rounds)]
    rec = {"stage": name, "median_us": round(statistics.median(ts), 2), "min_us": round(min(ts), 2)}
    if work:
        rec[unit] = round(work / (statistics.median(ts) * 1e-6) / 1e12, 3)   # T-units per second
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--att-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--compat-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    import torch
    from pointdsc_amd import ops, synthetic, PointDSC

    n, bs = args.n, args.bs
    only = set(filter(None, args.only.split(",")))
    want = lambda s: not only or s in only  # noqa: E731
    dev = "cuda:0"
    batch = synthetic.make_batch(bs, n, seed=1000, inlier_ratio=0.2)
    src, tgt = batch["src_keypts"].to(dev), batch["tgt_keypts"].to(dev)
    sig = torch.tensor([0.1], device=dev)
    gen = torch.Generator().manual_seed(0)
    M = bs * n

    if args.att_child:   # child process: one attention variant, several splits
        compat = ops.spatial_compat(src, tgt, sig)
        qkv = torch.randn(M, 384, generator=gen).to(dev)
        flops = 4.0 * 128 * n * n * bs
        for nsplit in [int(x) for x in args.att_child.split(",")]:
            bench_stage(f"attention[variant={os.environ.get('PDSC_ATT_VARIANT', 'default')},nsplit={nsplit}]",
                        lambda: ops.sc_attention(qkv, compat, bs, n, nsplit=nsplit), args.iters, args.rounds, flops, "TFLOP/s")
        return

    if args.compat_child:
        bench_stage(f"compat[variant={os.environ.get('PDSC_COMPAT_VARIANT', 'default')}]",
                    lambda: ops.spatial_compat(src, tgt, sig), args.iters, args.rounds, (4.0 * n * n + 24.0 * n) * bs, "TB/s")
        return
    if want("compat"):
        buf = torch.empty(bs, n, ops.compat_ld(n), device=dev)
        buf2 = torch.empty_like(buf)
        nbytes = float(buf.numel() * 4)
        bench_stage("hbm_fill(torch.fill_)", lambda: buf.fill_(1.0), args.iters, args.rounds, nbytes, "TB/s")
        bench_stage("hbm_copy(torch.copy_, read+write bytes)", lambda: buf2.copy_(buf), args.iters, args.rounds, 2 * nbytes, "TB/s")
        for variant in os.environ.get("PDSC_MB_COMPAT_VARIANTS", "0,1").split(","):
            env = dict(os.environ, PDSC_COMPAT_VARIANT=variant)
            subprocess.run([sys.executable, __file__, "--n", str(n), "--bs", str(bs), "--iters", str(args.iters),
                            "--rounds", str(args.rounds), "--compat-child"], env=env, check=False)
    if want("attention"):
        splits = "0,1,2,3,4,6"
        for variant in ("0", "1"):
            env = dict(os.environ, PDSC_ATT_VARIANT=variant)
            subprocess.run([sys.executable, __file__, "--n", str(n), "--bs", str(bs), "--iters", str(args.iters),
                            "--rounds", str(args.rounds), "--att-child", splits], env=env, check=False)
    if want("linear"):
        x128 = torch.randn(M, 128, generator=gen).to(dev)
        x64 = torch.randn(M, 64, generator=gen).to(dev)
        for (k, nout, x) in ((128, 128, x128), (128, 384, x128), (128, 64, x128), (64, 64, x64), (64, 128, x64), (128, 32, x128)):
            w = torch.randn(nout, k, generator=gen).to(dev)
            b = torch.randn(nout, generator=gen).to(dev)
            bench_stage(f"linear[K={k},Nout={nout}]", lambda: ops.linear(x, w, b, relu=True), args.iters, args.rounds,
                        2.0 * M * k * nout, "TFLOP/s")
    if want("layer"):
        rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)  # noqa: E731
        msg, res = rnd(M, 128), rnd(M, 128)
        tail_w = [rnd(64, 128), rnd(64), rnd(64, 64), rnd(64), rnd(128, 64), rnd(128)]
        head_w = [rnd(128, 128), rnd(128), rnd(384, 128), rnd(384)]
        bench_stage("layer_fused[tail+head]", lambda: ops.layer_fused(msg, res, None, tail_w, head_w), args.iters, args.rounds,
                    2.0 * M * 86016, "TFLOP/s")
        bench_stage("layer_fused[head]", lambda: ops.layer_fused(None, None, res, None, head_w), args.iters, args.rounds,
                    2.0 * M * (16384 + 49152), "TFLOP/s")
        bench_stage("layer_fused[tail]", lambda: ops.layer_fused(msg, res, None, tail_w, None), args.iters, args.rounds,
                    2.0 * M * 20480, "TFLOP/s")
    if want("tail"):
        model = PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.1,
                         sigma_d=0.1, k=40, nms_radius=0.1)
        model.load_state_dict(synthetic.make_state_dict(model.state_dict(), seed=6))
        model = model.eval().to(dev)
        data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        data["testing"] = True
        bench_stage("forward(total)", lambda: model(data), 3, args.rounds, float(bs), "Tpairs/s")
        S = int(n * 0.1)
        normed = torch.nn.functional.normalize(torch.randn(bs, n, 128, generator=gen), dim=-1).to(dev)
        conf = torch.randn(bs, n, generator=gen).to(dev)
        bench_stage("nms_keys", lambda: ops.nms_keys(src, conf, 0.1), args.iters, args.rounds)
        keys = ops.nms_keys(src, conf, 0.1)
        bench_stage("rank_select", lambda: ops.rank_select(keys, S), args.iters, args.rounds)
        seeds = ops.rank_select(keys, S)
        bench_stage("knn_seeds", lambda: ops.knn_seeds(normed, seeds, 40), args.iters, args.rounds)
        knn = ops.knn_seeds(normed, seeds, 40)
        bench_stage("seed_power_iteration", lambda: ops.seed_power_iteration(normed, src, tgt, knn, sig * 10, sig, 10),
                    args.iters, args.rounds)
        iters, mask, _ = ops.seed_power_iteration(normed, src, tgt, knn, sig * 10, sig, 10)
        bench_stage("seed_transforms", lambda: ops.seed_transforms(src, tgt, knn, iters, mask, 10), args.iters, args.rounds)
        trans, _ = ops.seed_transforms(src, tgt, knn, iters, mask, 10)
        bench_stage("score+select", lambda: ops.score_hypotheses(trans, src, tgt, 0.1), args.iters, args.rounds)
        init = batch["gt_trans"].to(dev).clone()
        init[:, :3, 3] += 0.02
        bench_stage("post_refinement", lambda: ops.post_refinement(init, src, tgt, 0.1, 20), args.iters, args.rounds)


if __name__ == "__main__":
    main()
