#!/usr/bin/env python3
"""Throughput bench of the PointDSC outlier-rejection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Metric (BASELINE.json): point-cloud pairs/sec at N=5000 correspondences.  One "step" = one pass of the whole
hot path (pdsc_forward_testing: compat build, 12 SCNonlocal layers, seeds, per-seed solver, scoring,
refinement) over one batch of 32 synthetic correspondence sets (BASELINE.json configs[2]) sharded over the
GPUs (32 / N per GPU: strong scaling; `--pairs-per-gpu` fixes the per-GPU batch instead), inputs already
resident in HBM.  Pairs are independent units: every rank processes its own shard and the only collective
is the final all_gather of the poses (RCCL), which is inside the timed region.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      -- the dominant kernel (sc_attention_kernel, MFMA-bound, fp32 in/fp32 acc): algorithmic
                   flops per launch / average launch duration measured with hipEvents on the launch stream
                   over the timed region (pdsc_profile_* in include/pointdsc_hip.h);
  roofline_compat -- same for the compat-matrix build (HBM-write-bound), the kernel north_star names;
  cpu_baseline  -- the CPU oracle (a torch-CPU restatement of the reference, kind "port") timed on this
                   host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL_KW = dict(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                sigma_d=0.10, k=40, nms_radius=0.10)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s achievable)


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline_worker(num_corr: int, pairs: int, threads: int) -> None:
    """Child process: time the CPU oracle on `pairs` pairs of the bench workload; prints one JSON line."""
    from oracle import pointdsc_oracle as O
    from pointdsc_amd import PointDSC, synthetic
    torch.set_num_threads(threads)
    model = PointDSC(**MODEL_KW)
    sd = synthetic.make_state_dict(model.state_dict(), seed=6)
    batch = synthetic.make_batch(pairs, num_corr, seed=1000, inlier_ratio=0.2)
    okw = {k: MODEL_KW[k] for k in ("num_layers", "num_channels", "num_iterations", "ratio", "inlier_threshold",
                                   "k", "nms_radius")}
    with torch.no_grad():
        first = O.forward_testing(sd, batch["corr_pos"][:1], batch["src_keypts"][:1], batch["tgt_keypts"][:1], **okw)
        t1 = time.perf_counter()
        for i in range(pairs):
            O.forward_testing(sd, batch["corr_pos"][i:i + 1], batch["src_keypts"][i:i + 1], batch["tgt_keypts"][i:i + 1], **okw)
        dt = time.perf_counter() - t1
    print(json.dumps({"pairs_per_s": pairs / dt, "seconds": dt, "threads": threads,
                      "first_trans": first["final_trans"][0].tolist()}), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--num-corr", type=int, default=5000, help="N correspondences per pair (headline: 5000)")
    ap.add_argument("--global-batch", type=int, default=32,
                    help="pairs per step over ALL GPUs (BASELINE.json configs[2]: 32 pairs sharded over the GPUs -> strong scaling)")
    ap.add_argument("--pairs-per-gpu", type=int, default=0,
                    help="override: fixed batch per GPU per step (weak scaling); 0 = global-batch / gpus")
    ap.add_argument("--attention-precision", choices=["bf16x3", "fp32", "bf16x3_all"], default="bf16x3",
                    help="arithmetic of the attention contractions: split-precision bf16 MFMA (default) or exact fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs timed on the CPU oracle (bounded sample)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = min(host cores, 32))")
    ap.add_argument("--cpu-timeout", type=float, default=240.0, help="wall-clock cap for the CPU baseline leg")
    ap.add_argument("--check", action="store_true", help="also verify rank-0's first pair against the oracle")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.num_corr, args.cpu_pairs, args.cpu_threads or (os.cpu_count() or 1))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from pointdsc_amd import PointDSC, _lib, sharding, synthetic
    lib = _lib.load()
    if args.pairs_per_gpu > 0:
        B, scaling = args.pairs_per_gpu, "weak"
    else:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} GPUs")
        B, scaling = args.global_batch // world, "strong"
    N = args.num_corr
    model = PointDSC(**MODEL_KW)
    sd = synthetic.make_state_dict(model.state_dict(), seed=6)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    model.attention_precision = args.attention_precision
    # each rank owns its shard of the global batch: pairs [rank*B, (rank+1)*B)
    batch = synthetic.make_batch(B, N, seed=1000 + rank * B, inlier_ratio=0.2)
    data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    total_pairs = B * world

    def step():
        with torch.no_grad():
            res = model(data)
        return sharding.gather_results(res["final_trans"], None, total_pairs)

    log(f"rank {rank}: model + {B} pairs (N={N}) resident on {dev}; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    log("warm-up done; timing")

    n_att = args.steps * MODEL_KW["num_layers"]
    _lib.check(lib.pdsc_profile_enable(n_att + 8), "pdsc_profile_enable")
    _lib.check(lib.pdsc_profile_reset(), "pdsc_profile_reset")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f}s for {args.steps} steps")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel, from the events recorded during the timed region ----
    def read(kind):
        ms, n = C.c_double(0), C.c_int(0)
        _lib.check(lib.pdsc_profile_read(kind, C.byref(ms), C.byref(n)), "pdsc_profile_read")
        return ms.value, n.value

    att_ms, att_n = read(0)
    cmp_ms, cmp_n = read(1)
    lay_ms, lay_n = read(2)
    _lib.check(lib.pdsc_profile_enable(0), "pdsc_profile_enable(0)")
    # fused layer launch (tail of layer i + head of layer i+1): 2 flop/MAC x (128*64 + 64*64 + 64*128 + 128*128 + 128*384)
    # MACs per point; matrix-pipe cycles per 32-point tile: 600 fp32 MFMAs x 64 + 288 bf16 MFMAs x 32 (q|k|v as bf16x3)
    lay_flops = 2.0 * 86016 * N * B
    lay_avg = lay_ms / max(lay_n, 1) * 1e-3
    lay_tflops = lay_flops / lay_avg / 1e12 if lay_n else None
    lay_pipe_cycles = (600 * 64 + 288 * 32) * math.ceil(N / 32) * B / 1024.0      # per SIMD (256 CUs x 4)
    att_flops = 4.0 * 128 * float(N) * float(N) * B           # 2 GEMMs x 2 flop/MAC x C x N^2 per pair, per launch
    att_avg = att_ms / max(att_n, 1) * 1e-3
    att_tflops = att_flops / att_avg / 1e12 if att_n else None
    cmp_bytes = (4.0 * N * N + 24.0 * N) * B                  # compat write + keypoint reads
    cmp_avg = cmp_ms / max(cmp_n, 1) * 1e-3
    cmp_gbs = cmp_bytes / cmp_avg / 1e9 if cmp_n else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = total_pairs * args.steps / elapsed
    line = {
        "metric": "point-cloud pairs/sec @ N=%d corr" % N,
        "value": round(value, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32" if args.attention_precision == "fp32" else "f32 (attention products as bf16x3 split, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "3DMatch-like synthetic correspondences (BASELINE.json configs[2]): N=%d corr, %d pairs per "
                               "step sharded over %d GPU(s) = %d per GPU, 12-layer PointDSC, seeded random weights"
                               % (N, total_pairs, world, B),
                   "num_corr": N, "pairs_per_gpu": B, "global_batch": total_pairs,
                   "parallelism": "pairs sharded over %d GPU(s), one all_gather of poses" % world},
        "roofline": ({"kernel": "sc_attention_kernel", "bound": "mfma",
                      "achieved": None if att_tflops is None else round(att_tflops, 2),
                      "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                      "frac": None if att_tflops is None else round(att_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                      "traffic": None, "launches": att_n,
                      "avg_launch_ms": round(att_avg * 1e3, 4), "flops_per_launch": att_flops}
                     if args.attention_precision == "fp32" else
                     # split precision: `achieved` counts ALGORITHMIC flops (4 C N^2 per pair per launch) against the
                     # dense bf16 MFMA peak; the kernel executes 3 bf16 MFMAs per algorithmic product (hi*hi, hi*lo,
                     # lo*hi), so its matrix-pipe utilisation is 3 x frac (`executed_frac`).
                     {"kernel": "sc_attention_split_kernel", "bound": "mfma",
                      "achieved": None if att_tflops is None else round(att_tflops, 2),
                      "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                      "frac": None if att_tflops is None else round(att_tflops / PEAK_BF16_MFMA_TFLOPS, 4),
                      "executed_tflops": None if att_tflops is None else round(3 * att_tflops, 2),
                      "executed_frac": None if att_tflops is None else round(3 * att_tflops / PEAK_BF16_MFMA_TFLOPS, 4),
                      "equivalent_fp32_mfma_frac": None if att_tflops is None else round(att_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                      "traffic": None, "launches": att_n,
                      "avg_launch_ms": round(att_avg * 1e3, 4), "flops_per_launch": att_flops}),
        "roofline_layer": {"kernel": "layer_wave_kernel", "bound": "mfma",
                           "achieved": None if lay_tflops is None else round(lay_tflops, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": None if lay_tflops is None else round(lay_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                           "matrix_pipe_busy_frac_at_2.4GHz": None if not lay_n else round(lay_pipe_cycles / (lay_avg * 2.4e9), 4),
                           "traffic": None, "launches": lay_n, "avg_launch_ms": round(lay_avg * 1e3, 4),
                           "flops_per_launch": lay_flops},
        "roofline_compat": {"kernel": "compat_sym_kernel", "bound": "hbm",
                            "achieved": None if cmp_gbs is None else round(cmp_gbs, 1), "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": None if cmp_gbs is None else round(cmp_gbs / PEAK_HBM_GBS, 4),
                            "traffic": None, "launches": cmp_n, "avg_launch_ms": round(cmp_avg * 1e3, 4),
                            "bytes_per_launch": cmp_bytes},
    }
    traffic_file = ROOT / "profiles" / "traffic.json"      # PMC-derived HBM bytes per launch, if collected
    if traffic_file.exists():
        try:
            tj = json.loads(traffic_file.read_text())
            key = f"N{N}_B{B}"
            if key in tj:
                line["roofline"]["traffic"] = tj[key].get(line["roofline"]["kernel"])
                line["roofline_compat"]["traffic"] = tj[key].get("compat_sym_kernel")
                line["roofline_layer"]["traffic"] = tj[key].get("layer_wave_kernel")
        except Exception:
            pass

    # ---- CPU baseline: the oracle on this host's cores, bounded sample (rank 0, N=1 only), in a child
    #      process with a wall-clock cap so the bench always finishes ----
    if world == 1 and not args.no_cpu_baseline:
        import subprocess
        # 32 intra-op threads is the fastest setting measured on the 256-core GPU box (16: 0.42, 32: 0.59, 64: 0.33,
        # 128: 0.19 pairs/s; 256 threads did not finish 3 pairs in 240 s), so that is what the baseline gets
        cores = args.cpu_threads or min(os.cpu_count() or 1, 32)
        n_cpu = max(1, min(args.cpu_pairs, B))
        log(f"CPU baseline: oracle on {n_cpu} pair(s), {cores} threads (cap {args.cpu_timeout:.0f}s)")
        cmd = [sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-worker", "--num-corr", str(N),
               "--cpu-pairs", str(n_cpu), "--cpu-threads", str(cores)]
        sample = ("%d pair(s) of the same N=%d workload after 1 warm-up, torch-CPU oracle "
                  "(oracle/pointdsc_oracle.py), %d intra-op threads" % (n_cpu, N, cores))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout)
            cj = json.loads(r.stdout.strip().splitlines()[-1])
            line["cpu_baseline"] = {"value": round(cj["pairs_per_s"], 4), "unit": "pairs/s", "cores": cores,
                                    "kind": "port", "sample": sample}
            if args.check:
                dT = float((out["final_trans"][0].cpu() - torch.tensor(cj["first_trans"])).abs().max())
                line["check"] = {"max_abs_dT_vs_oracle": dT}
        except subprocess.TimeoutExpired:
            line["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": cores, "kind": "port",
                                    "sample": sample + " -- did not finish within %.0fs" % args.cpu_timeout}
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": cores, "kind": "port",
                                    "sample": sample + " -- failed: %r" % (e,)}
        log("CPU baseline done")
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
