#!/usr/bin/env python3
"""Power and clock under the WHOLE forward: how much of the step runs on the board's power cap, and what a pair costs in joules.

    python tools/forward_power.py [--config n5000_b32] [--pairs 32] [--seconds 5]

Companion of tools/attention_power.py (same sampler: the hwmon of the GPU this process loads, found by calibration).  Arms:
  * the forward on one stream (InFlight depth 1),
  * the forward as bench.py times it (two forwards in flight, tail streams),
  * the attention launch alone and the compat build alone (the two kernels bench.py's roofline objects describe), for scale.
Per arm: steps/s, pairs/s, mean / max socket power, mean shader clock, joules per pair.  From the one-stream arm and the
attention arm the mean power of everything that is NOT attention follows (the attention launches are 12 x their measured
duration of the step and draw the cap); if the in-flight arm averages the cap, overlapping forwards has nothing left to win.
"""
import argparse
import importlib.util
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
_spec = importlib.util.spec_from_file_location("attention_power", ROOT / "tools" / "attention_power.py")
ap_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ap_mod)


def arm(name, step, sync, seconds, pairs_per_step, chunk=8):
    for _ in range(4):
        step()
    sync()
    smp = ap_mod.Sampler()
    smp.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(chunk):
            step()
        n += chunk
        if n % (4 * chunk) == 0:
            sync()                       # bounded queue depth; the forwards in flight keep overlapping between the syncs
    sync()
    el = time.perf_counter() - t0
    smp.stop_flag = True
    smp.join()
    s = [x for x in smp.samples if x[0] - t0 > 0.3 * seconds and x[1] == x[1]]
    pw = [x[1] for x in s]
    ck = [x[2] for x in s if x[2] == x[2]]
    mean_pw = sum(pw) / len(pw) if pw else float("nan")
    rate = n / el
    rec = {"arm": name, "steps_per_s": round(rate, 2), "ms_per_step": round(1e3 / rate, 4),
           "pairs_per_s": round(rate * pairs_per_step, 1) if pairs_per_step else None,
           "mean_power_w": round(mean_pw, 1), "max_power_w": round(max(pw), 1) if pw else None,
           "share_of_samples_at_the_cap": round(sum(1 for p in pw if p >= 1390.0) / max(len(pw), 1), 3),
           "mean_sclk_mhz": round(sum(ck) / len(ck), 0) if ck else None,
           "joule_per_step": round(mean_pw / rate, 4),
           "joule_per_pair": round(mean_pw / rate / pairs_per_step, 4) if pairs_per_step else None, "samples": len(s)}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="n5000_b32")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per step (default: the configuration's batch)")
    ap.add_argument("--seconds", type=float, default=5.0)
    a = ap.parse_args()
    import torch
    from pointdsc_amd import ops, workloads
    from pointdsc_amd.model import PointDSC
    from pointdsc_amd.pipeline import InFlight
    w = workloads.WORKLOADS[a.config]
    B = a.pairs or w["global_batch"]
    N = w["num_corr"]
    dev = "cuda:0"
    model = PointDSC(**w["model"]).eval().to(dev)
    model.load_state_dict(workloads.state_dict(a.config, model.state_dict()))
    batch = workloads.batch(a.config, 0, B)
    data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    sync = torch.cuda.synchronize
    with torch.no_grad():
        model(data)
    sync()

    src, tgt = data["src_keypts"], data["tgt_keypts"]
    sig = torch.tensor([float(w["model"].get("sigma_d", 0.1))], device=dev)
    c16 = ops.spatial_compat_u16(src, tgt, sig)
    gen = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * N, 384, generator=gen) * 0.3).to(dev)
    qs, kv = ops.pack_qkv_split(qkv, B, N)
    ap_mod.calibrate(lambda: ops.sc_attention_split(qs, kv, c16, B, N))

    out = []
    for depth in (1, 2):
        runner = InFlight(model, depth=depth)
        with torch.no_grad():
            out.append(arm(f"forward, {B} pairs of N={N}, {depth} in flight" + (" (bench.py's timed schedule)" if depth == 2 else " (one stream)"),
                           lambda: runner(data), sync, a.seconds, B))
        runner.synchronize()
        runner.close()
    out.append(arm("attention launch alone (random operands)", lambda: ops.sc_attention_split(qs, kv, c16, B, N, merge=False), sync, a.seconds, 0, chunk=50))
    out.append(arm("compat build alone (unorm16)", lambda: ops.spatial_compat_u16(src, tgt, sig), sync, a.seconds, 0, chunk=100))
    one, att = out[0], out[2]
    t_step, t_att = one["ms_per_step"], 12 * att["ms_per_step"]
    if t_step > t_att:
        rest = (one["mean_power_w"] * t_step - att["mean_power_w"] * t_att) / (t_step - t_att)
        print(json.dumps({"derived": "mean power of everything that is not attention, one stream",
                          "attention_ms_per_step": round(t_att, 3), "other_ms_per_step": round(t_step - t_att, 3), "other_mean_power_w": round(rest, 1)}))
    time.sleep(0.5)
    idle = ap_mod.Sampler()
    pw, ck = idle.read_once()
    print(json.dumps({"arm": "idle", "power_w": pw, "sclk_mhz": ck}))


if __name__ == "__main__":
    main()
