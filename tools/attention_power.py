#!/usr/bin/env python3
"""Power and clock under the attention launch: is sc_attention_split_kernel bounded by the chip's power budget?

    python tools/attention_power.py [--n 5000] [--bs 32] [--seconds 3] [--wide]

Runs the split-precision attention launch back to back for --seconds on (a) random operands, (b) all-zero operands (same
instruction stream, no data toggling: MI355X_MICROARCH.md "DVFS give-back"), optionally (c) the 64-queries-per-wave record kernel
(the r01-r04 record kernel `sc_attention_wide_kernel` was removed in r05: profiles/HISTORY.md), while a sampler thread reads the socket power and the shader clock every ~20 ms from
sysfs (hwmon power1_average / power1_input, freq1_input) or, failing that, `amd-smi metric` / `rocm-smi`.  For every arm it
prints launches/s, executed TFLOP/s (3 f16 MFMAs per algorithmic product), mean / max power, mean shader clock, and the
ENERGY PER EXECUTED MFMA (joules per v_mfma_f32_32x32x16_f16 wave-instruction) -- the figure that says whether two forms of the
kernel differ in what they ask of the power budget.  A chip that runs the random arm at its power cap with a lower clock than the
zero arm is power-bound on this kernel.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


_CHOSEN_HWMON = None      # set by calibrate(): the hwmon directory of the GPU this process actually loads


def all_hwmons():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(hw, name)):
                out.append((hw, os.path.join(hw, name)))
                break
    return out


def calibrate(launch):
    """A box may expose several GPUs in sysfs while this process sees one: pick the hwmon whose power moves most between idle and
    one second of the launch loop (and print every candidate's figures, so that the choice can be checked)."""
    global _CHOSEN_HWMON
    import torch
    cands = all_hwmons()
    if not cands:
        return
    rd = lambda p: int(open(p).read()) * 1e-6
    torch.cuda.synchronize()
    time.sleep(0.5)
    idle = [rd(p) for _, p in cands]
    t0 = time.perf_counter()
    peak = list(idle)
    while time.perf_counter() - t0 < 1.5:
        for _ in range(20):
            launch()
        peak = [max(a, rd(p)) for a, (_, p) in zip(peak, cands)]
        torch.cuda.synchronize()
    deltas = [b - a for a, b in zip(idle, peak)]
    k = max(range(len(cands)), key=lambda i: deltas[i])
    _CHOSEN_HWMON = cands[k][0]
    print(json.dumps({"calibration": [{"hwmon": hw, "idle_w": round(a, 1), "peak_under_load_w": round(b, 1)} for (hw, _), a, b in zip(cands, idle, peak)],
                      "chosen": _CHOSEN_HWMON}), flush=True)


def sysfs_sources():
    srcs = {}
    for hw in ([_CHOSEN_HWMON] if _CHOSEN_HWMON else glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("power1_average", "power1_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p) and "power" not in srcs:
                srcs["power"] = p
        p = os.path.join(hw, "freq1_input")
        if os.path.exists(p) and "sclk" not in srcs:
            srcs["sclk"] = p
        p = os.path.join(hw, "energy1_input")                  # accumulated microjoules, if the driver exposes it: exact mean power
        if os.path.exists(p) and "energy" not in srcs:
            srcs["energy"] = p
    return srcs


def read_energy_uj(srcs):
    try:
        return int(open(srcs["energy"]).read()) if "energy" in srcs else None
    except Exception:       # noqa: BLE001
        return None


class Sampler(threading.Thread):
    def __init__(self, period=0.02):
        super().__init__(daemon=True)
        self.period, self.stop_flag, self.samples = period, False, []
        self.srcs = sysfs_sources()
        self.mode = "sysfs" if "power" in self.srcs else "smi"

    def read_once(self):
        if self.mode == "sysfs":
            try:
                pw = int(open(self.srcs["power"]).read()) * 1e-6                      # microwatts
                ck = int(open(self.srcs["sclk"]).read()) * 1e-6 if "sclk" in self.srcs else float("nan")      # Hz -> MHz
                return pw, ck
            except Exception:       # noqa: BLE001
                self.mode = "smi"
        try:
            out = subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True,
                                 text=True, timeout=5).stdout
            j = json.loads(out)
            j = j[0] if isinstance(j, list) else j
            j = j.get("gpu_data", [j])[0] if isinstance(j, dict) and "gpu_data" in j else j
            pw = j["power"]["socket_power"]
            pw = float(pw["value"] if isinstance(pw, dict) else pw)
            ck = j["clock"]["gfx_0"]["clk"]
            ck = float(ck["value"] if isinstance(ck, dict) else ck)
            return pw, ck
        except Exception:       # noqa: BLE001
            return float("nan"), float("nan")

    def run(self):
        while not self.stop_flag:
            t = time.perf_counter()
            pw, ck = self.read_once()
            self.samples.append((t, pw, ck))
            time.sleep(self.period if self.mode == "sysfs" else 0.2)


def arm(name, launch, seconds, flops_exec, mfma_per_launch):
    import torch
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    smp = Sampler()
    e_start = read_energy_uj(smp.srcs)
    smp.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            launch()
        n += 50
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    e_end = read_energy_uj(smp.srcs)
    smp.stop_flag = True
    smp.join()
    s = [x for x in smp.samples if x[0] - t0 > 0.3 * seconds and x[1] == x[1]]       # drop the ramp
    pw = [x[1] for x in s]
    ck = [x[2] for x in s if x[2] == x[2]]
    mean_pw = sum(pw) / len(pw) if pw else float("nan")
    mean_ck = sum(ck) / len(ck) if ck else float("nan")
    rate = n / el
    rec = {"arm": name, "launches_per_s": round(rate, 1), "ms_per_launch": round(1e3 / rate, 4), "executed_tflops": round(flops_exec * rate / 1e12, 1),
           "mean_power_w": round(mean_pw, 1), "max_power_w": round(max(pw), 1) if pw else None, "mean_sclk_mhz": round(mean_ck, 0),
           "joule_per_launch": round(mean_pw / rate, 4), "nanojoule_per_executed_mfma": round(mean_pw / rate / mfma_per_launch * 1e9, 2),
           "samples": len(s), "sampler": smp.mode,
           "power_first_last_w": [round(pw[0], 1), round(pw[-1], 1)] if pw else None,
           "energy_counter_mean_power_w": round((e_end - e_start) * 1e-6 / el, 1) if (e_start is not None and e_end is not None) else None}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--wide", action="store_true", help="also the split kernel on the fp32 matrix")
    a = ap.parse_args()
    if a.wide:
        os.environ.setdefault("POINTDSC_HIP_LIB", str(ROOT / "pointdsc_amd" / "libpointdsc_hip_exp.so"))
    import torch
    from pointdsc_amd import ops, synthetic
    n, bs, dev = a.n, a.bs, "cuda:0"
    batch = synthetic.make_batch(min(bs, 4), n, seed=1)
    rep = (bs + min(bs, 4) - 1) // min(bs, 4)
    src = batch["src_keypts"].repeat(rep, 1, 1)[:bs].to(dev)
    tgt = batch["tgt_keypts"].repeat(rep, 1, 1)[:bs].to(dev)
    c16 = ops.spatial_compat_u16(src, tgt, torch.tensor([0.1], device=dev))
    c32 = ops.spatial_compat(src, tgt, torch.tensor([0.1], device=dev)) if a.wide else None
    gen = torch.Generator().manual_seed(0)
    flops_exec = 3 * 4.0 * 128 * n * n * bs                       # executed: three f16 MFMAs per algorithmic product
    mfma = flops_exec / (2.0 * 32 * 32 * 16)                      # v_mfma_f32_32x32x16_f16 wave-instructions per launch
    out = []
    for label, scale in (("random", 0.3), ("zeros", 0.0)):
        qkv = (torch.randn(bs * n, 384, generator=gen) * scale).to(dev)
        qs, kv = ops.pack_qkv_split(qkv, bs, n)
        if _CHOSEN_HWMON is None:
            calibrate(lambda: ops.sc_attention_split(qs, kv, c16, bs, n))
        out.append(arm(f"split kernel, unorm16 matrix, {label} operands", lambda: ops.sc_attention_split(qs, kv, c16, bs, n), a.seconds, flops_exec, mfma))
        if a.wide:
            out.append(arm(f"split kernel, fp32 matrix, {label} operands", lambda: ops.sc_attention_split(qs, kv, c32, bs, n), a.seconds, flops_exec, mfma))
    idle = Sampler()
    time.sleep(0.5)
    pw, ck = idle.read_once()
    print(json.dumps({"arm": "idle", "power_w": pw, "sclk_mhz": ck}))


if __name__ == "__main__":
    main()
