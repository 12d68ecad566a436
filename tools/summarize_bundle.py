#!/usr/bin/env python3
"""Markdown tables from the bench lines of an evidence bundle (tools/gpu_profile_run.sh <tag>).

    python tools/summarize_bundle.py profiles r03
"""
import json
import sys
from pathlib import Path

d, tag = Path(sys.argv[1]), sys.argv[2]


def load(name):
    p = d / f"{tag}_bench_line_{name}.json"
    try:
        return json.loads(p.read_text())
    except Exception:
        return None


def frac(x):
    return "—" if x is None else f"{x:.3f}"


print("| config | pairs/s (K=20 timed steps, forwards in flight) | single stream | sustained | ms/step | attention launch, executed frac of the 16-bit MFMA peak | "
      "fused layer launch, frac of 8 TB/s | compat build, frac of 8 TB/s | reference CPU path | check: max dT vs reference / oracle |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, label in (("n5000_b32", "`n5000_b32` (configs[2], headline)"), ("n1000_b1", "`n1000_b1` (configs[1])"),
                    ("kitti_n5000_b16", "`kitti_n5000_b16` (configs[3])"), ("lomatch_n10000_b8", "`lomatch_n10000_b8` (configs[4])")):
    r = load(name)
    if not r:
        continue
    c = r.get("check", {})
    cb = r.get("cpu_baseline", {})
    print(f"| {label} | **{r['value']:.0f}** ({r['in_flight']} in flight) | {r.get('single_stream', {}).get('value', float('nan')):.0f} | "
          f"{r.get('sustained', {}).get('value', float('nan')):.0f} | {r['ms_per_step']:.2f} | {r['roofline']['avg_launch_ms']:.3f} ms, "
          f"{frac(r['roofline'].get('executed_frac'))} | {r['roofline_layer']['avg_launch_ms']:.3f} ms, {frac(r['roofline_layer']['frac'])} | "
          f"{r['roofline_compat']['avg_launch_ms']:.3f} ms, {frac(r['roofline_compat']['frac'])} | "
          f"{cb.get('value', float('nan'))} ({cb.get('kind', '?')}, {cb.get('cores', '?')} thr) | "
          f"{c.get('max_abs_dT_vs_reference', float('nan')):.1e} / {c.get('max_abs_dT_vs_oracle', float('nan')):.1e} ({'ok' if c.get('ok') else c.get('ok')}) |")
print()
print("| configuration | pairs per GPU (GPUs of the run) | pairs/s per GPU | single stream | sustained | ms/step | attention executed frac | expected aggregate | expected efficiency |")
print("|---|---|---|---|---|---|---|---|---|")
for name, full, shares in (("n5000_b32", 32, (16, 8, 4)), ("kitti_n5000_b16", 16, (8, 4, 2)), ("lomatch_n10000_b8", 8, (4, 2, 1))):
    base = load(name)
    if not base:
        continue
    for B in (full,) + shares:
        r = base if B == full else load(f"{name}_{B}pairs")
        if not r:
            continue
        g = full // B
        print(f"| `{name}` | {B} ({g}) | {r['value']:.0f} | {r.get('single_stream', {}).get('value', float('nan')):.0f} | "
              f"{r.get('sustained', {}).get('value', float('nan')):.0f} | {r['ms_per_step']:.3f} | {frac(r['roofline'].get('executed_frac'))} | "
              f"{g * r['value']:.0f} | {r['value'] / base['value'] * 100:.0f} % |")
