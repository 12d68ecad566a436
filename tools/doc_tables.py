#!/usr/bin/env python3
"""Writes profiles/<tag>_tables.md -- the round's full result tables -- from the evidence bundle under profiles/
(tools/gpu_profile_run.sh <tag>).  DESIGN.md quotes the headline rows and points here (r06: the tables left DESIGN.md, which had
grown to 119 KB).   python tools/doc_tables.py [tag, default r06]"""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def L(name):
    return json.loads((ROOT / "profiles" / f"{TAG}_bench_line_{name}.json").read_text().strip().splitlines()[-1])


def bench_table():
    out = ["| config | pairs/s (K=20 timed steps, forwards in flight) | single stream | sustained | ms/step | attention launch, executed frac of the 16-bit MFMA peak | fused layer launch, frac of 8 TB/s on the minimal 3072 B/point (on the bytes moved) | compat build, frac of 8 TB/s | socket power in the sustained leg (share of the cap), J per pair | reference CPU path | check: max dT vs reference / oracle |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    names = [("n5000_b32", "configs[2], headline"), ("n1000_b1", "configs[1]"), ("kitti_n5000_b16", "configs[3]"), ("lomatch_n10000_b8", "configs[4]"),
             ("kitti_n12000_b4", "the reference's KITTI evaluation size"), ("multiway_n20000_b1", "the reference's multiway size"),
             ("trained_n5000_b32", "trained-like weights, N = 5000"), ("trained_n1000_b1", "trained-like weights, N = 1000"),
             ("trained_kitti_n5000_b16", "trained-like weights, KITTI scale"), ("trained_lomatch_n10000_b8", "trained-like weights, N = 10 000"),
             ("n5000_b32_per_launch_leaves", "headline with `att_leaves = per_launch`")]
    for n, lab in names:
        if not (ROOT / "profiles" / f"{TAG}_bench_line_{n}.json").exists():
            continue
        l = L(n); r = l["roofline"]; rl = l["roofline_layer"]; rc = l["roofline_compat"]; c = l["check"]; cb = l.get("cpu_baseline", {})
        pw = l.get("power")
        pws = "n/a" if not pw else f"{pw['mean_w']:.0f} W ({100 * pw['frac_of_cap']:.0f} %), {pw['joule_per_pair']:.2f} J"
        out.append(f"| `{n}` ({lab}) | **{l['value']:.0f}** ({l['in_flight']} in flight) | {l['single_stream']['value']:.0f} | {l['sustained']['value']:.0f} | "
                   f"{l['ms_per_step']:.2f} | {r['avg_launch_ms']:.3f} ms, {r['executed_frac']:.3f} | {rl['avg_launch_ms']:.3f} ms, {rl['frac']:.3f}" + (f" (moved {rl['moved_frac']:.3f})" if rl.get('moved_frac') else "") + f" | "
                   f"{rc['avg_launch_ms']:.3f} ms, {rc['frac']:.3f} | {pws} | {cb.get('value')} ({cb.get('kind')}, {cb.get('cores')} thr) | "
                   f"{c.get('max_abs_dT_vs_reference') or float('nan'):.1e} / {c.get('max_abs_dT_vs_oracle') or float('nan'):.1e} ({'ok' if c['ok'] else 'FAIL'}) |")
    return "\n".join(out)


def share_table():
    out = ["| configuration | pairs per GPU (GPUs of the run) | pairs/s per GPU | single stream | sustained | ms/step | attention executed frac | expected aggregate | expected efficiency |",
           "|---|---|---|---|---|---|---|---|---|"]
    for cfg, full, shares in (("n5000_b32", 32, (16, 8, 4)), ("kitti_n5000_b16", 16, (8, 4, 2)), ("lomatch_n10000_b8", 8, (4, 2, 1))):
        base = L(cfg)["value"]
        for B in (full,) + shares:
            l = L(cfg if B == full else f"{cfg}_{B}pairs"); g = full // B
            out.append(f"| `{cfg}` | {B} ({g}) | {l['value']:.0f} | {l['single_stream']['value']:.0f} | {l['sustained']['value']:.0f} | {l['ms_per_step']:.3f} | "
                       f"{l['roofline']['executed_frac']:.3f} | {l['value'] * g:.0f} | {100 * l['value'] / base:.0f} % |")
    return "\n".join(out)


def census(*paths):
    """census_table.py over the bundle's file and, behind it, the rows of records taken after the bundle (families added later)."""
    out = []
    for i, path in enumerate(paths):
        f = ROOT / "profiles" / path
        if not f.exists():
            continue
        t = subprocess.run([sys.executable, str(ROOT / "tools" / "census_table.py"), str(f)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
        out += t if not out else t[2:]
    return "\n".join(out)


def latency_table():
    out = ["| workload, ONE pair per call, pose + labels read back before the next call | ms per call | pairs/s | attention launch | fused layer launch | key split x leaves |",
           "|---|---|---|---|---|---|"]
    for n in ("n5000_b32", "n1000_b1", "trained_n1000_b1", "lomatch_n10000_b8"):
        f = ROOT / "profiles" / f"{TAG}_bench_line_{n}_latency_1pair.json"
        if not f.exists():
            continue
        l = json.loads(f.read_text().strip().splitlines()[-1])
        pl = l["config"]["attention_plan"]
        out.append(f"| `{n}` (N = {l['config']['num_corr']}) | **{l['ms_per_step']:.3f}** | {l['value']:.0f} | {l['roofline']['avg_launch_ms'] * 1e3:.1f} us | "
                   f"{l['roofline_layer']['avg_launch_ms'] * 1e3:.1f} us | {pl['key_split']} x {pl['leaves']} |")
    return "\n".join(out)


BLOCKS = {"bench": bench_table, "shares": share_table, "latency": latency_table, "census": lambda: census(f"{TAG}_parity_census.txt"),
          "census_fp32": lambda: census(f"{TAG}_parity_census_exact_fp32.txt")}

TITLES = {"bench": "One bench line per configuration (shipped defaults, one MI355X)",
          "shares": "Per-GPU shares of the 2- / 4- / 8-GPU strong-scaling runs, measured on one GPU (no curve has been measured on hardware)",
          "latency": "The number an unchanged caller sees: one pair per call, pose + labels read back (bench.py --latency)",
          "census": "Parity census, shipped arithmetic, every family at every batch size (tools/parity_census.py)",
          "census_fp32": "Parity census, exact-fp32 arithmetic"}

if __name__ == "__main__":
    out = [f"# Result tables of round {TAG[1:]} (generated by tools/doc_tables.py from profiles/{TAG}_*)", ""]
    for name, fn in BLOCKS.items():
        try:
            body = fn()
        except Exception as e:  # noqa: BLE001
            body = f"(not generated: {e!r})"
        out += [f"## {TITLES[name]}", "", body, ""]
    st = ROOT / "profiles" / f"{TAG}_stage_census.txt"
    if st.exists():
        out += ["## Stage census on the trained-like checkpoints (test_trained_checkpoint_stage_decisions_follow_the_reference)", "", "```", st.read_text().strip(), "```", ""]
    (ROOT / "profiles" / f"{TAG}_tables.md").write_text("\n".join(out))
    print(f"profiles/{TAG}_tables.md written")
