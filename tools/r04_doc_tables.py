#!/usr/bin/env python3
"""Regenerates the round-4 result tables of DESIGN.md from the evidence bundle under profiles/ (tools/gpu_profile_run.sh r04):
the text between `<!-- r04:<name> -->` and `<!-- /r04:<name> -->` is replaced.   python tools/r04_doc_tables.py"""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def L(name):
    return json.loads((ROOT / "profiles" / f"r04_bench_line_{name}.json").read_text().strip().splitlines()[-1])


def bench_table():
    out = ["| config | pairs/s (K=20 timed steps, forwards in flight) | single stream | sustained | ms/step | attention launch, executed frac of bf16 peak | fused layer launch, frac of 8 TB/s | compat build, frac of 8 TB/s | socket power in the sustained leg (share of the cap), J per pair | reference CPU path | check: max dT vs reference / oracle |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    names = [("n5000_b32", "configs[2], headline"), ("n1000_b1", "configs[1]"), ("kitti_n5000_b16", "configs[3]"), ("lomatch_n10000_b8", "configs[4]"),
             ("kitti_n12000_b4", "the reference's KITTI evaluation size"), ("multiway_n20000_b1", "the reference's multiway size")]
    for n, lab in names:
        l = L(n); r = l["roofline"]; rl = l["roofline_layer"]; rc = l["roofline_compat"]; c = l["check"]; cb = l.get("cpu_baseline", {})
        pw = l.get("power")
        pws = "n/a" if not pw else f"{pw['mean_w']:.0f} W ({100 * pw['frac_of_cap']:.0f} %), {pw['joule_per_pair']:.2f} J"
        out.append(f"| `{n}` ({lab}) | **{l['value']:.0f}** ({l['in_flight']} in flight) | {l['single_stream']['value']:.0f} | {l['sustained']['value']:.0f} | "
                   f"{l['ms_per_step']:.2f} | {r['avg_launch_ms']:.3f} ms, {r['executed_frac']:.3f} | {rl['avg_launch_ms']:.3f} ms, {rl['frac']:.3f} | "
                   f"{rc['avg_launch_ms']:.3f} ms, {rc['frac']:.3f} | {pws} | {cb.get('value')} ({cb.get('kind')}, {cb.get('cores')} thr) | "
                   f"{c.get('max_abs_dT_vs_reference'):.1e} / {c.get('max_abs_dT_vs_oracle'):.1e} ({'ok' if c['ok'] else 'FAIL'}) |")
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


def census(path):
    return subprocess.run([sys.executable, str(ROOT / "tools" / "census_table.py"), str(ROOT / "profiles" / path)], capture_output=True, text=True, check=True).stdout.strip()


BLOCKS = {"bench": bench_table, "shares": share_table, "census": lambda: census("r04_parity_census.txt"),
          "census_fp32": lambda: census("r04_parity_census_exact_fp32.txt")}

if __name__ == "__main__":
    p = ROOT / "DESIGN.md"
    s = p.read_text()
    for name, fn in BLOCKS.items():
        pat = re.compile(rf"(<!-- r04:{name} -->\n).*?(\n<!-- /r04:{name} -->)", re.S)
        assert pat.search(s), name
        s = pat.sub(lambda m: m.group(1) + fn() + m.group(2), s)
    p.write_text(s)
    print("DESIGN.md tables regenerated")
