#!/usr/bin/env python3
"""Markdown table from tools/parity_census.py output.   python tools/summarize_census.py profiles/r03_parity_census.txt"""
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"^(\S+) \(N=(\d+), attention (\S+), compat (\S+), layer_gemm (\S+)\) batches of (\d+): (\d+) pairs, FAIL (\[.*?\]), "
                 r"label flips vs fp32 ref (\d+), matched on the fp64 ref (\[.*?\]), max\|dT\| median (\S+) max (\S+), histogram (\{.*?\}); "
                 r"the reference's own fp32 and fp64 runs differ by >= 1e-4 on (\[.*?\])", line)
    if m:
        rows.append(m.groups())
print("| family (arithmetic) | batches of | pairs | outside the contract | of which the reference itself differs ≥ 1e-4 between fp32 and fp64 | "
      "matched on the fp64 reference only | label flips vs fp32 ref | median / max dT (closer reference) | dT histogram <1e-6 / <1e-5 / <2e-5 / <5e-5 / <1e-4 / ≥1e-4 |")
print("|---|---|---|---|---|---|---|---|---|")
for name, n, att, compat, gemm, step, pairs, fail, flips, m64, med, mx, hist, refself in rows:
    f, rs = eval(fail), set(eval(refself))
    h = eval(hist)
    print(f"| `{name}` ({att}, {compat}, {gemm}) | {step} | {pairs} | {f if f else '—'} | {[i for i in f if i in rs] if f else '—'} | {m64} | {flips} | "
          f"{med} / {mx} | {' / '.join(str(v) for v in h.values())} |")
