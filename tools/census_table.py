#!/usr/bin/env python3
"""Markdown table of a tools/parity_census.py record: per family and batch size the pairs outside the fp32 contract and what excuses them.

    python tools/census_table.py profiles/r04_parity_census.txt [more records ...]
"""
import json
import re
import sys

rows = []
for path in sys.argv[1:]:
    cur = None
    for line in open(path):
        m = re.match(r"^(\S+) \(N=(\d+), attention (\S+), compat (\S+), layer_gemm (\S+)\) batches of (\d+): (\d+) pairs.*max\|dT\| median (\S+) max (\S+),", line)
        if m:
            cur = dict(name=m.group(1), arith=f"{m.group(3)} / {m.group(4)} / {m.group(5)}", bs=int(m.group(6)), pairs=int(m.group(7)), median=m.group(8),
                       outside=[], unexcused=None, causes={}, ill=0, gaps=[])
            rows.append(cur)
            continue
        m = re.match(r"^\s+outside the fp32 contract: (\[.*?\]); unexcused by the reference's recorded decisions: (\[.*?\]|None)", line)
        if m and cur is not None:
            cur["outside"] = json.loads(m.group(1))
            cur["unexcused"] = None if m.group(2) == "None" else json.loads(m.group(2))
            continue
        if cur is not None and '"excused"' in line:
            d = json.loads(line.strip())
            why = d["why"]
            cause = ("knn-tie" if "knn-tie" in why else "zero-key tie" if why.startswith("zero-key tie") else "tie" if why.startswith("tie") else
                     "degenerate-solve" if why.startswith("degenerate-solve") else
                     "refinement" if why.startswith("refinement") else "label-edge" if why.startswith("label-edge") else "none")
            if not d["excused"]:
                # r06: no pass for "the reference does not reproduce itself" alone -- an un-named pair is inside only on the fp64 output
                cause = "fp64 reference" if d["pair"] not in (cur["unexcused"] or []) else "UNEXPLAINED"
            cur["causes"][cause] = cur["causes"].get(cause, 0) + 1
            cur["ill"] += bool(d["reference_not_self_consistent"])
            cur["gaps"] += [float(x) for x in re.findall(r"boundary gap is ([0-9.e+-]+)", why)]
print("| family | arithmetic (attention / compat / layer GEMM) | batch | pairs | median dT | outside the fp32 contract | of which the reference does not reproduce itself | recorded cause (count) | largest recorded kNN gap | unexplained |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    causes = ", ".join(f"{k} {v}" for k, v in sorted(r["causes"].items())) or "—"
    gap = f"{max(r['gaps']):.1e}" if r["gaps"] else "—"
    print(f"| `{r['name']}` | {r['arith']} | {r['bs']} | {r['pairs']} | {r['median']} | {len(r['outside'])} {r['outside'] if r['outside'] else ''} | {r['ill']} | {causes} | {gap} | "
          f"**{len(r['unexcused']) if r['unexcused'] is not None else 'n/a'}** |")
