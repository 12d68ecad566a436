#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite file (ROCm 7.2).

    rocprofv3 --pmc SQ_WAVE_CYCLES ... --kernel-trace -d gpurun_out/pmc1 -o p -- python bench.py ...
    python tools/rocpd_pmc_stats.py gpurun_out/pmc1/p_results.db [kernel-substring ...]

Counters are summed over the dimensions rocprofv3 reports (XCC/SE/instances) per dispatch, then averaged
over the dispatches of each kernel.
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filters):
    cur = sqlite3.connect(path).cursor()
    per_dispatch = defaultdict(float)
    meta = {}
    for disp, kname, cname, value, dur in cur.execute(
            "select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
        if filters and not any(f in kname for f in filters):
            continue
        per_dispatch[(disp, cname)] += value
        meta[disp] = (kname, dur)
    agg = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for (disp, cname), v in per_dispatch.items():
        agg[meta[disp][0]][cname].append(v)
    for disp, (kname, dur) in meta.items():
        durs[kname].append(dur)
    print(f"# source: {path}")
    for kname in sorted(agg, key=lambda k: -sum(durs[k])):
        d = durs[kname]
        print(f"{kname[:90]}  dispatches={len(d)} avg_duration_us={sum(d) / len(d) / 1e3:.1f}")
        for cname in sorted(agg[kname]):
            vals = agg[kname][cname]
            print(f"    {cname:36s} avg={sum(vals) / len(vals):.6g}  min={min(vals):.6g}  max={max(vals):.6g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
