#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / avg / min / max.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py ...
    python tools/rocpd_kernel_stats.py gpurun_out/prof/r1_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}   total kernel time {tot / 1e3:.3f} ms")
    print(f"{'kernel':64s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scr':>4s}")
    for r in rows:
        print(f"{r[0][:64]:64s} {r[1]:6d} {r[2]:11.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {100 * r[2] / tot:6.2f} "
              f"{r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
