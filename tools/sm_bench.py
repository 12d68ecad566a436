#!/usr/bin/env python3
"""Timing + HBM roofline of the spectral-matching baseline (csrc/spectral.hip): 4 N^2 bytes written once, 4 N^2 bytes read
per power iteration."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pointdsc_amd import baselines, synthetic  # noqa: E402


def main():
    dev = "cuda:0"
    for n, bs in ((1000, 1), (5000, 1), (5000, 8), (10000, 1), (20000, 1)):
        batch = synthetic.make_batch(bs, n, seed=3, inlier_ratio=0.2)
        c, s, t = (batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
        for form in ("resident", "streaming"):
            if form == "resident" and n > 5120:
                continue
            for _ in range(3):
                baselines.SM(c, s, t, 0.10, form=form)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                baselines.SM(c, s, t, 0.10, form=form)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            ld = (n + 63) // 64 * 64
            bytes_ = bs * 4.0 * n * ld * 11            # one write + ten reads of the matrix (streaming form)
            what = "matrix in registers" if form == "resident" else "matrix in HBM     "
            extra = "" if form == "resident" else f"  ({bytes_ / us / 1e6:5.2f} TB/s of matrix traffic)"
            print(f"N={n} bs={bs} {what}: {us:9.1f} us per call, {bs / us * 1e6:8.1f} pairs/s{extra}")


if __name__ == "__main__":
    main()
