#!/usr/bin/env python3
"""Where the time of the register-resident spectral-matching launch goes: time per call at 1 / 2 / 10 / 20 power iterations for both
forms (slope = one iteration incl. its grid barrier, intercept = matrix generation + fill + finish + pose)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pointdsc_amd import baselines, synthetic  # noqa: E402

dev = "cuda:0"
for n in (1000, 2048, 3000, 5000):
    batch = synthetic.make_batch(1, n, seed=3, inlier_ratio=0.2)
    c, s, t = (batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
    for form in ("resident", "streaming"):
        row = []
        for iters in (1, 2, 10, 20):
            for _ in range(3):
                baselines.SM(c, s, t, 0.10, num_iterations=iters, form=form)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                baselines.SM(c, s, t, 0.10, num_iterations=iters, form=form)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"N={n} {form:9s}: 1 it {row[0]:7.1f} us, 2 it {row[1]:7.1f}, 10 it {row[2]:7.1f}, 20 it {row[3]:7.1f}  -> per iteration {(row[3] - row[2]) / 10:6.1f} us, intercept {row[2] - 10 * (row[3] - row[2]) / 10:7.1f} us")
