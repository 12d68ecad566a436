#!/usr/bin/env python3
"""Experiments with two HIP streams (two module instances = two workspaces):
  * does splitting a small batch over two streams hide the latency-bound small kernels of one half under the attention of
    the other?  (r01/r02: no)
  * r03: two WHOLE steps in flight -- consecutive forwards of full batches alternate between two streams, so the
    latency-bound tail of step i (NMS, seed ranking, kNN, solver, scoring, refinement: ~1 ms at 32 pairs on a few CUs) can
    overlap the compat build and the first layers of step i+1.
python tools/overlap_probe.py [--n 5000] [--bs 4]"""
import argparse
import copy
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pointdsc_amd import PointDSC, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    dev = "cuda:0"
    kw = dict(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
              sigma_d=0.10, k=40, nms_radius=0.10)
    model = PointDSC(**kw)
    model.load_state_dict(synthetic.make_state_dict(model.state_dict(), seed=6))
    model = model.eval().to(dev)
    twin = copy.deepcopy(model)
    batch = synthetic.make_batch(args.bs, args.n, seed=1)
    data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    h = args.bs // 2
    lo = {k: (v[:h].contiguous() if torch.is_tensor(v) else v) for k, v in data.items()}
    hi = {k: (v[h:].contiguous() if torch.is_tensor(v) else v) for k, v in data.items()}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def whole():
        return model(data)["final_trans"]

    def halves_serial():
        return torch.cat([model(lo)["final_trans"], model(hi)["final_trans"]])

    def halves_two_streams():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = model(lo)["final_trans"]
        with torch.cuda.stream(s2):
            b = twin(hi)["final_trans"]
        cur.wait_stream(s1); cur.wait_stream(s2)
        return torch.cat([a, b])

    def two_steps_in_flight():
        # called once per step: even steps on s1 / model, odd steps on s2 / twin; nothing waits until the caller synchronises
        two_steps_in_flight.i ^= 1
        with torch.cuda.stream(s1 if two_steps_in_flight.i else s2):
            return (model if two_steps_in_flight.i else twin)(data)["final_trans"]
    two_steps_in_flight.i = 0

    with torch.no_grad():
        ref = whole()
        modes = (("one call", whole),) + ((("two halves, one stream", halves_serial), ("two halves, two streams", halves_two_streams)) if args.bs >= 2 else ()) \
            + (("whole steps alternating on two streams", two_steps_in_flight),)
        for name, fn in modes:
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            print(f"N={args.n} bs={args.bs}  {name:26s} {dt * 1e3:7.3f} ms per step   max|dT| vs one call {float((out - ref).abs().max()):.2e}")


if __name__ == "__main__":
    main()
