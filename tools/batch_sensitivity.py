#!/usr/bin/env python3
"""Per pair: final_trans of the same pair computed inside batches of different size (different attention key splits ->
different fp32 summation order) against the CPU oracle.   python tools/batch_sensitivity.py [--n 5000]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from oracle import pointdsc_oracle as O  # noqa: E402  (checker only)
from pointdsc_amd import PointDSC, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    torch.set_num_threads(32)
    dev = "cuda:0"
    kw = dict(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
              sigma_d=0.10, k=40, nms_radius=0.10)
    model = PointDSC(**kw)
    sd = synthetic.make_state_dict(model.state_dict(), seed=6)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    batch = synthetic.make_batch(args.bs, args.n, seed=args.seed)
    data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}

    initial = {}

    def run(sl):
        d = {k: v[sl].contiguous() for k, v in data.items()}
        d["testing"] = True
        with torch.no_grad():
            r = model(d)
        n_pairs = d["corr_pos"].shape[0]
        init = model.workspace_view("initial_trans", n_pairs, args.n)[: n_pairs * 16].reshape(n_pairs, 4, 4).cpu().clone()
        for k, i in enumerate(range(sl.start, sl.stop)):
            initial[(i, n_pairs)] = init[k]          # best seed hypothesis (before refinement) of pair i in this composition
        return r["final_trans"].cpu(), r["final_labels"].cpu()

    whole_T, whole_L = run(slice(0, args.bs))
    okw = {k: kw[k] for k in ("num_layers", "num_channels", "num_iterations", "ratio", "inlier_threshold", "k", "nms_radius")}
    for i in range(args.bs):
        one_T, one_L = run(slice(i, i + 1))
        j = (i // 2) * 2
        two_T, two_L = run(slice(j, j + 2))
        ref = O.forward_testing(sd, batch["corr_pos"][i:i + 1], batch["src_keypts"][i:i + 1], batch["tgt_keypts"][i:i + 1], **okw)
        rT, rL = ref["final_trans"][0], ref["final_labels"][0]
        d = lambda T: float((T - rT).abs().max())  # noqa: E731
        f = lambda L: int((L != rL).sum())  # noqa: E731
        # where does a difference come from?  refine both best-seed hypotheses with the oracle's post_refinement
        a, b = initial[(i, args.bs)], initial[(i, 1)]
        ra, na = O.post_refinement(a, batch["src_keypts"][i], batch["tgt_keypts"][i], kw["inlier_threshold"])
        rb, nb = O.post_refinement(b, batch["src_keypts"][i], batch["tgt_keypts"][i], kw["inlier_threshold"])
        print(f"pair {i}: best-seed hypotheses differ by {float((a - b).abs().max()):.1e}; oracle refinement from each: {na} vs {nb} "
              f"solves, results differ by {float((ra - rb).abs().max()):.1e}")
        print(f"pair {i}: |dT| vs oracle  in batch of {args.bs}: {d(whole_T[i]):.2e} ({f(whole_L[i])} flips)   of 2: {d(two_T[i - j]):.2e} "
              f"({f(two_L[i - j])})   alone: {d(one_T[0]):.2e} ({f(one_L[0])})   inliers {int(rL.sum())}")


if __name__ == "__main__":
    main()
