#!/usr/bin/env python3
"""Census fixtures, third part: how well-posed every Procrustes solve of the reference's post-refinement was.

Run in the BUILD container only (imports the unmodified reference from /root/reference):

    python oracle/make_census_refine_sv.py [family ...]        # -> tests/golden/census_refine_<family>.npz

For every pair of tests/golden/census_<name>.npz the reference's OWN ``PointDSC.post_refinement`` (models/PointDSC.py:403-438) is
called on the pose its forward handed to it -- ``initial_trans32`` / ``initial_trans64`` as recorded by
oracle/make_census_internals.py from ``cal_seed_trans``'s return value -- and on the pair's key points, in the run's dtype.  The
call must return the stored census pose bit for bit (asserted): the recorded input is the real input and the tap changes nothing.
``torch.svd`` is wrapped for the duration of the call (models/common.py:36 is its only call site on this path) and the
singular values IT RETURNED for every iteration's 3x3 covariance are stored, with the number of correspondences of that solve:

    refine_sv{32,64}      [pairs, 20, 3]   singular values of H per refinement solve, descending (NaN where no solve ran)
    refine_solves{32,64}  [pairs]          number of solves that ran
    refine_inliers{32,64} [pairs, 20]      correspondences entering the solve (-1 padded)

A weighted Procrustes problem determines the rotation only when H has rank >= 2 (models/common.py:36-41: with rank 1 -- two
correspondences, or collinear ones -- every rotation about the common line fits equally and U, V's null-space columns are whatever
LAPACK's gesdd leaves there; the reference's fp32 and fp64 runs then return poses O(1) apart).  tools/parity_census.py's
``degenerate-solve`` rule reads these records; it does not excuse the pair, it replaces the pose comparison by the comparison the
data still determines (where both poses send the correspondences of that solve).
"""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from pointdsc_amd import workloads  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"
MAX_SOLVES = 20


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    sys.path.insert(0, str(REF))
    import models.PointDSC as RP                      # the unmodified reference
    names = sys.argv[1:] or [n for n in workloads.WORKLOADS if (GOLDEN / f"census_internals_{n}.npz").exists()]
    real_svd = torch.svd
    for name in names:
        w = workloads.WORKLOADS[name]
        kw = dict(w["model"])
        census = np.load(GOLDEN / f"census_{name}.npz", allow_pickle=False)
        ix = np.load(GOLDEN / f"census_internals_{name}.npz", allow_pickle=False)
        total = census["ref32_final_trans"].shape[0]
        out = {}
        for tag, dt in (("32", torch.float32), ("64", torch.float64)):
            torch.set_default_dtype(dt)
            try:
                ref = RP.PointDSC(**kw).eval()        # post_refinement reads inlier_threshold only (models/PointDSC.py:415)
                sv = np.full((total, MAX_SOLVES, 3), np.nan, np.float64)
                cnt = np.full((total, MAX_SOLVES), -1, np.int32)
                solves = np.zeros(total, np.int32)
                for i in range(total):
                    one = workloads.batch(name, i, 1)
                    rec = []

                    def svd_tap(H, *a, **k):
                        r = real_svd(H, *a, **k)
                        rec.append(r[1][0].double().numpy().copy())
                        return r

                    counts = []
                    orig_rt = RP.rigid_transform_3d

                    def rt_tap(A, B, weights=None, weight_threshold=0):
                        counts.append(int(A.shape[1]))
                        return orig_rt(A, B, weights, weight_threshold)

                    torch.svd, RP.rigid_transform_3d = svd_tap, rt_tap
                    try:
                        with torch.no_grad():
                            T = ref.post_refinement(torch.from_numpy(ix[f"initial_trans{tag}"][i]).to(dt)[None],
                                                    one["src_keypts"].to(dt), one["tgt_keypts"].to(dt))
                    finally:
                        torch.svd, RP.rigid_transform_3d = real_svd, orig_rt
                    want = census[f"ref{tag}_final_trans"][i]
                    assert np.array_equal(T[0].numpy().astype(want.dtype), want), f"{name} pair {i} fp{tag}: post_refinement on the recorded pose differs from the census pose"
                    assert len(rec) == len(counts) <= MAX_SOLVES
                    solves[i] = len(rec)
                    if rec:
                        sv[i, :len(rec)] = np.stack(rec)
                        cnt[i, :len(rec)] = counts
                out.update({f"refine_sv{tag}": sv, f"refine_solves{tag}": solves, f"refine_inliers{tag}": cnt})
            finally:
                torch.set_default_dtype(torch.float32)
        np.savez_compressed(GOLDEN / f"census_refine_{name}.npz", input_checksum=census["input_checksum"], **out)
        s = out["refine_sv32"]
        last = np.array([s[i, max(out["refine_solves32"][i] - 1, 0)] for i in range(total)])
        ratio = last[:, 1] / last[:, 0]
        print(f"{name}: {total} pairs; last solve's s2/s1 below 1e-4 on pairs {np.flatnonzero(ratio < 1e-4).tolist()} "
              f"(fewer than 3 correspondences: {np.flatnonzero(np.array([out['refine_inliers32'][i, max(out['refine_solves32'][i] - 1, 0)] for i in range(total)]) < 3).tolist()}); "
              f"no solve ran on {np.flatnonzero(out['refine_solves32'] == 0).tolist()}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
