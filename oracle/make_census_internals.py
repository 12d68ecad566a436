#!/usr/bin/env python3
"""Census fixtures, second part: what the unmodified reference DECIDED on every census pair -- the discrete steps behind its pose.

Run in the BUILD container only (imports the unmodified reference from /root/reference):

    python oracle/make_census_internals.py n5000_b32 256 [--threads 4]    # -> tests/golden/census_internals_n5000_b32.npz

For every pair of tests/golden/census_<name>.npz (oracle/make_census_goldens.py) the reference forward runs again, fp32 and
fp64, with taps around four of its callables (wrappers installed at run time; no reference line is copied or edited):

  * ``PointDSC.cal_seed_trans`` (models/PointDSC.py:234-336): its argument ``seeds`` and its return values
    ``seedwise_fitness`` (-> integer inlier counts ``round(fitness * N)``, :328) and the chosen hypothesis (:329-332);
  * ``models.PointDSC.knn`` (the module-global bound at :5, called at :251): for every seed a 64-bit hash of its sorted
    neighbour set, and -- recomputed by the tap with knn's own expression ``2 - 2 x x^T`` (models/common.py:60-61) on the seed rows,
    in the run's dtype -- the gap between the last neighbour kept and the first one left out (``topk`` boundary, :68): a seed
    whose gap is at round-off level has a neighbour set, hence a hypothesis, that round-off decides;
  * ``PointDSC.pick_seeds`` (:199-217, r05): its argument ``scores`` -- the confidence logits.  When fewer than ``max_num`` keys
    (logit x is_local_max) are positive the rest of the seed list comes from the keys tied at 0 in the order ``torch.argsort``
    happens to leave them; the census recognises that regime from the recorded logits;
  * ``models.PointDSC.transform`` (the module-global bound at :6, called once per post-refinement iteration, :422): the warped
    source points, from which the tap recomputes ``L2_dis`` with the reference's own expression (:423) in the run's dtype and
    stores per iteration the inlier count (:424-425) and how close the nearest correspondence sits to the threshold
    (min |L2 - thr| / thr) -- i.e. whether one vote of that iteration hangs on round-off.

The poses returned by the tapped runs must equal the stored census outputs bit for bit (asserted), so the taps change nothing.

tests/test_gpu_parity.py::test_parity_census uses these records to turn "this pair is a near-tie" from an argument into a
checked property: a GPU result outside BASELINE.json's contract is excused ONLY when the hypothesis it chose is recorded
here as tied (within one vote) with the reference's winner, or when its refinement trace leaves the reference's at an
iteration recorded here as having a correspondence on the threshold.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from pointdsc_amd import workloads  # noqa: E402
from pointdsc_amd.model import PointDSC as AmdPointDSC  # noqa: E402  (state_dict template only)

GOLDEN = ROOT / "tests" / "golden"
MAX_IT = 21


def set_hash(idx) -> int:
    """64-bit FNV-1a over the ascending neighbour indices (as little-endian int32): equal sets <=> equal hashes, for all practical
    purposes.  tools/parity_census.py computes the same hash of the GPU's neighbour sets."""
    h = 0xcbf29ce484222325
    for b in np.sort(np.asarray(idx, dtype=np.int64)).astype("<i4").tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


class Taps:
    """Run-time wrappers around the reference's callables; `rec` holds what the current forward decided."""

    def __init__(self, ref_module, model, refine_thr: float):
        self.rp, self.model, self.thr = ref_module, model, refine_thr
        self.rec = None
        self.tgt = None
        self._orig_transform = ref_module.transform
        self._orig_cst = model.cal_seed_trans
        self._orig_knn = ref_module.knn
        self.knn_out = None

        def knn_tap(x, k, ignore_self=False, normalized=True):
            out = self._orig_knn(x, k, ignore_self=ignore_self, normalized=normalized)
            self.knn_out = (out, k, ignore_self, normalized)
            return out

        def transform_tap(pts, trans):
            out = self._orig_transform(pts, trans)
            if self.rec is not None and self.tgt is not None:
                l2 = torch.norm(out - self.tgt, dim=-1)[0]                     # models/PointDSC.py:423, in the run's dtype
                self.rec["refine_counts"].append(int((l2 < self.thr).sum()))    # :424-425
                self.rec["refine_margin"].append(float(((l2 - self.thr).abs() / self.thr).min()))
            return out

        def cst_tap(seeds, feats, src, tgt):
            out = self._orig_cst(seeds, feats, src, tgt)
            n = src.shape[1]
            self.rec["seeds"] = seeds[0].numpy().astype(np.int32)
            self.rec["counts"] = torch.round(out[1][0].double() * n).numpy().astype(np.int32)       # fitness = count / N (:328)
            self.rec["best"] = int(out[1][0].argmax())                                                # :329
            self.rec["initial_trans"] = out[2][0].double().numpy()
            self.tgt = tgt
            # neighbour sets of the seeds (what :252 gathers) and their topk boundary gaps
            idx, k, ignore_self, normalized = self.knn_out
            assert ignore_self and normalized
            rows = idx[0][seeds[0]]                                            # [S, k]
            self.rec["knn_hash"] = np.array([set_hash(r) for r in rows.numpy()], dtype=np.uint64)
            f = feats[0]
            dist = 2 - 2 * (f[seeds[0]] @ f.T)                                 # models/common.py:60-61 on the seed rows, run's dtype
            low = torch.topk(dist, k + 2, dim=-1, largest=False).values        # rank 0 = the seed itself (:68-69)
            self.rec["knn_gap"] = (low[:, k + 1] - low[:, k]).double().numpy().astype(np.float32)
            self.knn_out = None
            return out

        def pick_tap(dists, scores, R, max_num):
            # models/PointDSC.py:174,199-217: the confidence logits the seeds are picked from (r05).  With fewer than max_num POSITIVE
            # keys (key = logit x is_local_max, :211-216) the rest of the seed list is drawn from the keys tied at 0 in whatever order
            # torch.argsort leaves them (probe 2): the census needs the logits to recognise that regime.
            self.rec["conf"] = scores[0].double().numpy().astype(np.float32 if scores.dtype == torch.float32 else np.float64)
            return self._orig_pick(dists, scores, R, max_num)

        self._orig_pick = model.pick_seeds
        ref_module.transform = transform_tap
        ref_module.knn = knn_tap
        model.cal_seed_trans = cst_tap
        model.pick_seeds = pick_tap

    def run(self, data):
        self.rec = {"refine_counts": [], "refine_margin": []}
        self.tgt = None
        res = self.model(data)
        rec, self.rec, self.tgt = self.rec, None, None
        return res, rec

    def remove(self):
        self.rp.transform = self._orig_transform
        self.rp.knn = self._orig_knn
        del self.model.cal_seed_trans
        del self.model.pick_seeds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("pairs", type=int)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--no-write", action="store_true", help="fill the /tmp cache only (several processes can share a family)")
    a = ap.parse_args()
    warnings.filterwarnings("ignore")
    torch.set_num_threads(a.threads)
    sys.path.insert(0, str(REF))
    import models.PointDSC as RP                      # the unmodified reference
    RefPointDSC = RP.PointDSC

    name, w = a.name, workloads.WORKLOADS[a.name]
    kw = dict(w["model"])
    tmpl = AmdPointDSC(**kw).state_dict()
    sd = workloads.state_dict(name, tmpl)
    ref = RefPointDSC(**kw).eval()
    ref.load_state_dict(sd, strict=True)
    torch.set_default_dtype(torch.float64)
    ref64 = RefPointDSC(**kw).eval()
    ref64.load_state_dict(sd, strict=True)
    ref64 = ref64.double()
    torch.set_default_dtype(torch.float32)
    refine_thr = 0.10 if kw["inlier_threshold"] == 0.10 else 1.2           # models/PointDSC.py:415-418

    census = np.load(GOLDEN / f"census_{name}.npz", allow_pickle=False)
    cache = Path("/tmp") / f"census_internals_cache_v3_{name}"
    cache.mkdir(exist_ok=True)
    n = w["num_corr"]
    S = int(n * kw["ratio"])
    t_start = time.perf_counter()
    for i in range(a.first, a.first + a.pairs):
        f = cache / f"{i}.npz"
        if f.exists():
            continue
        one = workloads.batch(name, i, 1)
        data = {k: one[k] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        out = {}
        for tag, model, dt in (("32", ref, torch.float32), ("64", ref64, torch.float64)):
            torch.set_default_dtype(dt)
            taps = Taps(RP, model, refine_thr)
            try:
                with torch.no_grad():
                    res, rec = taps.run(dict({k: v.to(dt) for k, v in data.items()}, testing=True))
            finally:
                taps.remove()
                torch.set_default_dtype(torch.float32)
            want = census[f"ref{tag}_final_trans"][i]
            assert np.array_equal(res["final_trans"][0].numpy().astype(want.dtype), want), f"pair {i} fp{tag}: tapped run differs from the census fixture"
            rc = np.full(MAX_IT, -1, np.int32)
            rm = np.full(MAX_IT, np.nan, np.float32)
            k = len(rec["refine_counts"])
            rc[:k], rm[:k] = rec["refine_counts"], rec["refine_margin"]
            assert rec["seeds"].shape == (S,) and rec["counts"].shape == (S,)
            out.update({f"knn_hash{tag}": rec["knn_hash"], f"knn_gap{tag}": rec["knn_gap"], f"conf{tag}": rec["conf"].astype(np.float32)})
            out.update({f"seeds{tag}": rec["seeds"], f"counts{tag}": rec["counts"], f"best{tag}": np.int32(rec["best"]),
                        f"initial_trans{tag}": rec["initial_trans"], f"refine_counts{tag}": rc, f"refine_margin{tag}": rm})
        np.savez(f, **out)
        c = np.sort(out["counts32"])[::-1]
        print(f"{name} pair {i}: best seed {int(out['best32'])} (corr {int(out['seeds32'][out['best32']])}) votes {c[:4].tolist()}, "
              f"refinement {out['refine_counts32'][out['refine_counts32'] >= 0].tolist()} margin min {np.nanmin(out['refine_margin32']):.1e}, "
              f"elapsed {time.perf_counter() - t_start:.0f}s", flush=True)
    if a.no_write:
        return 0
    total = census["ref32_final_trans"].shape[0]
    rows = [np.load(cache / f"{i}.npz") for i in range(total)]
    keys = [f"{k}{t}" for t in ("32", "64") for k in ("seeds", "counts", "best", "initial_trans", "refine_counts", "refine_margin", "knn_hash", "knn_gap")]
    keys.append("conf32")                  # (fp32 logits only: 4 N bytes per pair)
    np.savez_compressed(GOLDEN / f"census_internals_{name}.npz", num_corr=np.int64(n), refine_threshold=np.float64(refine_thr),
                        input_checksum=census["input_checksum"], **{k: np.stack([r[k] for r in rows]) for k in keys})
    # summary for CENSUS_PINNING.json: how many pairs the reference itself decides by one vote
    c32 = np.stack([r["counts32"] for r in rows])
    top = np.sort(c32, axis=1)[:, ::-1]
    tie = np.flatnonzero(top[:, 0] - top[:, 1] <= 1)
    rm = np.stack([r["refine_margin32"] for r in rows])
    gaps = np.stack([r["knn_gap32"] for r in rows])
    best_gap = np.array([float(r["knn_gap32"][int(r["best32"])]) for r in rows])
    rep = {"pairs": total, "pairs_with_a_second_hypothesis_within_one_vote_of_the_winner_fp32": int(len(tie)),
           "median_number_of_hypotheses_within_one_vote_of_the_winner_fp32": float(np.median((top >= top[:, :1] - 1).sum(axis=1))),
           "knn_boundary_gap_fp32": {"median_over_all_seeds": float(np.median(gaps)), "share_of_seeds_below_2e-5": float((gaps < 2e-5).mean()),
                                     "share_of_seeds_below_2e-6": float((gaps < 2e-6).mean()),
                                     "pairs_whose_chosen_seed_gap_is_below_2e-5": int((best_gap < 2e-5).sum()),
                                     "pairs_whose_chosen_seed_gap_is_below_2e-6": int((best_gap < 2e-6).sum())},
           "refinement_margin_below_1e-4_fp32": [int(i) for i in np.flatnonzero(np.nanmin(rm, axis=1) < 1e-4)],
           "refinement_iterations_fp32": {str(k): int(v) for k, v in zip(*np.unique([(r["refine_counts32"] >= 0).sum() for r in rows], return_counts=True))}}
    p = GOLDEN / "CENSUS_PINNING.json"
    report = json.loads(p.read_text()) if p.exists() else {}
    report.setdefault(name, {})["internals"] = rep
    p.write_text(json.dumps(report, indent=1))
    print(json.dumps(rep))
    return 0


if __name__ == "__main__":
    sys.exit(main())
