#!/usr/bin/env python3
"""The reference's evaluation loop / demo around pointdsc_amd.PointDSC (pointdsc_amd/harness.py).

    python tools/eval_harness.py [--pcd1 a.ply --pcd2 b.ply] [--num-pairs 8] [--snapshot model_best.pkl] [--kitti]

Without --pcd1 the down-sampled demo cloud of tests/golden/demo_clouds_vox005.npz (reference demo_data/cloud_bin_0.ply at
0.05 m) is used.  Every pair = the cloud against a seeded second view of it (partial overlap, noise, random rigid motion),
stand-in descriptors with a known outlier share, GPU correspondence construction, forward, device-side stats row.
Registration Recall on 3DMatch-FCGF itself needs the released weights and the dataset (both absent here): pass
--snapshot / real descriptors when they exist; the loop is the same.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, harness, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pcd1", default=None, help="PLY file (binary LE / ascii, float xyz); default: the demo fixture")
    ap.add_argument("--voxel", type=float, default=0.05, help="config.downsample of the 3DMatch snapshot")
    ap.add_argument("--num-pairs", type=int, default=8)
    ap.add_argument("--outlier-share", type=float, default=0.6, help="share of stand-in descriptors replaced by noise")
    ap.add_argument("--snapshot", default=None, help="released model_best.pkl (load_state_dict(strict=False)); default: seeded weights")
    ap.add_argument("--mutual", action="store_true", help="mutual nearest neighbours only (datasets/ThreeDMatch.py:286-288)")
    ap.add_argument("--batch-size", type=int, default=1, help="pairs per (ragged) model call; the reference evaluates one pair per call")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    if a.pcd1:
        cloud = harness.voxel_down_sample(harness.read_ply_xyz(a.pcd1), a.voxel)
    else:
        cloud = np.load(ROOT / "tests" / "golden" / "demo_clouds_vox005.npz")["cloud_bin_0"]
    kw = dict(workloads.BASE_MODEL)                       # evaluation/test_3DMatch.py:215-224 with the snapshot's config.json
    model = PointDSC(**kw)
    if a.snapshot:
        print(model.load_state_dict(torch.load(a.snapshot, map_location="cpu"), strict=False))      # test_3DMatch.py:225-226
    else:
        model.load_state_dict(workloads.state_dict("n5000_b32", model.state_dict()))
    model = model.eval().cuda()
    stats = harness.eval_scene(model, harness.demo_pairs(cloud, a.num_pairs, corrupt=a.outlier_share), scene_ind=0,
                               inlier_threshold=kw["inlier_threshold"], use_mutual=a.mutual, batch_size=a.batch_size)
    summ = harness.summarize(stats)
    if a.json:
        print(json.dumps({"stats_columns": harness.STATS_NAMES, "stats": stats.tolist(), "summary": summ}))
        return
    print(f"{len(cloud)} points after {a.voxel} m voxel down-sampling; {a.num_pairs} pairs")
    print(" ".join(f"{n[:10]:>10s}" for n in harness.STATS_NAMES))
    for row in stats:
        print(" ".join(f"{v:10.4f}" for v in row))
    for k, v in summ.items():
        print(f"{k}: {v:.4f}" if isinstance(v, float) else f"{k}: {v}")


if __name__ == "__main__":
    main()
