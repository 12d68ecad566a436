"""Evaluation / demo harness (pointdsc_amd/harness.py, tools/eval_harness.py): the reference's callers around the path
(evaluation/test_3DMatch.py:20-103, demo_registration.py:37-44,101-117).  CPU part: PLY reading and open3d-style voxel
down-sampling; GPU part: the whole loop on the down-sampled demo cloud."""
import struct
from pathlib import Path

import numpy as np
import pytest
import torch

from pointdsc_amd import harness

ROOT = Path(__file__).resolve().parents[1]
FIXTURE = ROOT / "tests" / "golden" / "demo_clouds_vox005.npz"


def test_read_ply_binary_and_ascii(tmp_path):
    pts = np.random.RandomState(0).standard_normal((37, 3)).astype(np.float32)
    hdr = "ply\nformat binary_little_endian 1.0\ncomment x\nelement vertex 37\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
    (tmp_path / "b.ply").write_bytes(hdr.encode() + pts.tobytes())
    assert np.array_equal(harness.read_ply_xyz(tmp_path / "b.ply"), pts)
    # extra per-vertex properties (colour) interleaved, as CloudCompare / open3d write them
    hdr2 = ("ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
    body = b"".join(struct.pack("<fffBBB", *p, 1, 2, 3) for p in pts.tolist())
    (tmp_path / "c.ply").write_bytes(hdr2.encode() + body)
    assert np.array_equal(harness.read_ply_xyz(tmp_path / "c.ply"), pts)
    asc = "ply\nformat ascii 1.0\nelement vertex 37\nproperty float x\nproperty float y\nproperty float z\nend_header\n" + \
          "\n".join(" ".join(repr(float(v)) for v in p) for p in pts) + "\n"
    (tmp_path / "a.ply").write_text(asc)
    assert np.allclose(harness.read_ply_xyz(tmp_path / "a.ply"), pts, atol=0, rtol=1e-7)


def test_voxel_down_sample_is_the_mean_per_occupied_voxel():
    rs = np.random.RandomState(1)
    pts = (rs.random_sample((5000, 3)) * 2.0).astype(np.float32)
    vox = harness.voxel_down_sample(pts, 0.25)
    origin = pts.astype(np.float64).min(0) - 0.125
    idx = np.floor((pts.astype(np.float64) - origin) / 0.25).astype(np.int64)
    keys = {tuple(k) for k in idx.tolist()}
    assert len(vox) == len(keys)
    k0 = tuple(idx[0].tolist())
    members = pts[(idx == np.array(k0)).all(1)].astype(np.float64).mean(0)
    assert np.abs(vox.astype(np.float64) - members).sum(1).min() < 1e-6
    assert np.allclose(vox.astype(np.float64).mean(0), pts.mean(0), atol=0.05)


def test_demo_fixture_matches_the_survey_counts():
    fx = np.load(FIXTURE)
    assert int(fx["cloud_bin_0_raw_vertices"]) == 258342 and int(fx["cloud_bin_1_raw_vertices"]) == 268977
    assert 5000 < len(fx["cloud_bin_0"]) < 5600 and 5000 < len(fx["cloud_bin_1"]) < 5600                        # N ~ 5.3 k (section 0 item 4)


@pytest.mark.gpu
def test_eval_loop_on_the_demo_cloud():
    """eval_3DMatch_scene's loop end to end: down-sampled demo cloud -> second view -> stand-in descriptors -> GPU
    correspondence construction -> PointDSC.forward -> device-side stats row; every pair registers."""
    from pointdsc_amd import PointDSC, workloads
    cloud = np.load(FIXTURE)["cloud_bin_0"]
    model = PointDSC(**workloads.BASE_MODEL)
    model.load_state_dict(workloads.state_dict("n5000_b32", model.state_dict()))
    model = model.eval().cuda()
    stats = harness.eval_scene(model, harness.demo_pairs(cloud, 4), scene_ind=3)
    assert stats.shape == (4, 12) and (stats[:, 11] == 3).all()
    assert (stats[:, 0] == 1).all(), stats[:, :3]                       # success: RE < 15 deg and TE < 30 cm
    assert stats[:, 1].max() < 1.0 and stats[:, 2].max() < 3.0          # far inside: RE < 1 deg, TE < 3 cm
    assert (stats[:, 4] > 0.1).all() and (stats[:, 4] < 0.6).all()      # the stand-in descriptors leave a realistic outlier share
    assert (stats[:, 6] > 0.95).all() and (stats[:, 7] > 0.95).all()    # precision / recall of the inlier mask
    assert (stats[:, 9] > 0).all() and (stats[:, 10] > 0).all()
    summ = harness.summarize(stats)
    assert summ["registration_recall_pct"] == 100.0 and summ["num_pairs"] == 4
    # the demo's own direction (demo_registration.py:101-108): no mutual check, N = all source points
    one = next(iter(harness.demo_pairs(cloud, 1)))
    assert one["src_desc"].shape == (len(cloud), 33) and abs(float(np.linalg.norm(one["src_desc"][0])) - 1.0) < 1e-5
    assert torch.cuda.is_available()


@pytest.mark.gpu
def test_eval_loop_batches_pairs_of_different_size():
    """batch_size > 1: with the mutual check every pair keeps its own number of correspondences (as in the reference's evaluation,
    num_node='all'); four of them go through one ragged model call.  Same rows as the one-pair-per-call loop."""
    from pointdsc_amd import PointDSC, workloads
    cloud = np.load(FIXTURE)["cloud_bin_0"]
    model = PointDSC(**workloads.BASE_MODEL)
    model.load_state_dict(workloads.state_dict("n5000_b32", model.state_dict()))
    model = model.eval().cuda()
    one = harness.eval_scene(model, harness.demo_pairs(cloud, 5, corrupt=0.3), use_mutual=True, batch_size=1)
    many = harness.eval_scene(model, harness.demo_pairs(cloud, 5, corrupt=0.3), use_mutual=True, batch_size=4)
    assert one.shape == many.shape == (5, 12)
    assert len(set(one[:, 3].astype(int).tolist())) > 1, "the pairs were meant to differ"                # column 3: input inliers of the pair
    assert (one[:, 0] == many[:, 0]).all() and (one[:, 0] == 1).all()
    assert np.abs(one[:, 1] - many[:, 1]).max() < 0.05 and np.abs(one[:, 2] - many[:, 2]).max() < 0.1  # RE [deg], TE [cm]
    assert np.abs(one[:, [3, 5]] - many[:, [3, 5]]).max() <= 2                                           # inlier counts (in / true positives)
    assert np.abs(one[:, [4, 6, 7, 8]] - many[:, [4, 6, 7, 8]]).max() < 0.01                             # ratios / P / R / F1
