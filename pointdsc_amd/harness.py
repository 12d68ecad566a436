"""Evaluation / demo harness around ``PointDSC.forward`` -- what the reference's callers do on either side of the path.

Replicates, without open3d / easydict (absent here):
  * ``evaluation/test_3DMatch.py:20-103``  eval_3DMatch_scene: per-pair loop, ``model(data)`` under no_grad, the
    12-column stats row (0 success, 1 RE deg, 2 TE cm, 3 input inliers, 4 input inlier ratio, 5 output true positives,
    6 precision, 7 recall, 8 F1, 9 model time, 10 data time, 11 scene index) and the summary lines of
    ``eval_3DMatch`` (:141-176: recall, mean RE / TE over the successful pairs, mean P / R / F1, mean times);
  * ``demo_registration.py:37-44,101-117``  cloud -> voxel down-sampling -> descriptors -> nearest-neighbour matching
    -> ``corr_pos`` -> ``model(data)``.
The arithmetic on the path runs in libpointdsc_hip.so (correspondence construction f-2, forward a-*, stats row f-4);
this module is host plumbing: PLY reading (binary little-endian float xyz), open3d-style voxel
down-sampling in numpy, the pair loop, timers (``utils/timer.py`` semantics: wall clock, here with a device
synchronisation so that model time is the GPU's).

Descriptors: FPFH / FCGF extraction is upstream of the path and needs open3d / MinkowskiEngine (absent).  The
harness takes descriptors as input; ``standin_descriptors`` provides seeded unit vectors that agree for points that
coincide under the ground-truth motion (a stand-in with a controllable inlier ratio, NOT a feature extractor), so
that the whole loop runs and produces meaningful rows on the demo clouds.  With real descriptors + the released
weights the same loop is the Registration-Recall driver.
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .correspondences import build_correspondences

STATS_NAMES = ("success", "RE_deg", "TE_cm", "input_inliers", "input_inlier_ratio", "output_true_positives",
               "precision", "recall", "f1", "model_time_s", "data_time_s", "scene_ind")


# ---------------------------------------------------------------------------------------------------------------
# clouds
# ---------------------------------------------------------------------------------------------------------------
def read_ply_xyz(path) -> np.ndarray:
    """Vertices of a PLY file as float32 [n,3] (ascii or binary_little_endian; x, y, z may sit among other properties)."""
    raw = Path(path).read_bytes()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    fmt, n, props, in_vertex = None, 0, [], False
    np_types = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
                "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
                "uint": "u4", "uint32": "u4"}
    for line in raw[:end].decode("ascii", "replace").splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                n = int(tok[2])
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError("list property inside the vertex element")
            props.append((tok[2], np_types[tok[1]]))
    names = [p[0] for p in props]
    if not all(a in names for a in "xyz"):
        raise ValueError(f"{path}: no x/y/z vertex properties")
    if fmt == "binary_little_endian":
        rec = np.frombuffer(raw, dtype=np.dtype([(nm, "<" + ty) for nm, ty in props]), count=n, offset=end)
        return np.stack([rec["x"], rec["y"], rec["z"]], axis=1).astype(np.float32)
    if fmt == "ascii":
        rows = np.loadtxt(raw[end:].decode().splitlines()[:n], dtype=np.float64, ndmin=2)
        return rows[:, [names.index(a) for a in "xyz"]].astype(np.float32)
    raise ValueError(f"{path}: unsupported PLY format {fmt}")


def voxel_down_sample(points: np.ndarray, voxel: float) -> np.ndarray:
    """open3d ``PointCloud.voxel_down_sample``: voxel grid anchored at min_bound - voxel/2, one output point per occupied
    voxel = the mean of its points (output ordered by voxel index; open3d's order is a hash-map artefact)."""
    pts = np.asarray(points, dtype=np.float64)
    origin = pts.min(axis=0) - 0.5 * voxel
    idx = np.floor((pts - origin) / voxel).astype(np.int64)
    dims = idx.max(axis=0) + 1
    key = (idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2]
    order = np.argsort(key, kind="stable")
    key, pts = key[order], pts[order]
    first = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    sums = np.add.reduceat(pts, first, axis=0)
    counts = np.diff(np.r_[first, len(key)])
    return (sums / counts[:, None]).astype(np.float32)


def random_rigid(rs: np.random.RandomState, max_trans: float = 1.0) -> np.ndarray:
    q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = q
    T[:3, 3] = rs.uniform(-max_trans, max_trans, 3)
    return T


def standin_descriptors(points_in_common_frame: np.ndarray, dim: int, cell: float, seed: int, corrupt: float,
                        rs: np.random.RandomState) -> np.ndarray:
    """Seeded unit descriptors that agree for points falling into the same `cell`-sized cube of the COMMON frame (i.e.
    for true matches) -- a stand-in for FPFH / FCGF with a controllable outlier share (`corrupt` = fraction of points
    whose descriptor is replaced by an unrelated one).  Not a feature extractor: it needs the ground-truth frame."""
    q = np.floor(points_in_common_frame / cell).astype(np.int64)
    h = (q[:, 0] * 73856093) ^ (q[:, 1] * 19349663) ^ (q[:, 2] * 83492791) ^ seed
    out = np.empty((len(q), dim), dtype=np.float32)
    uniq, inv = np.unique(h, return_inverse=True)
    table = np.stack([np.random.RandomState(int(u) & 0x7FFFFFFF).standard_normal(dim) for u in uniq]).astype(np.float32)
    out[:] = table[inv]
    bad = rs.random_sample(len(q)) < corrupt
    out[bad] = rs.standard_normal((int(bad.sum()), dim)).astype(np.float32)
    out += rs.standard_normal(out.shape).astype(np.float32) * 0.05
    return out / np.linalg.norm(out, axis=1, keepdims=True)


def second_view(points: np.ndarray, seed: int, keep: float = 0.7, noise: float = 0.005) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """A second 'scan' of the same scene: a random sub-set of the points (partial overlap), sensor noise, moved by a
    random rigid motion.  Returns (tgt_points [m,3], gt_trans [4,4] with p_tgt = R p_src + t, tgt_points_in_src_frame)."""
    rs = np.random.RandomState(seed)
    T = random_rigid(rs)
    sel = rs.random_sample(len(points)) < keep
    in_src = points[sel] + rs.standard_normal((int(sel.sum()), 3)).astype(np.float32) * noise
    tgt = in_src @ T[:3, :3].T + T[:3, 3]
    return tgt.astype(np.float32), T, in_src.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# the pair loop (evaluation/test_3DMatch.py:20-103)
# ---------------------------------------------------------------------------------------------------------------
def gt_labels_from_trans(src_keypts: torch.Tensor, tgt_keypts: torch.Tensor, gt_trans: torch.Tensor, thr: float) -> torch.Tensor:
    """datasets/ThreeDMatch.py:292-296: a correspondence is an inlier if ||R src + t - tgt|| < inlier_threshold."""
    warped = src_keypts @ gt_trans[:3, :3].T + gt_trans[:3, 3]
    return ((warped - tgt_keypts).norm(dim=-1) < thr).float()


def eval_scene(model, pairs: Iterable[Dict[str, np.ndarray]], scene_ind: int = 0, re_thre: float = 15.0, te_thre: float = 30.0,
               inlier_threshold: float = 0.10, use_mutual: bool = False, device: str = "cuda:0", batch_size: int = 1) -> np.ndarray:
    """`pairs`: dicts with src_pts [ns,3], tgt_pts [nt,3], src_desc [ns,D], tgt_desc [nt,D], gt_trans [4,4] (numpy).
    Returns the [num_pair, 12] stats array of the reference's eval_3DMatch_scene.
    batch_size > 1 (r03): the correspondence sets of `batch_size` consecutive pairs -- every pair has its own N, as in the
    reference's evaluation (test_3DMatch.py:126 `num_node='all'`) -- go through ONE ragged call of the model (lists of
    per-pair tensors); model / data time are then the batch's time divided by its pairs."""
    rows: List[np.ndarray] = []
    dev = torch.device(device)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

    def flush(group):
        if not group:
            return
        t0 = time.perf_counter()
        if len(group) == 1:
            c = group[0]["corr"]
            res = model({"corr_pos": c["corr_pos"], "src_keypts": c["src_keypts"], "tgt_keypts": c["tgt_keypts"], "testing": True})   # test_3DMatch.py:53
            trans, labels = res["final_trans"], [res["final_labels"][0]]
        else:
            res = model({"corr_pos": [x["corr"]["corr_pos"][0] for x in group], "src_keypts": [x["corr"]["src_keypts"][0] for x in group],
                         "tgt_keypts": [x["corr"]["tgt_keypts"][0] for x in group], "testing": True})
            trans, labels = res["final_trans"], res["final_labels"]
        torch.cuda.synchronize(dev)
        model_time = (time.perf_counter() - t0) / len(group)
        for i, x in enumerate(group):
            st = ops.eval_stats(trans[i:i + 1], x["gt_trans"][None], labels[i][None], x["gt_labels"], re_thre, te_thre)[0].cpu().numpy()
            row = np.zeros(12)
            row[:9] = st
            row[9], row[10], row[11] = model_time, x["data_time"], scene_ind
            rows.append(row)

    with torch.no_grad():
        group = []
        for pair in pairs:
            t0 = time.perf_counter()
            corr = build_correspondences(g(pair["src_desc"]), g(pair["tgt_desc"]), g(pair["src_pts"]), g(pair["tgt_pts"]),
                                         use_mutual=use_mutual)
            gt_trans = g(pair["gt_trans"]).float()
            gt_labels = gt_labels_from_trans(corr["src_keypts"][0], corr["tgt_keypts"][0], gt_trans, inlier_threshold)[None]
            torch.cuda.synchronize(dev)
            group.append({"corr": corr, "gt_trans": gt_trans, "gt_labels": gt_labels, "data_time": time.perf_counter() - t0})
            if len(group) >= max(1, batch_size):
                flush(group)
                group = []
        flush(group)
    return np.stack(rows) if rows else np.zeros((0, 12))


def summarize(stats: np.ndarray) -> Dict[str, float]:
    """The aggregate lines of eval_3DMatch (evaluation/test_3DMatch.py:141-176)."""
    ok = stats[:, 0] > 0
    return {
        "num_pairs": int(len(stats)),
        "registration_recall_pct": float(stats[:, 0].mean() * 100.0) if len(stats) else float("nan"),
        "mean_RE_deg_success": float(stats[ok, 1].mean()) if ok.any() else float("nan"),
        "mean_TE_cm_success": float(stats[ok, 2].mean()) if ok.any() else float("nan"),
        "mean_input_inlier_ratio_pct": float(stats[:, 4].mean() * 100.0) if len(stats) else float("nan"),
        "mean_precision_pct": float(stats[:, 6].mean() * 100.0) if len(stats) else float("nan"),
        "mean_recall_pct": float(stats[:, 7].mean() * 100.0) if len(stats) else float("nan"),
        "mean_f1_pct": float(stats[:, 8].mean() * 100.0) if len(stats) else float("nan"),
        "mean_model_time_s": float(stats[:, 9].mean()) if len(stats) else float("nan"),
        "mean_data_time_s": float(stats[:, 10].mean()) if len(stats) else float("nan"),
    }


def demo_pairs(cloud: np.ndarray, num_pairs: int, dim: int = 33, cell: float = 0.05, corrupt: float = 0.6,
               seed: int = 0) -> Iterable[Dict[str, np.ndarray]]:
    """Pairs for the loop from ONE down-sampled cloud: view i = `second_view(cloud, seed + i)`, stand-in descriptors."""
    for i in range(num_pairs):
        rs = np.random.RandomState(10_000 + seed + i)
        tgt, T, tgt_in_src = second_view(cloud, seed + i)
        yield {"src_pts": cloud, "tgt_pts": tgt, "gt_trans": T,
               "src_desc": standin_descriptors(cloud, dim, cell, seed + i, corrupt, rs),
               "tgt_desc": standin_descriptors(tgt_in_src, dim, cell, seed + i, corrupt * 0.5, rs)}
