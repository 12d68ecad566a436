#!/usr/bin/env python3
"""Down-sampled demo clouds as a small fixture (build container only: reads /root/reference/demo_data/*.ply).

    python oracle/make_demo_fixture.py      # writes tests/golden/demo_clouds_vox005.npz

The reference's demo (demo_registration.py:37-44) voxelises the two clouds at config.downsample = 0.05 m (3DMatch
snapshot config.json:32) before describing them.  The fixture holds those down-sampled clouds (about 5.3 k / 5.1 k points,
SURVEY.md section 0 item 4) so that the harness test runs on the GPU box, where /root/reference does not exist.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import harness  # noqa: E402

out = {}
for i in (0, 1):
    raw = harness.read_ply_xyz(f"/root/reference/demo_data/cloud_bin_{i}.ply")
    vox = harness.voxel_down_sample(raw, 0.05)
    print(f"cloud_bin_{i}: {len(raw)} vertices -> {len(vox)} occupied 0.05 m voxels")
    out[f"cloud_bin_{i}"] = vox
    out[f"cloud_bin_{i}_raw_vertices"] = np.int64(len(raw))
np.savez_compressed(ROOT / "tests" / "golden" / "demo_clouds_vox005.npz", **out)
