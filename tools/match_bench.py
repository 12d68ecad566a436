#!/usr/bin/env python3
"""Timing of the correspondence construction (csrc/match.hip) at the sizes of the hot path's inputs."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pointdsc_amd import correspondences  # noqa: E402


def main():
    dev = "cuda:0"
    for n, d in ((5000, 32), (5000, 33), (10000, 32), (20000, 32)):
        rs = np.random.RandomState(0)
        a = rs.randn(n, d).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = rs.randn(n, d).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        kp = torch.rand(n, 3, device=dev)
        for mutual in (False, True):
            for _ in range(3):
                correspondences.build_correspondences(ta, tb, kp, kp, use_mutual=mutual)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                correspondences.build_correspondences(ta, tb, kp, kp, use_mutual=mutual)
            e1.record()
            torch.cuda.synchronize()
            gpu_us = e0.elapsed_time(e1) / 20 * 1e3
            # one call at a time, waited for (what a per-pair evaluation loop sees: launch count matters here, not in the back-to-back loops)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                correspondences.build_correspondences(ta, tb, kp, kp, use_mutual=mutual)
                torch.cuda.synchronize()
            lat_us = (time.perf_counter() - t0) / 50 * 1e6
            t0 = time.perf_counter()
            dist = np.sqrt(2 - 2 * (a @ b.T) + 1e-6)
            idx = np.argmin(dist, axis=1)
            if mutual:
                np.argmin(dist, axis=0)
            cpu_ms = (time.perf_counter() - t0) * 1e3
            flops = 2.0 * n * n * d * (2 if mutual else 1)
            print(f"N={n} D={d} mutual={mutual}: GPU {gpu_us:8.1f} us ({flops / gpu_us / 1e6:6.2f} TFLOP/s; one call at a time, waited for: {lat_us:7.1f} us)   numpy on the host {cpu_ms:8.1f} ms")


if __name__ == "__main__":
    main()
