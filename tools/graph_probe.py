#!/usr/bin/env python3
"""Does capturing one forward in a hipGraph help the small configuration?  

    python tools/graph_probe.py [--config n1000_b1]

pdsc_forward_testing only enqueues kernels (no sync, no allocation), so a forward is capturable as is: the probe captures
`model(data)` with torch.cuda.CUDAGraph (static inputs / outputs), checks the replay reproduces the eager result bit for
bit, and times eager calls against graph replays, each as a long back-to-back run on one stream.
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="n1000_b1")
ap.add_argument("--iters", type=int, default=2000)
a = ap.parse_args()
w = workloads.WORKLOADS[a.config]
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(a.config, model.state_dict()))
model = model.eval().cuda()
batch = workloads.batch(a.config, 0, w["global_batch"])
data = {k: batch[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
data["testing"] = True
with torch.no_grad():
    for _ in range(5):
        eager = model(data)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            model(data)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(data)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out["final_trans"], eager["final_trans"]) and torch.equal(out["final_labels"], eager["final_labels"])

    def timed(fn):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.iters * 1e3

    t_eager = timed(lambda: model(data))
    t_graph = timed(g.replay)
    t_eager2 = timed(lambda: model(data))
print(f"{a.config}: replay == eager bit for bit: {same}; eager {t_eager:.4f} ms / forward, graph replay {t_graph:.4f} ms, "
      f"eager again {t_eager2:.4f} ms  ({w['global_batch']} pair(s) per forward)")

# r03: several captured forwards in flight -- one graph per HIP stream / workspace slot, replayed round-robin (the eager form of
# this is pointdsc_amd.pipeline.InFlight; with graphs the ~45 launches per forward cost the host one call)
from pointdsc_amd.pipeline import InFlight  # noqa: E402
with torch.no_grad():
    for depth in (2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(depth)]
        graphs, outs = [], []
        for slot, st in enumerate(streams):
            model._ws_slot = slot
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(3):
                    model(data)
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                outs.append(model(data))
            graphs.append(gr)
        model._ws_slot = 0
        torch.cuda.synchronize()
        i = [0]

        def replay_next():
            k = i[0] % depth
            i[0] += 1
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()

        t = timed(replay_next)
        torch.cuda.synchronize()
        ok = all(torch.equal(o["final_trans"], eager["final_trans"]) for o in outs)
        runner = InFlight(model, depth=depth)
        t_eager_inflight = timed(lambda: runner(data))
        print(f"{a.config}: {depth} captured forwards in flight: {t:.4f} ms / forward (results equal: {ok}); eager InFlight depth {depth}: {t_eager_inflight:.4f} ms")

