#!/usr/bin/env python3
"""RCCL smoke of the N>1 bench path on ONE GPU (the pool gives this build single-GPU boxes only).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_probe.py

A one-rank "nccl" process group runs exactly the collective calls bench.py and pointdsc_amd/sharding.py make when
WORLD_SIZE > 1 -- init_process_group(backend="nccl", device_id=...), barrier, all_reduce(MAX) of a float64 device scalar,
all_gather_into_tensor of the [pairs, 16] fp32 poses and of the uint8 (pose bytes | labels) payload -- around a real
forward of the default workload's first 4 pairs.  It proves the calls, dtypes and device placement are accepted by RCCL
on this image; it says nothing about scaling (one rank, no xGMI traffic).
"""
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pointdsc_amd import PointDSC, sharding, workloads  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device("cuda", local_rank)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
world, rank = dist.get_world_size(), dist.get_rank()
print(f"process group up: backend {dist.get_backend()} world {world} rank {rank} device {dev}", flush=True)

name = workloads.DEFAULT
w = workloads.WORKLOADS[name]
B = 4
model = PointDSC(**w["model"])
model.load_state_dict(workloads.state_dict(name, model.state_dict()))
model = model.eval().to(dev)
batch = workloads.batch(name, rank * B, B)
data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
data["testing"] = True


def fence():
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()


fence()
t0 = time.perf_counter()
with torch.no_grad():
    res = model(data)
fence()
t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print(f"barrier + all_reduce(MAX, float64 on {t.device}): {float(t.item()) * 1e3:.2f} ms for one forward of {B} pairs", flush=True)

# the two payload shapes of sharding.gather_results, called directly (that function short-cuts at world == 1)
poses = res["final_trans"].reshape(B, 16).contiguous()
flat = torch.empty((world * B, 16), dtype=poses.dtype, device=dev)
dist.all_gather_into_tensor(flat, poses)
assert torch.equal(flat[rank * B:(rank + 1) * B], poses)
lab = res["final_labels"].to(torch.uint8)
payload = torch.cat([poses.view(torch.uint8).reshape(B, 64), lab], dim=1).contiguous()
flat8 = torch.empty((world * B,) + tuple(payload.shape[1:]), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(flat8, payload)
assert torch.equal(flat8[rank * B:(rank + 1) * B], payload)
back = flat8[:, :64].reshape(-1).clone().view(torch.float32).reshape(-1, 4, 4)
assert torch.equal(back, res["final_trans"])
out = sharding.gather_results(res["final_trans"], res["final_labels"], B * world)
assert torch.equal(out["final_trans"], res["final_trans"])
print(f"all_gather_into_tensor: fp32 poses {tuple(flat.shape)}, uint8 pose|label payload {tuple(flat8.shape)}: round trip exact", flush=True)
# r03: the bench's in-flight pattern -- the gather of step i rides behind forward i on forward i's side stream, the settle
# count is broadcast from rank 0, the timed region ends with one fence
from pointdsc_amd.pipeline import InFlight  # noqa: E402

t = torch.tensor([7], dtype=torch.int64, device=dev)
dist.broadcast(t, src=0)
assert int(t.item()) == 7
runner = InFlight(model, depth=2)
gathered = []


def gather_direct(r):
    p = r["final_trans"].reshape(B, 16).contiguous()
    o = torch.empty((world * B, 16), dtype=p.dtype, device=dev)
    dist.all_gather_into_tensor(o, p)        # under torch.cuda.stream(side): enqueued behind the forward on that stream
    return o


fence()
t0 = time.perf_counter()
for _ in range(6):
    gathered.append(runner(data, post=gather_direct))
fence()
dt = time.perf_counter() - t0
for g_ in gathered:
    assert torch.equal(g_["post"][rank * B:(rank + 1) * B].reshape(B, 4, 4), res["final_trans"])
print(f"6 forwards in flight on two streams, one all_gather_into_tensor behind each: {dt / 6 * 1e3:.2f} ms per step, gathers exact", flush=True)
dist.destroy_process_group()
print("RCCL_PROBE_OK", flush=True)
