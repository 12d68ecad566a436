"""Multi-GPU driver: pairs are independent units, so a batch of correspondence sets
is split contiguously over the ranks of one node (one process per GPU), every rank runs the whole hot
path on its shard with no data-path communication, and ONE small collective returns the results:
``all_gather`` of the [B/ws,4,4] poses (64 B per pair) and, optionally, the 0/1 labels as uint8.

The message is latency-bound (a few KB), so it is a single all_gather on RCCL (backend "nccl" on ROCm) --
never a ring of large chunks; xGMI link bandwidth is irrelevant here.  The same code runs on gloo/CPU
tensors, which is how the world_size-2 tests exercise it without GPUs.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``total`` pairs: the first ``total % world`` ranks get one extra pair."""
    per, rem = divmod(total, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def gather_results(local_trans: torch.Tensor, local_labels: Optional[torch.Tensor], total: int,
                   group=None) -> Dict[str, Optional[torch.Tensor]]:
    """all_gather the per-rank shards back into [total,4,4] (+ [total,N] labels) on every rank."""
    # no process group: nothing to gather.  A process group of ONE rank still takes the collective path below (a 64-byte all_gather
    # with itself): that is how the RCCL code path is exercised on a one-GPU box (tests/test_sharding_gloo.py).
    if not (dist.is_available() and dist.is_initialized()):
        return {"final_trans": local_trans, "final_labels": local_labels}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cap = -(-total // world)                       # shards are padded to the largest shard
    n_local = local_trans.shape[0]

    def pad(t: torch.Tensor) -> torch.Tensor:
        if t.shape[0] == cap:
            return t.contiguous()
        out = t.new_zeros((cap,) + tuple(t.shape[1:]))
        out[:n_local] = t
        return out

    payload = pad(local_trans.reshape(n_local, 16))
    if local_labels is not None:                   # one message: 16 pose floats + N label bytes per pair
        lab = pad(local_labels.to(torch.uint8)).view(torch.uint8)
        payload = torch.cat([payload.view(torch.uint8).reshape(cap, 64), lab], dim=1).contiguous()
    flat = torch.empty((world * cap,) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(flat, payload, group=group)      # output = concatenation along dim 0
    gathered = flat.view((world, cap) + tuple(payload.shape[1:]))
    trans_parts, label_parts = [], []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        part = gathered[r, : hi - lo]
        if local_labels is not None:
            trans_parts.append(part[:, :64].reshape(-1).clone().view(torch.float32).reshape(-1, 4, 4))
            label_parts.append(part[:, 64:].to(torch.float32))
        else:
            trans_parts.append(part.reshape(-1, 4, 4))
    return {"final_trans": torch.cat(trans_parts, 0),
            "final_labels": torch.cat(label_parts, 0) if local_labels is not None else None}


def forward_sharded(forward: Callable[[Dict[str, torch.Tensor]], Dict[str, torch.Tensor]],
                    data: Dict[str, torch.Tensor], gather_labels: bool = True, group=None):
    """Run ``forward`` (e.g. a ``pointdsc_amd.PointDSC`` module) on this rank's contiguous shard of
    ``data`` ([B,N,*] tensors present on every rank) and gather all results on every rank."""
    total = data["corr_pos"].shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(total, rank, world)
    local = {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in data.items()}
    if hi > lo:
        res = forward(local)
        lt, ll = res["final_trans"], res["final_labels"]
    else:  # more ranks than pairs
        ref = data["corr_pos"]
        lt = ref.new_zeros((0, 4, 4))
        ll = ref.new_zeros((0, ref.shape[1]))
    out = gather_results(lt, ll if gather_labels else None, total, group)
    out["M"] = None
    return out
