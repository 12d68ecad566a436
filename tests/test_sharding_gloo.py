"""N>1 path on CPU: two gloo ranks shard a batch of pairs, run a stand-in forward on their shard and
all_gather poses + labels (pointdsc_amd/sharding.py).  The real forward needs a GPU; the sharding /
gather logic is device-agnostic, which is what is covered here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointdsc_amd import sharding


def _fake_forward(data):
    """Deterministic per-pair 'result' so that every rank can verify the gathered tensors."""
    corr = data["corr_pos"]
    bs, n = corr.shape[0], corr.shape[1]
    trans = torch.eye(4).repeat(bs, 1, 1)
    trans[:, :3, 3] = corr[:, :, :3].sum(1)
    labels = (corr[:, :, 0] > 0).float()
    return {"final_trans": trans, "final_labels": labels, "M": None}


def _worker(rank, world, port, total, with_labels, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        data = {"corr_pos": torch.randn(total, 37, 6, generator=g), "testing": True}
        out = sharding.forward_sharded(_fake_forward, data, gather_labels=with_labels)
        want = _fake_forward(data)
        ok = torch.equal(out["final_trans"], want["final_trans"])
        if with_labels:
            ok = ok and torch.equal(out["final_labels"], want["final_labels"])
        else:
            ok = ok and out["final_labels"] is None
        q.put((rank, bool(ok), tuple(out["final_trans"].shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total,with_labels", [(4, True), (5, True), (1, True), (6, False)])
def test_two_rank_shard_and_gather(total, with_labels):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, with_labels, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] for r in results), results
    assert all(r[2] == (total, 4, 4) for r in results)


def test_shard_bounds_cover_everything_once():
    for total in (0, 1, 7, 32):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    data = {"corr_pos": torch.randn(3, 10, 6), "testing": True}
    out = sharding.forward_sharded(_fake_forward, data)
    assert torch.equal(out["final_trans"], _fake_forward(data)["final_trans"])
