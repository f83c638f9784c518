"""N>1 path on CPU: world_size-2 gloo run of the only collective of the path (one flat all-reduce of the norm
gradients, latte_amd.tta_step.allreduce_mean_grads) and of the sequence sharding rule bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from latte_amd.tta_step import allreduce_mean_grads, shard_scan_offset
    torch.manual_seed(0)  # same parameters on every rank
    params = [torch.nn.Parameter(torch.randn(n)) for n in (32, 32, 96, 1, 256)]
    g = torch.Generator().manual_seed(100 + rank)  # different gradients per rank
    for i, p in enumerate(params):
        if i != 3 or rank == 0:  # one parameter has a gradient on rank 0 only
            p.grad = torch.randn(p.shape, generator=g)
    mine = [None if p.grad is None else p.grad.clone() for p in params]
    allreduce_mean_grads(params, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    ok = True
    for i, p in enumerate(params):
        want = sum((gr[i] if gr[i] is not None else torch.zeros_like(p)) for gr in gathered) / world
        ok = ok and torch.allclose(p.grad, want, rtol=0, atol=1e-7)
    # identical update on every rank afterwards
    opt = torch.optim.Adam(params, lr=1e-3)
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    ok = ok and torch.equal(flat, ref)
    offs = [None] * world
    dist.all_gather_object(offs, shard_scan_offset(rank, 6))
    ok = ok and offs == [6 * r for r in range(world)]
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_allreduce_of_norm_grads_world2_gloo():
    world = 2
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_ranks_own_disjoint_sequences():
    from latte_amd import synth
    from latte_amd.tta_step import shard_scan_offset
    w = synth.workload("nuscenes_34k_w5")
    ids = [set(range(shard_scan_offset(r, w["B"]), shard_scan_offset(r, w["B"]) + w["B"])) for r in range(8)]
    assert all(ids[a].isdisjoint(ids[b]) for a in range(8) for b in range(a + 1, 8))
    # a rank's scans form whole sequences: B is a multiple of the sequence length
    assert w["B"] % w["seq_len"] == 0
