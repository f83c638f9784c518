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


def _shard_worker(rank, world, port, out):
    """what bench.py --gpus N does with BASELINE configs[3] (multiseq_64), minus the GPU: every rank derives its scans from
    shard_sequences, builds the per-scan metadata like latte_amd.workload.build_batch, and the ranks compare notes"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from latte_amd import synth
    from latte_amd.tta_step import shard_sequences
    w = synth.workload("multiseq_64")
    L = w["seq_len"]
    n_seq = w["B"] // L
    s0, ns = shard_sequences(n_seq, world, rank)
    scans = list(range(s0 * L, (s0 + ns) * L))
    scenes = [f"scene-{1000 + g // L:04d}" for g in scans]
    ids = [g % L for g in scans]
    mine = dict(scans=scans, scenes=scenes, ids=ids)
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    ok = True
    every = [g for r in allr for g in r["scans"]]
    ok = ok and sorted(every) == list(range(w["B"])) and len(set(every)) == w["B"]      # disjoint, complete
    for r in allr:                                                                        # whole sequences only
        for sc in set(r["scenes"]):
            fr = sorted(i for s, i in zip(r["scenes"], r["ids"]) if s == sc)
            ok = ok and fr == list(range(L))
    for a in range(world):
        for b in range(a + 1, world):
            ok = ok and set(allr[a]["scenes"]).isdisjoint(allr[b]["scenes"])              # no scene on two ranks
    # HR_Block's window rule (latte/models/hr_module.py:192-195) never reaches another rank's frames
    from latte_amd.hr_module import HR_Block
    from latte_amd.workload import hr_cfg_of
    hr = HR_Block(**hr_cfg_of(w))
    fake = dict(scene_name=scenes, scan_ID=ids, ori_points=[torch.zeros(1, 3)] * len(scans))
    segs, _, _ = hr._plan(fake, "ori_points", fake, "ori_points")
    ok = ok and all(scenes[i] == scenes[j] and abs(ids[i] - ids[j]) <= w["W"] // 2 for i, j, _, _, _ in segs) and all(src == row for _, _, src, _, row in segs)
    want = sum(min(L - 1, i + w["W"] // 2) - max(0, i - w["W"] // 2) + 1 for i in range(L)) * ns
    ok = ok and len(segs) == want
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multiseq_64_shard_gives_each_rank_whole_sequences_gloo(world):
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_shard_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {r: True for r in range(world)}


def test_shard_sequences_covers_every_world_size():
    from latte_amd.tta_step import shard_sequences
    for world in range(1, 9):
        runs = [shard_sequences(8, world, r) for r in range(world)]
        assert sum(n for _, n in runs) == 8 and all(n >= 1 for _, n in runs)
        assert all(runs[r][0] + runs[r][1] == runs[r + 1][0] for r in range(world - 1)) and runs[0][0] == 0
    with pytest.raises(ValueError):
        shard_sequences(8, 2, 2)
