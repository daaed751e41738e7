"""CPU (-m "not gpu"): the N>1 path on 2 gloo ranks — flat gradient bucket all-reduce (SUM), parameter
broadcast, ragged feature all-gather, bag sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mil_amd
        from mil_amd.dist import FlatParams, gather_features, shard_bags
        torch.manual_seed(100 + rank)                       # different init per rank on purpose
        net = mil_amd.Attention(3, device="cpu")
        flat = FlatParams(net)
        assert flat.numel == 640967
        flat.broadcast_params(src=0)
        ref = [torch.empty_like(flat.flat) for _ in range(world)]
        dist.all_gather(ref, flat.flat)
        assert all(torch.equal(ref[0], r) for r in ref)     # weights identical after broadcast
        # parameters are views of the flat buffer
        p0 = next(net.parameters())
        assert p0.data_ptr() == flat.flat.data_ptr()
        # rank-dependent gradients, written through the per-parameter .grad views
        flat.zero_grad()
        for i, p in enumerate(net.parameters()):
            p.grad.add_(float(rank + 1) * (i + 1))
        flat.allreduce_grads()
        want = sum(range(1, world + 1))
        for i, p in enumerate(net.parameters()):
            assert torch.all(p.grad == want * (i + 1))
        # ragged all-gather of features (tile-parallel inference of one bag)
        feats = torch.full((3 + rank, 80), float(rank))
        allf = gather_features(feats)
        assert allf.shape == (sum(3 + r for r in range(world)), 80)
        assert torch.all(allf[:3] == 0) and torch.all(allf[3:] == 1)
        mine = shard_bags(7, rank, world)
        assert mine == list(range(rank, 7, world))
        out.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_flat_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(out.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]
