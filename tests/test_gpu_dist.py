"""-m gpu: the N>1 paths with the MODEL in the loop, rehearsed as 2 ranks over gloo that share the one GPU of the test box
(the driver's 8-GPU run uses RCCL; the collectives and the sharding logic are the same code):

  * bag-parallel training step: rank r owns bags r, r+2, ...; after ONE all-reduce(SUM) of the flat gradient bucket every
    rank holds the gradient a single process computes over all bags (the reference sums bag gradients un-normalised,
    gbm/classify_combined.py:446-454), and after the fused Adam step the replicas hold identical parameters;
  * tile-parallel inference of one bag (BASELINE configs[4]; the reference's nn.DataParallel scatter/gather,
    gbm/model.py:132-135): `forward_tile_parallel` on ragged slices returns on every rank what one process returns for the
    whole bag;
  * bench.py launched through torch.distributed.run with 2 ranks: the JSON contract, the in-bench replica self-check, and
    `--infer` splitting ONE bag over the ranks.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bags():
    g = torch.Generator().manual_seed(404)
    sizes = [9, 5, 12, 7]
    return [torch.randn(n, 3, 64, 64, generator=g).clamp_(-1, 1) for n in sizes], torch.tensor([0, 1, 2, 1])


def _net(dtype=torch.float32):
    import mil_amd
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    net = mil_amd.Attention(3, compute_dtype=dtype).eval()
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
    return net


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mil_amd
        from mil_amd.dist import shard_bags
        bags, labels = _bags()
        net = _net()
        flat = mil_amd.FlatParams(net)
        flat.broadcast_params()
        opt = mil_amd.FlatAdam(flat, lr=1e-3)
        mine = shard_bags(len(bags), rank, world)
        flat.zero_grad()
        outs = net.forward_bags([bags[i].cuda() for i in mine], labels[mine])
        outs.loss.sum().backward()
        flat.allreduce_grads()
        grad_dp = flat.flat_grad.clone()
        opt.step()
        lo, hi = flat.flat.clone(), flat.flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_equal = bool(torch.equal(lo, hi))
        # the single-process answer over ALL bags
        ref = _net()
        rflat = mil_amd.FlatParams(ref)
        rflat.zero_grad()
        ref.forward_bags([b.cuda() for b in bags], labels).loss.sum().backward()
        scale = float(rflat.flat_grad.abs().max())
        grad_err = float((grad_dp - rflat.flat_grad).abs().max()) / scale
        # tile-parallel inference: ragged slices of one 23-tile bag
        x = torch.cat(bags[:3])[:23]
        cut = 14
        sl = x[:cut] if rank == 0 else x[cut:]
        tp = ref.forward_tile_parallel(sl.cuda(), torch.tensor([2]))
        with torch.no_grad():
            whole = ref(x.cuda(), torch.tensor([2]))
        tp_ok = all(torch.allclose(tp[k], whole[k], rtol=1e-5, atol=1e-7) for k in ("Aterm", "Mterm", "Fterm", "wROIs", "y_pred"))
        # the same check on the split-precision (default) and the bf16 paths: the slices run other launch sizes than the whole
        # bag, the gathered features must still give the whole-bag output (fp32 storage: 1e-4; bf16 storage: bf16 noise)
        for dt, rtol, atol in (("bf16x3", 1e-4, 1e-6), (torch.bfloat16, 5e-2, 1e-3)):
            net_dt = _net(dt)
            tp_dt = net_dt.forward_tile_parallel(sl.cuda(), torch.tensor([2]))
            with torch.no_grad():
                whole_dt = net_dt(x.cuda(), torch.tensor([2]))
            tp_ok = tp_ok and all(torch.allclose(tp_dt[k], whole_dt[k], rtol=rtol, atol=atol) for k in ("Aterm", "Mterm", "Fterm", "wROIs", "y_pred"))
            tp_ok = tp_ok and tuple(tp_dt["Aterm"].shape) == (3, 23)
        out.put((rank, replicas_equal, grad_err, scale > 0, tp_ok, tuple(tp["Aterm"].shape)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_gpu_gradients_and_tile_parallel():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(out.get(timeout=5) for _ in range(world))
    for rank, replicas_equal, grad_err, nonzero, tp_ok, ashape in got:
        assert replicas_equal, rank                      # identical parameters after all-reduce + Adam on every rank
        assert nonzero and grad_err < 1e-4, (rank, grad_err)     # fp32: summed shard gradients == single-process gradient
        assert tp_ok and ashape == (3, 23), rank


def _run_bench(extra):
    env = dict(os.environ, MIL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--size", "64", "--no-kernel-timer"] + extra
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_training_and_tile_parallel_inference():
    line = _run_bench(["--bags", "3", "--tiles", "16"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "tiles/s" and line["value"] > 0
    assert line["config"]["global_bags"] == 6 and "dp2" in line["config"]["parallelism"]
    assert "fp32_path" not in line and "cpu_baseline" not in line         # N=1 only
    line = _run_bench(["--infer", "--bags", "1", "--tiles", "50"])
    assert line["scaling"] == "strong" and line["config"]["global_bags"] == 1 and "tile-parallel tp2" in line["config"]["parallelism"]
    assert "fwd-only" in line["metric"] and "split over 2 ranks" in line["config"]["workload"]


def test_bench_gpus_n_without_a_launcher_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how a driver calls `--gpus 1`): bench.py starts torch.distributed.run as
    a child process before touching the GPU and relays rank 0's line, with the collective's own time in it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MIL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64",
           "--bags", "2", "--tiles", "16", "--no-kernel-timer"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_bags"] == 4
    assert line["allreduce_ms"] > 0 and line["step_ms_no_comm"] > 0


def _real_shape_worker(rank, world, port, out):
    """Bag-parallel step at the BENCHMARK's bag shape: one 256-tile bag @256x256 per rank, in the two fast compute modes."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mil_amd
        res = []
        for mode in (torch.bfloat16, mil_amd.BF16X3):
            bags = []
            for b in range(world):
                g = torch.Generator(device="cuda").manual_seed(20260104 + b)
                bags.append(torch.randn((256, 3, 256, 256), generator=g, device="cuda").clamp_(-1.0, 1.0))
            labels = torch.tensor([b % 3 for b in range(world)])
            net = _net(mode)
            flat = mil_amd.FlatParams(net)
            flat.broadcast_params()
            opt = mil_amd.FlatAdam(flat, lr=2e-4)
            flat.zero_grad()
            net.forward_bags([bags[rank]], labels[rank:rank + 1]).loss.sum().backward()
            flat.allreduce_grads()
            grad_dp = flat.flat_grad.clone()
            opt.step()
            lo, hi = flat.flat.clone(), flat.flat.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            replicas_equal = bool(torch.equal(lo, hi))
            ref = _net(mode)
            rflat = mil_amd.FlatParams(ref)
            rflat.zero_grad()
            ref.forward_bags(bags, labels).loss.sum().backward()
            scale = float(rflat.flat_grad.abs().max())
            # per parameter tensor: L2 error of the all-reduced gradient relative to the tensor's norm
            # (a tensor whose gradient is analytically zero — buffer.classifier.bias, tests/test_gpu_model.py — holds rounding
            # residue only: its error is measured against the bucket's scale, not against its own ~1e-8 norm)
            worst, wname, off = 0.0, "", 0
            for (k, p) in [(k, p) for k, p in ref.named_parameters() if p.requires_grad]:
                n = p.numel()
                a, r_ = grad_dp[off:off + n], rflat.flat_grad[off:off + n]
                e = float((a - r_).norm() / r_.norm().clamp_min(1e-4 * scale))
                if e > worst:
                    worst, wname = e, k
                off += n
            res.append((str(mode) + " (" + wname + ")", replicas_equal, worst, scale > 0))
            del bags, net, ref, flat, rflat
            torch.cuda.empty_cache()
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_two_ranks_at_the_benchmark_bag_shape_bf16_and_bf16x3():
    """Each rank encodes ONE 256-tile bag @256x256 (the unit BASELINE configs[1]/[3] are made of): the all-reduced (SUM)
    gradient equals what one process computes over both bags — same kernels on the same tiles, only the fp32 slab sums are
    grouped differently (a 512-tile launch walks its tiles on other workgroups) — and the replicas are bit-equal after Adam.
    gbm/model.py:132-135 is the strategy this replaces, gbm/classify_combined.py:446-454 the SUM semantics."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_real_shape_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(out.get(timeout=5) for _ in range(world))
    for rank, res in got:
        for mode, replicas_equal, worst, nonzero in res:
            print(f"rank {rank} {mode}: worst per-tensor L2 error of the all-reduced gradient {worst:.2e}")
            assert replicas_equal and nonzero, (rank, mode)
            # bf16: a kernel that picks another K order for another launch size would show up as bf16 roundings (1e-3..1e-2)
            assert worst < (1e-4 if "bf16x3" in mode else 2e-2), (rank, mode, worst)


def _rccl_worker(port, out):
    """ONE rank over the real backend ("nccl" = RCCL on ROCm) with the `device_id=` eager-init path bench.py uses: proves that
    librccl loads on the box, that a communicator comes up, and that the three collectives of the data path run on the flat
    buckets (forced: a one-rank group would otherwise short-circuit them)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import mil_amd
        from mil_amd.dist import gather_features
        net = _net()
        flat = mil_amd.FlatParams(net)
        before = flat.flat.clone()
        flat.broadcast_params(force=True)
        bags, labels = _bags()
        flat.zero_grad()
        net.forward_bags([b.cuda() for b in bags[:2]], labels[:2]).loss.sum().backward()
        g0 = flat.flat_grad.clone()
        flat.allreduce_grads(force=True)                       # SUM over one rank: the bucket itself
        feats = torch.randn(37, 80, device="cuda")
        gathered = gather_features(feats, force=True)          # all_gather of sizes + padded features
        torch.cuda.synchronize()
        out.put((dist.get_backend(), bool(torch.equal(before, flat.flat)), bool(torch.equal(g0, flat.flat_grad)),
                 float(g0.abs().max()) > 0, bool(torch.equal(gathered, feats)), int(flat.numel)))
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_collectives_on_the_flat_bucket():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0, p.exitcode
    backend, params_same, grads_same, nonzero, gather_same, numel = out.get(timeout=5)
    assert backend == "nccl"
    assert params_same and grads_same and nonzero and gather_same
    assert numel == 640967                                     # the 2.56 MB bucket
