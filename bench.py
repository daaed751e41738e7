#!/usr/bin/env python3
"""Headline benchmark: tiles/sec, forward+backward, ResNet-26 + attention-MIL on synthetic 256x256x3
bags (BASELINE.json metric, configs[1]: 8 bags x 256 tiles per GPU, bf16 operands / fp32 accumulate).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = zero the flat gradient bucket, re-pack the filters (they changed), one encoder pass over every tile of
this rank's bags (full-bag path: all tiles through the backbone, gradients enabled), the segmented MIL head, the
full backward, — for N>1 — one RCCL all-reduce (sum) of the flat gradient bucket, and one fused Adam step.  Weak scaling: every rank owns 8 bags.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GFLOP_PER_TILE_FWD_BWD_256 = 1.6298    # fwd + dgrad + wgrad of every conv, no stem dgrad
MFMA_PEAK_BF16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA
MFMA_PEAK_F32_TFLOPS = 157.3
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bags", type=int, default=8, help="bags per GPU per step")
    ap.add_argument("--tiles", type=int, default=256, help="tiles per bag")
    ap.add_argument("--size", type=int, default=256, help="tile edge in pixels")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="run the separate weight-gradient launches on a side stream "
                    "(measured: same throughput on this workload, +2%% at 512 tiles, -24%% at 64 tiles; off by default)")
    ap.add_argument("--no-overlap", action="store_true", help="(default now) weight-gradient launches on the main stream")
    ap.add_argument("--infer", action="store_true",
                    help="BASELINE config 5 instead of the headline metric: forward only, no_grad (attention-map extraction)")
    return ap.parse_args()


def cpu_baseline(size, weights_npz):
    """The oracle (CPU restatement of the reference, pinned by golden vectors) timed on this box's host
    cores on a bounded sample of the same workload: 1 bag x 64 tiles, forward+backward, full-bag path."""
    from oracle import mil_oracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    sd = orc.load_state(weights_npz, requires_grad=True)
    gen = torch.Generator().manual_seed(20260104)
    n = 64
    x = torch.randn(n, 3, size, size, generator=gen).clamp_(-1.0, 1.0)
    y = torch.tensor([1])
    times = []
    for it in range(6):
        for p in sd.values():
            p.grad = None
        t0 = time.perf_counter()
        out = orc.attention_forward(sd, x, y)
        out["loss"].backward()
        dt = time.perf_counter() - t0
        if it >= 1:
            times.append(dt)
        if sum(times) > 25.0:
            break
    med = float(np.median(times))
    return {"value": n / med, "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"oracle/mil_oracle.py (fp32 torch CPU, {cores} threads): 1 bag x {n} tiles @{size}x{size}, "
                      f"fwd+bwd, median of {len(times)} after 1 warm-up"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one process per GPU over RCCL; MIL_DIST_BACKEND=gloo (+ ranks sharing a device) exists only to rehearse the
        # multi-process path on a one-GPU box
        backend = os.environ.get("MIL_DIST_BACKEND", "nccl")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif args.gpus != 1:
        raise SystemExit("--gpus N>1 must be launched through torch.distributed.run (one process per GPU)")
    else:
        torch.cuda.set_device(0)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import mil_amd
    from mil_amd import ops

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    net = mil_amd.Attention(3, compute_dtype=dtype, device=dev).eval()     # eval = full-bag path (all tiles encoded)
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
    net.cnn.module.overlap_wgrad = bool(args.overlap) and not args.no_overlap
    flat = mil_amd.FlatParams(net)
    flat.broadcast_params()
    opt = mil_amd.FlatAdam(flat, lr=2e-4)               # reference optimizer: Adam(lr=2e-4), gbm/classify_combined.py:519

    # synthetic bags, generated on the device and resident before the timed region
    n_tiles = args.bags * args.tiles
    gen = torch.Generator(device=dev).manual_seed(20260104 + rank)
    x_all = torch.empty((n_tiles, 3, args.size, args.size), dtype=torch.float32, device=dev)
    for b in range(args.bags):          # bag by bag: bounded temporary memory
        x_all[b * args.tiles:(b + 1) * args.tiles] = torch.randn(
            (args.tiles, 3, args.size, args.size), generator=gen, device=dev).clamp_(-1.0, 1.0)
    sizes = [args.tiles] * args.bags
    labels = torch.tensor([(rank * args.bags + b) % 3 for b in range(args.bags)], device=dev)

    def infer_step():
        with torch.no_grad():
            return net.forward_bags((x_all, sizes), labels)

    def step():
        if args.infer:
            return infer_step()
        flat.zero_grad()
        outs = net.forward_bags((x_all, sizes), labels)
        outs.loss.sum().backward()      # == sum of the per-bag o["loss"] (the reference accumulates bag gradients un-normalised)
        flat.allreduce_grads()
        opt.step()                      # weights change every step: the next forward re-packs all filters
        return outs

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # calibration: what a plain device copy reaches on this box (read + write bytes / time), outside the timed region.
    # Every rank runs it, for about half a second: it doubles as the clock ramp of a freshly started box (three warm-up
    # steps are 35 ms of GPU work, far less than the power-state transition).  The W warm-up steps follow directly.
    copy_gbps = None
    if not args.no_kernel_timer:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)         # 1 GiB
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        best, t_start = 0.0, time.perf_counter()
        while time.perf_counter() - t_start < 0.5:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 20 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        copy_gbps = best
        del src, dst
    for _ in range(args.warmup):
        step()
    # Candidate dominant kernels, all on the 64x64 maps at 20(24)->20(24) channels (layer 1, 47% of the FLOPs):
    #   conv_block_fwd_kernel<24,2,4>   — the forward of a whole identity block (or conv_igemm_pf_kernel<24,2,3,4> per conv)
    #   conv_bwd_fused_kernel<24,2,3>   — the fused backward of one conv (data gradient + weight gradient in one pass)
    # Every launch of them is bracketed with HIP events on the launch stream; the one with the larger total time
    # in the timed region is reported as `roofline`.
    timer = None
    if not args.no_kernel_timer:
        timer = ops.KernelTimer(lambda label: label[:6] in (("conv", 24, 24, 3, 1, False), ("bwd_fused", 24, 24, 3, 1, False))
                                or label[:2] == ("block_fwd", 24))
        ops.TIMER = timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(torch.stack([o["loss"].detach() for o in outs]).mean())
    if not np.isfinite(loss_val):
        raise SystemExit("non-finite loss in benchmark step")

    if rank == 0:
        total_tiles = n_tiles * world * args.steps
        value = total_tiles / elapsed
        scale = (args.size / 256.0) ** 2
        achieved_model_tflops = value * (0.5754 if args.infer else GFLOP_PER_TILE_FWD_BWD_256) * scale / 1e3
        peak = MFMA_PEAK_BF16_TFLOPS if args.dtype == "bf16" else MFMA_PEAK_F32_TFLOPS
        roofline = None
        if timer is not None:
            spans = timer.durations_ms()
            fams = {}
            for label, d in spans:
                fams.setdefault(label[0], []).append((label, d))
            if fams:
                fam = max(fams, key=lambda k: sum(d for _l, d in fams[k]))
                label = fams[fam][0][0]
                n_img, ho, wo = (label[2], label[3], label[4]) if fam == "block_fwd" else (label[6], label[7], label[8])
                avg_ms = float(np.mean([d for _l, d in fams[fam]]))
                esz = 2 if args.dtype == "bf16" else 4
                conv_flops = 2.0 * 9 * 20 * 20 * n_img * ho * wo          # algorithmic: 20 real channels in and out
                px_bytes = n_img * ho * wo * 20 * esz                      # one 20-channel activation tensor
                if fam == "block_fwd":
                    flops, alg_bytes = 2 * conv_flops, 3 * px_bytes        # two convs; read x once, write o1 and out
                    kname = (f"conv_block_fwd_kernel<24,2,4> (whole identity block forward: conv-lrelu-conv-add-lrelu, 20 ch, "
                             f"{ho}x{wo} maps, {n_img} tiles/launch)")
                elif fam == "conv":
                    flops, alg_bytes = conv_flops, 2 * px_bytes            # read x once, write y once
                    kname = (f"conv_igemm_pf_kernel<24,2,3,4> (3x3 s1 forward conv, 20->20 ch, {ho}x{wo} maps, "
                             f"{n_img} tiles/launch)")
                else:
                    flops, alg_bytes = 2 * conv_flops, 3 * px_bytes        # dgrad + wgrad; read dz, x once, write dx once
                    kname = (f"conv_bwd_fused_kernel<24,2,3> (fused data+weight gradient of the 3x3 s1 conv, 20->20 ch, "
                             f"{ho}x{wo} maps, {n_img} tiles/launch)")
                ach = flops / (avg_ms * 1e-3) / 1e12
                traffic = None
                try:                                                       # rocprofv3 PMC passes of this same command
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
                    key = {"conv": "conv_igemm_pf_kernel<24, 2, 3, 4", "block_fwd": "conv_block_fwd_kernel<24, 2"}.get(fam, "conv_bwd_fused_kernel<24, 2, 3")
                    hit = [v for k, v in pmc.items() if key in k]           # all template variants of the kernel family
                    if hit and n_img == 2048:
                        traffic = sum(v["hbm_bytes"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit)
                except (OSError, KeyError, ValueError):
                    traffic = None
                # Which roof: arithmetic intensity of the kernel's ALGORITHMIC work against the ridge point
                # (dense MFMA peak / HBM peak = 312 FLOP/B at bf16).  The 20-channel convs sit far on the HBM side
                # (120 FLOP/B fused backward, 90 FLOP/B forward), so the fraction is priced against HBM bandwidth;
                # the MFMA-side numbers are carried along for reference.
                ai = flops / alg_bytes
                ridge = peak * 1e12 / (HBM_PEAK_GBPS * 1e9)
                alg_gbps = alg_bytes / (avg_ms * 1e-3) / 1e9
                extras = {
                    "kernel": kname, "launches_timed": len(fams[fam]), "avg_launch_ms": avg_ms,
                    "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                    "kernel_tflops": ach, "kernel_frac_of_mfma_peak": ach / peak,
                    "hbm_capped_attainable_tflops": min(peak, ai * HBM_PEAK_GBPS / 1e3),
                    "measured_copy_gbps": copy_gbps,
                    "whole_step_model_tflops": achieved_model_tflops,
                    "whole_step_frac_of_mfma_peak": achieved_model_tflops / peak,
                    "other_timed_kernels": {k: {"launches": len(v), "avg_launch_ms": float(np.mean([d for _l, d in v]))}
                                            for k, v in fams.items() if k != fam},
                }
                if ai < ridge:
                    roofline = {"bound": "hbm", "achieved": alg_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": alg_gbps / HBM_PEAK_GBPS, "traffic": traffic, **extras}
                else:
                    roofline = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                "traffic": traffic, **extras}
        if args.infer:
            roofline = None
        line = {
            "metric": ("tiles/sec fwd-only attention map, 256x256x3 bags, ResNet-26+attn" if args.infer else
                       "tiles/sec fwd+bwd, 256x256x3 bags, ResNet-26+attn"),
            "value": value, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.bags} bags x {args.tiles} tiles @{args.size}x{args.size}x3 per GPU, "
                                   "ResNet-26 (20/40/60/80) + attention-MIL head, fwd+bwd full-bag path "
                                   "(BASELINE.json configs[1])",
                       "global_bags": args.bags * world, "tiles_per_bag": args.tiles, "tile": args.size,
                       "parallelism": f"bag-parallel dp{world}, one RCCL all-reduce of the flat 2.56 MB gradient bucket"},
            "roofline": roofline,
        }
        if args.infer:
            line["model_tflops"] = achieved_model_tflops
        if world == 1 and not args.no_cpu_baseline and not args.infer:
            line["cpu_baseline"] = cpu_baseline(args.size, w)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
