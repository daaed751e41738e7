"""Which torch-level ops (copies, fills, adds) run inside one headline step beside the HIP kernels, with the Python frames that
issue them (torch.profiler): the glue DESIGN.md §7 keeps an eye on.     python tools/trace_step_glue.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda:0")
net = mil_amd.Attention(3, compute_dtype=torch.bfloat16, device=dev).eval()       # eval = full-bag path, as bench.py
import numpy as np  # noqa: E402
w = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "weights.npz"))
net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
flat = mil_amd.FlatParams(net)
opt = mil_amd.FlatAdam(flat, lr=2e-4)
bags, tiles = 8, 256
x = torch.randn((bags * tiles, 3, 256, 256), device=dev).clamp_(-1, 1)
sizes = [tiles] * bags
labels = torch.randint(0, 3, (bags,), device=dev)


def step():
    flat.zero_grad()
    outs = net.forward_bags((x, sizes), labels)
    outs.loss.sum().backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.events() if e.device_type.name == "CUDA" and ("copy" in e.name.lower() or "fill" in e.name.lower() or "at::native" in e.name)]
print(len(rows), "torch-issued device ops in one step")
by = {}
for e in prof.events():
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::zeros", "aten::empty_like", "aten::mul", "aten::sum"):
        stack = [s for s in (e.stack or []) if "mil_amd" in s or "network_amd" in s or "bench" in s][:2]
        key = (e.name, tuple(stack))
        by[key] = by.get(key, 0) + 1
for (name, stack), n in sorted(by.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n:3d} x {name:18s} {' <- '.join(stack)}")
