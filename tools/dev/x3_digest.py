"""sha256 of the split-precision path's outputs and gradients on a fixed synthetic bag (GPU box): run under two builds of
the library (MIL_LIB_PATH) to check that a kernel change is bit-neutral.  python tools/dev/x3_digest.py [tiles]"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("mil_amd", os.path.join(ROOT, "deep-convolutional-neural-network-resnet-26-and-attention-network_amd", "__init__.py"))
mil_amd = importlib.util.module_from_spec(spec)
sys.modules["mil_amd"] = mil_amd
spec.loader.exec_module(mil_amd)

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(7)
net = mil_amd.Attention(3, compute_dtype=mil_amd.BF16X3).cuda()
net.train(False)
x = torch.randn(tiles, 3, 256, 256, generator=torch.Generator().manual_seed(11)).clamp_(-1, 1).cuda()
out = net(x, torch.tensor([1]).cuda())
out["loss"].backward()
h = hashlib.sha256()
for _k, t in sorted(out.items()):
    if torch.is_tensor(t):
        h.update(t.detach().float().cpu().numpy().tobytes())
for n, p in sorted(net.named_parameters()):
    if p.grad is not None:
        h.update(p.grad.detach().float().cpu().numpy().tobytes())
print(h.hexdigest())
