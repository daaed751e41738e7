"""Fixed cost of the 24-channel fused backward launch: kernel time against the number of images (tiles per workgroup)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
h, c = 64, 24
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randn(20, 20, 3, 3, device="cuda", generator=g) * 0.05
wd, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dt)
for n in (32, 64, 128, 256, 512, 1024, 2048):
    def rnd():
        t = torch.randn(n, h, h, c, device="cuda", generator=g).to(dt); t[..., 20:] = 0; return t
    dz, x, add = rnd(), rnd(), rnd()
    need = ops.bwd_fused_workspace_bytes(n, h, h, 20, 20, 3, 1, dt)
    ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device="cuda")
    for _ in range(3): ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=add, mask=True, workspace=ws)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10): ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=add, mask=True, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"n={n:5d} tiles/wg={n*16/512:6.1f}  {min(ts):7.1f} us (min of 5 x 10 launches)   per tile-round {min(ts)/(n*16/512):6.2f} us")
