"""Which dW / db entries differ between one split (>2 GiB) fused-backward launch and two launches on halves."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
n, h, c = int(os.environ.get("N", 11200)), 64, 20
g = torch.Generator(device="cuda").manual_seed(9)
def rnd():
    t = torch.randn((n, h, h, 24), generator=g, device="cuda", dtype=torch.float32).to(dt); t[..., c:] = 0; return t
dz, x, add = rnd(), rnd(), rnd()
wd, _ = ops.pack_weights((torch.randn(c, c, 3, 3, generator=g, device="cuda") * 0.05), None, L.PACK_DGRAD, dt)
whole = ops.conv_bwd_fused(dz, wd, x, c, c, addend=add, mask=True)
half = n // 2
a = ops.conv_bwd_fused(dz[:half], wd, x[:half], c, c, addend=add[:half], mask=True)
b = ops.conv_bwd_fused(dz[half:], wd, x[half:], c, c, addend=add[half:], mask=True)
dw, ref = whole[1], a[1] + b[1]
print("dx equal", torch.equal(whole[0][:half], a[0]) and torch.equal(whole[0][half:], b[0]))
print("dW rel", float((dw - ref).norm() / ref.norm()), "db rel", float((whole[2] - a[2] - b[2]).norm() / (a[2] + b[2]).norm()))
d = (dw - ref).abs()
print("per tap max abs err", d.amax(dim=(0, 1)).flatten().tolist())
print("per co  max abs err", d.amax(dim=(1, 2, 3)).tolist())
print("per ci  max abs err", d.amax(dim=(0, 2, 3)).tolist())
print("ref scale", float(ref.abs().mean()))
