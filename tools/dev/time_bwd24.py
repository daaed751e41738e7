"""Launch time of the 24-channel fused backward on the benchmark's layer-1 shape (2048 x 64x64x24), three variants;
   MIL_BWD16=0 selects the generic kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
n, h, c = 2048, 64, 24
g = torch.Generator(device="cuda").manual_seed(1)
def rnd():
    t = torch.randn(n, h, h, c, device="cuda", generator=g).to(dt); t[..., 20:] = 0; return t
dz, x, add = rnd(), rnd(), rnd()
w = torch.randn(20, 20, 3, 3, device="cuda", generator=g) * 0.05
wd, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dt)
need = ops.bwd_fused_workspace_bytes(n, h, h, 20, 20, 3, 1, dt)
ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {}
for name, addend, mask in (("noadd+mask", None, True), ("add+mask", add, True), ("add", add, False)):
    res[name] = t(lambda: ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=addend, mask=mask, workspace=ws))
print("MIL_BWD16=" + os.environ.get("MIL_BWD16", "1"), {k: round(v, 1) for k, v in res.items()}, "us (incl. slab reduction launch)")
dx, dw, db = ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=add, mask=True, workspace=ws)
print("   checksum dx %.6f dw %.6f db %.6f" % (float(dx.float().abs().sum()), float(dw.abs().sum()), float(db.abs().sum())))
