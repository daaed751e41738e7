import sys, torch
sys.path.insert(0, '/root/repo')
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
n, h, c = 2048, 64, 24
dz = torch.randn(n, h, h, c, device='cuda').to(dt); dz[..., 20:] = 0
x = torch.randn(n, h, h, c, device='cuda').to(dt); x[..., 20:] = 0
add = torch.randn(n, h, h, c, device='cuda').to(dt); add[..., 20:] = 0
w = torch.randn(20, 20, 3, 3, device='cuda') * 0.05
wd, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dt)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
need = ops.bwd_fused_workspace_bytes(n, h, h, 20, 20, 3, 1, dt)
ws = torch.empty((need + 3) // 4, dtype=torch.float32, device='cuda')
print(f"fused bwd (mask)      {t(lambda: ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=None, mask=True, workspace=ws)):.1f} us")
print(f"fused bwd (add+mask)  {t(lambda: ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=add, mask=True, workspace=ws)):.1f} us")
