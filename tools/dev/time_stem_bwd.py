import sys, torch
sys.path.insert(0, '/root/repo')
import mil_amd
from mil_amd import ops
dt = torch.bfloat16
n = 2048
xs = torch.randn(n, 128, 128, 16, device='cuda').to(dt)
gp = torch.randn(n, 64, 64, 24, device='cuda').to(dt)
widx = torch.randint(0, 9, (n, 64, 64, 24), device='cuda', dtype=torch.uint8)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ws = [None]
def run():
    return ops.stem_bwd_fused(xs, gp, widx)
print(f"stem_bwd_fused {t(run):.1f} us")
