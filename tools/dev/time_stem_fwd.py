"""Launch time of the fused stem forward on the benchmark shape (2048 x 3 x 256 x 256 fp32 -> xs, pool, widx)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
n = 2048
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n, 3, 256, 256, device="cuda", generator=g).clamp_(-1, 1)
w = torch.randn(20, 3, 7, 7, device="cuda", generator=g) * 0.08
b = torch.randn(20, device="cuda", generator=g) * 0.1
wp, bp = ops.pack_weights(w, b, L.PACK_STEM, torch.bfloat16)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ts = [t(lambda: ops.stem_fwd_fused(x, wp, bp, 24)) for _ in range(3)]
xs, pool, widx = ops.stem_fwd_fused(x, wp, bp, 24)
print(f"stem_fwd_fused {min(ts):.1f} us (min of 3 x 10)   checksum pool {float(pool.float().abs().sum()):.3f} widx {int(widx.long().sum())}")
