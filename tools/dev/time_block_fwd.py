"""Launch time of the whole-block forward on the benchmark's layer-1 shape (2048 x 64x64x24)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
n, h, c = 2048, 64, 24
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n, h, h, c, device="cuda", generator=g).to(dt); x[..., 20:] = 0
w1 = torch.randn(20, 20, 3, 3, device="cuda", generator=g) * 0.05
w2 = torch.randn(20, 20, 3, 3, device="cuda", generator=g) * 0.05
b = torch.randn(20, device="cuda", generator=g) * 0.1
p1, b1 = ops.pack_weights(w1, b, L.PACK_FWD, dt)
p2, b2 = ops.pack_weights(w2, b, L.PACK_FWD, dt)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ts = [t(lambda: ops.conv_block_fwd(x, p1, b1, p2, b2)) for _ in range(3)]
o1, y = ops.conv_block_fwd(x, p1, b1, p2, b2)
print(f"block_fwd {min(ts):.1f} us (min of 3 x 20)  checksum o1 {float(o1.float().abs().sum()):.2f} y {float(y.float().abs().sum()):.2f}")
