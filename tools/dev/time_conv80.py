"""Launch times of the 80-channel 3x3 convs (8x8 maps, 2048 tiles): forward and data-gradient variants."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
c, n, h = (int(v) for v in os.environ.get("SHAPE", "80,2048,8").split(","))
g = torch.Generator(device="cuda").manual_seed(2)
x = torch.randn(n, h, h, c, device="cuda", generator=g).to(dt)
r = torch.randn(n, h, h, c, device="cuda", generator=g).to(dt)
w = torch.randn(c, c, 3, 3, device="cuda", generator=g) * 0.03
b = torch.randn(c, device="cuda", generator=g) * 0.1
pf, bp = ops.pack_weights(w, b, L.PACK_FWD, dt)
pd, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dt)
def t(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
fns = {"fwd lrelu": lambda: ops.conv(x, pf, bp, c, ks=3, stride=1, pad=1, lrelu=True),
       "fwd res+lrelu": lambda: ops.conv(x, pf, bp, c, ks=3, stride=1, pad=1, res=r, lrelu=True),
       "dgrad act": lambda: ops.conv(x, pd, None, c, ks=3, stride=1, pad=1, act=r),
       "dgrad res+act": lambda: ops.conv(x, pd, None, c, ks=3, stride=1, pad=1, res=r, act=r)}
fns["pair fwd block"] = lambda: ops.conv_pair(x, pf, bp, pf, bp, lreluA=True, resB=x, lreluB=True)
fns["pair dgrad chain"] = lambda: ops.conv_pair(x, pd, None, pd, None, actA=r, resB=x, actB=r)
print(" | ".join(f"{k} {t(f):.1f} us" for k, f in fns.items()))
print("   checksums", " ".join("none" if f() is None else f"{float((f()[1] if isinstance(f(), tuple) else f()).float().abs().sum()):.3f}" for f in fns.values()))
