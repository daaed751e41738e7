"""Launch times of the stage-entry weight-gradient pairs (generic wgrad_kernel, PROJ form) at the benchmark's shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops
dt = torch.bfloat16
n = 2048
g = torch.Generator(device="cuda").manual_seed(1)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
for cin, cout, h in ((20, 40, 64), (40, 60, 32), (60, 80, 16)):
    cp, cop = ops.cpad(cin), ops.cpad(cout)
    x = torch.randn(n, h, h, cp, device="cuda", generator=g).to(dt); x[..., cin:] = 0
    dz1 = torch.randn(n, h // 2, h // 2, cop, device="cuda", generator=g).to(dt); dz1[..., cout:] = 0
    dz2 = torch.randn(n, h // 2, h // 2, cop, device="cuda", generator=g).to(dt); dz2[..., cout:] = 0
    ws = [None]
    def run():
        r = ops.conv_wgrad_pair(x, dz1, dz2, cin, cout, workspace=ws[0]); ws[0] = r[3]; return r
    ts = min(t(run) for _ in range(3))
    r = run()
    out.append(f"{cin}->{cout}@{h}: {ts:.1f} us (chk {float(r[0].abs().sum()):.4f} {float(r[1].abs().sum()):.4f} {float(r[2].abs().sum()):.4f})")
# plain 80-channel wgrad (layer 4)
x = torch.randn(n, 8, 8, 80, device="cuda", generator=g).to(dt); dz = torch.randn(n, 8, 8, 80, device="cuda", generator=g).to(dt)
ts = min(t(lambda: ops.conv_wgrad(x, dz, 80, 80, ks=3, stride=1, pad=1)) for _ in range(3))
out.append(f"80->80@8 plain: {ts:.1f} us")
print(" | ".join(out))
