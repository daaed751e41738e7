import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mil_amd
from mil_amd import ops, _lib as L
from gpu_util import to_nhwc
torch.manual_seed(0)
n, h, w = 1, 16, 16
dt = torch.bfloat16
x = torch.randn(n, 20, h, w).to(dt).float()
dz = torch.randn(n, 20, h, w).to(dt).float()
wt = (torch.randn(20, 20, 3, 3) / 13).to(dt).float()
wd, _ = ops.pack_weights(wt.cuda(), None, L.PACK_DGRAD, dt)
ref = ops.conv(to_nhwc(dz, dt), wd, None, 24, ks=3, stride=1, pad=1)           # plain dgrad through the PF conv kernel
dx, dw, db = ops.conv_bwd_fused(to_nhwc(dz, dt), wd, to_nhwc(x, dt), 20, 20, mask=False)
err = (dx.float() - ref.float()).abs()          # [1,16,16,24]
print("max err", float(err.max()))
bad = (err > 0.05)
print("bad per channel", bad.sum(dim=(0, 1, 2)).tolist())
print("bad per row", bad.sum(dim=(0, 2, 3)).tolist())
print("bad per col", bad.sum(dim=(0, 1, 3)).tolist())
