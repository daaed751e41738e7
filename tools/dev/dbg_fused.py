import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mil_amd
from mil_amd import ops, _lib as L
from gpu_util import to_nhwc
torch.manual_seed(0)
n, h, w = 3, 16, 16
dt = torch.bfloat16
x = torch.randn(n, 20, h, w)
wt = torch.randn(20, 20, 3, 3) / 13
wd, _ = ops.pack_weights(wt.cuda(), None, L.PACK_DGRAD, dt)
for name, dz in (("ones", torch.ones(n, 20, h, w)),
                 ("img_index", torch.arange(n).float().view(n, 1, 1, 1).expand(n, 20, h, w) + 1),
                 ("row_index", torch.arange(h).float().view(1, 1, h, 1).expand(n, 20, h, w)),
                 ("col_index", torch.arange(w).float().view(1, 1, 1, w).expand(n, 20, h, w))):
    dx, dw, db = ops.conv_bwd_fused(to_nhwc(dz.contiguous(), dt), wd, to_nhwc(x, dt), 20, 20)
    print(name, "want", float(dz[:, 0].sum()), "got", db.cpu().tolist())
dz = (torch.arange(20).float() + 1).view(1, 20, 1, 1).expand(n, 20, h, w).contiguous()
dx, dw, db = ops.conv_bwd_fused(to_nhwc(dz, dt), wd, to_nhwc(x, dt), 20, 20)
print("chan", [round(v / 768, 3) for v in db.cpu().tolist()])
n1 = 1
dz = torch.ones(n1, 20, h, w)
dx, dw, db = ops.conv_bwd_fused(to_nhwc(dz, dt), wd, to_nhwc(x[:1], dt), 20, 20)
print("one image ones", db.cpu().tolist())
dz = torch.zeros(n1, 20, h, w); dz[0, :, 5, 7] = 1
dx, dw, db = ops.conv_bwd_fused(to_nhwc(dz, dt), wd, to_nhwc(x[:1], dt), 20, 20)
print("single pixel (5,7)", db.cpu().tolist())
dz = torch.zeros(n1, 20, h, w); dz[0, :, 0, 0] = 1
dx, dw, db = ops.conv_bwd_fused(to_nhwc(dz, dt), wd, to_nhwc(x[:1], dt), 20, 20)
print("single pixel (0,0)", db.cpu().tolist())
