import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mil_amd
from mil_amd import _lib as L, ops
from gpu_util import from_nhwc, round_to, rel_err
n, h, w, cout, slope = (2, 96, 80, 64, 0.0)
g = torch.Generator().manual_seed(17 + h + w)
x = torch.randn(n, 3, h, w, generator=g).cuda()
wt = (torch.randn(cout, 3, 7, 7, generator=g) * 0.1).cuda()
b = (torch.randn(cout, generator=g) * 0.1).cuda()
dt = torch.bfloat16
wp, bp = ops.pack_weights(wt, b, L.PACK_STEM, dt)
xs0 = ops.stem_s2d(x, dt)
stem = ops.conv(xs0, wp, bp, 64, ks=4, stride=1, pad=2, lrelu=True, slope=slope)
pool0, widx0 = ops.maxpool_fwd(stem)
xs1, pool1, widx1 = ops.stem_fwd_fused(x, wp, bp, 64, slope=slope, dtype=dt)
ref = F.max_pool2d(F.leaky_relu(F.conv2d(round_to(x.cpu(), dt), round_to(wt.cpu(), dt), b.cpu(), stride=2, padding=3), slope), 3, 2, 1)
print("pool0 vs ref", rel_err(from_nhwc(pool0, cout), ref), "pool1 vs ref", rel_err(from_nhwc(pool1, cout), ref), "equal", torch.equal(pool0, pool1))
print(b[:4], bp[:4])
