import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mil_amd
from mil_amd import ops, _lib as L
from gpu_util import to_nhwc, round_to, cpad
dt = torch.bfloat16
c, n, h, w = 20, 2, 64, 64
g = torch.Generator().manual_seed(307 + c + h)
x = round_to(torch.randn(n, c, h, w, generator=g), dt)
w1 = round_to(torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5, dt)
w2 = round_to(torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5, dt)
b1, b2 = torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1
xg = to_nhwc(x, dt)
p1, bp1 = ops.pack_weights(w1.cuda(), b1.cuda(), L.PACK_FWD, dt)
p2, bp2 = ops.pack_weights(w2.cuda(), b2.cuda(), L.PACK_FWD, dt)
def ne(a, b): return int((a.view(torch.int16) != b.view(torch.int16)).sum())
o1a, ya = ops.conv_block_fwd(xg, p1, bp1, p2, bp2)
o1b, yb = ops.conv_block_fwd(xg, p1, bp1, p2, bp2)
print("fused vs fused:", ne(o1a, o1b), ne(ya, yb))
os.environ["MIL_PF_MIN_TILES"] = "1"
z1p = ops.conv(xg, p1, bp1, cpad(c), ks=3, stride=1, pad=1, lrelu=True)
z1p2 = ops.conv(xg, p1, bp1, cpad(c), ks=3, stride=1, pad=1, lrelu=True)
os.environ["MIL_PF_MIN_TILES"] = "1000000000"
z1g = ops.conv(xg, p1, bp1, cpad(c), ks=3, stride=1, pad=1, lrelu=True)
torch.cuda.synchronize()
print("pf vs pf:", ne(z1p, z1p2), " pf vs generic:", ne(z1p, z1g), " fused vs pf:", ne(o1a, z1p), " fused vs generic:", ne(o1a, z1g))
