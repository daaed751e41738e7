import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mil_amd
from mil_amd import ops, _lib as L
from gpu_util import to_nhwc, round_to, cpad
import torch.nn.functional as F
dt = torch.bfloat16
for (c, n, h, w) in [(20, 1, 16, 16), (20, 2, 64, 64), (40, 1, 32, 32)]:
    g = torch.Generator().manual_seed(307 + c + h)
    x = round_to(torch.randn(n, c, h, w, generator=g), dt)
    w1 = round_to(torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5, dt)
    w2 = round_to(torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5, dt)
    b1, b2 = torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1
    xg = to_nhwc(x, dt)
    p1, bp1 = ops.pack_weights(w1.cuda(), b1.cuda(), L.PACK_FWD, dt)
    p2, bp2 = ops.pack_weights(w2.cuda(), b2.cuda(), L.PACK_FWD, dt)
    o1, y = ops.conv_block_fwd(xg, p1, bp1, p2, bp2)
    z1 = ops.conv(xg, p1, bp1, cpad(c), ks=3, stride=1, pad=1, lrelu=True)
    z2 = ops.conv(z1, p2, bp2, cpad(c), ks=3, stride=1, pad=1, res=xg, lrelu=True)
    torch.cuda.synchronize()
    for name, a, b in (("o1", o1, z1), ("y", y, z2)):
        d = (a.view(torch.int16) != b.view(torch.int16))
        print(c, h, name, "mismatch", int(d.sum()), "of", d.numel(), "maxabs", float((a.float() - b.float()).abs().max()))
        if d.any():
            idx = d.nonzero()[:6]
            print(idx.tolist(), [ (float(a[tuple(i)]), float(b[tuple(i)])) for i in idx])
            print("by y:", d.any(dim=3).any(dim=0).sum(1).tolist()[:20], "by ch:", d.sum(dim=(0,1,2)).tolist())
