import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
c = 80
g = torch.Generator(device="cuda").manual_seed(2)
w = torch.randn(c, c, 3, 3, device="cuda", generator=g) * 0.03
b = torch.randn(c, device="cuda", generator=g) * 0.1
pf, bp = ops.pack_weights(w, b, L.PACK_FWD, dt)
for n in (2, 8, 9, 17):
    x = torch.randn(n, 8, 8, c, device="cuda", generator=g).to(dt)
    r = torch.randn(n, 8, 8, c, device="cuda", generator=g).to(dt)
    for name, kw in (("plain", {}), ("res", {"res": r}), ("act", {"act": r}), ("res+act", {"res": r, "act": r})):
        y = ops.conv(x, pf, bp, c, ks=3, stride=1, pad=1, lrelu=True, **kw)
        bad = torch.isnan(y.float())
        print(n, name, "nan count", int(bad.sum()), "images with nan", sorted(set(torch.nonzero(bad)[:, 0].tolist()))[:10],
              "pixels", sorted(set((torch.nonzero(bad)[:, 1] * 8 + torch.nonzero(bad)[:, 2]).tolist()))[:12], "channels", sorted(set(torch.nonzero(bad)[:, 3].tolist()))[:12])
