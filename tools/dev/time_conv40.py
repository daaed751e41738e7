import sys, torch
sys.path.insert(0, '/root/repo')
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
for (c, n, h) in [(40, 2048, 32), (24, 2048, 64)]:
    cr = 40 if c == 40 else 20
    x = torch.randn(n, h, h, c, device='cuda').to(dt); x[..., cr:] = 0
    r = torch.randn(n, h, h, c, device='cuda').to(dt); r[..., cr:] = 0
    w = torch.randn(cr, cr, 3, 3, device='cuda') * 0.05; b = torch.zeros(cr, device='cuda')
    pf, bp = ops.pack_weights(w, b, L.PACK_FWD, dt)
    pd, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dt)
    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    print(f"c={c}: fwd lrelu {t(lambda: ops.conv(x, pf, bp, c, ks=3, stride=1, pad=1, lrelu=True)):.1f} us | fwd res+lrelu {t(lambda: ops.conv(x, pf, bp, c, ks=3, stride=1, pad=1, res=r, lrelu=True)):.1f} us | dgrad act {t(lambda: ops.conv(x, pd, None, c, ks=3, stride=1, pad=1, act=r)):.1f} us | dgrad res+act {t(lambda: ops.conv(x, pd, None, c, ks=3, stride=1, pad=1, res=r, act=r)):.1f} us")
