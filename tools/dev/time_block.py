import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
ops.BLOCK_FWD_CHANNELS = (24, 40)
for (c, n, h) in [(24, 2048, 64), (40, 2048, 32)]:
    x = torch.randn(n, h, h, c, device='cuda').to(dt)
    x[..., (20 if c == 24 else 40):] = 0
    cr = 20 if c == 24 else 40
    w1 = torch.randn(cr, cr, 3, 3, device='cuda') * 0.05; b = torch.zeros(cr, device='cuda')
    p1, bp1 = ops.pack_weights(w1, b, L.PACK_FWD, dt)
    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tb = t(lambda: ops.conv_block_fwd(x, p1, bp1, p1, bp1))
    def two():
        z = ops.conv(x, p1, bp1, c, ks=3, stride=1, pad=1, lrelu=True)
        return ops.conv(z, p1, bp1, c, ks=3, stride=1, pad=1, res=x, lrelu=True)
    t2 = t(two)
    print(f"c={c}: block {tb:.1f} us, two launches {t2:.1f} us (MIL_BLOCK_WAVES={os.environ.get('MIL_BLOCK_WAVES','8')})")
