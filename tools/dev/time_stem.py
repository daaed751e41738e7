"""Times the fused stem forward / backward alone (2048 tiles @256x256 by default): `python tools/dev/time_stem.py [bf16|bf16x3] [n] [size]`.
With MIL_LIB_PATH=<stamp build> the instrumented kernels print their phase shares."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mil_amd  # noqa: E402
from mil_amd import _lib as L, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = torch.bfloat16 if mode == "bf16" else torch.float32
code = L.MIL_DT_F32S if mode == "bf16x3" else L.MIL_DT_F32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((n, 3, size, size), generator=g, device="cuda").clamp_(-1, 1)
wt = torch.randn((20, 3, 7, 7), generator=g, device="cuda") * 0.08
b = torch.randn((20,), generator=g, device="cuda") * 0.1
with L.f32_mma(code):
    wp, bp = ops.pack_weights(wt, b, L.PACK_STEM, dt)

    def fwd():
        return ops.stem_fwd_fused(x, wp, bp, 24, dtype=dt, keep_s2d=False)

    _xs, pool, widx = fwd()
    gp = torch.randn(pool.shape[:3] + (20,), generator=g, device="cuda").to(dt)

    def bwd():
        return ops.stem_bwd_fused_nchw(x, gp, widx)

    for name, fn in (("stem_fwd", fwd), ("stem_bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = 3 if "stamp" in os.environ.get("MIL_LIB_PATH", "") else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{mode} {name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch ({n} tiles @{size})", flush=True)
