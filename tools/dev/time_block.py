"""Times the identity-block forward alone (2048 images of 64x64x20 by default), tiled form against the row-walk form, in split
precision and in bf16: `python tools/dev/time_block.py [n] [h]`.  With MIL_LIB_PATH=<stamp build> the kernels print their phase shares."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mil_amd  # noqa: E402,F401
from mil_amd import _lib as L, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
h = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.zeros((n, h, 64, 24), device="cuda")
x[..., :20] = torch.randn((n, h, 64, 20), generator=g, device="cuda")
w1 = torch.randn((20, 20, 3, 3), generator=g, device="cuda") / 13.4
w2 = torch.randn((20, 20, 3, 3), generator=g, device="cuda") / 13.4
b = torch.randn((20,), generator=g, device="cuda") * 0.1
reps = 3 if "stamp" in os.environ.get("MIL_LIB_PATH", "") else 20
for mode, dt, code in (("x3", torch.float32, L.MIL_DT_F32S), ("bf16", torch.bfloat16, L.MIL_DT_F32)):
    with L.f32_mma(code):
        xx = x.to(dt)
        p1, bp1 = ops.pack_weights(w1, b, L.PACK_FWD, dt)
        p2, bp2 = ops.pack_weights(w2, b, L.PACK_FWD, dt)
        outs = {}
        for form in ("0", "1", "0", "1"):
            os.environ["MIL_BLOCK_STRIP"] = form
            for _ in range(3):
                outs[form] = ops.conv_block_fwd(xx, p1, bp1, p2, bp2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv_block_fwd(xx, p1, bp1, p2, bp2)
            e1.record()
            torch.cuda.synchronize()
            print(f"block_fwd {mode} {'row walk' if form == '1' else 'tiled   '}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch ({n} images {h}x64)", flush=True)
        print(mode, "bit-identical:", all(torch.equal(a, b_) for a, b_ in zip(outs["0"], outs["1"])))
