"""Diagnostic (GPU box): per residual block, forward error and LeakyReLU-mask sign flips of the HIP fp32 / bf16x3 encoder
and of the fp32 CPU oracle, both against an fp64 run of the oracle.  Explains where gradient error enters.
    python tools/diag_mask_flips.py [tiles]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_inputs import synth_bag        # noqa: E402
from oracle import mil_oracle as orc        # noqa: E402
import mil_amd                              # noqa: E402
from mil_amd import _lib as L               # noqa: E402
from mil_amd import encoder                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.set_num_threads(min(64, os.cpu_count() or 1))
w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
x = synth_bag(n, 256, 256, 20260104)
sd32 = orc.load_state(w)
sd64 = {k: v.double() for k, v in sd32.items()}
a32, a64 = {}, {}
with torch.no_grad():
    orc.backbone(sd32, x, a32)
    orc.backbone(sd64, x.double(), a64)
names = [f"layer{li}.{b}" for li in range(1, 5) for b in range(3)]
for mode in (torch.float32, mil_amd.BF16X3):
    net = mil_amd.Attention(3, compute_dtype=mode).eval()
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
    with torch.no_grad(), L.f32_mma(L.mma_code(mode)):
        _f, saved = encoder.encoder_forward(net.cnn.module, x.cuda(), torch.float32)
    print(f"--- HIP {mode}: block output vs fp64: max-rel err, sign flips (of elements) | fp32 CPU oracle: same")
    for i, name in enumerate(names):
        ref = a64[name]
        c = ref.shape[1]
        hip = saved["blocks"][i][2][..., :c].permute(0, 3, 1, 2).double().cpu()
        o32 = a32[name].double()
        sc = float(ref.abs().max())
        fh = int(((hip > 0) != (ref > 0)).sum())
        fo = int(((o32 > 0) != (ref > 0)).sum())
        print(f"{name}: hip {float((hip - ref).abs().max()) / sc:.2e} flips {fh:5d} / {ref.numel()}   oracle32 {float((o32 - ref).abs().max()) / sc:.2e} flips {fo:5d}"
              f"   small |v|<1e-4*max: {int((ref.abs() < 1e-4 * sc).sum())}")
