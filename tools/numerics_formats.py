"""CPU experiment: which storage / arithmetic format lets the encoder meet 1e-3 on the logits?

Every candidate is emulated with fp32 torch convs on the CPU; errors are reported against an fp64 run of the same
arithmetic on the same weights and inputs.  Run:  python tools/numerics_formats.py [tiles]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mil_oracle as O          # noqa: E402
from fixture_inputs import synth_bag        # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def split(x):
    hi = bf(x)
    return hi, bf(x - hi)


def r_hilo(x):                  # a tensor stored as a bf16 hi + bf16 lo pair
    hi, lo = split(x)
    return hi + lo


def r_hi_lo8_fixed(x):         # 3 bytes: bf16 hi + int8 lo in units of 2^(exponent(hi) - 15): covers the half-ulp of hi, 15-16 significant bits in all
    hi = bf(x)
    e = torch.floor(torch.log2(hi.abs().clamp_min(1e-38)))
    unit = torch.exp2(e - 15)
    lo8 = torch.clamp(torch.round((x - hi) / unit), -128, 127)
    return hi + lo8 * unit


def r_hi_lo8_float(x):         # 3 bytes: bf16 hi + an 8-bit float lo (1 sign, 4 exponent, 3 mantissa bits; exponent relative to hi's)
    hi = bf(x)
    lo = x - hi
    m, e = torch.frexp(lo)
    return hi + torch.ldexp(torch.round(m * 16) / 16, e)


def r_fp16(x):
    return x.to(torch.float16).to(x.dtype)


class Fmt:
    """store(t): rounding applied where a tensor is written to HBM; conv(x, w, ...): the contraction."""
    def __init__(self, name, store, conv):
        self.name, self.store, self.conv = name, store, conv


def conv_exact(x, w, b, **kw):
    return F.conv2d(x, w, b, **kw)


def conv_bf16(x, w, b, **kw):
    return F.conv2d(bf(x), bf(w), b, **kw)


def conv_split3(x, w, b, **kw):           # hi*hi + lo*hi + hi*lo, fp32 accumulation
    xh, xl = split(x)
    wh, wl = split(w)
    return F.conv2d(xh, wh, b, **kw) + F.conv2d(xl, wh, None, **kw) + F.conv2d(xh, wl, None, **kw)


def conv_split4(x, w, b, **kw):
    xh, xl = split(x)
    wh, wl = split(w)
    return F.conv2d(xh, wh, b, **kw) + F.conv2d(xl, wh, None, **kw) + F.conv2d(xh, wl, None, **kw) + F.conv2d(xl, wl, None, **kw)


def ident(x):
    return x


FORMATS = [
    Fmt("fp32 store, fp32 mul (today's exact path)", ident, conv_exact),
    Fmt("bf16 store, bf16 mul (today's fast path)", bf, conv_bf16),
    Fmt("fp16 store, exact mul", r_fp16, conv_exact),
    Fmt("bf16 hi+lo store, exact mul", r_hilo, conv_exact),
    Fmt("fp32 store, bf16x3 split mul", ident, conv_split3),
    Fmt("fp32 store, bf16x4 split mul", ident, conv_split4),
    Fmt("bf16 hi+lo store, bf16x3 split mul", r_hilo, conv_split3),
    Fmt("fp32 block in/out, bf16 mid, bf16x3 mul", "mid", conv_split3),
    Fmt("3-byte store (hi + int8 lo), bf16x3 mul", r_hi_lo8_fixed, conv_split3),
    Fmt("3-byte store (hi + e4m3-like lo), bf16x3 mul", r_hi_lo8_float, conv_split3),
]


def backbone(sd, x, fmt, p="cnn.module."):
    st = fmt.store if callable(fmt.store) else ident
    mid = bf if fmt.store == "mid" else st
    t = F.leaky_relu(fmt.conv(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=2, padding=3), O.LEAK)
    t = st(F.max_pool2d(st(t), 3, 2, 1))
    for li, _pl, stride in O.STAGES:
        for b in range(O.BLOCKS_PER_STAGE):
            q = f"{p}layer{li}.{b}."
            s = stride if b == 0 else 1
            o = mid(F.leaky_relu(fmt.conv(t, sd[q + "conv1.weight"], sd[q + "conv1.bias"], stride=s, padding=1), O.LEAK))
            o = fmt.conv(o, sd[q + "conv2.weight"], sd[q + "conv2.bias"], stride=1, padding=1)
            key = q + "downsample.0.weight"
            short = st(fmt.conv(t, sd[key], None, stride=s)) if key in sd else t
            t = st(F.leaky_relu(o + short, O.LEAK))
    t = t.mean(dim=(2, 3))
    return t @ sd[p + "fc.weight"].t()


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.set_num_threads(8)
    z = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    sd = O.load_state(z)
    x = synth_bag(tiles, 256, 256, 20260104)
    label = torch.tensor([0])
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        ref = O.mil_head(sd64, backbone(sd64, x.double(), FORMATS[0]), label)
        print(f"tiles={tiles}  |Mterm| = {ref['Mterm'].abs().max():.3f}  |Bterm| = {ref['Bterm'].abs().max():.3f}  |F| = {ref['Fterm'].abs().max():.1f}")
        print(f"{'format':48s} {'Mterm':>9s} {'Aterm':>9s} {'Aterm rel':>9s} {'y_pred':>9s} {'loss':>9s} {'Bterm':>9s} {'Fterm rel':>9s}")
        for fmt in FORMATS:
            out = O.mil_head(sd, backbone(sd, x, fmt), label)
            d = lambda k: float((out[k].double() - ref[k]).abs().max())
            print(f"{fmt.name:48s} {d('Mterm'):9.2e} {d('Aterm'):9.2e} {d('Aterm') / float(ref['Aterm'].abs().max()):9.2e} {d('y_pred'):9.2e} "
                  f"{d('loss'):9.2e} {d('Bterm'):9.2e} {d('Fterm') / float(ref['Fterm'].abs().max()):9.2e}")


if __name__ == "__main__":
    main()
