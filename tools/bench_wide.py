"""Per-launch timing of the wide-layer kernels (csrc/conv_wide.hip) on the shapes of an alt_resnet [3,3,3,3] step at 256 tiles
of 256x256: 3x3 stride-1 forward / data gradient and the weight gradient at 128 ch @32x32, 256 @16x16, 512 @8x8, plus the
stride-2 entries.  Prints TFLOP/s per shape; `--check` compares each output with torch's conv on the CPU for a small batch.

    python tools/bench_wide.py [--iters 20] [--n 256] [--check] [--only conv|wgrad|s2]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa: E402
from mil_amd import _lib as L, ops  # noqa: E402

SHAPES = [(128, 128, 32, 1), (256, 256, 16, 1), (512, 512, 8, 1), (64, 128, 64, 2), (128, 256, 32, 2), (256, 512, 16, 2)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3       # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--gconv", action="store_true", help="forward / data gradient on the gather-GEMM kernel (mil_gconv)")
    a = ap.parse_args()
    dt = torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(3)
    for cin, cout, hw, stride in SHAPES:
        n = a.n
        ho = hw // stride
        x = torch.randn((n, hw, hw, cin), generator=gen, device="cuda").to(dt)
        w = torch.randn((cout, cin, 3, 3), generator=gen, device="cuda") * (2.0 / (9 * cin)) ** 0.5
        dz = torch.randn((n, ho, ho, cout), generator=gen, device="cuda").to(dt)
        if a.gconv and not ops.gconv_supported(cin, cout, 3, stride):
            continue
        if a.gconv:
            wf = ops.gconv_pack_weights(w, L.PACK_FWD)
            wb = ops.gconv_pack_weights(w, L.PACK_DGRAD) if ops.gconv_supported(cout, cin, 3, stride) else None

            def fwd(xx, relu=True):
                return ops.gconv(xx, wf, cout, ks=3, stride=stride, pad=1, relu=relu)

            def dgrad(zz):
                return ops.gconv(zz, wb, cin, ks=3, stride=stride, pad=1, transposed=True, out_hw=(hw, hw))
            if not ops.gconv_supported(cout, cin, 3, stride):          # e.g. the 128 -> 64 channel gradient: a 64-wide output block
                wb = ops.wide_pack_weights(w, L.PACK_DGRAD, dt)

                def dgrad(zz):
                    return ops.wide_conv(zz, wb, cin, ks=3, stride=1, pad=1, zero_insert=stride == 2, out_hw=(hw, hw))
        else:
            wf = ops.wide_pack_weights(w, L.PACK_FWD, dt)
            wb = ops.wide_pack_weights(w, L.PACK_DGRAD, dt)

            def fwd(xx, relu=True):
                return ops.wide_conv(xx, wf, cout, ks=3, stride=stride, pad=1, relu=relu)

            def dgrad(zz):
                return ops.wide_conv(zz, wb, cin, ks=3, stride=1, pad=1, zero_insert=stride == 2, out_hw=(hw, hw))
        flops = 2.0 * n * ho * ho * cout * cin * 9
        tag = f"{cin:3d}->{cout:3d} @{hw:2d} s{stride}"
        if a.only in ("", "conv") or (a.only == "s2" and stride == 2):
            t = timed(lambda: fwd(x), a.iters)
            print(f"{tag} forward   {t:8.1f} us  {flops / t * 1e-6:7.1f} TFLOP/s", flush=True)
            t = timed(lambda: dgrad(dz), a.iters)
            print(f"{tag} dgrad     {t:8.1f} us  {flops / t * 1e-6:7.1f} TFLOP/s", flush=True)
        if a.only in ("", "wgrad") or (a.only == "s2" and stride == 2):
            ws = [None]

            def wg():
                _, ws[0] = ops.wide_wgrad(x, dz, cin, cout, ks=3, stride=stride, pad=1, workspace=ws[0])
            t = timed(wg, a.iters)
            print(f"{tag} wgrad     {t:8.1f} us  {flops / t * 1e-6:7.1f} TFLOP/s", flush=True)
        if a.check:
            m = 6
            xs, dzs = x[:m].contiguous(), dz[:m].contiguous()
            y = fwd(xs, relu=False).float().cpu()
            wq = w.to(dt).float().cpu()
            ref = F.conv2d(xs.float().cpu().permute(0, 3, 1, 2), wq, stride=stride, padding=1).permute(0, 2, 3, 1)
            e_f = float((y - ref).abs().max() / ref.abs().max())
            dx = dgrad(dzs).float().cpu()
            refd = F.conv_transpose2d(dzs.float().cpu().permute(0, 3, 1, 2), wq, stride=stride, padding=1,
                                      output_padding=stride - 1).permute(0, 2, 3, 1)
            e_d = float((dx - refd).abs().max() / refd.abs().max())
            dw, _ = ops.wide_wgrad(xs, dzs, cin, cout, ks=3, stride=stride, pad=1)
            refw = torch.nn.grad.conv2d_weight(xs.float().cpu().permute(0, 3, 1, 2), (cout, cin, 3, 3),
                                               dzs.float().cpu().permute(0, 3, 1, 2), stride=stride, padding=1)
            e_w = float((dw.cpu() - refw).abs().max() / refw.abs().max())
            print(f"{tag} check: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e}", flush=True)
            assert e_f < 1e-2 and e_d < 1e-2 and e_w < 1e-3, "mismatch"


if __name__ == "__main__":
    main()
