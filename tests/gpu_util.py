"""Helpers shared by the -m gpu parity tests (layout conversion between the reference's NCHW fp32
tensors and the kernels' channel-padded NHWC tensors)."""
import torch

X3 = "bf16x3"        # compute mode of the split-precision path: fp32 tensors, three bf16 MFMAs per k-step (mil_amd._lib.BF16X3)


def storage(dtype):
    return torch.float32 if dtype == X3 else dtype


def cpad(c):
    return (c + 7) // 8 * 8


def to_nhwc(x_nchw, dtype, device="cuda"):
    n, c, h, w = x_nchw.shape
    out = torch.zeros((n, h, w, cpad(c)), dtype=torch.float32)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return out.to(device=device, dtype=storage(dtype)).contiguous()


def from_nhwc(y, c):
    return y[..., :c].float().cpu().permute(0, 3, 1, 2).contiguous()


def round_to(x, dtype):
    """What the kernel sees after the operand is stored in `dtype`."""
    return x.to(storage(dtype)).float()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
