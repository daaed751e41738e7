"""Development check: the split-precision encoder with the row-walk block forward on / off (MIL_BLOCK_STRIP), per block launch."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import mil_amd
from mil_amd import ops
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(n, 3, 256, 256, device="cuda").clamp_(-1, 1)
net = mil_amd.Attention(3, compute_dtype=mil_amd.BF16X3).cuda().eval()
orig = ops.conv_block_fwd
calls = []
def both(xx, *a, **k):
    os.environ["MIL_BLOCK_STRIP"] = "0"
    r0 = orig(xx, *a, **k)
    os.environ["MIL_BLOCK_STRIP"] = "1"
    r1 = orig(xx, *a, **k)
    torch.cuda.synchronize()
    if r0 is not None:
        for name, t0, t1 in zip(("o1", "y"), r0, r1):
            bad = (t0 != t1) | ~torch.isfinite(t1)
            if bool(bad.any()):
                idx = bad.nonzero()
                imgs = idx[:, 0].unique()
                print(f"call {len(calls)} {name} shape {tuple(t0.shape)}: {int(bad.sum())} bad elements in {len(imgs)} images; first {idx[0].tolist()} last {idx[-1].tolist()}; "
                      f"images {imgs[:8].tolist()}..{imgs[-4:].tolist()} rows {idx[:,1].unique()[:10].tolist()} cols {idx[:,2].unique()[:10].tolist()} ch {idx[:,3].unique().tolist()}")
                print("  input finite:", bool(torch.isfinite(xx).all()), "pad channels max", float(xx[..., 20:].abs().max()))
            else:
                print(f"call {len(calls)} {name}: identical")
    calls.append(1)
    return r0
ops.conv_block_fwd = both
with torch.no_grad():
    h = net.cnn(x)
print("finite:", bool(torch.isfinite(h).all()))
