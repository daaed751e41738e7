"""Forward + backward time of the MIL head alone (segmented kernels of mil_head.hip) on the benchmark's batch: 8 bags x 256
instances, and on one 4096-instance bag."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd.head import BagLayout, head_apply
torch.manual_seed(3)
net = mil_amd.Attention(3).cuda()
ws = net.head_weights()
for sizes in ([256] * 8, [4096]):
    lay = BagLayout(sizes, "cuda")
    n = sum(sizes)
    H = torch.randn(n, 80, device="cuda").requires_grad_(True)
    y = torch.zeros(len(sizes), dtype=torch.int64, device="cuda")
    keep = (torch.rand(n, 80, device="cuda") > 0.25).to(torch.uint8)
    def step():
        H.grad = None
        out = head_apply(H, lay, y, keep, None, ws)
        out[0].sum().backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): step()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"bags {len(sizes)} x {sizes[0]}: head fwd+bwd {min(ts):.1f} us (incl. launch gaps), dH checksum {float(H.grad.abs().sum()):.6f}")
