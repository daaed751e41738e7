import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MIL_PF_MIN_TILES"] = "1"
import mil_amd
w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
gen = torch.Generator(device="cuda").manual_seed(2024)
x = torch.randn((96, 3, 128, 128), generator=gen, device="cuda").clamp_(-1, 1)
sizes, labels = [40, 30, 26], torch.tensor([0, 1, 2], device="cuda")
res = {}
for name, dt, fused in (("bf16-fused", torch.bfloat16, True), ("bf16-unfused", torch.bfloat16, False), ("fp32", torch.float32, True)):
    net = mil_amd.Attention(3, compute_dtype=dt).eval()
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
    enc = net.cnn.module
    enc.fuse_backward = fused
    outs = net.forward_bags((x, sizes), labels)
    outs.loss.sum().backward()
    torch.cuda.synchronize()
    res[name] = {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()}
for k in ("cnn.module.conv1.bias", "cnn.module.conv1.weight", "cnn.module.layer1.0.conv1.bias"):
    a, b, c = res["bf16-fused"][k], res["bf16-unfused"][k], res["fp32"][k]
    s = float(c.abs().max())
    print(k, "scale(fp32)", s, "fused-unfused", float((a - b).abs().max()) / float(b.abs().max()), "fused-fp32", float((a - c).abs().max()) / s, "unfused-fp32", float((b - c).abs().max()) / s)
    if a.numel() <= 24:
        print(" fused  ", a.numpy().round(4)); print(" unfused", b.numpy().round(4)); print(" fp32   ", c.numpy().round(4))
