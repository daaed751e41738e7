import sys, torch, collections
sys.path.insert(0, '/root/repo')
import numpy as np
import mil_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
net = mil_amd.Attention(3).eval()
flat = mil_amd.FlatParams(net); opt = mil_amd.FlatAdam(flat)
x = torch.randn(8 * 32, 3, 128, 128, device=dev).clamp_(-1, 1)
sizes = [32] * 8
labels = torch.tensor([b % 3 for b in range(8)], device=dev)
def step():
    flat.zero_grad()
    outs = net.forward_bags((x, sizes), labels)
    outs.loss.sum().backward()
    flat.allreduce_grads(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        c[e.name[:70]] += 1
for k, v in c.most_common(25): print(v, k)
print("---- cpu ops")
c = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::"))
for k, v in c.most_common(30): print(v, k)
