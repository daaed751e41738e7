import sys, torch, collections
sys.path.insert(0, '/root/repo')
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::add_", "aten::zero_", "aten::clone", "aten::_to_copy", "aten::zeros", "aten::empty") and e.stack:
        fr = [s for s in e.stack if "/root/repo" in s or "mil_amd" in s]
        c[(e.name, fr[0] if fr else e.stack[0])] += 1
for (n, s), v in sorted(c.items(), key=lambda kv: -kv[1])[:40]: print(v, n, s[-110:])
