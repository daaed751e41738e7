import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
dev = torch.device('cuda')
net = mil_amd.Attention(3).cuda().eval()
flat = mil_amd.FlatParams(net); opt = mil_amd.FlatAdam(flat)
x = torch.randn(8 * 256, 3, 256, 256, device=dev).clamp_(-1, 1)
sizes = [256] * 8
labels = torch.tensor([b % 3 for b in range(8)], device=dev)
def step():
    flat.zero_grad()
    outs = net.forward_bags((x, sizes), labels)
    outs.loss.sum().backward()
    flat.allreduce_grads(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
# phases
torch.cuda.synchronize(); t0 = time.perf_counter(); flat.zero_grad(); outs = net.forward_bags((x, sizes), labels); t1 = time.perf_counter(); outs.loss.sum().backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter(); torch.cuda.synchronize()
print(f"host: forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms, opt {1e3*(t3-t2):.2f} ms")
