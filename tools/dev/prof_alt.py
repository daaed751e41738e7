"""alt_resnet [3,3,3,3] forward+backward on 256 tiles @256x256 (the bench's alt_resnet_path), for rocprofv3 --stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
torch.manual_seed(77)
net = mil_amd.alt_resnet.ResNet(mil_amd.alt_resnet.BasicBlock, [3, 3, 3, 3], num_classes=80, compute_dtype=torch.bfloat16).cuda()
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((256, 3, 256, 256), generator=g, device="cuda").clamp_(-1, 1)
dfe = torch.randn((256, 80), generator=g, device="cuda")
for _ in range(6):
    for p in net.parameters(): p.grad = None
    f = net(x); f.backward(dfe)
torch.cuda.synchronize()
