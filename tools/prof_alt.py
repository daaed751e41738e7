"""alt_resnet.ResNet(BasicBlock,[3,3,3,3]) forward+backward on 256 tiles @256x256 (the bench's alt_resnet_path step), for
rocprofv3:  rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_alt.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(77)
net = mil_amd.alt_resnet.ResNet(mil_amd.alt_resnet.BasicBlock, [3, 3, 3, 3], num_classes=80, compute_dtype=torch.bfloat16).cuda()
gen = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((256, 3, 256, 256), generator=gen, device="cuda").clamp_(-1.0, 1.0)
dfe = torch.randn((256, 80), generator=gen, device="cuda")
for _ in range(steps):
    for p in net.parameters():
        p.grad = None
    net(x).backward(dfe)
torch.cuda.synchronize()
