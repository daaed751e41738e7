import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mil_amd
from mil_amd import ops, _lib as L
from oracle import mil_oracle as orc
orig = ops.wide_conv
def wc(x, wp, cout, **kw):
    print("wide_conv", tuple(x.shape), x.dtype, cout, {k: (tuple(v.shape) if hasattr(v, 'shape') else v) for k, v in kw.items()}, flush=True)
    return orig(x, wp, cout, **kw)
ops.wide_conv = wc
z = np.load('tests/golden/alt_l1111_n4_64.npz')
net = mil_amd.alt_resnet.ResNet(layers=[1,1,1,1], num_classes=80, compute_dtype=torch.float32)
net.load_state_dict(orc.alt_seeded_state((1,1,1,1), 80, 555)); net.cuda()
f = net(torch.from_numpy(z['x']).cuda())
print((f.cpu() - torch.from_numpy(z['feats'])).abs().max())
