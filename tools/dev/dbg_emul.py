import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from oracle import mil_oracle as orc
g = np.load('tests/golden/eval_n8_64.npz'); w = np.load('tests/golden/weights.npz')
x = torch.tensor(g['x']); dfe = torch.randn(8, 80, generator=torch.Generator().manual_seed(3))
res = {}
for emu in (False, True):
    sd = orc.load_state(w, requires_grad=True)
    f = orc.backbone(sd, x, emulate_bf16=emu); f.backward(dfe)
    res[emu] = (f.detach(), {k: v.grad.clone() for k, v in sd.items() if v.grad is not None})
f0, g0 = res[False]; f1, g1 = res[True]
print('feats rel', float((f0-f1).abs().max()/f0.abs().max()))
errs = {k: float((g0[k]-g1[k]).abs().max()/g0[k].abs().max()) for k in g0}
cos = {k: float(torch.nn.functional.cosine_similarity(g0[k].flatten(), g1[k].flatten(), dim=0)) for k in g0}
print('grad relmax: median', np.median(list(errs.values())), 'worst', max(errs.items(), key=lambda kv: kv[1]))
print('grad cosine: median', np.median(list(cos.values())), 'worst', min(cos.items(), key=lambda kv: kv[1]))
