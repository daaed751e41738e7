import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mil_amd
from mil_amd import encoder
from oracle import mil_oracle as orc
g = np.load('tests/golden/eval_n8_64.npz'); w = np.load('tests/golden/weights.npz')
x = torch.tensor(g['x'])
sd = orc.load_state(w, requires_grad=True)
acts = {}
feats_ref = orc.backbone(sd, x, acts=acts, emulate_bf16=True)
net = mil_amd.Attention(3, compute_dtype=torch.bfloat16).eval()
net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
enc = net.cnn.module
with torch.no_grad():
    feats, saved = encoder.encoder_forward(enc, x.cuda(), torch.bfloat16)
def cmp(name, mine, ref, c):
    m = mine[..., :c].float().permute(0,3,1,2).cpu()
    r = ref.detach()
    d = (m - r).abs()
    print(f"{name}: max|ref|={r.abs().max():.3f} maxdiff={d.max():.4g} frac_mismatch={(d>0).float().mean():.4g} meanabs diff={d.mean():.3g}")
cmp('stem', saved['stem'], acts['stem'], 20)
for li, bi in ((1,2),(2,5),(3,8),(4,11)):
    cmp(f'layer{li}', saved['blocks'][bi][2], acts[f'layer{li}'], (20,40,60,80)[li-1])
print('feats rel', float((feats.cpu()-feats_ref.detach()).abs().max()/feats_ref.abs().max()))
