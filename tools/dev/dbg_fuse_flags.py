"""Which fused forward kernel is not bit-identical to the un-fused sequence, where (block, image), by how much."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MIL_PF_MIN_TILES"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import mil_amd
from mil_amd import encoder
w = np.load("tests/golden/weights.npz")
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ntiles = int(sys.argv[3]) if len(sys.argv) > 3 else 96
gen = torch.Generator(device="cuda").manual_seed(2024)
x = torch.randn((ntiles, 3, size, size), generator=gen, device="cuda").clamp_(-1, 1)

def run(**flags):
    net = mil_amd.Attention(3, compute_dtype=torch.bfloat16).eval()
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
    enc = net.cnn.module
    for k in ("fuse_backward", "fuse_stem_forward", "fuse_stage_entry", "fuse_block_forward"):
        setattr(enc, k, flags.get(k, False))
    with torch.no_grad():
        feats, saved = encoder.encoder_forward(enc, x, torch.bfloat16)
    torch.cuda.synchronize()
    return feats, saved

base_f, base_s = run()
for flag in ("fuse_stem_forward", "fuse_stage_entry", "fuse_block_forward"):
    f, s = run(**{flag: True})
    print(flag, "feats equal:", torch.equal(f, base_f))
    if not torch.equal(s["xs"], base_s["xs"]): print("   xs differs")
    if not torch.equal(s["widx"], base_s["widx"]): print("   widx differs")
    for bi, (a, b) in enumerate(zip(s["blocks"], base_s["blocks"])):
        for name, ta, tb in zip(("in", "o1", "out"), a, b):
            if not torch.equal(ta, tb):
                d = (ta.float() - tb.float()).abs()
                imgs = torch.nonzero(d.flatten(1).max(1).values > 0).flatten().tolist()
                nbad = int((d > 0).sum())
                print(f"   block {bi} {name}: shape {tuple(ta.shape)} differing elems {nbad} max {float(d.max()):.4g} (ref max {float(tb.float().abs().max()):.4g}) images {imgs[:20]}{'...' if len(imgs) > 20 else ''}")
                if name == "out" or name == "o1":
                    i0 = imgs[0]
                    dd = d[i0]
                    ys, xs_, cs = torch.nonzero(dd > 0, as_tuple=True)
                    print(f"      image {i0}: rows {sorted(set(ys.tolist()))[:12]} cols {sorted(set(xs_.tolist()))[:12]} chans {sorted(set(cs.tolist()))[:12]}")
                break
        else:
            continue
        break
