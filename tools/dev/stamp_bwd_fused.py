"""Phase shares of conv_bwd_fused_kernel<24,...> from a -DMIL_STAMP build (see the macro block in conv_bwd_fused.hip):
   make -C <pkg>/csrc clean && make -C <pkg>/csrc -j16 EXTRA=-DMIL_STAMP && python tools/dev/stamp_bwd_fused.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mil_amd
from mil_amd import ops, _lib as L
dt = torch.bfloat16
n, h, c = 2048, 64, 24
g = torch.Generator(device="cuda").manual_seed(1)
def rnd():
    t = torch.randn(n, h, h, c, device="cuda", generator=g).to(dt); t[..., 20:] = 0; return t
dz, x, add = rnd(), rnd(), rnd()
w = torch.randn(20, 20, 3, 3, device="cuda", generator=g) * 0.05
wd, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dt)
need = ops.bwd_fused_workspace_bytes(n, h, h, 20, 20, 3, 1, dt)
ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device="cuda")
names = ["commit", "barrier1", "dgrad", "epilogue", "barrier2", "wgrad"]
for addend, mask in ((None, True), (add, True)):
    for _ in range(3):
        ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=addend, mask=mask, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv_bwd_fused(dz, wd, x, 20, 20, addend=addend, mask=mask, workspace=ws); e1.record(); torch.cuda.synchronize()
    raw = ws.view(torch.int64)
    # stamps sit behind the slab: grid*NW*8 u64 at the end of the workspace
    NW = 8
    tail = raw[-(need // 8):]           # whole ws as i64; take the last grid*NW*8 entries
    grid = 512
    st = raw[raw.numel() - grid * NW * 8:].view(grid, NW, 8).cpu().numpy().astype(np.float64)
    tiles = st[..., 6]
    ok = tiles > 0
    per_tile = st[..., :6] / np.maximum(tiles[..., None], 1)
    print(f"addend={'yes' if addend is not None else 'no'}: launch {e0.elapsed_time(e1)*1e3:.0f} us; tiles/wg mean {tiles[ok].mean():.1f}")
    tot = per_tile[ok].sum(-1).mean()
    cyc = st[..., :6].sum(-1)[ok]; rt = st[..., 7][ok]
    print(f"  in-kernel clock: {float((cyc / np.maximum(rt, 1)).mean()) * 100:.0f} MHz (sum of phase cycles / s_memrealtime ticks x 100 MHz); loop wall {float(rt.mean()) / 100:.0f} us")
    print("  cycles per tile per wave (mean over waves):", {k: round(float(per_tile[ok][:, i].mean())) for i, k in enumerate(names)}, "total", round(float(tot)))
    for wv in range(NW):
        sel = ok[:, wv]
        print(f"  wave {wv}:", {k: round(float(per_tile[:, wv][sel][:, i].mean())) for i, k in enumerate(names)})
