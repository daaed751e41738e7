import sys, torch, time
sys.path.insert(0, '/root/repo')
import mil_amd
T = 512
rois = torch.randint(0, 256, (T, 1200, 1200, 3), dtype=torch.uint8, device='cuda')
for res in (256, 300):
    prep = mil_amd.TilePreprocessor(1200, res, pad=100)
    p = prep.draw_params(T)
    out = prep(rois, p); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = prep(rois, p)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"res {res}: {ms:.3f} ms for {T} tiles -> {T/ms*1e3:.0f} tiles/s, input {rois.numel()/ms/1e6:.0f} GB/s")
