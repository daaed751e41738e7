"""Seeded INPUT builders shared by the fixture generators under tests/golden/ (which run only in the build container,
next to the reference) and by the tests that check against those fixtures (which run anywhere).  Nothing here touches
the reference: fixtures store expected outputs, these functions rebuild the inputs they were computed from."""
import numpy as np
import torch


def prep_inputs(z):
    """uint8 ROIs of a prep_* fixture (tests/golden/make_golden.py run_prep_case): noise, with a ramp in the green
    channel of every second tile (resampling bugs that noise hides show up on ramps)."""
    n, roi, seed = int(z["n_tiles"]), int(z["roi"]), int(z["seed"])
    rng = np.random.default_rng(seed)
    rois = rng.integers(0, 256, (n, roi, roi, 3), dtype=np.uint8)
    ramp = (np.add.outer(np.arange(roi), 2 * np.arange(roi)) % 256).astype(np.uint8)
    rois[::2, :, :, 1] = ramp
    return rois


def synth_bag(n, h, w, seed):
    """x = clamp(N(0,1), -1, 1) fp32 [n,3,h,w] from a seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, h, w, generator=g).clamp_(-1.0, 1.0)


def dla_inputs(seed, n):
    """(attn [n,3], activations [n,3], raster [n,2] = (row, col) pixel origins) of a tests/golden/dla fixture."""
    g = torch.Generator().manual_seed(seed)
    attn = torch.softmax(torch.randn(n, 3, generator=g) * 2, dim=0)              # columns sum to 1, like Aterm.t()
    activations = torch.randn(n, 3, generator=g) * attn                          # like wROIs.t()
    rng = np.random.default_rng(seed)
    raster = np.stack([rng.integers(0, 90, n) * 1200, rng.integers(0, 120, n) * 1200], axis=1)
    return attn, activations, raster


DLA_CASES = (("slideA", 11, 37), ("slideB", 12, 5))      # (name, seed, tiles); "slideC" is the constant-map case (seed 13, 4 tiles)
