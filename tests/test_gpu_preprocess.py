"""-m gpu: the device tile pre-processing (csrc/preprocess.hip through the C ABI) against outputs of Pillow itself
(tests/golden/prep_*.npz) and against the oracle restatement — bit-exact (integer resampling, exact fp32 normalisation)."""
import os

import numpy as np
import pytest
import torch

import mil_amd
from fixture_inputs import prep_inputs
from oracle import preprocess_oracle as po

pytestmark = pytest.mark.gpu
CASES = ["prep_s120_r32_train", "prep_s100_r37_flat", "prep_s50_r80_train", "prep_s1200_r300_train"]


@pytest.mark.parametrize("name", CASES)
def test_preprocess_matches_pillow_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rois = prep_inputs(z)
    train, res, pad, roi = bool(int(z["train"])), int(z["res"]), int(z["pad"]), int(z["roi"])
    prep = mil_amd.TilePreprocessor(roi, res, pad=pad)
    out = prep(torch.from_numpy(rois).cuda(), torch.from_numpy(z["params"]) if train else None).cpu().numpy()
    want = z["out"] if "out" in z.files else np.stack([po.to_tensor_normalize(u) for u in z["out_u8"]])
    assert out.shape == want.shape and out.dtype == np.float32
    assert np.array_equal(out, want)


def test_preprocess_matches_oracle_at_benchmark_sizes():
    """1200x1200 ROIs -> 256x256 tiles (BASELINE tile size), train and flat chains, several tiles per launch."""
    rng = np.random.default_rng(77)
    rois = rng.integers(0, 256, (3, 1200, 1200, 3), dtype=np.uint8)
    prep = mil_amd.TilePreprocessor(1200, 256, pad=100)
    params = torch.tensor([[0, 200, 1, 1], [200, 0, 0, 1], [37, 141, 1, 0]], dtype=torch.int32)
    got = prep(torch.from_numpy(rois).cuda(), params).cpu().numpy()
    for t in range(3):
        assert np.array_equal(got[t], po.finalize_tile(rois[t], 256, params[t].numpy(), pad=100)), t
    flat = prep(torch.from_numpy(rois[:1]).cuda()).cpu().numpy()
    assert np.array_equal(flat[0], po.finalize_tile(rois[0], 256))
    assert float(np.abs(got).max()) <= 1.0


def test_preprocess_feeds_the_encoder(golden_dir):
    """End to end: uint8 ROIs -> device finalisation -> Attention.forward (the call of gbm/classify_combined.py:432)."""
    rng = np.random.default_rng(5)
    rois = torch.from_numpy(rng.integers(0, 256, (6, 160, 160, 3), dtype=np.uint8)).cuda()
    prep = mil_amd.TilePreprocessor(160, 64, pad=20)
    tiles = prep(rois, prep.draw_params(6))
    assert tiles.shape == (6, 3, 64, 64) and tiles.is_cuda
    net = mil_amd.Attention(3).eval()
    out = net(tiles, torch.tensor([1]))
    assert torch.isfinite(out["loss"]) and abs(float(out["Aterm"].sum(1)[0]) - 1.0) < 1e-4


def test_preprocess_errors():
    prep = mil_amd.TilePreprocessor(64, 32, pad=8)
    with pytest.raises(ValueError):
        prep(torch.zeros((2, 60, 60, 3), dtype=torch.uint8).cuda())
    with pytest.raises(ValueError):
        prep(torch.zeros((2, 64, 64, 3), dtype=torch.uint8).cuda(), torch.tensor([[17, 0, 0, 0], [0, 0, 0, 0]]))
    with pytest.raises(RuntimeError):
        prep(torch.zeros((2, 64, 64, 3), dtype=torch.uint8))
