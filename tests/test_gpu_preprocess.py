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


def _s2d_of(tiles):
    """fp32 [T,3,H,W] -> the bf16 space-to-depth tensor [T,H/2,W/2,16] (channel = c*4 + dy*2 + dx, 12 real)."""
    t, _c, h, w = tiles.shape
    v = tiles.view(t, 3, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(t, h // 2, w // 2, 12)
    out = torch.zeros((t, h // 2, w // 2, 16), dtype=torch.bfloat16, device=tiles.device)
    out[..., :12] = v.to(torch.bfloat16)
    return out


@pytest.mark.parametrize("name", ["prep_s120_r32_train", "prep_s50_r80_train", "prep_s1200_r300_train"])
def test_preprocess_s2d_output_is_the_bf16_rounding_of_the_tile_stack(golden_dir, name):
    """mil_tile_preprocess_s2d writes the same tiles (Pillow-exact, flips included) as the bf16 space-to-depth tensor the
    stem kernels read: bit for bit the bf16 rounding of the fp32 stack, padding channels zero."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rois = torch.from_numpy(prep_inputs(z)).cuda()
    res, pad, roi = int(z["res"]), int(z["pad"]), int(z["roi"])
    prep = mil_amd.TilePreprocessor(roi, res, pad=pad)
    params = torch.from_numpy(z["params"])
    tiles = prep(rois, params)
    s2d = prep(rois, params, out="s2d")
    assert isinstance(s2d, mil_amd.S2dTiles) and tuple(s2d.shape) == tuple(tiles.shape)
    assert torch.equal(s2d.xs.view(torch.int16), _s2d_of(tiles).view(torch.int16))
    flat = prep(rois[:2], None, out="s2d")
    assert torch.equal(flat.xs.view(torch.int16), _s2d_of(prep(rois[:2])).view(torch.int16))


def test_s2d_feed_is_bit_identical_to_the_fp32_tensor_api(golden_dir, monkeypatch):
    """uint8 ROIs -> TilePreprocessor(out="s2d") -> Attention.forward_bags on the S2dTiles handle: the fused stem reads the
    space-to-depth records directly (no fp32 stack ever exists).  Every output and every gradient must equal, bit for bit,
    the run on the fp32 [T,3,R,R] stack (the bf16 stem rounds its input to exactly these values), in eval mode over ragged
    bags and in train mode with injected subsample indices; and the chain agrees with the CPU oracles end to end
    (preprocess_oracle -> mil_oracle) within the bf16 path's tolerance."""
    from oracle import mil_oracle as orc
    monkeypatch.setenv("MIL_PF_MIN_TILES", "1")
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    rng = np.random.default_rng(11)
    n, roi, res, pad = 24, 300, 128, 30
    rois_np = rng.integers(0, 256, (n, roi, roi, 3), dtype=np.uint8)
    rois = torch.from_numpy(rois_np).cuda()
    prep = mil_amd.TilePreprocessor(roi, res, pad=pad)
    params = prep.draw_params(n, torch.Generator().manual_seed(2))
    tiles, s2d = prep(rois, params), prep(rois, params, out="s2d")
    sizes, labels = [14, 10], torch.tensor([2, 0])
    runs = []
    for feed in (tiles, s2d):
        net = mil_amd.Attention(3, compute_dtype=torch.bfloat16).eval()
        net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
        outs = net.forward_bags((feed, sizes), labels)
        outs.loss.sum().backward()
        torch.cuda.synchronize()
        runs.append(([{k: v.detach().clone() for k, v in o.items()} for o in outs],
                     {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    for oa, ob in zip(runs[0][0], runs[1][0]):
        for k in oa:
            assert torch.equal(oa[k], ob[k]), k
    for k, g in runs[0][1].items():
        assert torch.equal(g, runs[1][1][k]), k
    # one bag through the reference-style call, S2dTiles handle
    net = mil_amd.Attention(3, compute_dtype=torch.bfloat16).eval()
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()})
    one = net(s2d[:14], torch.tensor([2]))
    assert torch.equal(one["Aterm"], runs[0][0][0]["Aterm"])
    # train mode: per-bag subsample of an S2dTiles bag
    net.train()
    net.rng_override = {"indices": torch.tensor([3, 0, 7, 9]), "keep_mask": torch.ones(4, 80, dtype=torch.uint8)}
    tr_s = net(s2d[:14], torch.tensor([2]))
    tr_f = net(tiles[:14], torch.tensor([2]))
    assert torch.equal(tr_s["Aterm"], tr_f["Aterm"]) and torch.equal(tr_s["loss"], tr_f["loss"])
    # end to end against the CPU oracles (bf16 path tolerance: attention weights)
    x_ref = torch.from_numpy(np.stack([po.finalize_tile(rois_np[t], res, params[t].numpy(), pad=pad) for t in range(14)]))
    assert np.array_equal(tiles[:14].cpu().numpy(), x_ref.numpy())
    sd = orc.load_state(w)
    with torch.no_grad():
        ref = orc.attention_forward(sd, x_ref, torch.tensor([2]))
    a = runs[1][0][0]["Aterm"].cpu()
    assert float((a - ref["Aterm"]).abs().max()) < 2e-2 and int(runs[1][0][0]["y_pred_hat"]) in (0, 1, 2)
    # the fp32 compute modes refuse the handle
    net32 = mil_amd.Attention(3, compute_dtype=torch.float32).eval()
    with pytest.raises(ValueError):
        net32(s2d[:14], torch.tensor([2]))
