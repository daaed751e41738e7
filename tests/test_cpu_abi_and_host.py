"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/mil_hip.h declares (no
compute calls without a GPU), host-side logic (module surface, state-dict layout, error behaviour,
shape/argument checks that must fire before a kernel is launched)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mil_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mil_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import mil_amd
    assert os.path.exists(mil_amd.LIB_PATH), "libmil_hip.so missing: run __graft_entry__.build()"
    handle = ctypes.CDLL(mil_amd.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/mil_hip.h but not exported"
    from mil_amd import _lib
    assert sorted(_lib.EXPORTS) == declared           # the ctypes binding covers exactly the header
    assert mil_amd.lib().mil_abi_version() == 2
    assert mil_amd.lib().mil_head_grad_floats() == 6807   # 11 head tensors


def test_host_side_queries_need_no_gpu():
    import mil_amd
    lib = mil_amd.lib()
    n = ctypes.c_size_t(0)
    assert lib.mil_packed_weight_elems(ctypes.byref(n), 20, 20, 3, 0) == 0
    assert n.value == (7 + 6) * 2 * 64 * 8             # 27 channel groups -> 7 k-steps, 2 column tiles; + the K-packed order's 6
    assert lib.mil_packed_weight_elems(ctypes.byref(n), 40, 40, 3, 0) == 0
    assert n.value == 12 * 3 * 64 * 8                  # 45 channel groups -> 12 k-steps, 3 column tiles (no second order)
    assert lib.mil_packed_weight_elems(ctypes.byref(n), 20, 3, 7, 2) == 0
    assert n.value == (8 + 6) * 2 * 64 * 8             # stem as 4x4 over 16 s2d channels: 8 k-steps; + the K-packed SK6 order's 6 (round 5)
    assert lib.mil_packed_weight_elems(ctypes.byref(n), 64, 3, 7, 2) == 0
    assert n.value == 8 * 4 * 64 * 8                   # alt_resnet's 64-channel stem: the standard order only
    assert lib.mil_conv_wgrad_workspace(ctypes.byref(n), 8, 64, 64, 20, 64, 64, 20, 3, 1, 1, 0, 1) == 0
    assert n.value > 0
    assert lib.mil_conv_wgrad_workspace(ctypes.byref(n), 8, 64, 64, 33, 64, 64, 20, 3, 1, 1, 0, 1) == 2   # unsupported width
    assert lib.mil_head_workspace_floats(ctypes.byref(n), 100, 3) == 0 and n.value > 0
    # argument errors are status codes, never crashes
    assert lib.mil_conv_igemm(None, None, None, None, None, None, 1, 8, 8, 24, 8, 8, 24, 3, 1, 1, 0, 0, 0.1, 1, None) == 1


def test_module_surface_matches_reference(golden_dir):
    import mil_amd
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    net = mil_amd.Attention(3, device="cpu")
    sd = net.state_dict()
    assert list(sd.keys()) == list(w.keys())                       # 65 keys, reference order
    assert all(tuple(sd[k].shape) == w[k].shape for k in sd)
    assert sum(p.numel() for p in net.parameters()) == 640967
    missing, unexpected = net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()}, strict=False)
    assert not missing and not unexpected
    # attributes the reference driver touches
    assert isinstance(net.weight_mask, torch.nn.Parameter) and net.weight_mask.shape == (3,)
    assert hasattr(net.cnn, "module") and hasattr(net, "context") and hasattr(net, "attention") and hasattr(net, "buffer")
    assert "off_diag" not in sd and tuple(net.off_diag.shape) == (3, 3)
    assert net.train().training and not net.eval().training
    assert "ResNet" in str(net) and len(list(net.named_parameters())) == 65
    # transfer-style filtering on key names works as in the reference driver (conv-only keys)
    assert len([k for k in sd if "cnn" in k and "conv" in k]) == 50


def test_no_cpu_fallback():
    import mil_amd
    net = mil_amd.Attention(3, device="cpu")
    with pytest.raises(RuntimeError):
        net(torch.zeros(4, 3, 32, 32), torch.tensor([0]))
    with pytest.raises(RuntimeError):
        net.cnn.module.layer1[0](torch.zeros(1, 20, 8, 8))
    with pytest.raises(ValueError):
        mil_amd.Attention(5, device="cpu")


def test_bag_layout_and_sharding():
    from mil_amd.dist import shard_bags
    from mil_amd.head import BagLayout
    lay = BagLayout([3, 5, 2], torch.device("cpu"))
    assert lay.offsets.tolist() == [0, 3, 8, 10] and lay.inst_bag.tolist() == [0] * 3 + [1] * 5 + [2] * 2
    with pytest.raises(ValueError):
        BagLayout([4, 1], torch.device("cpu"))           # single-instance bag: batch-norm refuses it
    with pytest.raises(ValueError):
        BagLayout([], torch.device("cpu"))
    assert shard_bags(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard_bags(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_lr_schedule_matches_reference_setstage():
    """gbm/classify_combined.py:110-138: warm-up base/(10-epoch), main base, check base/2, freeze base/10, stop >340."""
    from mil_amd.train import stage_for_epoch
    base = 2e-4
    assert stage_for_epoch(0) == ("Warmup", base / 10, True, False)
    assert stage_for_epoch(9) == ("Warmup", base / 1, True, False)
    assert stage_for_epoch(10) == ("Main", base, True, False)
    assert stage_for_epoch(149) == ("Main", base, True, False)
    assert stage_for_epoch(150) == ("Check", base / 2, True, False)
    assert stage_for_epoch(150, test=True) == ("Check", base / 2, False, False)
    assert stage_for_epoch(250, test=True) == ("Freeze", base / 10, False, False)
    assert stage_for_epoch(339) == ("Freeze", base / 10, True, False)
    assert stage_for_epoch(340) == ("Hold", None, None, False)
    assert stage_for_epoch(341)[3] is True


def test_set_stage_and_attention_map_export(tmp_path):
    import mil_amd
    from mil_amd.train import set_stage, write_attention_map
    net = mil_amd.Attention(3, device="cpu")
    opt = torch.optim.Adam(net.parameters(), lr=1.0)
    name, stop = set_stage(opt, net, 3)
    assert name == "Warmup" and not stop and abs(opt.param_groups[0]["lr"] - 2e-4 / 7) < 1e-12 and net.training
    set_stage(opt, net, 200, test=True)
    assert not net.training and abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12
    path = tmp_path / "prediction-AGMIL-ATTN.slide.dla"
    write_attention_map(str(path), [(0, 5), (1, 6), (2, 7)], torch.tensor([0.2, 0.6, 0.4]))
    lines = path.read_text().strip().split("\n")
    assert [ln.split()[:2] for ln in lines] == [["5", "0"], ["6", "1"], ["7", "2"]]
    assert [round(float(ln.split()[2]), 6) for ln in lines] == [0.0, 1.0, 0.5]
    with pytest.raises(ValueError):
        write_attention_map(str(path), [(0, 0)], torch.tensor([0.1, 0.2]))


def test_dla_export_is_byte_exact_with_reference_statements(tmp_path, golden_dir):
    """gbm/classify.py:207-225 writes four `.dla` files per slide (ATTN from `plt.Normalize()` in float32, ACTF1-3 raw),
    each value printed as the repr of the float32 widened to a Python float.  The fixtures under tests/golden/dla were
    produced by executing those statements on matplotlib itself (tests/golden/make_dla_golden.py)."""
    import mil_amd
    from fixture_inputs import DLA_CASES, dla_inputs
    cases = [(name, *dla_inputs(seed, n)) for name, seed, n in DLA_CASES]
    _, act_c, raster_c = dla_inputs(13, 4)
    cases.append(("slideC", torch.full((4, 3), 0.25), act_c, raster_c))          # constant map: vmin == vmax -> zeros
    for name, attn, activations, raster in cases:
        paths = mil_amd.write_map({"basename": name}, 0, raster, attn, activations, output_dir=str(tmp_path))
        assert [os.path.basename(p) for p in paths] == [f"prediction-AGMIL-{t}.{name}.dla" for t in ("ATTN", "ACTF1", "ACTF2", "ACTF3")]
        for p in paths:
            want = open(os.path.join(golden_dir, "dla", os.path.basename(p)), "rb").read()
            assert open(p, "rb").read() == want, p
    # the single-file helper writes the same bytes as the ATTN file when given the same column of a normalised map
    name, attn, activations, raster = cases[0]
    one = tmp_path / "one.dla"
    from mil_amd.train import _minmax_f32
    mil_amd.write_attention_map(str(one), [tuple(r) for r in raster], torch.from_numpy(_minmax_f32(attn)[:, 0]), normalise=False)
    assert one.read_bytes() == open(os.path.join(golden_dir, "dla", f"prediction-AGMIL-ATTN.{name}.dla"), "rb").read()
    with pytest.raises(ValueError):
        mil_amd.write_map({"basename": "x"}, 0, raster, attn[:, 0], activations, output_dir=str(tmp_path))


def test_init_parity_with_reference_seed(golden_dir):
    """gbm/model.py:118-187: construction order, default-init RNG consumption and `reset_params` (name-dependent init,
    in `named_modules()` order) must match the reference draw for draw: under `torch.manual_seed(1234)` every conv /
    linear WEIGHT equals the reference's (tests/golden/weights.npz holds `Attention(3)` built under that seed; the
    golden generator then overwrote only biases, the batch-norm affine and weight_mask, which reset to constants)."""
    import mil_amd
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    torch.manual_seed(1234)
    net = mil_amd.Attention(3, device="cpu")
    sd = net.state_dict()
    overwritten = [k for k in w.keys() if (k.endswith(".bias") or k in ("context.bn.weight", "weight_mask"))]
    assert len(overwritten) == 32
    for k in w.keys():
        if k in overwritten:
            continue
        assert np.array_equal(sd[k].numpy(), w[k]), k                 # 33 weight tensors, bit for bit
    for k in overwritten:                                              # and the constants the reference init leaves
        if k == "weight_mask":
            assert torch.equal(sd[k], torch.tensor([0.25, 0.25, 0.25]))
        elif k == "context.bn.weight":
            assert torch.equal(sd[k], torch.ones(80))
        else:
            assert torch.equal(sd[k], torch.zeros_like(sd[k])), k
    # reset_linear (gbm/model.py:183-187) re-draws every Linear with the tanh gain and zero bias
    torch.manual_seed(7)
    net.reset_linear()
    torch.manual_seed(7)
    for lin in (net.cnn.module.fc, net.attention.lin1, net.attention.lin2, net.buffer.lin1, net.buffer.classifier):   # modules() order
        ref = torch.empty_like(lin.weight)
        torch.nn.init.kaiming_normal_(ref, mode="fan_in", nonlinearity="tanh")
        assert torch.equal(lin.weight.detach(), ref)
        assert lin.bias is None or torch.equal(lin.bias.detach(), torch.zeros_like(lin.bias))


def test_forward_hooks_are_accepted_and_unmaterialised_children_refuse():
    """Children are real nn.Modules: hooks register on them.  Firing is covered on the GPU
    (tests/test_gpu_hooks.py); here: the bookkeeping that needs no kernel."""
    import mil_amd
    from mil_amd import hooks
    net = mil_amd.Attention(3, device="cpu")
    enc = net.cnn.module
    assert enc.child_hooks() is None and net._hooked_head_modules() == []
    h1 = enc.layer1.register_forward_hook(lambda m, i, o: None)
    h2 = net.context.register_forward_hook(lambda m, i, o: None)
    assert enc.child_hooks() == [enc.layer1] and net._hooked_head_modules() == [net.context]
    h1.remove(); h2.remove()
    assert enc.child_hooks() is None
    h3 = net.attention.lin1.register_forward_hook(lambda m, i, o: None)
    with pytest.raises(RuntimeError, match="never"):
        net._hooked_head_modules()
    h3.remove()
    h4 = enc.layer2[0].conv1.register_forward_hook(lambda m, i, o: None)
    with pytest.raises(RuntimeError, match="never"):
        hooks.refuse(enc.layer2[0], ["conv1", "conv2"])
    h4.remove()
    assert [enc.block_position(i) for i in (0, 2, 3, 11)] == [(0, 0, 3), (0, 2, 3), (1, 0, 3), (3, 2, 3)]
    with pytest.raises(RuntimeError):
        net.train().forward_tile_parallel(torch.zeros(4, 3, 32, 32))          # inference path only


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """gbm/classify_combined.py:468-474 writes {'classifier', 'optimizer'}; :521-535 reads it (full / conv-only transfer).
    The file written here must load into a plain torch.optim.Adam over the same parameters, and back."""
    import torch
    import mil_amd
    from mil_amd import train
    torch.manual_seed(5)
    model = mil_amd.Attention(3, device="cpu")
    flat = mil_amd.FlatParams(model)
    opt = mil_amd.FlatAdam(flat, lr=1e-4)
    opt.t = 7
    opt.exp_avg.normal_()
    opt.exp_avg_sq.uniform_()
    path = str(tmp_path / "train_step-001.model")
    train.save_checkpoint(path, model, opt)

    ckpt = torch.load(path, weights_only=True)
    assert set(ckpt) == {"classifier", "optimizer"}
    assert list(ckpt["classifier"].keys()) == list(model.state_dict().keys())
    ref_model = mil_amd.Attention(3, device="cpu")
    ref_model.load_state_dict(ckpt["classifier"], strict=False)
    ref_opt = torch.optim.Adam(ref_model.parameters(), betas=(0.9, 0.999), lr=0.0002)      # the reference's optimizer
    ref_opt.load_state_dict(ckpt["optimizer"])
    p0 = next(iter(ref_model.parameters()))
    assert float(ref_opt.state[p0]["step"]) == 7 and ref_opt.param_groups[0]["lr"] == 1e-4
    n0 = p0.numel()
    assert torch.equal(ref_opt.state[p0]["exp_avg"].reshape(-1), opt.exp_avg[:n0])

    # a checkpoint written by torch's Adam loads into the fused optimizer
    path2 = str(tmp_path / "from_torch.model")
    torch.save({"classifier": ref_model.state_dict(), "optimizer": ref_opt.state_dict()}, path2)
    model2 = mil_amd.Attention(3, device="cpu")
    flat2 = mil_amd.FlatParams(model2)
    opt2 = mil_amd.FlatAdam(flat2)
    missing, unexpected = train.load_checkpoint(path2, model2, opt2)
    assert not missing and not unexpected
    assert torch.equal(flat2.flat, flat.flat) and opt2.t == 7 and opt2.lr == 1e-4
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert all(p.data_ptr() >= flat2.flat.data_ptr() for p in model2.parameters())         # still views of the bucket

    # transfer: only the encoder's conv tensors move
    model3 = mil_amd.Attention(3, device="cpu")
    before = {k: v.clone() for k, v in model3.state_dict().items()}
    train.load_checkpoint(path, model3, transfer=True)
    for k, v in model3.state_dict().items():
        moved = "cnn" in k and "conv" in k
        assert torch.equal(v, ckpt["classifier"][k]) if moved else torch.equal(v, before[k]), k


# ---- round 3 host logic -----------------------------------------------------------------------------------------------
def test_compute_mode_plumbing():
    """`compute_dtype=mil_amd.BF16X3` = fp32 tensors whose convolution entry points get MIL_DT_F32S; the pointwise entry points
    keep MIL_DT_F32; the mode is scoped (context manager) and restored on exceptions."""
    import mil_amd
    from mil_amd import _lib as L
    assert L.storage_dtype(mil_amd.BF16X3) == torch.float32 and L.storage_dtype(torch.bfloat16) == torch.bfloat16
    assert L.mma_code(mil_amd.BF16X3) == L.MIL_DT_F32S == 3 and L.mma_code(torch.float32) == L.MIL_DT_F32 == 0
    assert L.dt_code(torch.float32) == L.MIL_DT_F32 and L.dt_code(torch.float32, mma=True) == L.MIL_DT_F32
    with L.f32_mma(L.MIL_DT_F32S):
        assert L.dt_code(torch.float32, mma=True) == L.MIL_DT_F32S
        assert L.dt_code(torch.float32) == L.MIL_DT_F32                 # pooling / s2d / head: plain fp32 entry points
        assert L.dt_code(torch.bfloat16, mma=True) == L.MIL_DT_BF16
    with pytest.raises(RuntimeError):
        with L.f32_mma(L.MIL_DT_F32S):
            raise RuntimeError("x")
    assert L.dt_code(torch.float32, mma=True) == L.MIL_DT_F32
    with pytest.raises(ValueError):
        L.dt_code(torch.float32, dense_grads=True)
    net = mil_amd.Attention(3, compute_dtype=mil_amd.BF16X3, device="cpu")
    assert net.compute_dtype == mil_amd.BF16X3


def test_split_precision_queries_need_no_gpu():
    """MIL_DT_F32S is accepted by the workspace queries of the kernels that have a split-precision form and refused by the rest."""
    import mil_amd
    lib = mil_amd.lib()
    n = ctypes.c_size_t(0)
    assert lib.mil_conv_wgrad_workspace(ctypes.byref(n), 8, 64, 64, 20, 64, 64, 20, 3, 1, 1, 0, 3) == 0 and n.value > 0
    # (the fused kernels' queries raise their LDS limit through the HIP runtime and need a device: covered by -m gpu)
    assert lib.mil_conv_bwd_fused_workspace(ctypes.byref(n), 8, 32, 32, 40, 40, 3, 1, 3) == 2                     # 40 channels: does not fit LDS
    assert lib.mil_conv_bwd_fused_workspace(ctypes.byref(n), 8, 64, 64, 20, 20, 3, 1, 0) == 2                     # exact fp32: no fused kernel
    assert lib.mil_stem_bwd_fused_workspace(ctypes.byref(n), 8, 128, 128, 3) == 2                                 # the s2d feed is bf16
    assert lib.mil_stream_copy(None, None, 16, None) == 1


def test_s2d_tiles_handle():
    import mil_amd
    xs = torch.zeros((5, 8, 6, 16), dtype=torch.bfloat16)
    t = mil_amd.S2dTiles(xs)
    assert tuple(t.shape) == (5, 3, 16, 12) and t.dim() == 4 and len(t) == 5
    assert tuple(t[torch.tensor([0, 3])].shape) == (2, 3, 16, 12) and tuple(t[1:4].shape) == (3, 3, 16, 12)
    assert tuple(mil_amd.S2dTiles.cat([t, t[:2]]).shape) == (7, 3, 16, 12)
    with pytest.raises(ValueError):
        mil_amd.S2dTiles(torch.zeros((5, 8, 6, 16)))            # fp32: not the bf16 record tensor
    with pytest.raises(ValueError):
        mil_amd.S2dTiles(torch.zeros((5, 8, 6, 12), dtype=torch.bfloat16))


def test_head_gradient_run_detection():
    """The head adds its gradient block into the flat bucket with ONE add when the parameters' .grad tensors sit back to back
    (FlatParams); otherwise autograd accumulates per parameter as before."""
    import mil_amd
    from mil_amd.head import _contiguous_grad_run
    net = mil_amd.Attention(3, device="cpu")
    ws = net.head_weights()
    assert _contiguous_grad_run(ws) is None                       # no .grad yet
    flat = mil_amd.FlatParams(net)
    first, count, total = _contiguous_grad_run(ws)
    assert first == 0 and count == 10 and total == 6807 - 3       # context.bn.weight .. buffer.classifier.bias; weight_mask sits elsewhere
    ws[3].grad = torch.zeros_like(ws[3])                          # a foreign .grad tensor breaks the run there
    assert _contiguous_grad_run(ws) == (0, 3, 80 + 80 + 40 * 80)
    flat.zero_grad()                                              # re-attaches
    assert _contiguous_grad_run(ws)[1] == 10


def test_bench_roofline_accounting():
    """bench.py's per-family algorithmic cost (what `roofline.achieved` is computed from) and the launch filter."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    px = 2048 * 64 * 64
    fl, by, shape = bench._family_cost(("bwd_fused", 24, 24, 3, 1, False, 2048, 64, 64, True), 2)
    assert fl == 2 * 2.0 * 9 * 20 * 20 * px and by == 4 * px * 20 * 2 and shape == (2048, 64, 64)
    fl, by, _ = bench._family_cost(("bwd_fused", 24, 24, 3, 1, False, 2048, 64, 64, False), 4)
    assert by == 3 * px * 20 * 4
    fl, by, _ = bench._family_cost(("stem_fwd", 24, 2048, 256, 256), 2)
    assert fl == 2.0 * 147 * 20 * 2048 * 128 * 128 and by == 2048 * 3 * 256 * 256 * 4 + 2048 * 64 * 64 * 20 * 3
    fl, by, _ = bench._family_cost(("stem_fwd_xs", 24, 2048, 256, 256), 2)
    assert by == 2048 * 128 * 128 * 24 + 2048 * 64 * 64 * 20 * 3
    assert bench.timer_wants(("conv", 24, 24, 3, 1, False, 2048, 64, 64)) and not bench.timer_wants(("conv", 40, 40, 3, 1, False, 2048, 32, 32))
    assert bench.timer_wants(("stem_bwd", 2048, 256, 256)) and bench.timer_wants(("wgrad", 24, 24, 3, 1, 2048, 64, 64))
    assert not bench.timer_wants(("wgrad", 16, 24, 4, 1, 2048, 128, 128)) and not bench.timer_wants(("chain", 80, 2048, 8, 8, 5))
    for mode in ("bf16", "bf16x3"):
        for fam in ("block_fwd", "conv", "bwd_fused", "wgrad", "stem_fwd", "stem_bwd", "stem_fwd_xs", "stem_bwd_xs"):
            name, keys = bench._family_kernel(fam, mode)
            assert isinstance(name, str) and len(keys) >= 1
    assert "fused16x3" in bench._family_kernel("bwd_fused", "bf16x3")[0] and "fused16_kernel" in bench._family_kernel("bwd_fused", "bf16")[0]
    assert "x3" in bench._family_kernel("block_fwd", "bf16x3")[0] and "x3" not in bench._family_kernel("block_fwd", "bf16")[0]
