"""-m gpu: the drop-in `Attention` module on the HIP kernels against (a) the golden vectors captured
from the reference and (b) the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): logits / attention weights within 1e-3 absolute of the fp32 CPU
reference — asserted for the exact-fp32 kernel path, together with a relative check because attention
weights are O(1/N).  The bf16 path (fp32 accumulate) is held to a looser, stated bound."""
import os

import numpy as np
import pytest
import torch

from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu

OUT_KEYS = ["Aterm", "wROIs", "Bterm", "Mterm", "Fterm", "Aterm_mu", "Aterm_var", "loss", "l2", "KLD", "y_pred",
            "y_pred_hat", "error"]
CASES = ["eval_n8_64", "eval_n8_64_cw", "eval_n5_50x70", "train_n40_64", "eval_n2_256"]


@pytest.fixture(autouse=True, params=["generic", "persistent"])
def kernel_path(request, monkeypatch):
    """Every model-level case runs twice: through the generic kernels (what these small launches would pick) and
    through the persistent prefetch-pipelined kernels that benchmark-sized launches use."""
    monkeypatch.setenv("MIL_PF_MIN_TILES", "1" if request.param == "persistent" else "1000000000")
    return request.param


def _model(golden_dir, dtype, class_weights=None):
    import mil_amd
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    net = mil_amd.Attention(3, class_weights=class_weights, compute_dtype=dtype)
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()}, strict=True)
    return net


def _run_case(golden_dir, name, dtype):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cw = torch.tensor(g["class_weights"]) if "class_weights" in g else None
    net = _model(golden_dir, dtype, cw)
    if "rec.indices" in g:
        net.train()
        net.rng_override = {"indices": torch.tensor(g["rec.indices"]), "keep_mask": torch.tensor(g["rec.keep_mask"])}
    else:
        net.eval()
    out = net(torch.tensor(g["x"]).cuda(), torch.tensor(g["y"]).cuda())
    out["loss"].backward()
    return net, g, out


def _maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def _rel(a, b):
    return _maxabs(a, b) / max(float(np.abs(np.asarray(b, np.float64)).max()), 1e-30)


# d loss / d buffer.classifier.bias = sum_k dM_k * sum_n A1[n,k] = sum_k dM_k = 0 analytically (the
# attention maps sum to 1 and the soft-target CE gradient sums to 0): the reference's own value is
# ~1e-8 rounding residue, so it is compared absolutely, not relatively.
ZERO_GRADS = ("buffer.classifier.bias",)


def _grad_close(k, got, ref, rtol):
    if k in ZERO_GRADS:
        return _maxabs(got, ref) < 1e-5
    return _rel(got, ref) < rtol


@pytest.mark.parametrize("name", CASES)
def test_fp32_outputs_match_reference_golden(golden_dir, name):
    net, g, out = _run_case(golden_dir, name, torch.float32)
    assert list(out.keys()) == OUT_KEYS
    for k in OUT_KEYS:
        ref, got = g["out." + k], out[k].detach().cpu().numpy()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert got.dtype == ref.dtype, k
    assert [k for k in OUT_KEYS if out[k].requires_grad] == ["loss", "l2"]
    assert np.array_equal(out["y_pred_hat"].cpu().numpy(), g["out.y_pred_hat"])
    assert np.array_equal(out["error"].cpu().numpy(), g["out.error"])
    # the north-star gate: logits (Mterm), attention weights, class probabilities within 1e-3 absolute
    for k in ("Mterm", "Aterm", "y_pred", "loss", "wROIs", "Bterm"):
        assert _maxabs(out[k].detach().cpu().numpy(), g["out." + k]) < 1e-3, k
    # and relative, since attention weights are ~1/N
    for k in ("Aterm", "Fterm", "Mterm", "Bterm", "KLD", "Aterm_mu", "l2", "loss"):
        assert _rel(out[k].detach().cpu().numpy(), g["out." + k]) < 2e-4, k
    assert abs(float(out["Aterm"].sum(1).sub(1).abs().max())) < 1e-5          # each map sums to 1


@pytest.mark.parametrize("name", CASES)
def test_bf16x3_outputs_match_reference_golden(golden_dir, name):
    """The split-precision path (fp32 tensors, bf16x3 products) on every reference golden case — eval and train mode (recorded
    subsample indices and dropout mask), class weights, 50x70 tiles (W % 4 != 0: the un-fused stem), the two-tile bag: same
    output dict, and the north-star gate, 1e-3 absolute, on logits / attention weights / probabilities / loss."""
    from gpu_util import X3
    net, g, out = _run_case(golden_dir, name, X3)
    assert list(out.keys()) == OUT_KEYS
    for k in OUT_KEYS:
        ref, got = g["out." + k], out[k].detach().cpu().numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype, k
    assert np.array_equal(out["y_pred_hat"].cpu().numpy(), g["out.y_pred_hat"])
    err = {k: _maxabs(out[k].detach().cpu().numpy(), g["out." + k]) for k in ("Mterm", "Aterm", "y_pred", "loss", "wROIs", "Bterm")}
    print(f"bf16x3 golden {name}: {err}")
    for k, e in err.items():
        assert e < 1e-3, (k, e)
    assert _rel(out["Fterm"].detach().cpu().numpy(), g["out.Fterm"]) < 5e-5
    assert abs(float(out["Aterm"].sum(1).sub(1).abs().max())) < 1e-5
    params = dict(net.named_parameters())
    worst = 0.0
    for k, n_ref in zip(list(g["gradnorm.names"]), g["gradnorm.l2"]):
        worst = max(worst, abs(float(params[k].grad.double().norm()) - n_ref) / max(n_ref, 1e-3))
    print(f"bf16x3 golden {name}: worst gradient-norm deviation {worst:.2e}")
    assert worst < (0.2 if name == "eval_n2_256" else 5e-2)      # branch flips; N=2: ill-conditioned BN backward


def test_default_constructed_module_is_inside_the_reference_tolerance(golden_dir):
    """`Attention(n_classes, class_weights)` exactly as gbm/classify_combined.py:518 calls it — no keyword — must land inside
    the north-star gate (1e-3 absolute on logits / attention weights / probabilities / loss vs the reference golden): the
    default compute mode is the split-precision one; bf16 is the opt-in fast mode."""
    import mil_amd
    g = np.load(os.path.join(golden_dir, "eval_n8_64_cw.npz"))
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    net = mil_amd.Attention(3, torch.tensor(g["class_weights"]))
    assert net.compute_dtype == mil_amd.BF16X3
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()}, strict=True)
    net.eval()
    out = net(torch.tensor(g["x"]).cuda(), torch.tensor(g["y"]).cuda())
    assert list(out.keys()) == OUT_KEYS
    for k in ("Mterm", "Aterm", "y_pred", "loss", "wROIs", "Bterm"):
        assert _maxabs(out[k].detach().cpu().numpy(), g["out." + k]) < 1e-3, k
    assert np.array_equal(out["y_pred_hat"].cpu().numpy(), g["out.y_pred_hat"])


@pytest.mark.parametrize("name", CASES)
def test_fp32_gradients_match_reference_golden(golden_dir, name):
    net, g, out = _run_case(golden_dir, name, torch.float32)
    params = dict(net.named_parameters())
    names = list(g["gradnorm.names"])
    assert names == list(params.keys())
    tol = 5e-3 if name == "eval_n2_256" else 5e-4        # N=2: ill-conditioned BN backward (see oracle test)
    for k, n_ref in zip(names, g["gradnorm.l2"]):
        gr = params[k].grad
        assert gr is not None, k
        assert abs(float(gr.double().norm()) - n_ref) <= tol * max(n_ref, 1e-3), k
        if "grad." + k in g:
            assert _grad_close(k, gr.cpu().numpy(), g["grad." + k], 2e-3), k


@pytest.mark.parametrize("name", ["eval_n8_64", "eval_n5_50x70"])
def test_fp32_stage_activations(golden_dir, name):
    """Encoder internals against the reference's hooked activations."""
    from mil_amd import encoder
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    net = _model(golden_dir, torch.float32).eval()
    enc = net.cnn.module
    with torch.no_grad():
        feats, saved = encoder.encoder_forward(enc, torch.tensor(g["x"]).cuda(), torch.float32)
    for li, bi in ((1, 2), (2, 5), (3, 8), (4, 11)):
        c = (20, 40, 60, 80)[li - 1]
        act = saved["blocks"][bi][2][..., :c].permute(0, 3, 1, 2).cpu().numpy()
        assert _rel(act, g[f"act.layer{li}"]) < 2e-5, li
    assert _rel(feats.cpu().numpy(), g["out.Fterm"]) < 5e-5


@pytest.mark.parametrize("name", ["eval_n8_64", "eval_n5_50x70"])
def test_bf16_encoder_forward_backward_vs_emulating_oracle(golden_dir, name):
    """bf16 kernel correctness, isolated from the head's ill-conditioned batch-norm: the encoder's
    features and — for one fixed upstream gradient — every encoder parameter gradient, against the oracle
    run with bf16-STORAGE EMULATION (same fp32 arithmetic, tensors rounded to bf16 exactly where the HIP
    path stores them).  What remains is fp32 accumulation order and the rare 1-ulp bf16 flip it causes."""
    from mil_amd import encoder
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    x = torch.tensor(g["x"])
    dfe = torch.randn(x.shape[0], 80, generator=torch.Generator().manual_seed(3))
    sd = orc.load_state(w, requires_grad=True)
    feats_ref = orc.backbone(sd, x, emulate_bf16=True)
    feats_ref.backward(dfe)
    net = _model(golden_dir, torch.bfloat16).eval()
    enc = net.cnn.module
    feats = enc(x.cuda())
    feats.backward(dfe.cuda())
    assert _rel(feats.detach().cpu().numpy(), feats_ref.detach().numpy()) < 1e-2
    # Two correct bf16-storage realisations are not bit-identical: a 1e-7 accumulation-order difference
    # flips a bf16 rounding now and then, and through 26 un-normalised layers the flips multiply (measured:
    # 3e-5 of the stem outputs differ, 75% of layer4's, each by <= 1 ulp).  So gradients are compared by
    # direction and size: cosine >= 0.97 and max-relative <= 0.35 per tensor, which is what the emulating
    # oracle itself shows against the fp32 oracle (median cosine 0.994, worst 0.969) on this case.
    worst_cos, worst_rel = (2.0, ""), (0.0, "")
    for k, p in enc.named_parameters():
        a = p.grad.detach().cpu().double().flatten()
        b = sd["cnn.module." + k].grad.double().flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
        worst_cos = min(worst_cos, (cos, k))
        worst_rel = max(worst_rel, (_rel(a.numpy(), b.numpy()), k))
    print("bf16 encoder grads vs emulating oracle: worst cosine", worst_cos, "worst rel", worst_rel)
    assert worst_cos[0] > 0.97, worst_cos
    assert worst_rel[0] < 0.35, worst_rel


@pytest.mark.parametrize("name", ["eval_n8_64", "train_n40_64", "eval_n5_50x70"])
def test_bf16_end_to_end_stated_tolerance(golden_dir, name):
    """Precision of the bf16-storage choice itself, end to end against the fp32 reference golden: features
    within 3% of the largest feature, class probabilities within 2e-2, loss within 2%.  The batch-statistics
    BatchNorm over a handful of near-identical random-weight instances amplifies feature error into the
    attention weights and the gradients, which is why the north star's 1e-3 gate is asserted on the fp32
    path (above) and the bf16 kernels are checked tightly against the emulating oracle instead."""
    net, g, out = _run_case(golden_dir, name, torch.bfloat16)
    assert _rel(out["Fterm"].cpu().numpy(), g["out.Fterm"]) < 3e-2
    assert _maxabs(out["y_pred"].cpu().numpy(), g["out.y_pred"]) < 2e-2
    assert _rel(out["loss"].detach().cpu().numpy(), g["out.loss"]) < 2e-2
    assert _maxabs(out["Aterm"].cpu().numpy(), g["out.Aterm"]) < 2e-2
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_batched_bags_equal_per_bag_calls(golden_dir):
    """forward_bags (one encoder pass, segmented head) == one call per bag; gradients add up."""
    g = np.load(os.path.join(golden_dir, "eval_n8_64.npz"))
    x = torch.tensor(g["x"]).cuda()
    bags = [x[:3], x[3:8], x[1:7]]
    labels = torch.tensor([0, 1, 2])
    net = _model(golden_dir, torch.float32).eval()
    outs = net.forward_bags(bags, labels)
    torch.stack([o["loss"] for o in outs]).sum().backward()
    g_batched = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad(set_to_none=True)
    outs2 = net.forward_bags(bags, labels)              # the un-split loss vector: same values, same gradients
    assert torch.equal(outs2.loss.detach(), torch.stack([o["loss"].detach() for o in outs]))
    outs2.loss.sum().backward()
    for k, p in net.named_parameters():
        assert torch.equal(p.grad, g_batched[k]), k
    net.zero_grad(set_to_none=True)
    for b, (xb, yb) in enumerate(zip(bags, labels)):
        o = net(xb, yb.view(1))
        o["loss"].backward()
        for k in ("Aterm", "Mterm", "loss", "y_pred", "wROIs"):
            assert torch.allclose(o[k], outs[b][k], rtol=1e-5, atol=1e-7), (b, k)
    for k, p in net.named_parameters():
        assert _grad_close(k, p.grad.cpu().numpy(), g_batched[k].cpu().numpy(), 1e-4), k


def test_head_matches_oracle_many_instances(golden_dir):
    """The segmented head alone vs the oracle on a large ragged batch (N up to 1500 per bag)."""
    from mil_amd.head import BagLayout, head_apply
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    sd = orc.load_state(w, requires_grad=True)
    gen = torch.Generator().manual_seed(5)
    sizes = [1500, 2, 257, 64]
    H = torch.randn(sum(sizes), 80, generator=gen) * 20
    keep = (torch.rand(sum(sizes), 80, generator=gen) >= 0.25).to(torch.uint8)
    labels = torch.tensor([2, 0, 1, 1])
    cw = torch.tensor([0.5, 1.0, 2.0])
    net = _model(golden_dir, torch.float32, cw)
    Hg = H.cuda().requires_grad_(True)
    layout = BagLayout(sizes, Hg.device)
    loss, l2, a1, wrois, bterm, kld, rec = head_apply(Hg, layout, labels.cuda(), keep.cuda(), cw.cuda(), net.head_weights())
    (loss.sum() + 0.3 * l2).backward()
    Hc = H.clone().requires_grad_(True)
    tot, off = 0.0, 0
    for b, n in enumerate(sizes):
        o = orc.mil_head(sd, Hc[off:off + n], labels[b:b + 1], keep_mask=keep[off:off + n], class_weights=cw)
        assert _rel(a1[off:off + n].t().cpu().numpy(), o["Aterm"].numpy()) < 1e-4
        assert _rel(rec[b, 0:3].cpu().numpy(), o["Mterm"].numpy().ravel()) < 1e-4
        assert abs(float(loss[b].detach()) - float(o["loss"].detach())) < 1e-5 * max(1.0, abs(float(o["loss"])))
        assert abs(float(rec[b, 9]) - float(o["Aterm_var"])) < 1e-5
        assert abs(float(rec[b, 8]) - float(o["Aterm_mu"])) < 1e-4 * max(1.0, float(o["Aterm_mu"]))
        assert abs(float(kld[b]) - float(o["KLD"])) < 1e-4 * float(o["KLD"])
        tot = tot + o["loss"]
        last = o
        off += n
    (tot + 0.3 * last["l2"]).backward()
    assert _rel(Hg.grad.cpu().numpy(), Hc.grad.numpy()) < 2e-3
    for p, k in zip(net.head_weights(), ["context.bn.weight", "context.bn.bias", "attention.lin1.weight",
                                         "attention.lin1.bias", "attention.lin2.weight", "attention.lin2.bias",
                                         "buffer.lin1.weight", "buffer.lin1.bias", "buffer.classifier.weight",
                                         "buffer.classifier.bias", "weight_mask"]):
        assert _grad_close(k, p.grad.cpu().numpy(), sd[k].grad.numpy(), 2e-3), k


def test_errors_are_exceptions(golden_dir):
    net = _model(golden_dir, torch.float32).eval()
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 32, 32).cuda(), torch.tensor([0]))          # single-instance bag (BatchNorm)
    with pytest.raises(ValueError):
        net(torch.zeros(4, 1, 32, 32).cuda(), torch.tensor([0]))          # wrong channel count


def test_flat_adam_matches_torch_adam(golden_dir):
    """mil_adam_step (one launch over the flat buckets) == torch.optim.Adam for several steps, and the packed
    filters follow the updated weights (outputs change, stay finite, and equal a freshly built model's)."""
    import mil_amd
    g = np.load(os.path.join(golden_dir, "eval_n8_64.npz"))
    x, y = torch.tensor(g["x"]).cuda(), torch.tensor(g["y"]).cuda()
    net = _model(golden_dir, torch.float32).eval()
    flat = mil_amd.FlatParams(net)
    opt = mil_amd.FlatAdam(flat, lr=1e-3)
    ref_params = [p.detach().clone().cpu().requires_grad_(True) for p in net.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3)
    losses = []
    for _ in range(3):
        flat.zero_grad()
        out = net(x, y)
        out["loss"].backward()
        losses.append(float(out["loss"].detach()))
        for rp, p in zip(ref_params, net.parameters()):
            rp.grad = p.grad.detach().cpu().clone()
        opt.step()
        ref_opt.step()
        for rp, p in zip(ref_params, net.parameters()):
            assert torch.allclose(p.detach().cpu(), rp.detach(), rtol=1e-5, atol=1e-7)
    assert losses[2] < losses[0]                       # three Adam steps on one bag reduce its loss
    fresh = _model(golden_dir, torch.float32).eval()
    fresh.load_state_dict(net.state_dict())
    a, b = net(x, y), fresh(x, y)
    assert torch.allclose(a["Mterm"], b["Mterm"], rtol=1e-6, atol=1e-7)


def test_benchmark_sized_launch_properties(golden_dir, monkeypatch):
    """Size-independent checks at a launch size that takes the persistent kernels on its own (2 bags x 256 tiles
    @256x256, default tile-count threshold): every attention map sums to 1; features of a tile do not depend on
    which other tiles share the launch (same tiles re-encoded alone through the generic kernels, fp32 exactly
    comparable up to accumulation order); two runs are bitwise identical (no atomics anywhere)."""
    monkeypatch.delenv("MIL_PF_MIN_TILES", raising=False)
    gen = torch.Generator(device="cuda").manual_seed(123)
    x = torch.randn((512, 3, 256, 256), generator=gen, device="cuda").clamp_(-1, 1)
    labels = torch.tensor([0, 2])
    for dtype, tol in ((torch.float32, 2e-4), (torch.bfloat16, 3e-2)):
        net = _model(golden_dir, dtype).eval()
        outs = net.forward_bags((x, [256, 256]), labels)
        torch.stack([o["loss"] for o in outs]).sum().backward()
        g1 = [p.grad.clone() for p in net.parameters()]
        for o in outs:
            assert torch.allclose(o["Aterm"].sum(dim=1), torch.ones(3, device="cuda"), atol=1e-5)
            assert torch.isfinite(o["loss"]) and torch.isfinite(o["Fterm"]).all()
        assert all(torch.isfinite(g).all() for g in g1)
        with torch.no_grad():
            small = net.cnn(x[:6])                      # 6 tiles: far below the persistent-kernel threshold
        big = outs[0]["Fterm"][:6]
        assert float((small - big).abs().max() / big.abs().max()) < tol, dtype
        net.zero_grad(set_to_none=True)
        outs2 = net.forward_bags((x, [256, 256]), labels)
        torch.stack([o["loss"] for o in outs2]).sum().backward()
        assert all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))
        assert torch.equal(outs[1]["Aterm"], outs2[1]["Aterm"])


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_row_walk_kernels_inside_the_model(golden_dir, mode, monkeypatch):
    """The row-walk kernels (block forward, stem forward, bf16 stem backward) forced on against forced off through the whole model
    at 2 bags x 256 tiles @256x256: every forward output bit for bit — the forward row walks are bit-identical to their tiled forms,
    so a result does not depend on which form a launch size selects —, every gradient bit for bit in split precision (its stem
    backward has one form), and in bf16 to the fp32 summation order of the stem's weight gradient (the stem has no input gradient,
    so nothing else can differ)."""
    import mil_amd
    monkeypatch.delenv("MIL_PF_MIN_TILES", raising=False)
    dtype = torch.bfloat16 if mode == "bf16" else mil_amd.BF16X3
    gen = torch.Generator(device="cuda").manual_seed(321)
    x = torch.randn((512, 3, 256, 256), generator=gen, device="cuda").clamp_(-1, 1)
    labels = torch.tensor([1, 2])
    res = {}
    for forced in ("0", "1"):
        monkeypatch.setenv("MIL_BLOCK_STRIP", forced)
        monkeypatch.setenv("MIL_STEM_WALK", forced)
        net = _model(golden_dir, dtype).eval()
        outs = net.forward_bags((x, [256, 256]), labels)
        torch.stack([o["loss"] for o in outs]).sum().backward()
        torch.cuda.synchronize()
        res[forced] = ([{k: o[k].detach().clone() for k in ("Aterm", "Mterm", "Fterm", "loss", "y_pred")} for o in outs],
                       {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    for o0, o1 in zip(res["0"][0], res["1"][0]):
        for k in o0:
            assert torch.equal(o0[k], o1[k]), k
    for k, g0 in res["0"][1].items():
        g1 = res["1"][1][k]
        if mode == "bf16x3" or not k.startswith("cnn.module.conv1."):
            assert torch.equal(g0, g1), k
        else:
            assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max()), k


@pytest.mark.parametrize("dtype", [torch.float32, "bf16x3"], ids=["fp32", "bf16x3"])
def test_live_driver_tile_size_and_ragged_bags(golden_dir, dtype):
    """The reference's live driver feeds 300x300 tiles (gbm/classify_combined.py:412; maps 150->75->38->19->10, no
    power of two anywhere) and bags of any size up to 2500 (RoiBuilder.py:230).  The exact-fp32 path AND the split-precision
    path (whose odd maps take other kernels than every power-of-two size: ragged edge tiles, generic fall-backs) vs the CPU
    oracle on ragged bags encoded in one launch: logits / attention weights / probabilities / loss within 1e-3, four
    gradients; bf16 path for shape/finite checks on a larger ragged batch."""
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    gen = torch.Generator().manual_seed(31)
    sizes = [5, 2, 9]
    x = torch.randn(sum(sizes), 3, 300, 300, generator=gen).clamp_(-1, 1)
    labels = torch.tensor([1, 0, 2])
    net = _model(golden_dir, dtype).eval()
    outs = net.forward_bags((x.cuda(), sizes), labels)
    torch.stack([o["loss"] for o in outs]).sum().backward()
    sd = orc.load_state(w, requires_grad=True)
    off, tot = 0, 0.0
    for b, n in enumerate(sizes):
        ref = orc.attention_forward(sd, x[off:off + n], labels[b:b + 1])
        for k in ("Aterm", "Mterm", "y_pred", "loss"):
            assert _maxabs(outs[b][k].detach().cpu().numpy(), ref[k].detach().numpy()) < 1e-3, (b, k)
        assert _rel(outs[b]["Fterm"].cpu().numpy(), ref["Fterm"].numpy()) < (2e-4 if dtype == torch.float32 else 1e-4)
        tot = tot + ref["loss"]
        off += n
    tot.backward()
    params = dict(net.named_parameters())
    for k in ("cnn.module.conv1.weight", "cnn.module.layer2.0.downsample.0.weight", "cnn.module.layer4.2.conv2.weight",
              "attention.lin1.weight"):
        # 1e-2: one of the bags has 2 instances, whose batch-norm backward is ill-conditioned (see the oracle test);
        # split precision: LeakyReLU branch flips on top
        assert _rel(params[k].grad.cpu().numpy(), sd[k].grad.numpy()) < (1e-2 if dtype == torch.float32 else 5e-2), k
    if dtype != torch.float32:
        return
    net16 = _model(golden_dir, torch.bfloat16).eval()
    sizes16 = [37, 3, 60]
    x16 = torch.randn(sum(sizes16), 3, 300, 300, generator=gen).clamp_(-1, 1).cuda()
    outs16 = net16.forward_bags((x16, sizes16), torch.tensor([0, 1, 2]))
    for o, n in zip(outs16, sizes16):
        assert o["Aterm"].shape == (3, n) and torch.isfinite(o["Aterm"]).all()
        assert torch.allclose(o["Aterm"].sum(dim=1), torch.ones(3, device="cuda"), atol=1e-5)


def test_large_bag_attention_map_inference(golden_dir):
    """BASELINE config 5 in miniature: one large bag, forward only, attention weights vs the CPU reference
    arithmetic (fp32 kernels, 1e-3 absolute as the north star asks, and 1% relative since weights are ~1/N);
    the tile-parallel split (encode slices separately, gather features, run the head once) gives the same map."""
    from mil_amd.head import BagLayout, head_apply
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    n = 768
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(n, 3, 64, 64, generator=gen).clamp_(-1, 1)
    y = torch.tensor([1])
    sd = orc.load_state(w)
    with torch.no_grad():
        feats = torch.cat([orc.backbone(sd, x[i:i + 128]) for i in range(0, n, 128)])
        ref = orc.mil_head(sd, feats, y)
    net = _model(golden_dir, torch.float32).eval()
    with torch.no_grad():
        out = net(x.cuda(), y.cuda())
    assert _maxabs(out["Aterm"].cpu().numpy(), ref["Aterm"].numpy()) < 1e-3
    assert _rel(out["Aterm"].cpu().numpy(), ref["Aterm"].numpy()) < 1e-2
    assert int(out["y_pred_hat"]) == int(ref["y_pred_hat"])
    with torch.no_grad():          # what 8 ranks would do: each encodes a slice, features are gathered, head replicated
        parts = [net.cnn(x[i:i + 96].cuda()) for i in range(0, n, 96)]
        H = torch.cat(parts)
        _loss, _l2, a1, *_ = head_apply(H, BagLayout([n], H.device), y.cuda(), None, None, net.head_weights())
    assert torch.allclose(a1.t(), out["Aterm"], rtol=1e-4, atol=1e-7)


def test_direct_gradient_accumulation_equals_autograd(golden_dir):
    """FlatParams switches the encoder to accumulate weight gradients straight into the flat bucket; two backward
    passes must leave exactly the sum autograd's own accumulation would (and twice one pass)."""
    import mil_amd
    g = np.load(os.path.join(golden_dir, "eval_n8_64.npz"))
    x, y = torch.tensor(g["x"]).cuda(), torch.tensor(g["y"]).cuda()
    ref = _model(golden_dir, torch.float32).eval()
    for _ in range(2):
        ref(x, y)["loss"].backward()
    net = _model(golden_dir, torch.float32).eval()
    flat = mil_amd.FlatParams(net)
    assert net.cnn.module.direct_grad
    flat.zero_grad()
    for _ in range(2):
        net(x, y)["loss"].backward()
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        assert p.grad.data_ptr() >= flat.flat_grad.data_ptr()
        assert _grad_close(k, p.grad.cpu().numpy(), q.grad.cpu().numpy(), 1e-5), k


def test_autograd_grad_returns_gradients_and_never_touches_grad_storage(golden_dir):
    """Without the FlatParams opt-in the in-place accumulation paths stay off even when the .grad tensors happen to be one
    contiguous buffer: `torch.autograd.grad` gets every gradient back and that buffer is not written.  (Under the opt-in —
    a FlatParams owns the gradients — the contract is plain `loss.backward()`: autograd offers a custom Function no way to tell
    an accumulating backward from `autograd.grad`, needs_input_grad being fixed at forward time.)"""
    import mil_amd
    g = np.load(os.path.join(golden_dir, "eval_n8_64.npz"))
    x, y = torch.tensor(g["x"]).cuda(), torch.tensor(g["y"]).cuda()
    ref = _model(golden_dir, torch.float32).eval()
    ref(x, y)["loss"].backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    net = _model(golden_dir, torch.float32).eval()
    params = dict(net.named_parameters())
    buf = torch.full((sum(p.numel() for p in params.values()),), 7.0, device="cuda")
    off = 0
    for p in params.values():                    # adjacent, contiguous .grad views — what FlatParams builds, minus the opt-in
        p.grad = buf[off:off + p.numel()].view(p.shape)
        off += p.numel()
    assert not net.direct_grad and not net.cnn.module.direct_grad
    got = torch.autograd.grad(net(x, y)["loss"], list(params.values()))
    assert bool((buf == 7.0).all())              # nothing accumulated behind autograd's back
    for (k, _p), gk in zip(params.items(), got):
        assert gk is not None and _grad_close(k, gk.cpu().numpy(), want[k].cpu().numpy(), 1e-5), k
    # opt-in active (FlatParams): plain loss.backward() accumulates in place; a FROZEN head parameter (requires_grad False at
    # forward time -> ctx.needs_input_grad False) ends the in-place run in front of it and gets no gradient
    flat = mil_amd.FlatParams(net)
    flat.zero_grad()
    net.attention.lin1.bias.requires_grad_(False)
    net(x, y)["loss"].backward()
    for k, p in net.named_parameters():
        if k == "attention.lin1.bias":
            assert float(p.grad.abs().max()) == 0.0          # untouched slot of the bucket
        else:
            assert _grad_close(k, p.grad.cpu().numpy(), want[k].cpu().numpy(), 1e-5), k


def test_whole_step_is_bitwise_reproducible(golden_dir, monkeypatch):
    """Two passes from the same state give bit-identical outputs and gradients (no atomics, fixed reduction trees, no
    races in the double-buffered / early-load pipelines): large enough for the persistent kernels and several tiles per
    workgroup."""
    import mil_amd
    monkeypatch.setenv("MIL_PF_MIN_TILES", "1")
    net = _model(golden_dir, torch.bfloat16).eval()
    flat = mil_amd.FlatParams(net)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((96, 3, 128, 128), generator=g, device="cuda").clamp_(-1, 1)
    sizes, labels = [40, 30, 26], torch.tensor([0, 1, 2], device="cuda")
    results = []
    for _ in range(3):
        flat.zero_grad()
        outs = net.forward_bags((x, sizes), labels)
        outs.loss.sum().backward()
        torch.cuda.synchronize()
        results.append((outs.loss.detach().clone(), torch.cat([o["Aterm"].reshape(-1) for o in outs]).clone(), flat.flat_grad.clone()))
    for r in results[1:]:
        assert torch.equal(r[0], results[0][0]) and torch.equal(r[1], results[0][1])
        assert torch.equal(r[2], results[0][2])
    assert float(results[0][2].abs().max()) > 0


def test_side_stream_weight_gradients_equal_main_stream(golden_dir):
    """`overlap_wgrad` (separate weight-gradient launches on a side stream) changes scheduling only: same bits."""
    import mil_amd
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((24, 3, 64, 64), generator=g, device="cuda").clamp_(-1, 1)
    sizes, labels = [10, 14], torch.tensor([2, 0], device="cuda")
    grads = []
    for overlap in (False, True):
        net = _model(golden_dir, torch.bfloat16).eval()
        net.cnn.module.overlap_wgrad = overlap
        flat = mil_amd.FlatParams(net)
        flat.zero_grad()
        net.forward_bags((x, sizes), labels).loss.sum().backward()
        torch.cuda.synchronize()
        grads.append(flat.flat_grad.clone())
    assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0


def test_input_modified_between_forward_and_backward_raises(golden_dir):
    """Without a kept space-to-depth copy the fused stem backward re-reads the caller's input tiles: an in-place change between
    forward and backward must raise (the tensor's version counter is checked) instead of silently corrupting conv1's gradient;
    keep_s2d=True keeps a library-owned copy and is immune."""
    import mil_amd
    for mode in (torch.bfloat16, mil_amd.BF16X3):        # bf16x3 keeps an fp32 clone instead of s2d records
        net = _model(golden_dir, mode).eval()
        x = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(1)).clamp_(-1, 1).cuda()
        out = net(x, torch.tensor([1]).cuda())
        x.mul_(0.5)
        with pytest.raises(RuntimeError, match="modified in place"):
            out["loss"].backward()
        net.cnn.module.keep_s2d = True
        x0 = x.clone()
        net(x0, torch.tensor([1]).cuda())["loss"].backward()
        want = net.cnn.module.conv1.weight.grad.clone()
        net.cnn.module.conv1.weight.grad = None
        out = net(x, torch.tensor([1]).cuda())
        x.mul_(0.5)
        out["loss"].backward()
        assert float(want.abs().max()) > 0 and torch.equal(net.cnn.module.conv1.weight.grad, want), mode


def test_weight_update_through_data_needs_invalidate_packed_weights(golden_dir):
    """The encoder keeps MFMA-order copies of its filters, refreshed on parameter version / storage / epoch changes.  A write
    through `.data` (the reference does this at nnBlocks.py:375) bumps none of them: `mil_amd.invalidate_packed_weights()` is
    the documented hook, after which the next forward sees the new weights; an in-place op on the Parameter needs nothing."""
    import mil_amd
    net = _model(golden_dir, torch.bfloat16).eval()
    x = torch.randn(6, 3, 64, 64, generator=torch.Generator().manual_seed(4)).clamp_(-1, 1).cuda()
    with torch.no_grad():
        f0 = net.cnn(x).clone()
        net.cnn.module.layer1[0].conv1.weight.data.mul_(0.5)
        stale = net.cnn(x).clone()                      # packed copy not refreshed: documented behaviour
        mil_amd.invalidate_packed_weights()
        f1 = net.cnn(x).clone()
        net.cnn.module.layer1[0].conv1.weight.mul_(2.0)   # in-place on the Parameter: version counter moves
        f2 = net.cnn(x).clone()
    assert torch.equal(stale, f0) and not torch.equal(f1, f0)
    assert float((f2 - f0).abs().max()) <= 2e-2 * float(f0.abs().max())
