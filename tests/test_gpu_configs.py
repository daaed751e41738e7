"""-m gpu: the BASELINE.json configurations at (or near) their own sizes, through the drop-in module.

  configs[0]  1 bag x 64 tiles @256^2          HIP fp32 AND bf16 vs the committed reference golden (eval_n64_256_cfg1.npz)
  configs[1]  bags of 256 tiles @256^2 (bf16)  one full 256-tile bag vs the fp32 CPU oracle: the north star's 1e-3 gate,
                                               measured on the path bench.py times (bf16) and on the fp32 path
  configs[2]  512^2 tiles, 128 tiles/bag       fp32 vs oracle on 6 tiles; bf16 8x128 tiles: properties (crosses 2 GiB launches)
  configs[4]  1 bag x 4096 tiles, fwd only     attention weights vs the CPU oracle on the FULL bag; tile-sliced encode == whole
  fused vs un-fused kernel sequencing (bf16)   forward bit-identical, every gradient within one extra bf16 rounding

Tolerances: north star = 1e-3 absolute on logits (Mterm) / attention weights (Aterm) / class probabilities against the
fp32 CPU reference.  It is asserted on the fp32 kernel path everywhere (measured: 1e-6..5e-5).  The bf16 path stores every
activation with 8 significant bits through 26 un-normalised layers; the CPU oracle run with bf16 STORAGE EMULATED (plain
fp32 arithmetic, tensors rounded where the HIP path stores bf16) shows the same deviation from the fp32 reference as the
HIP kernels do (Mterm ~9e-2 of |2|, Aterm ~1e-3 of 1.5e-2 at 64 tiles; it enters in every stage about equally — keeping
layers 3-4 in fp32 would only halve it), so it is the price of the storage format, not of the kernels.  bf16 is therefore
asserted against stated bounds of ~2x the measured deviation, printed on every run, and BASELINE.md carries the numbers:
attention weights meet 1e-3 absolute from 256 tiles per bag up; class probabilities meet it at every size; the logits do not.
"""
import os

import numpy as np
import pytest
import torch

from fixture_inputs import synth_bag
from gpu_util import X3
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu


def _weights(golden_dir):
    return np.load(os.path.join(golden_dir, "weights.npz"))


def _model(golden_dir, dtype, class_weights=None):
    import mil_amd
    w = _weights(golden_dir)
    net = mil_amd.Attention(3, class_weights=class_weights, compute_dtype=dtype)
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()}, strict=True)
    return net.eval()


def _maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def _rel(a, b):
    return _maxabs(a, b) / max(float(np.abs(np.asarray(b, np.float64)).max()), 1e-30)


def _np(t):
    return t.detach().float().cpu().numpy()


def _oracle_features(sd, x, chunk=64):
    with torch.no_grad():
        return torch.cat([orc.backbone(sd, x[i:i + chunk]) for i in range(0, x.shape[0], chunk)])


# ---- configs[0]: the committed reference golden at the configuration's own size ---------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3], ids=["fp32", "bf16", "bf16x3"])
def test_config1_golden_64_tiles_256(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, "eval_n64_256_cfg1.npz"))
    x = synth_bag(64, 256, 256, 20260104)                    # the fixture stores no input: rebuilt from its seed
    net = _model(golden_dir, dtype)
    out = net(x.cuda(), torch.tensor(g["y"]).cuda())
    out["loss"].backward()
    err = {k: _maxabs(_np(out[k]), g["out." + k]) for k in ("Aterm", "Mterm", "y_pred", "loss", "wROIs", "Bterm")}
    frel = _rel(_np(out["Fterm"]), g["out.Fterm"])
    names = list(g["gradnorm.names"])
    params = dict(net.named_parameters())
    assert names == list(params.keys())
    gerr = {k: abs(float(params[k].grad.double().norm()) - n) / max(n, 1e-3) for k, n in zip(names, g["gradnorm.l2"])}
    worst = max(gerr, key=gerr.get)
    print(f"cfg1[{dtype}]: abs err {err}  Fterm rel {frel:.2e}  worst grad-norm rel {gerr[worst]:.2e} ({worst})")
    assert int(out["y_pred_hat"]) == int(g["out.y_pred_hat"]) and float(out["error"]) == float(g["out.error"][0])
    assert abs(float(out["Aterm"].sum(1).sub(1).abs().max())) < 1e-5
    if dtype == torch.float32:
        for k, e in err.items():
            assert e < 1e-3, (k, e)                                           # north-star gate
        for k in ("Aterm", "Mterm", "Bterm", "loss", "KLD", "Aterm_mu", "l2"):
            assert _rel(_np(out[k]), g["out." + k]) < 2e-4, k
        assert frel < 5e-5
        # gradient norms: 5e-3 — the bias gradients are cancellation-heavy sums over 64*64*64 pixels, where fp32 summation
        # order alone moves the reference itself by 5e-4 (tests/test_oracle_golden.py); measured here 1.2e-3
        assert gerr[worst] < 5e-3, (worst, gerr[worst])
    elif dtype == X3:
        # split-precision path (fp32 tensors, bf16x3 products): the north-star gate on every output, with margin — the CPU
        # emulation of this arithmetic (tools/numerics_formats.py) gives Mterm 2.5e-4, Aterm 1.6e-6, y_pred 8e-7, Bterm 5.6e-4
        for k, e in err.items():
            assert e < 1e-3, (k, e)
        assert err["Aterm"] < 2e-5 and err["y_pred"] < 2e-5 and err["loss"] < 2e-5, err
        assert frel < 5e-5
        assert gerr[worst] < 2e-2, (worst, gerr[worst])     # measured 5.4e-3 on a bias gradient (cancellation-heavy pixel sum)
    else:
        # measured on MI355X: Aterm 1.1e-3, Mterm 9.8e-2, y_pred 9.4e-4, loss 2.0e-3, wROIs 5.6e-3, Bterm 0.20, Fterm 4.7e-3,
        # worst gradient norm 12 % (a bias gradient) — the emulating oracle gives the same figures (see module docstring)
        assert err["Aterm"] < 3e-3 and err["wROIs"] < 1.2e-2 and err["Bterm"] < 0.5, err
        assert err["y_pred"] < 3e-3 and err["Mterm"] < 0.2 and err["loss"] < 5e-3, err
        assert frel < 1e-2
        assert gerr[worst] < 0.3, (worst, gerr[worst])


# ---- configs[1]: one bag of the benchmark's shape, against the fp32 oracle --------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3], ids=["fp32", "bf16", "bf16x3"])
def test_config2_bag_256_tiles_vs_oracle(golden_dir, dtype):
    """One 256-tile bag @256x256 — the unit bench.py's step is made of — through the path bench.py times (bf16) and the
    fp32 path, against the fp32 CPU oracle, forward and backward."""
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    n = 256
    x = synth_bag(n, 256, 256, 20260111)
    y = torch.tensor([2])
    sd = orc.load_state(_weights(golden_dir), requires_grad=True)
    ref = orc.attention_forward(sd, x, y)
    ref["loss"].backward()
    net = _model(golden_dir, dtype)
    out = net(x.cuda(), y.cuda())
    out["loss"].backward()
    err = {k: _maxabs(_np(out[k]), ref[k].detach().numpy()) for k in ("Aterm", "Mterm", "y_pred", "loss", "wROIs", "Bterm")}
    arel = _rel(_np(out["Aterm"]), ref["Aterm"].numpy())
    frel = _rel(_np(out["Fterm"]), ref["Fterm"].numpy())
    gerr = {}
    for k, p in net.named_parameters():
        nref = float(sd[k].grad.double().norm())
        gerr[k] = abs(float(p.grad.double().norm()) - nref) / max(nref, 1e-3)
    worst = max(gerr, key=gerr.get)
    print(f"cfg2-bag[{dtype}]: abs err {err}  Aterm rel {arel:.2e}  Fterm rel {frel:.2e}  worst grad-norm rel "
          f"{gerr[worst]:.2e} ({worst})")
    assert int(out["y_pred_hat"]) == int(ref["y_pred_hat"])
    if dtype in (torch.float32, X3):
        for k, e in err.items():
            assert e < 1e-3, (k, e)
        assert arel < 1e-3 and frel < 5e-5, (arel, frel)
        assert gerr[worst] < (5e-3 if dtype == torch.float32 else 2e-2), (worst, gerr[worst])     # fp32: measured 2.4e-3 on a bias gradient (16.8 M-term fp32 sums)
        return
    # bf16: measured on MI355X — Aterm 3.2e-4 (3.5 % of the largest weight), Mterm 9.3e-2, y_pred 7.8e-4, loss 6.6e-4,
    # wROIs 2.1e-3, Bterm 0.25, Fterm 4.4e-3, worst gradient norm 23 % (a layer-1 bias gradient)
    assert err["Aterm"] < 1e-3, err                                       # attention weights: north-star gate met in bf16
    assert arel < 8e-2                                                    # and relative, since weights are ~1/256
    assert err["y_pred"] < 2e-3 and err["loss"] < 2e-3, err               # class probabilities: met within 2x
    assert err["Mterm"] < 0.2 and err["Bterm"] < 0.5 and err["wROIs"] < 5e-3, err     # logits: NOT met in bf16 (stated bound)
    assert frel < 1e-2 and gerr[worst] < 0.45, (frel, worst, gerr[worst])
    # the same arithmetic with bf16 storage emulated on the CPU deviates from fp32 by the same amounts: what is left
    # between the HIP path and that emulation is accumulation order (and the bf16 flips it triggers downstream)
    sde = orc.load_state(_weights(golden_dir), requires_grad=True)
    emu = orc.attention_forward(sde, x, y, emulate_bf16=True)
    emu["loss"].backward()
    dev_emu = {k: _maxabs(emu[k].detach().numpy(), ref[k].detach().numpy()) for k in ("Aterm", "Mterm", "Bterm")}
    vs_emu = {k: _maxabs(_np(out[k]), emu[k].detach().numpy()) for k in ("Aterm", "Mterm", "Bterm")}
    cos = {}
    for k, p in net.named_parameters():
        if not (k.startswith("cnn.module.") and p.dim() == 4):
            # the head's buffer-branch gradients are cancellation-dominated (dB_n = sum_k dM_k A1[n,k] with sum_k dM_k = 0
            # and three nearly uniform maps: `buffer.classifier.bias` is analytically zero, `buffer.lin1.bias` flips sign
            # under a 6e-3 change of the logits), and conv biases are cancellation-heavy pixel sums: direction is
            # compared on the conv FILTER gradients, which is where a kernel error would show
            continue
        a, b = p.grad.detach().cpu().double().flatten(), sde[k].grad.double().flatten()
        cos[k] = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
    wc = min(cos, key=cos.get)
    print(f"cfg2-bag[bf16]: emulating oracle vs fp32 reference {dev_emu}; HIP vs emulating oracle {vs_emu}; "
          f"worst gradient cosine vs emulation {cos[wc]:.4f} ({wc})")
    for k in dev_emu:
        assert vs_emu[k] < 1.5 * max(dev_emu[k], err[k]), (k, vs_emu[k], dev_emu[k])    # no further from the emulation than bf16 is from fp32
    assert cos[wc] > 0.8, (wc, cos[wc])


# ---- gradient tolerances anchored to fp64 ----------------------------------------------------------------------------
def _grad_errors(grads, g64):
    """Per parameter tensor: L2 error relative to the fp64 gradient's norm."""
    out = {}
    for k, g in g64.items():
        n = float(g.norm())
        out[k] = float((grads[k].double() - g).norm()) / max(n, 1e-300)
    return out


@pytest.mark.parametrize("dtype", [torch.float32, X3], ids=["fp32", "bf16x3"])
@pytest.mark.parametrize("case", [(64, 20260104, 1), (256, 20260111, 2)], ids=["cfg1-64", "bag-256"])
def test_gradients_anchored_to_fp64(golden_dir, dtype, case):
    """How far may a gradient be from the reference?  The fp32 CPU oracle is itself a rounded computation: its bias gradients
    are cancellation-heavy sums over up to 16.8 M pixels and move by ~1e-3 with the summation order alone.  So the yardstick
    is an fp64 run of the same arithmetic (oracle with double weights and input): for every one of the 65 parameter
    gradients the distance of the HIP result from fp64 is printed next to the fp32 oracle's own (MIL_TEST_VERBOSE=1: all 65).
    What these numbers are made of: where no LeakyReLU branch flips the two agree within 2x (1e-5..1e-4); ONE flipped element
    (its pre-activation within 1e-6 of zero relative to the map's scale) lifts everything upstream to 2e-3..6e-3 — in the HIP
    path and in the fp32 oracle alike (measured: HIP one flip at layer4.0, the oracle one at layer3.2).  Asserted: 3x the
    oracle's own distance or 2e-2, whichever is larger."""
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    n, seed, label = case
    x = synth_bag(n, 256, 256, seed)
    y = torch.tensor([label])
    w = _weights(golden_dir)
    sd32 = orc.load_state(w, requires_grad=True)
    orc.attention_forward(sd32, x, y)["loss"].backward()
    sd64 = {k: torch.tensor(w[k], dtype=torch.float64, requires_grad=True) for k, _s in orc.state_dict_spec()}
    ref64 = orc.attention_forward(sd64, x.double(), y)
    ref64["loss"].backward()
    g64 = {k: v.grad for k, v in sd64.items()}
    net = _model(golden_dir, dtype)
    out = net(x.cuda(), y.cuda())
    out["loss"].backward()
    e_orc = _grad_errors({k: v.grad for k, v in sd32.items()}, g64)
    e_hip = _grad_errors({k: p.grad.detach().cpu() for k, p in net.named_parameters()}, g64)
    # analytically zero gradients (buffer.classifier.bias: sum_k dM_k = 0; buffer.lin1.bias with it): both sides hold rounding noise
    gmax = max(float(g.norm()) for g in g64.values())
    live = [k for k, g in g64.items() if float(g.norm()) > 1e-9 * gmax]
    worst_h = sorted(live, key=e_hip.get, reverse=True)[:6]
    print(f"fp64 anchor [{dtype}, {n} tiles]: Mterm vs fp64 {_maxabs(_np(out['Mterm']), ref64['Mterm'].detach().numpy()):.2e}; "
          "worst HIP gradients (L2 rel. to fp64; fp32 oracle in brackets): " +
          ", ".join(f"{k} {e_hip[k]:.2e} [{e_orc[k]:.2e}]" for k in worst_h))
    if os.environ.get("MIL_TEST_VERBOSE"):
        for k in live:
            print(f"    {k:44s} hip {e_hip[k]:.2e}  oracle32 {e_orc[k]:.2e}  |g64| {float(g64[k].norm()):.3e}")
    for k in live:
        # a single flipped LeakyReLU branch (pre-activation within rounding of zero) costs up to ~1e-2 here, in the HIP path
        # and in the fp32 oracle alike; the kernels' own error is bounded in test_encoder_gradients_on_its_own_activation_pattern
        # (bf16x3: forward error 1e-5 of the scale instead of 1e-6 -> 5-20 flipped elements per map instead of 0-2: measured 3.4e-2)
        assert e_hip[k] <= max(3.0 * e_orc[k], 2e-2 if dtype == torch.float32 else 8e-2), (k, e_hip[k], e_orc[k])


def _hip_patterns(saved):
    """Activation pattern of a HIP encoder run (the tensors its forward saved for the backward) in the form
    `orc.backbone(patterns=...)` takes."""
    widx = saved["widx"][..., :20].permute(0, 3, 1, 2).cpu()
    pat = {"stem_tap": (widx & 15).long(), "stem_pos": ((widx >> 4) & 1) == 0}
    names = [f"layer{li}.{b}" for li in range(1, 5) for b in range(3)]
    for name, (_xin, o1, out) in zip(names, saved["blocks"]):
        c = orc.STAGES[int(name[5]) - 1][1]
        pat[name + ".o1"] = (o1[..., :c].float() > 0).permute(0, 3, 1, 2).cpu()
        pat[name] = (out[..., :c].float() > 0).permute(0, 3, 1, 2).cpu()
    return pat


@pytest.mark.parametrize("n", [24, 40], ids=["24-tiles-persistent-forced", "40-tiles-grids-wrap"])
@pytest.mark.parametrize("dtype", [torch.float32, X3, torch.bfloat16], ids=["fp32", "bf16x3", "bf16"])
def test_encoder_gradients_on_its_own_activation_pattern(golden_dir, dtype, n, monkeypatch):
    """The encoder's vector-Jacobian product against fp64 ON THE SAME LINEAR PIECE.  ResNet-26 with LeakyReLU and max-pool is
    piecewise linear; two correct fp32 evaluations differ in the branch of the few elements whose pre-activation is within
    rounding of zero, and ONE such element in a 327,680-element map moves the gradients of everything upstream by 5e-3 of
    their norm (test_gradients_anchored_to_fp64 prints it; tools/diag_mask_flips.py counts the flips).  That is a property
    of the function, not of the kernels.  Here the fp64 oracle is evaluated on the activation pattern the HIP forward
    actually took (pool winners, both LeakyReLU masks of every block — read from the tensors the forward saved), so what is
    left is the arithmetic of the kernels: every one of the 54 encoder gradients (L2 error relative to the tensor's norm)
    within 1e-5 of fp64 in fp32 (measured 1.4e-6), 2e-4 with bf16x3 products (measured 5e-5), 4e-2 with bf16 storage
    (measured 1.3e-2: eight significant bits per stored tensor through 26 layers — the 12-23 % gradient-norm deviations the
    end-to-end bf16 tests see are branch flips, not bf16-rounded sums).
    n = 24 forces the persistent kernels onto these small launches (MIL_PF_MIN_TILES=1); n = 40 runs the production dispatch
    at a size where the stem (2560 tiles) and the 20-channel kernels (640 tiles) have more tiles than resident workgroups
    (512), so every persistent workgroup walks several tiles with its prefetch in flight."""
    from mil_amd import _lib as L
    from mil_amd import encoder
    if n == 24:
        monkeypatch.setenv("MIL_PF_MIN_TILES", "1")
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    x = synth_bag(n, 256, 256, 20260131)
    w = _weights(golden_dir)
    net = _model(golden_dir, dtype)
    enc = net.cnn.module
    st = L.storage_dtype(dtype)
    dfe = torch.randn(n, 80, generator=torch.Generator().manual_seed(8))
    with L.f32_mma(L.mma_code(dtype)):
        with torch.no_grad():
            feats, saved = encoder.encoder_forward(enc, x.cuda(), st)
        pat = _hip_patterns(saved)
        grads = encoder.encoder_backward(enc, saved, dfe.cuda(), st)
    torch.cuda.synchronize()
    names = [k[len("cnn.module."):] for k, _s in orc.state_dict_spec() if k.startswith("cnn.module.")]
    assert len(names) == len(grads)
    sd64 = {k: torch.tensor(w[k], dtype=torch.float64, requires_grad=True) for k, _s in orc.state_dict_spec()}
    f64 = orc.backbone(sd64, x.double(), patterns=pat)
    f64.backward(dfe.double())
    ferr = float((feats.double().cpu() - f64.detach()).abs().max() / f64.detach().abs().max())
    errs = {}
    for k, g in zip(names, grads):
        g64 = sd64["cnn.module." + k].grad
        errs[k] = float((g.double().cpu() - g64).norm() / g64.norm())
    worst = sorted(errs, key=errs.get, reverse=True)[:3]
    print(f"own-pattern VJP [{dtype}]: features {ferr:.2e}; worst gradients " + ", ".join(f"{k} {errs[k]:.2e}" for k in worst))
    # measured on MI355X: fp32 4.2e-7 / 1.35e-6 (conv1.weight); bf16x3 4.2e-6 / 5.0e-5; bf16 4.3e-3 / 1.3e-2
    ftol, gtol = {torch.float32: (2e-6, 1e-5), X3: (2e-5, 2e-4), torch.bfloat16: (1e-2, 4e-2)}[dtype]
    assert ferr < ftol, ferr
    for k, e in errs.items():
        assert e < gtol, (k, e)


# ---- configs[2]: 512x512 tiles ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, X3], ids=["fp32", "bf16x3"])
def test_config3_512_tiles_fp32_vs_oracle(golden_dir, dtype):
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    x = synth_bag(6, 512, 512, 20260112)
    y = torch.tensor([0])
    sd = orc.load_state(_weights(golden_dir), requires_grad=True)
    ref = orc.attention_forward(sd, x, y)
    ref["loss"].backward()
    net = _model(golden_dir, dtype)
    out = net(x.cuda(), y.cuda())
    out["loss"].backward()
    for k in ("Aterm", "Mterm", "y_pred", "loss", "wROIs", "Bterm"):
        e = _maxabs(_np(out[k]), ref[k].detach().numpy())
        print(f"cfg3[{dtype}] {k}: abs err {e:.2e}")
        assert e < 1e-3, k
    for k in ("Aterm", "Fterm", "Mterm", "Bterm", "KLD", "loss"):
        assert _rel(_np(out[k]), ref[k].detach().numpy()) < (2e-4 if dtype == torch.float32 else 1e-3), k
    for k, p in net.named_parameters():
        nref = float(sd[k].grad.double().norm())
        assert abs(float(p.grad.double().norm()) - nref) <= (2e-3 if dtype == torch.float32 else 2e-2) * max(nref, 1e-3), k


@pytest.mark.parametrize("dtype", [torch.bfloat16, X3], ids=["bf16", "bf16x3"])
def test_config2_full_size_properties(golden_dir, dtype):
    """8 bags x 256 tiles @256x256 = 2048 tiles in ONE launch sequence — BASELINE configs[1], the headline workload, exactly as
    bench.py steps it (grid sizes, 256 resident image groups, slab counts of the 2048-tile launches).  Size-independent
    properties: finite everywhere, every attention map sums to 1, two runs bit-identical (no atomics, fixed-order slab
    reductions), the first and last tiles' features equal their stand-alone encoding, and bag 0 of the 8-bag step equals the
    same bag run alone: logits within 1e-3 on the split-precision path (fp32 storage: other tile shapes only regroup fp32
    sums), within bf16 noise on the bf16 path."""
    import mil_amd
    net = _model(golden_dir, dtype)
    flat = mil_amd.FlatParams(net)
    x = torch.empty((2048, 3, 256, 256), dtype=torch.float32, device="cuda")
    for b in range(8):
        gen = torch.Generator(device="cuda").manual_seed(20260104 + b)
        x[b * 256:(b + 1) * 256] = torch.randn((256, 3, 256, 256), generator=gen, device="cuda").clamp_(-1, 1)
    sizes, labels = [256] * 8, torch.tensor([b % 3 for b in range(8)], device="cuda")
    runs = []
    for _ in range(2):
        flat.zero_grad()
        outs = net.forward_bags((x, sizes), labels)
        outs.loss.sum().backward()
        torch.cuda.synchronize()
        runs.append((outs.loss.detach().clone(), torch.cat([o["Aterm"].reshape(-1) for o in outs]).clone(),
                     flat.flat_grad.clone(), outs[7]["Fterm"][-4:].clone(), outs[0]["Mterm"].clone(), outs[0]["Aterm"].clone()))
    for o in outs:
        assert torch.isfinite(o["loss"]) and torch.isfinite(o["Fterm"]).all() and torch.isfinite(o["Aterm"]).all()
        assert torch.allclose(o["Aterm"].sum(dim=1), torch.ones(3, device="cuda"), atol=1e-5)
    assert torch.isfinite(runs[0][2]).all() and float(runs[0][2].abs().max()) > 0
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
    ftol = 1e-4 if dtype == X3 else 3e-2
    with torch.no_grad():
        alone_first, alone_last = net.cnn(x[:4]), net.cnn(x[-4:])              # generic (non-persistent) kernels
        bag0 = net(x[:256], labels[:1])                                       # the same bag as a 256-tile launch
    for a, b in ((alone_first, outs[0]["Fterm"][:4]), (alone_last, runs[0][3])):
        assert float((a - b).abs().max() / b.abs().max()) < ftol
    m_err = float((bag0["Mterm"] - runs[0][4]).abs().max())
    a_err = float((bag0["Aterm"] - runs[0][5]).abs().max())
    print(f"cfg2-full[{dtype}]: bag 0 alone vs inside the 8-bag step: Mterm {m_err:.2e}  Aterm {a_err:.2e}")
    assert a_err < 1e-3
    assert m_err < (1e-3 if dtype == X3 else 0.2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, X3], ids=["bf16", "bf16x3"])
def test_config3_full_size_properties(golden_dir, dtype):
    """8 bags x 128 tiles @512x512 in ONE launch sequence (BASELINE configs[2] as bench.py --size 512 --tiles 128 runs it):
    layer-1 tensors are 2.1 GiB in bf16 and 4.3 GiB on the split-precision path (fp32 tensors: twice across the 2 GiB reach of
    a buffer descriptor), so the kernels that address through 32-bit buffer offsets split their launches.
    Size-independent properties: finite everywhere, every attention map sums to 1, a tile's features do not depend on
    the launch it shares (first tiles re-encoded alone; 1e-4 on the split-precision path), two runs bit-identical (no atomics,
    fixed-order reductions)."""
    import mil_amd
    net = _model(golden_dir, dtype)
    flat = mil_amd.FlatParams(net)
    gen = torch.Generator(device="cuda").manual_seed(321)
    x = torch.empty((1024, 3, 512, 512), dtype=torch.float32, device="cuda")
    for b in range(8):
        x[b * 128:(b + 1) * 128] = torch.randn((128, 3, 512, 512), generator=gen, device="cuda").clamp_(-1, 1)
    sizes, labels = [128] * 8, torch.tensor([b % 3 for b in range(8)], device="cuda")
    runs = []
    for _ in range(2):
        flat.zero_grad()
        outs = net.forward_bags((x, sizes), labels)
        outs.loss.sum().backward()
        torch.cuda.synchronize()
        runs.append((outs.loss.detach().clone(), torch.cat([o["Aterm"].reshape(-1) for o in outs]).clone(),
                     flat.flat_grad.clone(), outs[7]["Fterm"][-4:].clone()))
    for o in outs:
        assert torch.isfinite(o["loss"]) and torch.isfinite(o["Fterm"]).all() and torch.isfinite(o["Aterm"]).all()
        assert torch.allclose(o["Aterm"].sum(dim=1), torch.ones(3, device="cuda"), atol=1e-5)
    assert torch.isfinite(runs[0][2]).all() and float(runs[0][2].abs().max()) > 0
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
    with torch.no_grad():
        alone_first, alone_last = net.cnn(x[:4]), net.cnn(x[-4:])              # generic (non-persistent) kernels
    big_first, big_last = outs[0]["Fterm"][:4], runs[0][3]
    for a, b in ((alone_first, big_first), (alone_last, big_last)):           # last tiles sit beyond the 2 GiB mark
        assert float((a - b).abs().max() / b.abs().max()) < (1e-4 if dtype == X3 else 3e-2)


# ---- configs[4]: one 4096-tile bag, forward only ---------------------------------------------------------------------
def test_config5_full_bag_attention_map(golden_dir):
    """1 bag x 4096 tiles @256x256, forward only: the attention weights the reference's CPU arithmetic gives for the WHOLE
    bag (oracle, chunked encode under no_grad) against the fp32 and the bf16 HIP paths; and the tile-parallel form
    (slices encoded separately as 8 ranks would, features concatenated, one head) equals the single-launch form bit for bit."""
    from mil_amd.head import BagLayout, head_apply
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    n = 4096
    y = torch.tensor([1])
    sd = orc.load_state(_weights(golden_dir))
    xg = torch.empty((n, 3, 256, 256), dtype=torch.float32, device="cuda")
    feats = []
    for i in range(0, n, 256):                                # 256 tiles at a time: bounded host memory
        xc = synth_bag(256, 256, 256, 20260120 + i)
        xg[i:i + 256] = xc.cuda()
        feats.append(_oracle_features(sd, xc))
    with torch.no_grad():
        ref = orc.mil_head(sd, torch.cat(feats), y)
    for dtype, a_abs, a_rel in ((torch.float32, 1e-3, 1e-3), (X3, 1e-3, 1e-3), (torch.bfloat16, 1e-3, 0.1)):      # measured: 1.2e-8 / 1.8e-5; 3.0e-5 / 4.7e-2 (bf16)
        net = _model(golden_dir, dtype)
        with torch.no_grad():
            out = net(xg, y.cuda())
            e_abs = _maxabs(_np(out["Aterm"]), ref["Aterm"].numpy())
            e_rel = _rel(_np(out["Aterm"]), ref["Aterm"].numpy())
            m_abs = _maxabs(_np(out["Mterm"]), ref["Mterm"].numpy())
            print(f"cfg5[{dtype}]: Aterm abs {e_abs:.2e} rel {e_rel:.2e}  Mterm abs {m_abs:.2e}")
            assert e_abs < a_abs and e_rel < a_rel, (dtype, e_abs, e_rel)
            assert m_abs < (0.2 if dtype == torch.bfloat16 else 1e-3)      # bf16 logits: stated bound (measured 8.9e-2)
            assert int(out["y_pred_hat"]) == int(ref["y_pred_hat"])
            assert torch.allclose(out["Aterm"].sum(dim=1), torch.ones(3, device="cuda"), atol=1e-5)
            # what 8 ranks do (Attention.forward_tile_parallel): encode a slice each, gather, replicated head
            H = torch.cat([net.cnn(xg[r * 512:(r + 1) * 512]) for r in range(8)])
            _l, _l2, a1, *_ = head_apply(H, BagLayout([n], H.device), y.cuda(), None, None, net.head_weights())
            if dtype == torch.float32:          # fp32: the same bits whatever the launch a tile shares
                assert torch.equal(H, out["Fterm"]) and torch.equal(a1.t(), out["Aterm"])
            elif dtype == X3:                   # split products: other kernels / tile shapes for a 512-tile launch, fp32 storage
                assert float((H - out["Fterm"]).abs().max() / out["Fterm"].abs().max()) < 1e-4
                assert _maxabs(_np(a1.t()), ref["Aterm"].numpy()) < a_abs
            else:
                # bf16: a 512-tile launch picks other tile shapes / kernels for the small late maps than a 4096-tile one
                # (different MFMA summation order -> a bf16 store lands on the neighbouring value now and then): same map
                # within bf16 noise, and as close to the oracle as the single launch is
                assert float((H - out["Fterm"]).abs().max() / out["Fterm"].abs().max()) < 1e-2
                assert _rel(_np(a1.t()), _np(out["Aterm"])) < 5e-2
                assert _maxabs(_np(a1.t()), ref["Aterm"].numpy()) < a_abs
            one = net.forward_tile_parallel(xg, y.cuda())                 # world size 1: same entry point, no collective
            assert torch.equal(one["Aterm"], out["Aterm"]) and torch.equal(one["Mterm"], out["Mterm"])


# ---- fused fast path vs the un-fused kernel sequence (bf16) -------------------------------------------------------
def _bags_128():
    gen = torch.Generator(device="cuda").manual_seed(2024)
    x = torch.randn((96, 3, 128, 128), generator=gen, device="cuda").clamp_(-1, 1)
    return x, [40, 30, 26], torch.tensor([0, 1, 2], device="cuda")


def _set_flags(enc, **kw):
    for k in ("fuse_backward", "fuse_stem_forward", "fuse_stage_entry", "fuse_block_forward"):
        setattr(enc, k, kw.get(k, False))


def test_fused_forward_kernels_vs_unfused_sequence_bf16(golden_dir, monkeypatch):
    """encoder_forward chooses between fused kernels and the plain persistent conv sequence.  Sized so that every launch
    takes the persistent kernels and a workgroup walks several tiles.
      * whole-block forward: same MFMA order, same bf16 stores -> every saved activation and the features BIT-identical to
        the un-fused sequence;
      * fused stem (round 5: s2d + conv + max-pool of the fp32 accumulators in registers + lrelu; the un-fused chain pools the
        bf16-ROUNDED stem tensor): the pooled map agrees except where the position code in the low mantissa bits moves a bf16
        rounding (<= one bf16 step on < 1e-3 of the elements); features within 1e-2;
      * stage-entry pair (3x3/s2 conv + 1x1/s2 projection from one staged tile): its K order differs from the
        generic stride-2 conv, so a handful of o1 elements land on the neighbouring bf16 value (measured: 70 of 983 k,
        1 ulp); asserted as <= 1e-3 of the elements off by <= 1 ulp, features within 1e-2."""
    from mil_amd import encoder
    monkeypatch.setenv("MIL_PF_MIN_TILES", "1")
    x, _sizes, _labels = _bags_128()
    runs = {}
    for name, kw in (("none", {}), ("block", dict(fuse_block_forward=True)), ("stem", dict(fuse_stem_forward=True)),
                     ("entry", dict(fuse_stage_entry=True))):
        net = _model(golden_dir, torch.bfloat16)
        _set_flags(net.cnn.module, **kw)
        with torch.no_grad():
            runs[name] = encoder.encoder_forward(net.cnn.module, x, torch.bfloat16)
    f0, s0 = runs["none"]
    f1, s1 = runs["block"]
    assert torch.equal(f1, f0) and torch.equal(s1["widx"], s0["widx"])
    for bi, (a, b) in enumerate(zip(s1["blocks"], s0["blocks"])):
        assert all(torch.equal(ta, tb) for ta, tb in zip(a, b)), bi

    def one_ulp(a, b):
        a, b = a.float(), b.float()
        d = (a - b).abs()
        ulp = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -126).log2().floor().exp2() * 2.0 ** -7     # bf16: 8 significant bits
        return d, ulp

    f3, s3 = runs["stem"]
    d, ulp = one_ulp(s3["blocks"][0][0], s0["blocks"][0][0])           # the pooled map = block 0's input
    # (+ 2e-5: the bias joins the sum at the pooled value instead of in the accumulator — fp32 summation noise where conv and bias cancel)
    assert bool((d <= ulp + 2e-5).all()) and float((d > 0).float().mean()) < 1e-3
    assert float((f3 - f0).abs().max() / f0.abs().max()) < 1e-2
    f2, s2 = runs["entry"]
    assert torch.equal(s2["blocks"][3][0], s0["blocks"][3][0])      # layer-2 entry: identical input in both runs
    d, ulp = one_ulp(s2["blocks"][3][1], s0["blocks"][3][1])
    # one bf16 step, or — for results that nearly cancel — the fp32 summation noise of the accumulation itself
    assert bool((d <= ulp + 2e-5).all()), float((d - ulp).max())
    assert float((d > 0).float().mean()) < 1e-3
    for bi in (6, 9):                                               # later entries see inputs that already carry those flips
        a, b = s2["blocks"][bi][1].float(), s0["blocks"][bi][1].float()
        assert float((a - b).abs().max() / b.abs().max()) < 2e-2, bi
    assert float((f2 - f0).abs().max() / f0.abs().max()) < 1e-2


def test_fused_backward_sequencing_vs_unfused_bf16(golden_dir, monkeypatch):
    """encoder_backward's fused branches (one-pass dgrad+wgrad with addend/mask variants, parity-class stride-2 dgrad,
    paired stage-entry weight gradients, fused stem backward) against the plain dgrad / wgrad / pool-backward sequence ON
    THE SAME SAVED ACTIVATIONS (identical forward flags, so the forward is bit-identical and only the backward differs).
    The un-fused sequence rounds dz1 and the projection's addend to bf16 between launches where the fused kernels keep
    fp32 registers: per parameter tensor the gradients must agree within that one extra rounding — asserted as
    max-relative <= 5e-2 and cosine >= 0.999 on every FILTER (measured: 6e-3 / 0.99998 on conv1.weight) — which is what guards
    *which addend, which mask, which dz* (a wrong operand leaves no cosine to speak of).  BIAS gradients are sums of ~10^6
    bf16-rounded terms that cancel to a few percent of their mass, taken from dz tensors that already differ by those extra
    roundings: the two sequences land 3.6e-2 (layer-1) to 7.1e-2 (conv1.bias, cosine 0.9988: round 5, after the stem's pooling
    winners became the fp32 maxima — another routing, another sample of the same noise; both sequences sit 0.41 from the fp32
    gradient there, branch flips) apart: asserted as <= 0.12 and cosine >= 0.995."""
    monkeypatch.setenv("MIL_PF_MIN_TILES", "1")
    x, sizes, labels = _bags_128()
    res = {}
    for fused in (True, False):
        net = _model(golden_dir, torch.bfloat16)
        _set_flags(net.cnn.module, fuse_backward=fused, fuse_stem_forward=True, fuse_stage_entry=True, fuse_block_forward=True)
        outs = net.forward_bags((x, sizes), labels)
        outs.loss.sum().backward()
        torch.cuda.synchronize()
        res[fused] = (torch.cat([o["Fterm"] for o in outs]).clone(), outs.loss.detach().clone(),
                      {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])     # same forward
    worst, wcos = {1: (0.0, ""), 2: (0.0, "")}, {1: (2.0, ""), 2: (2.0, "")}
    for k, gf in res[True][2].items():
        gu = res[False][2][k]
        scale = float(gu.abs().max())
        if scale == 0.0:
            assert float(gf.abs().max()) == 0.0, k
            continue
        kind = 1 if gf.dim() == 1 else 2                    # bias-like vectors / filters and matrices
        worst[kind] = max(worst[kind], (float((gf - gu).abs().max()) / scale, k))
        a, b = gf.double().flatten(), gu.double().flatten()
        wcos[kind] = min(wcos[kind], (float(torch.dot(a, b) / (a.norm() * b.norm())), k))
    print("fused vs un-fused backward: worst max-relative gradient difference", worst, "worst cosine", wcos)
    assert worst[2][0] < 5e-2 and wcos[2][0] > 0.999, (worst, wcos)
    assert worst[1][0] < 0.12 and wcos[1][0] > 0.995, (worst, wcos)


def test_dense_gradient_layout_and_dropped_s2d_copy_change_no_bit(golden_dir, monkeypatch):
    """Two byte-saving choices of the bf16 encoder are pure layout: (a) the gradient tensors of the 20-channel stage at 20
    channels per pixel (MIL_DT_BF16_DGRAD) instead of the padded 24, (b) no space-to-depth copy of the input kept — the
    fused stem backward rebuilds its tiles from the fp32 input.  Every gradient must be BIT-identical with either choice
    switched off, at the benchmark's tile size (64x64 first-stage maps) and at 128x128 tiles (32x32 maps), and the dense
    chain must really have run (a [T,H,W,20] tensor reaches the fused stem backward)."""
    from mil_amd import ops
    monkeypatch.setenv("MIL_PF_MIN_TILES", "1")
    gen = torch.Generator().manual_seed(77)
    for size, sizes in ((256, [20, 12]), (128, [24, 40]), (300, [5, 4])):       # 300: the live driver's 75x75 first-stage maps (ragged tiles)
        x = torch.randn(sum(sizes), 3, size, size, generator=gen).clamp_(-1, 1).cuda()
        labels = torch.tensor([2, 0])
        grads, seen = {}, {}
        real = ops.stem_bwd_fused_nchw

        def spy(xx, g_pool, *a, **k):
            seen["c"] = g_pool.shape[-1]
            return real(xx, g_pool, *a, **k)
        monkeypatch.setattr(ops, "stem_bwd_fused_nchw", spy)
        for dense, keep in ((True, False), (False, False), (True, True)):
            net = _model(golden_dir, torch.bfloat16)
            net.cnn.module.dense_grads = dense
            net.cnn.module.keep_s2d = keep
            seen.clear()
            net.forward_bags((x, sizes), labels).loss.sum().backward()
            torch.cuda.synchronize()
            grads[(dense, keep)] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
            if not keep and size != 300:                    # (at 300x300 either layout may be chosen: only the equality below matters)
                assert seen.get("c") == (20 if dense else 24), (size, dense, seen)
            elif not keep and not dense:
                assert seen.get("c", 24) == 24, (size, seen)
        monkeypatch.setattr(ops, "stem_bwd_fused_nchw", real)
        ref = grads[(True, False)]
        for key in ((False, False), (True, True)):
            for k, g in grads[key].items():
                assert torch.equal(g, ref[k]), (size, key, k)
        assert all(float(g.abs().max()) > 0 for k, g in ref.items() if "conv" in k)


def test_batched_slab_reductions_are_bit_identical(golden_dir, monkeypatch):
    """The 28 weight-gradient slab reductions of a backward pass run as ONE launch (mil_reduce_defer_begin / _end /
    mil_wgrad_reduce_all) instead of one launch behind every producer: same summation trees, so every gradient must be
    BIT-identical to the per-call reductions — bf16 (fused kernels) and fp32 (plain dgrad / wgrad), persistent and
    generic launch sizes, accumulating into the flat bucket and through autograd's own accumulation."""
    import mil_amd
    x, sizes, labels = _bags_128()
    for dtype, pf in ((torch.bfloat16, "1"), (torch.float32, "1000000000"), (torch.bfloat16, "1000000000")):
        monkeypatch.setenv("MIL_PF_MIN_TILES", pf)
        grads = []
        for batched in (False, True):
            net = _model(golden_dir, dtype)
            net.cnn.module.batch_reductions = batched
            flat = mil_amd.FlatParams(net)
            for _ in range(2):                               # second pass accumulates on top of the first
                net.forward_bags((x[:48], [20, 28]), labels[:2]).loss.sum().backward()
            torch.cuda.synchronize()
            grads.append(flat.flat_grad.clone())
            assert (net.cnn.module._reduce_batch is not None) == batched
        assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0, dtype
    net = _model(golden_dir, torch.bfloat16)                 # autograd path (fresh dW tensors every call)
    out = net(x[:16], labels[:1])
    out["loss"].backward()
    ref = _model(golden_dir, torch.bfloat16)
    ref.cnn.module.batch_reductions = False
    ref(x[:16], labels[:1])["loss"].backward()
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        assert torch.equal(p.grad, q.grad), k
