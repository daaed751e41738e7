#!/usr/bin/env python3
"""Generate the golden vectors that pin the oracle and the HIP path.

Runs ONLY in the build container, where the upstream reference is mounted
read-only at /root/reference.  It imports the reference's own `gbm/model.py`
(`Attention`, `ResNet`) and `nnBlocks.py` (`BasicResBlock`,
`CrossEntropyWithProbs`) on CPU through the two-line shim of SURVEY.md §8(c)
(a stub `PyTorchHelpers` module; `.cuda()` made the identity), runs it on
seeded synthetic bags and stores *tensors only* (inputs, the 65 state-dict
entries, the 13 output-dict entries, stage activations, parameter gradients)
as .npz files next to this script.  Nothing from the reference (source,
bytecode, pickled modules) is stored.

    python tests/golden/make_golden.py

The fixtures are committed; the GPU box never runs this script.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    sys.path[:0] = [REF, os.path.join(REF, "gbm")]
    sys.modules["PyTorchHelpers"] = types.ModuleType("PyTorchHelpers")
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    with contextlib.redirect_stdout(io.StringIO()):
        import model  # noqa: the reference's gbm/model.py
    return model


def make_weights(model):
    """Reference init under seed 1234, then seeded non-zero biases / BN affine /
    weight_mask so that every term of the arithmetic is exercised."""
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(1234)
        net = model.Attention(3)
    g = torch.Generator().manual_seed(4321)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith(".bias") and "bn" not in k:
            v.copy_(0.05 * torch.randn(v.shape, generator=g))
    sd["context.bn.weight"].copy_(1.0 + 0.1 * torch.randn(80, generator=g))
    sd["context.bn.bias"].copy_(0.1 * torch.randn(80, generator=g))
    sd["weight_mask"].copy_(torch.tensor([0.25, -0.10, 0.05]))
    net.load_state_dict(sd)
    return net


def synth_bag(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, h, w, generator=g).clamp_(-1.0, 1.0)


def stage_hooks(net):
    acts = {}
    cnn = net.cnn.module
    hs = []

    def keep(name):
        def hook(_m, _i, o):
            acts[name] = o.detach().clone()
        return hook
    # conv1's output is modified in place by the LeakyReLU that follows, so hook the relu.
    hs.append(cnn.relu.register_forward_hook(keep("stem")))
    hs.append(cnn.maxpool.register_forward_hook(keep("pool")))
    for i in (1, 2, 3, 4):
        hs.append(getattr(cnn, f"layer{i}").register_forward_hook(keep(f"layer{i}")))
    return acts, hs


def run_case(model, net, name, x, y, *, train=False, seed=None, class_weights=None,
             full_grads=False, stages=False, store_x=True):
    if class_weights is not None:
        # the loss module reads a plain attribute (nnBlocks.py:65)
        net.loss.weight = class_weights
    else:
        net.loss.weight = None
    net.train(train)
    net.zero_grad(set_to_none=True)
    rec = {}
    extra_hooks = []
    if train:
        # gbm/model.py:193 draws randperm from the global RNG, then Dropout draws its mask.
        torch.manual_seed(seed)
        n0 = x.shape[0]
        rec["indices"] = torch.randperm(n0)[: int(n0 * 0.2)].clone()
        torch.manual_seed(seed)

        def do_hook(_m, i, o):
            rec["keep_mask"] = (o != 0).to(torch.uint8)
        extra_hooks.append(net.context.do.register_forward_hook(do_hook))
    acts, hs = stage_hooks(net) if stages else ({}, [])
    out = net(x, y)
    out["loss"].backward()
    for h in hs + extra_hooks:
        h.remove()

    blob = {}
    if store_x:
        blob["x"] = x.numpy()
    blob["y"] = y.numpy()
    for k, v in out.items():
        blob["out." + k] = v.detach().numpy()
    for k, v in acts.items():
        blob["act." + k] = v.numpy()
    for k, v in rec.items():
        blob["rec." + k] = v.numpy()
    if class_weights is not None:
        blob["class_weights"] = class_weights.numpy()
    names, norms, sums = [], [], []
    for k, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(k)
        norms.append(float(g.double().norm()))
        sums.append(float(g.double().sum()))
        if full_grads:
            blob["grad." + k] = g.numpy()
    blob["gradnorm.names"] = np.array(names)
    blob["gradnorm.l2"] = np.array(norms, dtype=np.float64)
    blob["gradnorm.sum"] = np.array(sums, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    print(f"{name}: loss={out['loss'].item():.6f} y_pred={out['y_pred'].numpy().round(4)} "
          f"|Fterm|max={out['Fterm'].abs().max().item():.4f}")


def load_reference_alt():
    """alt_resnet.py has a relative import of `.utils` (a URL loader it only needs for pretrained=True): give it a
    package context with a stub whose loader raises."""
    import importlib.util
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = [REF]
    sys.modules["refpkg"] = pkg
    utils = types.ModuleType("refpkg.utils")

    def _offline(*_a, **_k):
        raise RuntimeError("pretrained weights are a network fetch; unavailable offline")
    utils.load_state_dict_from_url = _offline
    sys.modules["refpkg.utils"] = utils
    spec = importlib.util.spec_from_file_location("refpkg.alt_resnet", os.path.join(REF, "alt_resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refpkg.alt_resnet"] = mod
    spec.loader.exec_module(mod)
    return mod


def run_alt_case(alt, name, layers, num_classes, wseed, x):
    """Reference alt_resnet on seeded weights (oracle/mil_oracle.alt_seeded_state regenerates them, nothing to store)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import mil_oracle as orc
    net = alt.ResNet(alt.BasicBlock, list(layers), num_classes=num_classes)
    sd = orc.alt_seeded_state(layers, num_classes, wseed)
    assert list(sd.keys()) == list(net.state_dict().keys())
    net.load_state_dict(sd)
    feats = net(x)
    g = torch.Generator().manual_seed(wseed + 1)
    dfe = torch.randn(feats.shape, generator=g)
    feats.backward(dfe)
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x.numpy(), feats=feats.detach().numpy(), dfeats=dfe.numpy(),
                        layers=np.array(layers), num_classes=np.array(num_classes), wseed=np.array(wseed),
                        **{"gradnorm.names": np.array(names), "gradnorm.l2": np.array(norms, dtype=np.float64),
                           "grad.conv1.weight": net.conv1.weight.grad.numpy(),
                           "grad.layer2.0.downsample.0.weight": net.layer2[0].downsample[0].weight.grad.numpy(),
                           "grad.fc.bias": net.fc.bias.grad.numpy()})
    print(f"{name}: |feats|max={feats.abs().max().item():.4f}")


def run_prep_case(name, n_tiles, roi, pad, res, seed, train, store="f32"):
    """The tile pre-processing chain of RoiBuilder.py:193-210 run with Pillow itself (the library the reference's
    torchvision transforms call for PIL images) and torch's own ToTensor/Normalize arithmetic.  torchvision is not
    importable here, so its PIL-backend calls are spelled out: Pad(p) = ImageOps.expand(border=p, fill=0); RandomCrop =
    Image.crop; Resize(r) = Image.resize((r, r), BILINEAR); flips = Image.transpose.  Inputs are regenerated from `seed`."""
    from PIL import Image, ImageOps
    rng = np.random.default_rng(seed)
    rois = rng.integers(0, 256, (n_tiles, roi, roi, 3), dtype=np.uint8)
    # smooth gradients in half of the tiles: resampling bugs that noise hides show up on ramps
    ramp = (np.add.outer(np.arange(roi), 2 * np.arange(roi)) % 256).astype(np.uint8)
    rois[::2, :, :, 1] = ramp
    params = np.zeros((n_tiles, 4), dtype=np.int32)
    if train:
        params[:, 0] = rng.integers(0, 2 * pad + 1, n_tiles)
        params[:, 1] = rng.integers(0, 2 * pad + 1, n_tiles)
        params[:, 2] = rng.integers(0, 2, n_tiles)
        params[:, 3] = rng.integers(0, 2, n_tiles)
        params[0] = (0, 2 * pad, 1, 0)                      # extreme crops: all padding on one side
        if n_tiles > 1:
            params[1] = (2 * pad, 0, 0, 1)
    outs_u8, outs = [], []
    for t in range(n_tiles):
        img = Image.fromarray(rois[t])
        if train:
            top, left, hf, vf = (int(v) for v in params[t])
            img = ImageOps.expand(img, border=pad, fill=0)
            img = img.crop((left, top, left + roi, top + roi))
        img = img.resize((res, res), Image.BILINEAR)
        if train and hf:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if train and vf:
            img = img.transpose(Image.FLIP_TOP_BOTTOM)
        arr = np.array(img)
        outs_u8.append(arr)
        ten = torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)      # ToTensor
        ten.sub_(0.5).div_(0.5)                                                                   # Normalize(.5,.5)
        outs.append(ten.numpy())
    blob = dict(roi=np.array(roi), pad=np.array(pad), res=np.array(res), seed=np.array(seed), train=np.array(int(train)),
                n_tiles=np.array(n_tiles), params=params)
    if store == "f32":
        blob["out"] = np.stack(outs)
    else:
        blob["out_u8"] = np.stack(outs_u8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    print(f"{name}: {n_tiles} tiles {roi}->{res}")


def main():
    run_prep_case("prep_s120_r32_train", 6, 120, 10, 32, 31, True)
    run_prep_case("prep_s100_r37_flat", 3, 100, 10, 37, 32, False)
    run_prep_case("prep_s50_r80_train", 2, 50, 7, 80, 33, True)                  # up-sampling
    run_prep_case("prep_s1200_r300_train", 2, 1200, 100, 300, 34, True, store="u8")    # the reference driver's sizes
    alt = load_reference_alt()
    run_alt_case(alt, "alt_l1111_n4_64", (1, 1, 1, 1), 80, 555, synth_bag(4, 64, 64, 20260201))
    run_alt_case(alt, "alt_l2222_n2_96x80", (2, 2, 2, 2), 80, 556, synth_bag(2, 96, 80, 20260202))
    model = load_reference()
    net = make_weights(model)
    np.savez_compressed(os.path.join(HERE, "weights.npz"),
                        **{k: v.numpy() for k, v in net.state_dict().items()})
    print("weights:", len(net.state_dict()), "tensors,",
          sum(v.numel() for v in net.state_dict().values()), "params")

    y1 = torch.tensor([1])
    y2 = torch.tensor([2])
    y0 = torch.tensor([0])
    x8 = synth_bag(8, 64, 64, 20260104)
    run_case(model, net, "eval_n8_64", x8, y1, full_grads=True, stages=True)
    run_case(model, net, "eval_n8_64_cw", x8, y2,
             class_weights=torch.tensor([0.5, 1.0, 2.0]))
    run_case(model, net, "eval_n5_50x70", synth_bag(5, 50, 70, 20260105), y0, stages=True)
    run_case(model, net, "train_n40_64", synth_bag(40, 64, 64, 20260106), y2,
             train=True, seed=77, full_grads=True)
    run_case(model, net, "eval_n2_256", synth_bag(2, 256, 256, 20260107), y1)
    # BASELINE.json configs[0]: 1 bag x 64 tiles @256x256; input is regenerated from its seed.
    run_case(model, net, "eval_n64_256_cfg1", synth_bag(64, 256, 256, 20260104), y1,
             store_x=False)


if __name__ == "__main__":
    main()
