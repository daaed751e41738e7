"""CPU oracle for the ResNet-26 + attention-MIL hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  The shipped path
(`mil_amd`) never falls back to it and raises if its HIP library is missing.

It restates, in plain fp32 PyTorch on the CPU and as pure functions over a state
dict, the arithmetic of the reference (paths relative to the upstream repository):

  * tile encoder ............ gbm/model.py:14-61   (`ResNet`, layers [3,3,3,3], widths 20/40/60/80)
  * residual block .......... nnBlocks.py:157-189  (`BasicResBlock`)
  * context layer ........... gbm/model.py:89-111  (`ContextLayer`)
  * MIL head + output dict .. gbm/model.py:189-264 (`Attention.forward`)
  * soft-target CE .......... nnBlocks.py:47-134   (`CrossEntropyWithProbs`)

Parity is PINNED: `tests/test_oracle_golden.py` checks this file against golden
vectors captured from the reference itself (tests/golden/make_golden.py): the 13
output-dict entries, stage activations and all 65 parameter gradients, eval and
train mode (with the reference's recorded subsample indices and dropout mask).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LEAK = 0.1                      # nnBlocks.py:170, gbm/model.py:25
STAGES = ((1, 20, 1), (2, 40, 2), (3, 60, 2), (4, 80, 2))   # gbm/model.py:27-30
BLOCKS_PER_STAGE = 3            # gbm/model.py:133
SMOOTHING = 0.25                # gbm/model.py:128
DROP_P = 0.25                   # gbm/model.py:107
SUBSAMPLE = 0.2                 # gbm/model.py:193
BN_EPS = 1e-5                   # torch BatchNorm1d default used at gbm/model.py:105


def state_dict_spec():
    """Ordered (key, shape) list of the reference state dict."""
    spec = [("weight_mask", (3,)),
            ("cnn.module.conv1.weight", (20, 3, 7, 7)), ("cnn.module.conv1.bias", (20,))]
    cin = 20
    for li, planes, stride in STAGES:
        for b in range(BLOCKS_PER_STAGE):
            p = f"cnn.module.layer{li}.{b}."
            c_in = cin if b == 0 else planes
            spec += [(p + "conv1.weight", (planes, c_in, 3, 3)), (p + "conv1.bias", (planes,)),
                     (p + "conv2.weight", (planes, planes, 3, 3)), (p + "conv2.bias", (planes,))]
            if b == 0 and (stride != 1 or c_in != planes):
                spec.append((p + "downsample.0.weight", (planes, c_in, 1, 1)))
        cin = planes
    spec += [("cnn.module.fc.weight", (80, 80)),
             ("context.bn.weight", (80,)), ("context.bn.bias", (80,)),
             ("attention.lin1.weight", (40, 80)), ("attention.lin1.bias", (40,)),
             ("attention.lin2.weight", (3, 40)), ("attention.lin2.bias", (3,)),
             ("buffer.lin1.weight", (40, 80)), ("buffer.lin1.bias", (40,)),
             ("buffer.classifier.weight", (1, 40)), ("buffer.classifier.bias", (1,))]
    return spec


class _RoundFwd(torch.autograd.Function):
    """y = bf16(x) in the forward pass, identity gradient (a tensor STORED in bf16)."""
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """identity in the forward pass, g = bf16(g) in the backward pass (a GRADIENT stored in bf16)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _ident(t):
    return t


def _lrelu(v, pos=None):
    """LeakyReLU(0.1); `pos` (bool, same shape) forces the branch per element instead of the sign of v — see `patterns`
    of `backbone`."""
    return F.leaky_relu(v, LEAK) if pos is None else torch.where(pos, v, LEAK * v)


def residual_block(sd, prefix, x, stride, rf=_ident, rb=_ident, pat=None, name=""):
    """nnBlocks.py:175-189.  rf/rb are identity for the fp32 reference arithmetic; the bf16-storage
    emulation passes rounding functions at the points where the HIP path stores bf16 tensors."""
    p1 = None if pat is None else pat[name + ".o1"]
    p2 = None if pat is None else pat[name]
    o = _lrelu(rb(F.conv2d(x, rf(sd[prefix + "conv1.weight"]), sd[prefix + "conv1.bias"],
                           stride=stride, padding=1)), p1)
    o = rf(o)
    o = F.conv2d(o, rf(sd[prefix + "conv2.weight"]), sd[prefix + "conv2.bias"], stride=1, padding=1)
    key = prefix + "downsample.0.weight"
    shortcut = rf(F.conv2d(rb(x), rf(sd[key]), None, stride=stride)) if key in sd else x
    return rf(_lrelu(rb(o + shortcut), p2))


def backbone(sd, x, acts=None, prefix="cnn.module.", emulate_bf16=False, patterns=None):
    """gbm/model.py:50-61.  x: [T,3,H,W] fp32 -> [T,80].

    emulate_bf16=True is NOT reference arithmetic: it is the same fp32 computation with every tensor the
    HIP bf16 path keeps in HBM (activations, their gradients, MFMA weight operands) rounded to bf16 at
    the point where that path stores it, so that the bf16 kernels can be checked tightly instead of
    only against a loose bf16-vs-fp32 bound.

    patterns (NOT reference arithmetic either): the piecewise-linear network evaluated on a GIVEN activation pattern —
    {"stem_tap": int64 [T,20,Hp,Wp] winning tap (ky*3+kx) of every max-pool window, "stem_pos": bool, its winner > 0,
    "layerL.B.o1" / "layerL.B": bool [T,C,H,W], sign of the block's two LeakyReLU inputs}.  An element whose
    pre-activation is within rounding of zero flips its branch between two correct fp32 evaluations and moves the
    gradients by up to 1e-2 of their norm (one element in 327,680 is enough: tools/diag_mask_flips.py), which says nothing
    about the kernels; with the pattern taken from the implementation under test, an fp64 run of this function is the
    exact gradient of the SAME linear piece and kernel error is all that is left."""
    rf = _RoundFwd.apply if emulate_bf16 else _ident
    rb = _RoundBwd.apply if emulate_bf16 else _ident
    pre = rb(F.conv2d(rf(x), rf(sd[prefix + "conv1.weight"]), sd[prefix + "conv1.bias"], stride=2, padding=3))
    if patterns is None:
        t = rf(F.leaky_relu(pre, LEAK))
        if acts is not None:
            acts["stem"] = t
        t = rb(F.max_pool2d(t, kernel_size=3, stride=2, padding=1))
    else:                       # LeakyReLU is monotonic: pool the pre-activation at the given winners, then the given branch
        n, c, h, w = pre.shape
        hp, wp = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        win = F.unfold(F.pad(pre, (1, 1, 1, 1), value=float("-inf")), kernel_size=3, stride=2).view(n, c, 9, hp, wp)
        t = _lrelu(win.gather(2, patterns["stem_tap"].view(n, c, 1, hp, wp)).squeeze(2), patterns["stem_pos"])
    if acts is not None:
        acts["pool"] = t
    for li, _planes, stride in STAGES:
        for b in range(BLOCKS_PER_STAGE):
            t = residual_block(sd, f"{prefix}layer{li}.{b}.", t, stride if b == 0 else 1, rf, rb, patterns, f"layer{li}.{b}")
            if acts is not None:
                acts[f"layer{li}.{b}"] = t
        if acts is not None:
            acts[f"layer{li}"] = t
    t = t.mean(dim=(2, 3))
    return t @ sd[prefix + "fc.weight"].t()


def soft_target_ce(logit, label, class_weights=None, classes=3, smoothing=SMOOTHING):
    """nnBlocks.py:71-85 (smooth one-hot) + :121-134 (weighted soft-target CE, mean)."""
    target = torch.full((label.shape[0], classes), smoothing / (classes - 1), dtype=logit.dtype)
    target.scatter_(1, label.view(-1, 1), 1.0 - smoothing)
    nll = -F.log_softmax(logit, dim=1)
    if class_weights is not None:
        nll = nll * class_weights.view(1, -1).to(nll.dtype)
    return (target * nll).sum(dim=1).mean()


def mil_head(sd, feats, label, *, keep_mask=None, class_weights=None):
    """gbm/model.py:198-264 from the bag features H [N,80] on.  `keep_mask` [N,80]
    (1 = kept) switches the Dropout(0.25) of gbm/model.py:107 on, as in training."""
    n = feats.shape[0]
    if n < 2:
        # torch's batch-statistics BatchNorm1d raises for a single instance (gbm/model.py:105)
        raise ValueError("Expected more than 1 value per channel when training, got input size "
                         f"{tuple(feats.shape)}")
    kld = 0.5 * feats.pow(2).mean()
    mean = feats.mean(dim=0, keepdim=True)
    var = (feats - mean).pow(2).mean(dim=0, keepdim=True)          # biased, batch statistics
    hz = (feats - mean) / torch.sqrt(var + BN_EPS) * sd["context.bn.weight"] + sd["context.bn.bias"]
    hm = F.leaky_relu(feats, LEAK)
    if keep_mask is not None:
        hm = hm * keep_mask.to(hm.dtype) / (1.0 - DROP_P)

    a_raw = torch.tanh(hz @ sd["attention.lin1.weight"].t() + sd["attention.lin1.bias"])
    a_raw = a_raw @ sd["attention.lin2.weight"].t() + sd["attention.lin2.bias"]      # [N,3]
    w = sd["weight_mask"]
    a_mask = torch.sigmoid(-10.0 * w) * F.softplus(a_raw) + torch.sigmoid(10.0 * w)
    a1 = a_mask / a_mask.abs().sum(dim=0, keepdim=True).clamp_min(1e-12)
    aterm = a1.t()                                                                   # [3,N]
    a2 = a_raw / a_raw.pow(2).sum(dim=0, keepdim=True).sqrt().clamp_min(1e-12)
    off_diag = 1.0 - torch.eye(3, dtype=feats.dtype)
    aterm_var = ((a2.t() @ a2) * off_diag).mean()
    aterm_mu = 0.5 * a_raw.mean(dim=0).pow(2).sum()

    b = F.leaky_relu(hm @ sd["buffer.lin1.weight"].t() + sd["buffer.lin1.bias"], LEAK)
    b = b @ sd["buffer.classifier.weight"].t() + sd["buffer.classifier.bias"]       # [N,1]
    mterm = aterm @ b                                                                # [3,1]
    wrois = aterm * b.view(1, n)
    logit = mterm.view(1, 3)
    y_pred = F.softmax(logit, dim=1)
    y_hat = torch.argmax(y_pred).long()
    label = label.long().view(-1)
    loss = soft_target_ce(logit, label, class_weights)
    error = 1.0 - y_hat.eq(label).float()
    l2 = torch.stack([sd["buffer.lin1.weight"].norm(), sd["buffer.classifier.weight"].norm()]).mean()
    return OrderedDict([
        ("Aterm", aterm.detach()), ("wROIs", wrois.detach()), ("Bterm", b.detach()),
        ("Mterm", mterm.detach()), ("Fterm", feats.detach()), ("Aterm_mu", aterm_mu.detach()),
        ("Aterm_var", aterm_var.detach()), ("loss", loss), ("l2", l2), ("KLD", kld.detach()),
        ("y_pred", y_pred.detach()), ("y_pred_hat", y_hat.detach()), ("error", error)])


def attention_forward(sd, full_input, label, *, training=False, indices=None, keep_mask=None,
                      class_weights=None, acts=None, emulate_bf16=False):
    """gbm/model.py:189-264.  In training mode the caller supplies what the reference draws
    from the global RNG: `indices` (the randperm subsample, :193) and `keep_mask` (Dropout)."""
    x = full_input.detach()
    if training:
        if indices is None:
            indices = torch.randperm(x.shape[0])[: int(x.shape[0] * SUBSAMPLE)]
        x = x[indices]
        if keep_mask is None:
            keep_mask = (torch.rand(x.shape[0], 80) >= DROP_P)
    else:
        keep_mask = None
    feats = backbone(sd, x, acts, emulate_bf16=emulate_bf16)
    return mil_head(sd, feats, label, keep_mask=keep_mask, class_weights=class_weights)


def load_state(npz, requires_grad=False):
    """numpy archive (tests/golden/weights.npz) -> ordered dict of fp32 tensors."""
    sd = OrderedDict()
    for k, _shape in state_dict_spec():
        sd[k] = torch.tensor(npz[k], dtype=torch.float32, requires_grad=requires_grad)
    return sd


# ---- second backbone configuration: alt_resnet.py (BatchNorm-free torchvision ResNet, widths 64..512) ----------
ALT_WIDTHS = (64, 128, 256, 512)            # alt_resnet.py:86-89


def alt_state_dict_spec(layers, num_classes):
    """Ordered (key, shape) list of alt_resnet.ResNet(BasicBlock, layers, num_classes).state_dict()."""
    spec = [("conv1.weight", (64, 3, 7, 7))]
    cin = 64
    for li, (planes, depth) in enumerate(zip(ALT_WIDTHS, layers), start=1):
        for b in range(depth):
            c_in = cin if b == 0 else planes
            spec += [(f"layer{li}.{b}.conv1.weight", (planes, c_in, 3, 3)), (f"layer{li}.{b}.conv2.weight", (planes, planes, 3, 3))]
            if b == 0 and (li > 1 or c_in != planes):
                spec.append((f"layer{li}.{b}.downsample.0.weight", (planes, c_in, 1, 1)))
        cin = planes
    spec += [("fc.weight", (num_classes, 512)), ("fc.bias", (num_classes,))]
    return spec


def alt_seeded_state(layers, num_classes, seed, requires_grad=False):
    """Deterministic weights (He-scaled normal from a seeded CPU generator) so that 5-20 M parameters never have to be
    stored: the golden generator loads exactly these into the reference model, the tests regenerate them."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shape in alt_state_dict_spec(layers, num_classes):
        if k.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        sd[k] = t.requires_grad_(requires_grad)
    return sd


def alt_backbone(sd, x, layers):
    """alt_resnet.py:128-145 (`_forward_impl`) with BasicBlock of :54-66: conv-relu-conv-(+identity/projection)-relu."""
    t = F.relu(F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3))
    t = F.max_pool2d(t, kernel_size=3, stride=2, padding=1)
    for li, depth in enumerate(layers, start=1):
        for b in range(depth):
            p = f"layer{li}.{b}."
            stride = 2 if (b == 0 and li > 1) else 1
            o = F.relu(F.conv2d(t, sd[p + "conv1.weight"], None, stride=stride, padding=1))
            o = F.conv2d(o, sd[p + "conv2.weight"], None, stride=1, padding=1)
            key = p + "downsample.0.weight"
            shortcut = F.conv2d(t, sd[key], None, stride=stride) if key in sd else t
            t = F.relu(o + shortcut)
    t = t.mean(dim=(2, 3))
    return t @ sd["fc.weight"].t() + sd["fc.bias"]
