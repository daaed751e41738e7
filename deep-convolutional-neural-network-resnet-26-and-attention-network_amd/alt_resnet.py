"""The wide tile-encoder variant of the reference (`alt_resnet.py`: a torchvision-style ResNet with every
BatchNorm removed — bias-free 3x3 / 1x1 convs, ReLU, widths 64/128/256/512, `Linear(512, num_classes)` with
bias; `alt_resnet.py:24-33` conv helpers, `:35-66` BasicBlock, `:70-145` ResNet, `:157-165` resnet18) on the HIP
kernels: same module tree and state-dict keys (`conv1.weight`, `layerL.B.conv{1,2}.weight`,
`layerL.0.downsample.0.weight`, `fc.weight`, `fc.bias`), forward and hand-written backward.

64-channel layers run on the resident-filter kernels of the 20–80-channel path; 128/256/512-channel layers run on
the channel-blocked kernels of `csrc/conv_wide.hip`.  `pretrained=True` of the reference is a URL fetch and is not
offered; its `zero_init_residual=True` branch raises in the reference (`alt_resnet.py:104` touches a `bn2` that no
longer exists) and is rejected here as well.
"""
import torch
from torch import nn

from . import _lib as L
from . import ops

WIDTHS = (64, 128, 256, 512)          # alt_resnet.py:86-89
GATHER_GEMM = [True]                  # A/B switch: False keeps every wide conv on the channel-blocked kernels of conv_wide.hip


class BasicBlock(nn.Module):
    """Parameter container of alt_resnet.py:35-66 (two bias-free 3x3 convs, optional 1x1 projection)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("BasicBlock is executed by the fused HIP encoder (ResNet.forward)")


class _Conv:
    """One convolution of the network: packed operands + the three kernel calls, narrow or wide path.  The packed
    (MFMA fragment order) copies are built once and re-used until the weight changes: `_packed_conv` keys them on
    the parameter's storage, its version counter and the optimizer epoch, as `encoder.ResNet.refresh_packed` does."""

    def __init__(self, conv, dtype):
        w = conv.weight
        self.w, self.cout, self.cin, self.ks = w, w.shape[0], w.shape[1], w.shape[2]
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.wide = self.cout > 80 or self.cin > 80
        # bf16 wide layers whose channel counts fit its 64-channel K-steps / 128-channel output blocks run the gather-GEMM
        # kernel (csrc/conv_gather.hip); the 128 -> 64 channel gradient of the stage-2 entry does not and stays on conv_wide
        bf = dtype == torch.bfloat16 and self.wide and GATHER_GEMM[0]
        self.g_fwd = bf and ops.gconv_supported(self.cin, self.cout, self.ks, self.stride)
        self.g_bwd = bf and ops.gconv_supported(self.cout, self.cin, self.ks, self.stride)
        if self.wide:
            self.fwd_w = ops.gconv_pack_weights(w, L.PACK_FWD) if self.g_fwd else ops.wide_pack_weights(w, L.PACK_FWD, dtype)
            self.bwd_w = ops.gconv_pack_weights(w, L.PACK_DGRAD) if self.g_bwd else ops.wide_pack_weights(w, L.PACK_DGRAD, dtype)
        else:
            self.fwd_w, self.fwd_b = ops.pack_weights(w, None, L.PACK_FWD, dtype)
            self.bwd_w, _ = ops.pack_weights(w, None, L.PACK_DGRAD, dtype)

    def forward(self, x, res=None, relu=True):
        if self.g_fwd:
            return ops.gconv(x, self.fwd_w, self.cout, ks=self.ks, stride=self.stride, pad=self.pad, res=res, relu=relu)
        if self.wide:
            return ops.wide_conv(x, self.fwd_w, self.cout, ks=self.ks, stride=self.stride, pad=self.pad, res=res, relu=relu)
        return ops.conv(x, self.fwd_w, None, self.cout, ks=self.ks, stride=self.stride, pad=self.pad, res=res, lrelu=relu,
                        slope=0.0)

    def dgrad(self, dz, in_hw, addend=None, act=None):
        zi = self.stride == 2
        if self.g_bwd:
            return ops.gconv(dz, self.bwd_w, self.cin, ks=self.ks, stride=self.stride, pad=self.pad, transposed=True, out_hw=in_hw,
                             res=addend, act=act)
        if self.wide:
            return ops.wide_conv(dz, self.bwd_w, self.cin, ks=self.ks, stride=1, pad=self.pad, res=addend, act=act,
                                 zero_insert=zi, out_hw=in_hw)
        return ops.conv(dz, self.bwd_w, None, self.cin, ks=self.ks, stride=1, pad=self.pad, res=addend, act=act,
                        zero_insert=zi, out_hw=in_hw, slope=0.0)

    def wgrad(self, x, dz, ws):
        if self.wide:
            return ops.wide_wgrad(x, dz, self.cin, self.cout, ks=self.ks, stride=self.stride, pad=self.pad, workspace=ws)
        dw, _ = ops.conv_wgrad(x, dz, self.cin, self.cout, ks=self.ks, stride=self.stride, pad=self.pad, want_bias=False)
        return dw, ws


class ResNet(nn.Module):
    """[T,3,H,W] fp32 -> [T,num_classes] fp32 (alt_resnet.py:70-145)."""

    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), num_classes=1000, zero_init_residual=False, groups=1,
                 width_per_group=64, compute_dtype=torch.bfloat16):
        super().__init__()
        if block is not BasicBlock or groups != 1 or width_per_group != 64:
            raise ValueError("the HIP encoder implements BasicBlock with groups=1, base_width=64")
        if zero_init_residual:
            raise AttributeError("'BasicBlock' object has no attribute 'bn2' (as in the reference, alt_resnet.py:104)")
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, (width, depth) in enumerate(zip(WIDTHS, layers)):
            setattr(self, f"layer{i + 1}", self._make_layer(width, depth, stride=1 if i == 0 else 2))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")     # alt_resnet.py:95-97
        self.compute_dtype = compute_dtype

    def _make_layer(self, planes, blocks, stride=1):
        shortcut = None
        if stride != 1 or self.inplanes != planes:
            shortcut = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False))
        seq = [BasicBlock(self.inplanes, planes, stride, shortcut)]
        self.inplanes = planes
        seq += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def blocks(self):
        for i in range(4):
            for blk in getattr(self, f"layer{i + 1}"):
                yield blk

    def encoder_params(self):
        ps = [self.conv1.weight]
        for blk in self.blocks():
            ps += [blk.conv1.weight, blk.conv2.weight]
            if blk.downsample is not None:
                ps.append(blk.downsample[0].weight)
        ps += [self.fc.weight, self.fc.bias]
        return ps

    def forward(self, x):
        return _AltFn.apply(self, x, *self.encoder_params())


def _weight_tag(w, dtype):
    from .encoder import WEIGHT_EPOCH
    return (w.data_ptr(), w._version, WEIGHT_EPOCH[0], dtype)


def _packed_conv(net, conv, dtype):
    """The `_Conv` of `conv`, re-packed only when its weight (or the compute dtype) changed since the last call."""
    cache = net.__dict__.setdefault("_conv_cache", {})
    tag = _weight_tag(conv.weight, dtype)
    hit = cache.get(id(conv))
    if hit is None or hit[0] != tag:
        hit = cache[id(conv)] = (tag, _Conv(conv, dtype))
    return hit[1]


def _packed_stem(net, dtype):
    cache = net.__dict__.setdefault("_conv_cache", {})
    tag = _weight_tag(net.conv1.weight, dtype)
    hit = cache.get("stem")
    if hit is None or hit[0] != tag:
        hit = cache["stem"] = (tag, ops.pack_weights(net.conv1.weight, None, L.PACK_STEM, dtype))
    return hit[1]


def _forward(net, x, dtype):
    wp, bp = _packed_stem(net, dtype)
    fused = ops.stem_fwd_fused(x, wp, bp, 64, slope=0.0, dtype=dtype)
    if fused is not None:
        xs, pool, widx = fused
        stem_hw = tuple(xs.shape[1:3])
    else:
        xs = ops.stem_s2d(x, dtype)
        stem = ops.conv(xs, wp, bp, 64, ks=4, stride=1, pad=2, lrelu=True, slope=0.0)
        pool, widx = ops.maxpool_fwd(stem)
        stem_hw = tuple(stem.shape[1:3])
    saved = {"xs": xs, "stem_hw": stem_hw, "widx": widx, "blocks": []}
    t = pool
    for blk in net.blocks():
        c1, c2 = _packed_conv(net, blk.conv1, dtype), _packed_conv(net, blk.conv2, dtype)
        ds = _packed_conv(net, blk.downsample[0], dtype) if blk.downsample is not None else None
        o1 = c1.forward(t)
        short = ds.forward(t, relu=False) if ds is not None else t
        out = c2.forward(o1, res=short)
        saved["blocks"].append((t, o1, out, c1, c2, ds))
        t = out
    pooled, feats = ops.avgpool_fc_fwd(t, net.fc.weight.detach(), 512, bias=net.fc.bias.detach())
    saved["pooled"] = pooled
    return feats, saved


def _backward(net, saved, dfeats, dtype):
    blocks = saved["blocks"]
    last_out = blocks[-1][2]
    dz, dwfc, dbfc = ops.avgpool_fc_bwd(dfeats.contiguous(), net.fc.weight.detach(), saved["pooled"], last_out, 512,
                                        slope=0.0, want_bias=True)
    grads, ws = [], None
    for bi in range(len(blocks) - 1, -1, -1):
        xin, o1, _out, c1, c2, ds = blocks[bi]
        g2, ws = c2.wgrad(o1, dz, ws)
        dz1 = c2.dgrad(dz, o1.shape[1:3], act=o1)
        g1, ws = c1.wgrad(xin, dz1, ws)
        if ds is not None:
            gd, ws = ds.wgrad(xin, dz, ws)
            addend = ds.dgrad(dz, xin.shape[1:3])
        else:
            gd, addend = None, dz
        mask = xin if bi > 0 else None            # block 0 reads the max-pool output
        dz = c1.dgrad(dz1, xin.shape[1:3], addend=addend, act=mask)
        grads.append((g1, g2, gd))
    dstem = ops.maxpool_bwd(dz, saved["widx"], saved["stem_hw"], slope=0.0)
    dwstem, _ = ops.conv_wgrad(saved["xs"], dstem, 3, 64, ks=4, stride=1, pad=2, stem=True, want_bias=False)
    flat = [dwstem]
    for g1, g2, gd in reversed(grads):
        flat += [g1, g2]
        if gd is not None:
            flat.append(gd)
    flat += [dwfc, dbfc]
    return flat


class _AltFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        dtype = net.compute_dtype
        feats, saved = _forward(net, x.detach(), dtype)
        ctx.net, ctx.saved, ctx.dtype = net, saved, dtype
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        grads = _backward(ctx.net, ctx.saved, dfeats, ctx.dtype)
        ctx.saved = None
        return (None, None, *grads)


def resnet18(pretrained=False, progress=True, **kwargs):
    """alt_resnet.py:157-165.  The reference's pretrained=True downloads torchvision weights; not offered here."""
    if pretrained:
        raise RuntimeError("pretrained weights are a network fetch in the reference (alt_resnet.py:150-153); load a state dict instead")
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)
