"""ResNet-26 tile encoder on the HIP kernels: forward and hand-written backward.

Mirrors the reference's `ResNet` (gbm/model.py:14-61) built from `BasicResBlock`
(nnBlocks.py:157-189) with layers [3,3,3,3] and widths 20/40/60/80 — same module tree, same
state-dict keys — but `forward` never calls a torch conv: it sequences the C-ABI kernels of
include/mil_hip.h (NHWC, channel-padded, fp32 or bf16 operands) and a custom autograd Function
provides the backward (dgrad / wgrad / pooling backward) from saved NHWC activations.
"""
import contextlib
import os

import torch
from torch import nn

from . import _lib as L
from . import hooks
from . import ops

STAGE_WIDTHS = (20, 40, 60, 80)        # gbm/model.py:27-30
WEIGHT_EPOCH = [0]                     # bumped by optimizers that update weights through raw kernels (FlatAdam)
STEM_WIDTH = 20                        # gbm/model.py:20


def invalidate_packed_weights():
    """Call after changing conv weights in a way autograd's version counters do not see — `conv.weight.data.normal_()` (as the
    reference itself does, nnBlocks.py:375), EMA / clipping code writing through `.data`, raw-pointer updates: the encoders
    keep MFMA-fragment-order copies of every filter, re-packed only when a parameter's `_version`, its storage or this
    epoch changes.  (`FlatAdam.step` bumps the epoch itself; `load_state_dict` and in-place ops on the Parameter bump
    `_version`.)"""
    WEIGHT_EPOCH[0] += 1


class BasicResBlock(nn.Module):
    """Parameter container with the reference block's layout (nnBlocks.py:157-173): conv1 (3x3,
    stride s, bias), conv2 (3x3, bias), optional `downsample` = Sequential(1x1 stride-s conv, no bias).
    The arithmetic of nnBlocks.py:175-189 is executed by the fused HIP kernels in `encoder_forward`."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=True)
        self.relu = nn.LeakyReLU(ops.LEAK)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("BasicResBlock is executed by the fused HIP encoder (ResNet.forward); "
                           "it has no stand-alone torch path")


class ResNet(nn.Module):
    """Tile encoder [T,3,H,W] fp32 -> [T,num_classes] fp32 (gbm/model.py:14-61)."""

    def __init__(self, block=BasicResBlock, layers=(3, 3, 3, 3), num_classes=80, zero_init_residual=False,
                 groups=1, width_per_group=64, compute_dtype=L.BF16X3):
        super().__init__()
        if block is not BasicResBlock or groups != 1 or width_per_group != 64:
            raise ValueError("the HIP encoder implements BasicResBlock with groups=1, base_width=64")
        self.inplanes = STEM_WIDTH
        self.conv1 = nn.Conv2d(3, STEM_WIDTH, kernel_size=7, stride=2, padding=3)
        self.relu = nn.LeakyReLU(ops.LEAK, inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, (width, depth) in enumerate(zip(STAGE_WIDTHS, layers)):
            setattr(self, f"layer{i + 1}", self._make_layer(width, depth, stride=1 if i == 0 else 2))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(STAGE_WIDTHS[-1], num_classes, bias=False)
        self.compute_dtype = compute_dtype
        # separate weight-gradient launches on a side stream: no longer a win once the hot layers run fused backward
        # kernels (same throughput at 2048 tiles, slower for small bags) — available, off by default
        self.overlap_wgrad = False
        self.direct_grad = False        # accumulate parameter gradients straight into existing .grad tensors
        self.n_side_streams = 1
        self.fuse_backward = True
        self.fuse_stem_forward = True
        self.keep_s2d = False           # True: the fused stem forward also writes the bf16 space-to-depth copy of the input (bf16 mode)
        self.fuse_stage_entry = True
        self.fuse_block_forward = True
        # gradient tensors of the 20-channel stage (produced and consumed only by the fused backward kernels) at 20 channels
        # per pixel instead of the padded 24: 17 % fewer bytes on three of the four tensor passes of its fused backward
        self.dense_grads = True
        # the 28 slab reductions of a backward pass recorded and run as ONE launch (ops.ReduceBatch) instead of one ~10 us
        # launch behind every weight-gradient kernel
        self.batch_reductions = True
        self._reduce_batch = None
        self._pack_table = None
        self._pack_version = None
        self._side = None

    def _make_layer(self, planes, blocks, stride=1):
        shortcut = None
        if stride != 1 or self.inplanes != planes:
            shortcut = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False))
        seq = [BasicResBlock(self.inplanes, planes, stride, shortcut)]
        self.inplanes = planes
        seq += [BasicResBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    # ---- execution ---------------------------------------------------------------------------
    def blocks(self):
        for i in range(4):
            for blk in getattr(self, f"layer{i + 1}"):
                yield blk

    def encoder_params(self):
        """Flat, ordered parameter list handed to the autograd Function."""
        ps = [self.conv1.weight, self.conv1.bias]
        for blk in self.blocks():
            ps += [blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias]
            if blk.downsample is not None:
                ps.append(blk.downsample[0].weight)
        ps.append(self.fc.weight)
        return ps

    def _side_streams(self):
        if self._side is None:
            self._side = [torch.cuda.Stream() for _ in range(self.n_side_streams)]
        return self._side

    # ---- packed (MFMA fragment order) copies of every filter, refreshed by ONE launch -------------------
    def _pack_specs(self):
        """(key, weight, bias, mode) for every packed filter the forward and backward passes use."""
        specs = [("stem", self.conv1.weight, self.conv1.bias, L.PACK_STEM)]
        for bi, blk in enumerate(self.blocks()):
            for name, conv in (("c1", blk.conv1), ("c2", blk.conv2)):
                specs.append((f"b{bi}.{name}", conv.weight, conv.bias, L.PACK_FWD))
                specs.append((f"b{bi}.{name}", conv.weight, None, L.PACK_DGRAD))
            if blk.downsample is not None:
                specs.append((f"b{bi}.ds", blk.downsample[0].weight, None, L.PACK_FWD))
                specs.append((f"b{bi}.ds", blk.downsample[0].weight, None, L.PACK_DGRAD))
                if blk.stride == 2:       # conv1's and the projection's transposed convs in one parity-class filter
                    specs.append((f"b{bi}.c1", blk.conv1.weight, blk.downsample[0].weight, L.PACK_DGRAD_S2))
        return specs

    def refresh_packed(self, dtype):
        """Make the packed filters current: rebuild the job table if storage moved, re-run the single pack launch
        if any weight changed (parameter version counters, or WEIGHT_EPOCH bumped by FlatAdam.step)."""
        import ctypes
        lib = L.lib()
        specs = self._pack_specs()
        ptr_tag = (dtype, L.dt_code(dtype, mma=True)) + tuple(w.data_ptr() for _k, w, _b, _m in specs) + tuple(0 if b is None else b.data_ptr() for _k, _w, b, _m in specs)
        if self._pack_table is None or self._pack_table[0] != ptr_tag:
            dev = self.conv1.weight.device
            rec = lib.mil_pack_job_bytes()
            host = (ctypes.c_char * (rec * len(specs)))()
            store = {}
            for i, (key, w, b, mode) in enumerate(specs):
                if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
                    raise ValueError("conv weights must be contiguous CUDA fp32 tensors")
                cout, cin, ks, _ = w.shape
                elems = ctypes.c_size_t(0)
                L.check(lib.mil_packed_weight_elems(ctypes.byref(elems), cout, cin, ks, mode), "mil_packed_weight_elems")
                packed = torch.empty(elems.value, dtype=dtype, device=dev)
                n_out = cin if mode == L.PACK_DGRAD else cout
                bias_pad = torch.empty((ops.cpad(n_out) + 15) // 16 * 16, dtype=torch.float32, device=dev)
                L.check(lib.mil_pack_job_fill(ctypes.byref(host, i * rec), w.data_ptr(), L.ptr(b), packed.data_ptr(),
                                              bias_pad.data_ptr(), cout, cin, ks, mode, L.dt_code(dtype, mma=True)), "mil_pack_job_fill")
                store[(key, mode)] = (packed, bias_pad)
            table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
            self._pack_table = (ptr_tag, table, store, len(specs))
            self._pack_version = None
        ver = (WEIGHT_EPOCH[0],) + tuple(w._version for _k, w, _b, _m in specs) + tuple(0 if b is None else b._version for _k, _w, b, _m in specs)
        if self._pack_version != ver:
            _tag, table, _store, njobs = self._pack_table
            L.check(lib.mil_pack_all(table.data_ptr(), njobs, L.stream_ptr()), "mil_pack_all")
            self._pack_version = ver

    def _packed(self, key, weight, bias, mode, dtype):
        return self._pack_table[2][(key, mode)]

    # ---- forward hooks on children (the reference's children are live modules: SURVEY.md §8b) ----------------
    def block_position(self, bi):
        """(stage index, index inside the stage, stage depth) of block `bi` of `blocks()`."""
        for li in range(4):
            depth = len(getattr(self, f"layer{li + 1}"))
            if bi < depth:
                return li, bi, depth
            bi -= depth
        raise IndexError(bi)

    def child_hooks(self):
        """The children that carry a forward (pre-)hook, or None: the un-hooked forward builds no views."""
        hk = [m for m in self.modules() if m is not self and hooks.hooked(m)]
        return hk or None

    def _fire_block_hooks(self, bi, blk, stage_in, xin, o1, out):
        """Hooks of block `bi`, of its LeakyReLU (called twice per block upstream, nnBlocks.py:180,187) and — after a
        stage's last block — of the stage's Sequential, with NCHW fp32 views of the saved activations."""
        cin, cout = blk.conv1.in_channels, blk.conv1.out_channels
        if hooks.hooked(blk.relu):
            v1 = hooks.nchw(o1, cout)
            hooks.fire(blk.relu, v1, v1)
            v2 = hooks.nchw(out, cout)
            hooks.fire(blk.relu, v2, v2)
        if hooks.hooked(blk):
            hooks.fire(blk, hooks.nchw(xin, cin), hooks.nchw(out, cout))
        li, j, depth = self.block_position(bi)
        if j == depth - 1:
            stage = getattr(self, f"layer{li + 1}")
            if hooks.hooked(stage):
                hooks.fire(stage, hooks.nchw(stage_in, stage[0].conv1.in_channels), hooks.nchw(out, cout))

    def forward(self, x):
        """x: fp32 [T,3,H,W] tiles (the reference's tensor), or `preprocess.S2dTiles` — the same tiles as the bf16
        space-to-depth tensor the stem kernels read (bf16 compute mode only)."""
        from .preprocess import S2dTiles
        if isinstance(x, S2dTiles):
            x = x.xs
        return _EncoderFn.apply(self, x, *self.encoder_params())


def encoder_forward(net, x, dtype):
    """Runs the kernels; returns (feats [T,80] fp32, saved-state dict for the backward)."""
    net.refresh_packed(dtype)
    x = x.contiguous()                  # the tensor the kernels read — and, without a kept s2d copy, the one the backward re-reads
    hk = net.child_hooks()              # None unless a forward hook sits on a child module (then views are built for it)
    wp, bp = net._packed("stem", net.conv1.weight, net.conv1.bias, L.PACK_STEM, dtype)
    stem_hooked = hk is not None and hooks.any_hooked((net.conv1, net.relu, net.maxpool))
    if x.dtype == torch.bfloat16:       # the tiles arrive as the bf16 space-to-depth tensor [T,H/2,W/2,16] (preprocess.S2dTiles)
        if x.dim() != 4 or x.shape[3] != 16:
            raise ValueError(f"a bf16 input must be the space-to-depth tensor [T,H/2,W/2,16], got {tuple(x.shape)}")
        if dtype != torch.bfloat16:
            raise ValueError("space-to-depth bf16 tiles feed the bf16 compute mode only (the fp32 modes take fp32 [T,3,H,W] tiles)")
        return _encoder_forward_from(net, x, dtype, hk, stem_hooked, wp, bp, xs_in=x)
    return _encoder_forward_from(net, x, dtype, hk, stem_hooked, wp, bp, xs_in=None)


def _encoder_forward_from(net, x, dtype, hk, stem_hooked, wp, bp, xs_in):
    # no space-to-depth copy is kept (keep_s2d False): the fused stem backward rebuilds its tiles from x itself
    split = dtype == torch.float32 and L.dt_code(dtype, mma=True) == L.MIL_DT_F32S       # bf16x3: never an s2d copy
    if split and net.keep_s2d and xs_in is None:
        x = x.clone()                   # this mode has no s2d form: `keep_s2d` keeps a library-owned fp32 copy for the backward instead
    fused = None
    if xs_in is None:
        fused = ops.stem_fwd_fused(x, wp, bp, ops.cpad(STEM_WIDTH), dtype=dtype,
                                   keep_s2d=(net.keep_s2d or not net.fuse_backward) and not split) if (net.fuse_stem_forward and not stem_hooked) else None
    else:                               # no fp32 stack exists: the fused stem reads the s2d records themselves
        fused = ops.stem_fwd_fused_xs(xs_in, wp, bp, ops.cpad(STEM_WIDTH)) if (net.fuse_stem_forward and not stem_hooked) else None
        if fused is not None:
            fused = (xs_in,) + fused
        x = hooks.s2d_to_nchw(xs_in) if stem_hooked else None          # a hooked stem sees the reference's tensor
    if fused is not None:
        xs, pool, widx = fused
        stem_hw = (xs_in.shape[1], xs_in.shape[2]) if xs_in is not None else (x.shape[2] // 2, x.shape[3] // 2)
    else:
        xs = xs_in if xs_in is not None else ops.stem_s2d(x, dtype)
        stem = ops.conv(xs, wp, bp, ops.cpad(STEM_WIDTH), ks=4, stride=1, pad=2, lrelu=True)
        pool, widx = ops.maxpool_fwd(stem)
        stem_hw = tuple(stem.shape[1:3])
        if stem_hooked:                 # the hooked children see what the reference's would (NCHW fp32, 20 channels)
            if hooks.hooked(net.conv1):   # pre-activation output: one extra launch, only ever paid under a hook
                pre = ops.conv(xs, wp, bp, ops.cpad(STEM_WIDTH), ks=4, stride=1, pad=2, lrelu=False)
                hooks.fire(net.conv1, x, hooks.nchw(pre, STEM_WIDTH))
            stem_v = hooks.nchw(stem, STEM_WIDTH)
            hooks.fire(net.relu, stem_v, stem_v)            # in place upstream (gbm/model.py:25): input is the output
            hooks.fire(net.maxpool, stem_v, hooks.nchw(pool, STEM_WIDTH))
    # the stem output itself is not kept.  Without an s2d copy the backward rebuilds its tiles from the INPUT tensor: its
    # version counter is recorded so that an in-place change between forward and backward raises instead of silently
    # giving a wrong conv1 gradient (keep_s2d=True keeps a library-owned copy where the caller cannot promise that)
    src = xs_in if xs_in is not None else x             # the caller's tensor the backward re-reads
    saved = {"xs": xs, "x": x if xs is None else None, "x_src": src if (xs is None or xs_in is not None) else None,
             "x_version": src._version, "stem_hw": stem_hw, "widx": widx, "blocks": []}
    t = pool
    stage_in = pool
    all_blocks = list(net.blocks())
    chained = 0                         # blocks already run (and recorded) by a stage-wide conv chain
    for bi, blk in enumerate(all_blocks):
        if chained:
            chained -= 1
            continue
        cout = blk.conv1.out_channels
        s = blk.stride
        if hk is not None:
            if net.block_position(bi)[1] == 0:
                stage_in = t
            hooks.refuse(blk, ["conv1", "conv2"] + (["downsample", "downsample.0"] if blk.downsample is not None else []))
        w1, b1 = net._packed(f"b{bi}.c1", blk.conv1.weight, blk.conv1.bias, L.PACK_FWD, dtype)
        w2, b2 = net._packed(f"b{bi}.c2", blk.conv2.weight, blk.conv2.bias, L.PACK_FWD, dtype)
        if s == 1 and blk.downsample is None and net.fuse_block_forward:      # whole block in one pass (24/40 channels)
            both = ops.conv_block_fwd(t, w1, b1, w2, b2)
            if both is None:                             # 80 channels on 8x8 maps: both convs on the LDS-resident images
                both = ops.conv_pair(t, w1, b1, w2, b2, lreluA=True, resB=t, lreluB=True)
            if both is not None:
                o1, out = both
                saved["blocks"].append((t, o1, out))
                if hk is not None:
                    net._fire_block_hooks(bi, blk, stage_in, t, o1, out)
                t = out
                continue
        pair = None
        if blk.downsample is not None:
            wd, _ = net._packed(f"b{bi}.ds", blk.downsample[0].weight, None, L.PACK_FWD, dtype)
            if s == 2 and net.fuse_stage_entry:          # both stride-2 convs in one pass over the block input
                pair = ops.conv_s2_entry(t, w1, b1, wd, ops.cpad(cout))
        if pair is not None:
            o1, short = pair
        else:
            o1 = ops.conv(t, w1, b1, ops.cpad(cout), ks=3, stride=s, pad=1, lrelu=True)
            short = ops.conv(t, wd, None, ops.cpad(cout), ks=1, stride=s, pad=0) if blk.downsample is not None else t
        # A stage whose maps fit the pixel-resident kernel (whole images in LDS: the 80-channel stage at 256x256 tiles): this
        # entry block's conv2 and BOTH convs of every identity block behind it as ONE chain launch
        depth = net.block_position(bi)[2] if blk.downsample is not None else 0
        rest = all_blocks[bi + 1:bi + depth] if depth else []
        if (rest and net.fuse_block_forward and net.block_position(bi)[1] == 0 and len(rest) <= 2 and
                all(b.stride == 1 and b.downsample is None for b in rest)):
            convs = [dict(w=w2, bias=b2, res=short, lrelu=True)]
            for j, b in enumerate(rest):
                wa, ba = net._packed(f"b{bi + 1 + j}.c1", b.conv1.weight, b.conv1.bias, L.PACK_FWD, dtype)
                wb, bb = net._packed(f"b{bi + 1 + j}.c2", b.conv2.weight, b.conv2.bias, L.PACK_FWD, dtype)
                convs += [dict(w=wa, bias=ba, lrelu=True), dict(w=wb, bias=bb, res=2 * j, lrelu=True)]
            outs = ops.conv_chain(o1, convs)
            if outs is not None:
                saved["blocks"].append((t, o1, outs[0]))
                if hk is not None:
                    net._fire_block_hooks(bi, blk, stage_in, t, o1, outs[0])
                for j, b in enumerate(rest):
                    xin_j, o1_j, out_j = outs[2 * j], outs[2 * j + 1], outs[2 * j + 2]
                    saved["blocks"].append((xin_j, o1_j, out_j))
                    if hk is not None:
                        hooks.refuse(b, ["conv1", "conv2"])
                        net._fire_block_hooks(bi + 1 + j, b, stage_in, xin_j, o1_j, out_j)
                t = outs[-1]
                chained = len(rest)
                continue
        out = ops.conv(o1, w2, b2, ops.cpad(cout), ks=3, stride=1, pad=1, res=short, lrelu=True)
        saved["blocks"].append((t, o1, out))
        if hk is not None:
            net._fire_block_hooks(bi, blk, stage_in, t, o1, out)
        t = out
    pooled, feats = ops.avgpool_fc_fwd(t, net.fc.weight.detach(), STAGE_WIDTHS[-1])
    saved["pooled"] = pooled
    if hk is not None:
        if hooks.hooked(net.avgpool):
            hooks.fire(net.avgpool, hooks.nchw(t, STAGE_WIDTHS[-1]), pooled.view(pooled.shape[0], -1, 1, 1))
        if hooks.hooked(net.fc):
            hooks.fire(net.fc, pooled, feats)
    return feats, saved


def encoder_backward(net, saved, dfeats, dtype, allow_direct=True):
    """Gradients of every encoder parameter, in `encoder_params()` order (the input is detached in
    the reference, gbm/model.py:194-196, so no data-gradient is produced for the tiles)."""
    blocks = list(net.blocks())
    grads = {}
    if saved["x_src"] is not None and saved["x_src"]._version != saved["x_version"]:
        raise RuntimeError("the input tiles were modified in place between the encoder's forward and backward: conv1's gradient is "
                           "computed from them (no space-to-depth copy is kept).  Keep the tensor untouched until backward, or set "
                           "`net.cnn.module.keep_s2d = True` to have the forward keep its own copy (bf16: the space-to-depth records; "
                           "bf16x3: an fp32 clone of the tiles)")
    last_out = saved["blocks"][-1][2]
    dz, dwfc = ops.avgpool_fc_bwd(dfeats.contiguous(), net.fc.weight.detach(), saved["pooled"], last_out,
                                  STAGE_WIDTHS[-1],
                                  out=net.fc.weight.grad if (net.direct_grad and allow_direct and net.fc.weight.grad is not None and
                                                             all(p.grad is not None and p.grad.is_contiguous() for p in net.encoder_params())) else None)
    # Weight gradients only consume (x, dz) and nothing downstream waits for them, so they run on side
    # streams (round-robin, one slab workspace each) beside the sequential dgrad chain: the small late-layer
    # and stride-2 launches do not fill 256 CUs alone.
    main = torch.cuda.current_stream()
    sides = net._side_streams()
    use_side = net.overlap_wgrad
    ws = [None] * (len(sides) + 1)
    rr = [0]

    # in-place accumulation into the parameters' .grad: opt-in (dist.FlatParams sets direct_grad; plain loss.backward() only),
    # and only when autograd wants a gradient for every encoder parameter (allow_direct, from ctx.needs_input_grad)
    direct = net.direct_grad and allow_direct and all(p.grad is not None and p.grad.is_contiguous() for p in net.encoder_params())
    batch = None
    if net.batch_reductions and not use_side:
        if net._reduce_batch is None or net._reduce_batch.device != dfeats.device:
            net._reduce_batch = ops.ReduceBatch(dfeats.device)
        batch = net._reduce_batch

    def gout(*params):
        """Destination gradient tensors (the parameters' own .grad) when accumulating in place, else None."""
        return tuple(None if p is None else p.grad for p in params) if direct else None

    def wgrad(xin, dzz, cin, cout, key=None, **kw):
        n, h, w, _ = xin.shape
        _, ho, wo, _ = dzz.shape
        need = ops.wgrad_workspace_bytes(n, h, w, cin, ho, wo, cout, kw["ks"], kw["stride"], kw["pad"],
                                         kw.get("stem", False), xin.dtype)
        if batch is not None:                # deferred reduction: a slab buffer of its own, alive until the batched launch
            assert key is not None, "deferred reductions need one workspace per call site: pass key="
            return ops.conv_wgrad(xin, dzz, cin, cout, workspace=batch.workspace(("w", key), need), **kw)
        if not use_side:
            if ws[-1] is None or ws[-1].numel() * 4 < need:
                ws[-1] = torch.empty((need + 3) // 4, dtype=torch.float32, device=xin.device)
            return ops.conv_wgrad(xin, dzz, cin, cout, workspace=ws[-1], **kw)
        k = rr[0] % len(sides)
        rr[0] += 1
        side = sides[k]
        side.wait_stream(main)                       # dz was produced on the main stream
        with torch.cuda.stream(side):
            if ws[k] is None or ws[k].numel() * 4 < need:   # launches on one stream are serialised: one buffer each
                ws[k] = torch.empty((need + 3) // 4, dtype=torch.float32, device=xin.device)
            out = ops.conv_wgrad(xin, dzz, cin, cout, workspace=ws[k], **kw)
        xin.record_stream(side)
        dzz.record_stream(side)
        return out

    fws = None

    def is_dense(t, c):
        return t.shape[-1] == c and ops.cpad(c) != c

    def fused_bwd(dzz, wd, xin, cin, cout, addend, mask, out, key=None):
        nonlocal fws
        n, h, w, _ = dzz.shape
        dense = is_dense(dzz, cout)
        need = ops.bwd_fused_workspace_bytes(n, h, w, cout, cin, 3, 1, dzz.dtype, dense)
        if need is None:
            if dense:
                raise RuntimeError("dense gradient layout chosen for a shape without a fused backward kernel")
            return None
        if batch is not None:
            return ops.conv_bwd_fused(dzz, wd, xin, cin, cout, addend=addend, mask=mask, workspace=batch.workspace(("f", key), need), out=out)
        if fws is None or fws.numel() * 4 < need:
            fws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dzz.device)
        return ops.conv_bwd_fused(dzz, wd, xin, cin, cout, addend=addend, mask=mask, workspace=fws, out=out)

    # weight-gradient producers record their slab reductions; leaving the block runs them all in one launch
    def stage_dgrad_chain(bi, dz_out):
        """Every 3x3 stride-1 data gradient of the stage that ENDS with block bi in ONE launch (ops.conv_chain: whole images
        resident in LDS — the 80-channel stage at 256x256 tiles), plus the weight gradients they feed.  Returns
        {block: "done"} for its identity blocks and {entry block: (dz of its output, dz1)} — or None when the stage / shape
        has no such kernel (then the per-block path below runs)."""
        li, j, depth = net.block_position(bi)
        e = bi - depth + 1
        split_ = dtype == torch.float32 and L.dt_code(dtype, mma=True) == L.MIL_DT_F32S
        if j != depth - 1 or depth < 2 or depth > 3 or not net.fuse_backward or not (dtype == torch.bfloat16 or split_):
            return None
        ent = blocks[e]
        if ent.stride != 2 or ent.downsample is None or any(b.stride != 1 or b.downsample is not None for b in blocks[e + 1:bi + 1]):
            return None
        cout = ent.conv1.out_channels
        n, h, w, cp = dz_out.shape
        if (cp, h, w) not in ops.RESIDENT_SHAPES or ops.bwd_fused_workspace_bytes(n, h, w, cout, cout, 3, 1, dtype) is not None:
            return None                         # widths with a fused dgrad+wgrad kernel keep it
        convs, src = [], None                   # src: chain output that is the gradient entering the current block (None: dz_out)
        for k in range(bi, e, -1):
            xin_k, o1_k, _ = saved["blocks"][k]
            w2d_k, _ = net._packed(f"b{k}.c2", blocks[k].conv2.weight, None, L.PACK_DGRAD, dtype)
            w1d_k, _ = net._packed(f"b{k}.c1", blocks[k].conv1.weight, None, L.PACK_DGRAD, dtype)
            convs.append(dict(w=w2d_k, act=o1_k))                                            # dmid_k = lrelu'(o1) * conv2^T(dz_k)
            convs.append(dict(w=w1d_k, res=dz_out if src is None else src, act=xin_k))       # dz_{k-1} = lrelu'(x) * (conv1^T(dmid_k) + dz_k)
            src = len(convs) - 1
        _xin_e, o1_e, _ = saved["blocks"][e]
        w2d_e, _ = net._packed(f"b{e}.c2", ent.conv2.weight, None, L.PACK_DGRAD, dtype)
        convs.append(dict(w=w2d_e, act=o1_e))                                                # the entry block's dz1
        outs = ops.conv_chain(dz_out, convs)
        if outs is None:
            return None
        done = {}
        dz_k = dz_out
        for i, k in enumerate(range(bi, e, -1)):
            xin_k, o1_k, _ = saved["blocks"][k]
            dmid_k, dz_prev = outs[2 * i], outs[2 * i + 1]
            grads[f"b{k}.c2"] = wgrad(o1_k, dz_k, cout, cout, key=(k, 2), ks=3, stride=1, pad=1, out=gout(blocks[k].conv2.weight, blocks[k].conv2.bias))
            grads[f"b{k}.c1"] = wgrad(xin_k, dmid_k, cout, cout, key=(k, 1), ks=3, stride=1, pad=1, out=gout(blocks[k].conv1.weight, blocks[k].conv1.bias))
            done[k] = "done"
            dz_k = dz_prev
        grads[f"b{e}.c2"] = wgrad(o1_e, dz_k, cout, cout, key=(e, 2), ks=3, stride=1, pad=1, out=gout(ent.conv2.weight, ent.conv2.bias))
        done[e] = (dz_k, outs[-1])
        return done

    pre = {}                                    # blocks whose data gradients a stage-wide chain has already produced
    with (batch if batch is not None else contextlib.nullcontext()):
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            xin, o1, _out = saved["blocks"][bi]
            cin, cout, s = blk.conv1.in_channels, blk.conv1.out_channels, blk.stride
            if bi not in pre:
                chain = stage_dgrad_chain(bi, dz)
                if chain is not None:
                    pre.update(chain)
            if pre.get(bi) == "done":
                continue
            w2d, _ = net._packed(f"b{bi}.c2", blk.conv2.weight, None, L.PACK_DGRAD, dtype)
            if bi in pre:                               # stage entry behind a chain: dz of its output and dz1 are there, dW2/db2 too
                dz, dz1 = pre[bi]
                fused = "chain"
            else:
                fused = fused_bwd(dz, w2d, o1, cout, cout, None, True, gout(blk.conv2.weight, blk.conv2.bias), key=(bi, 2)) if net.fuse_backward else None
            if fused == "chain":
                pass
            elif fused is not None:                     # one pass: dz1 and dW2/db2
                dz1, grads[f"b{bi}.c2"] = fused[0], (fused[1], fused[2])
            elif is_dense(dz, cout):
                raise RuntimeError("dense gradient layout without the fused backward")
            else:
                grads[f"b{bi}.c2"] = wgrad(o1, dz, cout, cout, key=(bi, 2), ks=3, stride=1, pad=1, out=gout(blk.conv2.weight, blk.conv2.bias))
                if s == 1 and blk.downsample is None and net.fuse_backward:
                    # 80 channels on 8x8 maps: the block's two transposed convs on the LDS-resident images, one launch:
                    # dz1 = lrelu'(o1) * conv2^T(dz),  dz(prev) = lrelu'(x) * (conv1^T(dz1) + dz)
                    w1d_, _ = net._packed(f"b{bi}.c1", blk.conv1.weight, None, L.PACK_DGRAD, dtype)
                    chain = ops.conv_pair(dz, w2d, None, w1d_, None, actA=o1, resB=dz, actB=xin if bi > 0 else None)
                    if chain is not None:
                        dz1, dz_prev = chain
                        grads[f"b{bi}.c1"] = wgrad(xin, dz1, cin, cout, key=(bi, 1), ks=3, stride=1, pad=1, out=gout(blk.conv1.weight, blk.conv1.bias))
                        dz = dz_prev
                        continue
                dz1 = ops.conv(dz, w2d, None, ops.cpad(cout), ks=3, stride=1, pad=1, act=o1)
            w1d, _ = net._packed(f"b{bi}.c1", blk.conv1.weight, None, L.PACK_DGRAD, dtype)
            mask = xin if bi > 0 else None          # block 0 reads the max-pool output (no activation in between)
            if s == 1 and blk.downsample is None and net.fuse_backward:
                fused = fused_bwd(dz1, w1d, xin, cin, cout, dz, mask is not None, gout(blk.conv1.weight, blk.conv1.bias), key=(bi, 1))
                if fused is not None:                   # one pass: previous block's dz and dW1/db1
                    dz, grads[f"b{bi}.c1"] = fused[0], (fused[1], fused[2])
                    continue
            pair = None
            if s == 2 and blk.downsample is not None and net.fuse_backward and not use_side:
                # both weight gradients of the stage-entry convs from one pass over the block input
                o = gout(blk.conv1.weight, blk.conv1.bias, blk.downsample[0].weight)
                pws = batch.ws.get(("p", bi)) if batch is not None else ws[-1]
                pair = ops.conv_wgrad_pair(xin, dz1, dz, cin, cout, workspace=pws, out=o)
            if pair is not None:
                if batch is not None:
                    batch.ws[("p", bi)] = pair[3]           # kept until the batched reduction has run
                else:
                    ws[-1] = pair[3]
                grads[f"b{bi}.c1"], grads[f"b{bi}.ds"] = (pair[0], pair[1]), (pair[2], None)
            else:
                grads[f"b{bi}.c1"] = wgrad(xin, dz1, cin, cout, key=(bi, 1), ks=3, stride=s, pad=1, out=gout(blk.conv1.weight, blk.conv1.bias))
            if blk.downsample is not None:
                if pair is None:
                    grads[f"b{bi}.ds"] = wgrad(xin, dz, cin, cout, key=(bi, 0), ks=1, stride=s, pad=0, want_bias=False,
                                               out=gout(blk.downsample[0].weight, None))
                if s == 2 and net.fuse_backward:     # both transposed convs + the mask in one pass over the compact dz maps
                    ws2, _ = net._packed(f"b{bi}.c1", blk.conv1.weight, blk.downsample[0].weight, L.PACK_DGRAD_S2, dtype)
                    # the gradient chain of the first (20-channel) stage below this point runs on kernels that read the
                    # dense layout, when every one of them exists for these shapes: the fused backward and the fused stem backward
                    dense_cx = None
                    split = dtype == torch.float32 and L.dt_code(dtype, mma=True) == L.MIL_DT_F32S
                    if (net.dense_grads and cin == STEM_WIDTH and ops.cpad(cin) != cin and (dtype == torch.bfloat16 or split) and bi > 0 and
                            all(b.stride == 1 and b.downsample is None for b in blocks[:bi]) and
                            ops.bwd_fused_workspace_bytes(xin.shape[0], xin.shape[1], xin.shape[2], cin, cin, 3, 1, dtype, True) is not None and
                            ops.stem_bwd_dense_ok(saved["x"] if saved["xs"] is None else saved["xs"], dtype)):
                        dense_cx = cin
                    fused = ops.conv_dgrad_s2(dz1, dz, ws2, ops.cpad(cin), xin.shape[1:3], act=mask, dense_cx=dense_cx)
                    if fused is not None:
                        dz = fused
                        continue
                wdd, _ = net._packed(f"b{bi}.ds", blk.downsample[0].weight, None, L.PACK_DGRAD, dtype)
                if s == 2:
                    addend = ops.conv(dz, wdd, None, ops.cpad(cin), ks=1, stride=1, pad=0, zero_insert=True,
                                      out_hw=xin.shape[1:3])
                else:
                    addend = ops.conv(dz, wdd, None, ops.cpad(cin), ks=1, stride=1, pad=0)
            else:
                addend = dz
            if s == 2:
                dz = ops.conv(dz1, w1d, None, ops.cpad(cin), ks=3, stride=1, pad=1, zero_insert=True,
                              out_hw=xin.shape[1:3], res=addend, act=mask)
            else:
                dz = ops.conv(dz1, w1d, None, ops.cpad(cin), ks=3, stride=1, pad=1, res=addend, act=mask)
        fused_stem = None
        if net.fuse_backward:               # pool backward + lrelu backward + stem wgrad in one pass (bf16 path)
            stem_ws = (lambda nb: batch.workspace(("s", 0), nb)) if batch is not None else None
            if saved["xs"] is None:
                fused_stem = ops.stem_bwd_fused_nchw(saved["x"], dz, saved["widx"], out=gout(net.conv1.weight, net.conv1.bias),
                                                     ws_alloc=stem_ws)
            else:
                fused_stem = ops.stem_bwd_fused(saved["xs"], dz, saved["widx"], out=gout(net.conv1.weight, net.conv1.bias),
                                                ws_alloc=stem_ws)
        if fused_stem is not None:
            grads["stem"] = fused_stem
        elif is_dense(dz, STEM_WIDTH):
            raise RuntimeError("dense gradient layout without the fused stem backward")
        else:
            dstem = ops.maxpool_bwd(dz, saved["widx"], saved["stem_hw"])
            if saved["xs"] is None:
                saved["xs"] = ops.stem_s2d(saved["x"], dz.dtype)
            grads["stem"] = wgrad(saved["xs"], dstem, 3, STEM_WIDTH, key=("stem", 0), ks=4, stride=1, pad=2, stem=True,
                                  out=gout(net.conv1.weight, net.conv1.bias))

    if use_side:
        for side in sides:
            main.wait_stream(side)
    flat = [grads["stem"][0], grads["stem"][1]]
    for bi, blk in enumerate(blocks):
        flat += [grads[f"b{bi}.c1"][0], grads[f"b{bi}.c1"][1], grads[f"b{bi}.c2"][0], grads[f"b{bi}.c2"][1]]
        if blk.downsample is not None:
            flat.append(grads[f"b{bi}.ds"][0])
    flat.append(dwfc)
    if direct:
        return [None] * len(flat)          # already accumulated into the parameters' .grad tensors
    return flat


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        mode = net.compute_dtype            # torch.bfloat16, torch.float32 (exact-f32 MFMA) or L.BF16X3 (fp32 tensors, split products)
        dtype = L.storage_dtype(mode)
        with L.f32_mma(L.mma_code(mode)):
            feats, saved = encoder_forward(net, x.detach(), dtype)
        ctx.net, ctx.saved, ctx.dtype, ctx.mode = net, saved, dtype, mode
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        with L.f32_mma(L.mma_code(ctx.mode)):
            grads = encoder_backward(ctx.net, ctx.saved, dfeats, ctx.dtype, allow_direct=all(ctx.needs_input_grad[2:]))
        ctx.saved = None
        return (None, None, *grads)
