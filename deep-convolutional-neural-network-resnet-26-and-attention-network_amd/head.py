"""Attention-MIL head on the HIP kernels (mil_head_fwd / mil_head_bwd of include/mil_hip.h) as a
torch.autograd.Function.  Reference arithmetic: gbm/model.py:198-246; segmented over bags so that a
batch of bags is one launch sequence while every reduction over instances stays inside its bag."""
import ctypes

import torch

from . import _lib as L
from . import ops

DROP_P = 0.25       # gbm/model.py:107
SMOOTHING = 0.25    # gbm/model.py:128
BN_EPS = 1e-5       # BatchNorm1d default, gbm/model.py:105
N_FEATS = 80


class BagLayout:
    """Device-side description of how the rows of H split into bags."""

    def __init__(self, sizes, device):
        sizes = [int(s) for s in sizes]
        if len(sizes) == 0 or min(sizes) < 2:
            # torch's batch-statistics BatchNorm1d refuses a single instance (gbm/model.py:105)
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{min(sizes) if sizes else 0}, {N_FEATS}])")
        self.sizes = sizes
        self.nbags = len(sizes)
        self.ntot = sum(sizes)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        self.offsets_host = offs
        # built on the host and uploaded once (a device-side repeat_interleave has to synchronise to size its output)
        import numpy as np
        self.offsets = torch.tensor(offs, dtype=torch.int32, device=device)
        self.inst_bag = torch.from_numpy(np.repeat(np.arange(self.nbags, dtype=np.int32), sizes)).to(device)

    _cache = {}

    @classmethod
    def cached(cls, sizes, device):
        """Layouts are immutable: training loops present the same bag sizes step after step."""
        key = (tuple(int(s) for s in sizes), str(device))
        lay = cls._cache.get(key)
        if lay is None:
            if len(cls._cache) > 64:
                cls._cache.clear()
            lay = cls._cache[key] = cls(sizes, device)
        return lay


def _weight_array(ws):
    arr = (ctypes.c_void_p * 11)()
    for i, w in enumerate(ws):
        if w.dtype != torch.float32 or not w.is_cuda:
            raise ValueError("head parameters must be CUDA fp32")
        arr[i] = w.data_ptr()
    return arr


class _HeadFn(torch.autograd.Function):
    """(H, 11 head parameters) -> (loss [nbags], l2 []) with grad; a1/wrois/bterm/kld/rec without."""

    @staticmethod
    def forward(ctx, H, layout, labels, keep_mask, class_weights, internals, direct, *weights):
        lib = L.lib()
        H = H.contiguous()
        ws = [w.detach().contiguous() for w in weights]
        dev = H.device
        ntot, nbags = layout.ntot, layout.nbags
        if tuple(H.shape) != (ntot, N_FEATS) or H.dtype != torch.float32:
            raise ValueError(f"H must be fp32 [{ntot},{N_FEATS}], got {tuple(H.shape)} {H.dtype}")
        if keep_mask is not None and (tuple(keep_mask.shape) != (ntot, N_FEATS) or keep_mask.dtype != torch.uint8):
            raise ValueError("keep_mask must be uint8 [N,80]")
        need = ctypes.c_size_t(0)
        L.check(lib.mil_head_workspace_floats(ctypes.byref(need), ntot, nbags), "mil_head_workspace_floats")
        work = torch.empty(need.value, dtype=torch.float32, device=dev)
        a1 = torch.empty((ntot, 3), dtype=torch.float32, device=dev)
        wrois = torch.empty(3 * ntot, dtype=torch.float32, device=dev)
        bterm = torch.empty(ntot, dtype=torch.float32, device=dev)
        kld = torch.empty(nbags, dtype=torch.float32, device=dev)
        rec = torch.zeros((nbags, lib.mil_head_rec_floats()), dtype=torch.float32, device=dev)
        arr = _weight_array(ws)
        L.check(lib.mil_head_fwd(H.data_ptr(), layout.offsets.data_ptr(), layout.inst_bag.data_ptr(), labels.data_ptr(),
                                 L.ptr(keep_mask), L.ptr(class_weights), arr, work.data_ptr(), a1.data_ptr(),
                                 wrois.data_ptr(), bterm.data_ptr(), kld.data_ptr(), rec.data_ptr(), ntot, nbags,
                                 ops.LEAK, DROP_P, SMOOTHING, BN_EPS, L.stream_ptr()), "mil_head_fwd")
        ctx.layout, ctx.keep_mask, ctx.ws, ctx.work = layout, keep_mask, ws, work
        ctx.params = weights               # the Parameters themselves: their .grad may be one flat bucket (dist.FlatParams)
        ctx.direct = bool(direct)          # opt-in (FlatParams sets Attention.direct_grad): accumulate into that bucket in place
        if internals is not None:          # forward hooks on head children read the kernels' own intermediates
            o = nbags * 2 * N_FEATS
            internals["stats"] = work[:o].view(nbags, 2, N_FEATS)                      # per bag: mean, 1/sqrt(var+eps)
            internals["t"] = work[o:o + ntot * 40].view(ntot, 40); o += ntot * 40       # tanh(attention.lin1)
            internals["v"] = work[o:o + ntot * 40].view(ntot, 40); o += ntot * 40       # buffer.lin1 output
            internals["araw"] = work[o:o + ntot * 3].view(ntot, 3)                      # attention output
        ctx.save_for_backward(H, bterm, rec)
        loss = rec[:, 6].clone()
        l2 = rec[0, 17].clone()
        ctx.mark_non_differentiable(a1, wrois, bterm, kld, rec)
        return loss, l2, a1, wrois, bterm, kld, rec

    @staticmethod
    def backward(ctx, g_loss, g_l2, *_unused):
        lib = L.lib()
        H, bterm, rec = ctx.saved_tensors
        layout = ctx.layout
        dev = H.device
        if g_loss is None:
            g_loss = torch.zeros(layout.nbags, dtype=torch.float32, device=dev)
        g_loss = g_loss.contiguous().float()
        g_l2 = None if g_l2 is None else g_l2.reshape(1).contiguous().float()
        dH = torch.empty_like(H)
        grads = torch.empty(lib.mil_head_grad_floats(), dtype=torch.float32, device=dev)
        arr = _weight_array(ctx.ws)
        L.check(lib.mil_head_bwd(H.data_ptr(), layout.offsets.data_ptr(), layout.inst_bag.data_ptr(), L.ptr(ctx.keep_mask),
                                 arr, ctx.work.data_ptr(), bterm.data_ptr(), rec.data_ptr(), g_loss.data_ptr(),
                                 L.ptr(g_l2), dH.data_ptr(), grads.data_ptr(), layout.ntot, layout.nbags, ops.LEAK,
                                 DROP_P, L.stream_ptr()), "mil_head_bwd")
        # Parameters whose .grad tensors sit back to back in one buffer, in this order (dist.FlatParams: the ten tensors from
        # context.bn.weight to buffer.classifier.bias): ONE add of the kernel's gradient block into that run instead of one
        # autograd accumulation (a torch add launch) per parameter — the kernel writes its gradients in the same order.
        # Only under the explicit opt-in (a FlatParams owns the gradients and the caller runs plain `loss.backward()`), and
        # only for parameters that required a gradient at forward time (a frozen one ends the run); without the opt-in every
        # gradient is RETURNED, so that torch.autograd.grad / backward(inputs=...) / gradient hooks see them and no .grad is
        # touched behind their back.
        n_fixed = 7                          # positional inputs in front of the weights
        wanted = [bool(ctx.needs_input_grad[n_fixed + i]) for i in range(len(ctx.ws))]
        out, o, run = [], 0, (_contiguous_grad_run(ctx.params, wanted) if ctx.direct else None)
        if run is not None:
            first, count, total = run
            g0 = ctx.params[first].grad
            off0 = sum(w.numel() for w in ctx.ws[:first])
            torch.empty(0, dtype=torch.float32, device=dev).set_(g0.untyped_storage(), g0.storage_offset(), (total,)).add_(grads[off0:off0 + total])
        for i, w in enumerate(ctx.ws):
            direct = run is not None and run[0] <= i < run[0] + run[1]
            out.append(None if direct else grads[o:o + w.numel()].view(w.shape))
            o += w.numel()
        return (dH, None, None, None, None, None, None, *out)


def _contiguous_grad_run(params, wanted=None):
    """(first index, count, total elements) of the longest prefix run of `params` (from index 0) whose existing fp32 .grad
    tensors are contiguous and adjacent in memory — and for which autograd wants a gradient (`wanted[i]`, from
    ctx.needs_input_grad) —, or None when fewer than two qualify."""
    run, nxt, total = 0, None, 0
    for i, p in enumerate(params):
        g = getattr(p, "grad", None)
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or not p.requires_grad:
            break
        if wanted is not None and not wanted[i]:
            break
        if nxt is not None and g.data_ptr() != nxt:
            break
        nxt = g.data_ptr() + g.numel() * 4
        total += g.numel()
        run += 1
    return (0, run, total) if run >= 2 else None


def head_apply(H, layout, labels, keep_mask, class_weights, weights, internals=None, direct=False):
    """direct=True (set through `Attention.direct_grad` by dist.FlatParams): the backward adds the head gradients straight
    into the parameters' existing .grad storage where that is one contiguous run and returns None for them — valid for plain
    `loss.backward()` accumulation only; leave it False for torch.autograd.grad / backward(inputs=...) / gradient hooks."""
    return _HeadFn.apply(H, layout, labels, keep_mask, class_weights, internals, direct, *weights)
