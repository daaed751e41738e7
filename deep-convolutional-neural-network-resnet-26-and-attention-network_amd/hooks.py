"""Forward hooks on the children of the fused modules.

The reference's children are live `nn.Module`s, so a driver may attach forward hooks to them
(`prime_activation_summary`, gbm/classify_combined.py:418).  Here the children are parameter
containers whose arithmetic runs inside fused HIP kernels, so the fused encoder / head call the hooks themselves
with the tensors the kernels have materialised, converted to the reference's layout (NCHW fp32, real channels
only) — and only when a hook is registered: an un-hooked forward never builds these views.

A hook on a child whose output the fused path never materialises (e.g. the pre-activation output of a conv
inside a block) raises at forward time instead of silently not firing.
"""
import torch


def hooked(module):
    """True when `module` carries a forward hook or a forward pre-hook."""
    return bool(module._forward_hooks) or bool(module._forward_pre_hooks)


def any_hooked(modules):
    return any(hooked(m) for m in modules)


def nchw(t, channels):
    """Channel-padded NHWC activation of any compute dtype -> what the reference module would have returned."""
    return t[..., :channels].permute(0, 3, 1, 2).float()


def fire(module, args, output):
    """Run `module`'s forward pre-hooks and forward hooks as `nn.Module.__call__` would; returns the output (a hook may
    replace it for downstream *hooks*, but it cannot alter the fused arithmetic — such a return value raises)."""
    if not isinstance(args, tuple):
        args = (args,)
    for hook_id, hook in list(module._forward_pre_hooks.items()):
        if hook_id in module._forward_pre_hooks_with_kwargs:
            res = hook(module, args, {})
        else:
            res = hook(module, args)
        if res is not None:
            raise RuntimeError(f"forward pre-hook on {type(module).__name__} returned a replacement input: the fused HIP "
                               "path cannot re-route its operands through a hook")
    for hook_id, hook in list(module._forward_hooks.items()):
        if hook_id in module._forward_hooks_with_kwargs:
            res = hook(module, args, {}, output)
        else:
            res = hook(module, args, output)
        if res is not None:
            raise RuntimeError(f"forward hook on {type(module).__name__} returned a replacement output: the fused HIP "
                               "path cannot re-route its results through a hook")
    return output


def refuse(owner, names):
    """Raise for hooks on children whose tensors never exist outside a fused kernel."""
    for name in names:
        child = owner.get_submodule(name)
        if hooked(child):
            raise RuntimeError(f"a forward hook is registered on '{name}', whose output the fused HIP kernels never "
                               "materialise (it lives in registers/LDS only); hook the enclosing block or stage instead")


def s2d_to_nchw(xs):
    """bf16 space-to-depth tiles [T,H/2,W/2,16] (channel = c*4 + dy*2 + dx) -> the fp32 [T,3,H,W] stack they stand for."""
    t, h2, w2, _ = xs.shape
    return xs[..., :12].float().view(t, h2, w2, 3, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(t, 3, 2 * h2, 2 * w2).contiguous()
