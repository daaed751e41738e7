"""Thin tensor-level wrappers over the C ABI (include/mil_hip.h).  Each wrapper checks shapes on the
host (a wrong extent in a hand-written kernel is a GPU fault, not an exception), allocates outputs with
torch (plumbing) and enqueues the HIP kernel on torch's current stream."""
import ctypes

import torch

from . import _lib as L

LEAK = 0.1   # LeakyReLU slope of the reference (nnBlocks.py:170, gbm/model.py:25)


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    `want(label)` selects which launches are bracketed; everything else runs untouched."""

    def __init__(self, want):
        self.want = want
        self.spans = []

    def bracket(self, label):
        if not self.want(label):
            return None
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        self.spans.append((label, start, end))
        return end

    def durations_ms(self):
        torch.cuda.synchronize()
        return [(label, s.elapsed_time(e)) for label, s, e in self.spans]


TIMER = None   # set to a KernelTimer by bench.py


class ReduceBatch:
    """Deferred slab reductions (mil_reduce_defer_begin / _end / mil_wgrad_reduce_all): inside `with batch:` every
    weight-gradient producer records its reduction instead of launching it; leaving the block runs them all in ONE
    launch.  The producers' workspaces must stay untouched until then: `workspace(key, nbytes)` hands out one persistent
    buffer per call site."""
    MAX_JOBS = 64

    def __init__(self, device):
        self.device = device
        self.rec = L.lib().mil_reduce_job_bytes()
        self.host = (ctypes.c_char * (self.rec * self.MAX_JOBS))()
        self.dev = torch.empty(self.rec * self.MAX_JOBS, dtype=torch.uint8, device=device)
        self.uploaded = None                 # bytes of the table the device copy holds
        self.ws = {}                         # persistent slab buffers, one per call site (about 0.7 GB at 2048 tiles of 256x256)
        self.handed = None                   # keys handed out inside the current `with` block

    def workspace(self, key, nbytes):
        if self.handed is not None:
            if key in self.handed:
                raise RuntimeError(f"slab workspace {key!r} handed out twice inside one deferred-reduction block: the second "
                                   "producer would overwrite slabs the batched reduction has not read yet")
            self.handed.add(key)
        t = self.ws.get(key)
        if t is None or t.numel() * 4 < nbytes:
            t = self.ws[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
        return t

    def __enter__(self):
        L.check(L.lib().mil_reduce_defer_begin(ctypes.addressof(self.host), self.MAX_JOBS), "mil_reduce_defer_begin")
        self.handed = set()
        return self

    def __exit__(self, exc_type, exc, tb):
        n = ctypes.c_int(0)
        self.handed = None
        L.check(L.lib().mil_reduce_defer_end(ctypes.byref(n)), "mil_reduce_defer_end")
        if exc_type is not None or n.value == 0:
            return False
        raw = bytes(self.host[: self.rec * n.value])
        if raw != self.uploaded:             # pointers and shapes repeat step after step: the table is uploaded once
            self.dev[: len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            self.uploaded = raw
        L.check(L.lib().mil_wgrad_reduce_all(self.dev.data_ptr(), ctypes.addressof(self.host), n.value, L.stream_ptr()),
                "mil_wgrad_reduce_all")
        return False


def cpad(c):
    return (c + 7) // 8 * 8


def _need(t, shape, dtype, name):
    if t is None:
        return
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous CUDA tensor")
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype:
        raise ValueError(f"{name}: expected {tuple(shape)} {dtype}, got {tuple(t.shape)} {t.dtype}")


def stem_s2d(x, dtype):
    """[n,3,H,W] fp32 NCHW -> [n,ceil(H/2),ceil(W/2),16] NHWC space-to-depth of `dtype`."""
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError(f"expected a CUDA fp32 [N,3,H,W] tile stack, got {tuple(x.shape)} {x.dtype} on {x.device}")
    x = x.contiguous()
    n, _, h, w = x.shape
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2, 16), dtype=dtype, device=x.device)
    L.check(L.lib().mil_stem_s2d(x.data_ptr(), out.data_ptr(), n, h, w, L.dt_code(dtype), L.stream_ptr()), "mil_stem_s2d")
    return out


def pack_weights(w, bias, mode, dtype):
    """fp32 [Cout,Cin,k,k] -> (packed MFMA-fragment weights, zero-padded fp32 bias)."""
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_cuda or w.dim() != 4:
        raise ValueError("weights must be CUDA fp32 [Cout,Cin,k,k]")
    w = w.contiguous()
    cout, cin, ks, _ = w.shape
    elems = ctypes.c_size_t(0)
    L.check(L.lib().mil_packed_weight_elems(ctypes.byref(elems), cout, cin, ks, mode), "mil_packed_weight_elems")
    if dtype == L.BF16X3:                 # the split path's filters: fp32-sized fragments [hi | lo], packed under MIL_DT_F32S
        with L.f32_mma(L.MIL_DT_F32S):
            return pack_weights(w, bias, mode, torch.float32)
    packed = torch.empty(elems.value, dtype=dtype, device=w.device)
    n_out = cin if mode == L.PACK_DGRAD else cout
    nt = (cpad(n_out) + 15) // 16
    bias_pad = torch.empty(nt * 16, dtype=torch.float32, device=w.device)
    b = None if bias is None else bias.detach().contiguous()
    L.check(L.lib().mil_pack_conv_weights(w.data_ptr(), L.ptr(b), packed.data_ptr(), bias_pad.data_ptr(), cout, cin, ks,
                                          mode, L.dt_code(dtype, mma=True), L.stream_ptr()), "mil_pack_conv_weights")
    return packed, bias_pad


def conv(x, wpack, bias_pad, cout_p, *, ks, stride, pad, out_hw=None, res=None, act=None, lrelu=False,
         zero_insert=False, slope=LEAK):
    """y = mask(lrelu?(conv(x)+bias+res)) — see mil_conv_igemm in include/mil_hip.h."""
    n, h, w, cin_p = x.shape
    if zero_insert:
        if out_hw is None:
            raise ValueError("zero_insert needs the full-resolution output size")
        ho, wo = out_hw
    else:
        ho = (h + 2 * pad - ks) // stride + 1 if ks != 4 else h
        wo = (w + 2 * pad - ks) // stride + 1 if ks != 4 else w
    y = torch.empty((n, ho, wo, cout_p), dtype=x.dtype, device=x.device)
    _need(x, x.shape, x.dtype, "x")
    _need(res, y.shape, x.dtype, "res")
    _need(act, y.shape, x.dtype, "act")
    end = TIMER.bracket(("conv", cin_p, cout_p, ks, stride, bool(zero_insert), n, ho, wo)) if TIMER else None
    L.check(L.lib().mil_conv_igemm(x.data_ptr(), wpack.data_ptr(), L.ptr(bias_pad), L.ptr(res), L.ptr(act), y.data_ptr(),
                                   n, h, w, cin_p, ho, wo, cout_p, ks, 1 if zero_insert else stride, pad,
                                   1 if zero_insert else 0, 1 if lrelu else 0, slope, L.dt_code(x.dtype, mma=True), L.stream_ptr()),
            "mil_conv_igemm")
    if end is not None:
        end.record()
    return y


def wgrad_workspace_bytes(n, h, w, cin, ho, wo, cout, ks, stride, pad, stem, dtype):
    need = ctypes.c_size_t(0)
    L.check(L.lib().mil_conv_wgrad_workspace(ctypes.byref(need), n, h, w, cin, ho, wo, cout, ks, stride, pad,
                                             1 if stem else 0, L.dt_code(dtype, mma=True)), "mil_conv_wgrad_workspace")
    return need.value


def conv_wgrad(x, dz, cin, cout, *, ks, stride, pad, stem=False, want_bias=True, workspace=None, out=None):
    """(dW [cout,cin,k,k] fp32, db [cout] fp32 or None) — see mil_conv_wgrad."""
    n, h, w, cin_p = x.shape
    _, ho, wo, cout_p = dz.shape
    _need(dz, (n, ho, wo, cpad(cout)), x.dtype, "dz")
    _need(x, (n, h, w, 16 if stem else cpad(cin)), x.dtype, "x")
    need = wgrad_workspace_bytes(n, h, w, cin, ho, wo, cout, ks, stride, pad, stem, x.dtype)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    kk = 7 if stem else ks
    if out is None:
        dw = torch.empty((cout, cin, kk, kk), dtype=torch.float32, device=x.device)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    else:                         # accumulate straight into caller-owned gradient tensors (p.grad views)
        dw, db = out
        _need(dw, (cout, cin, kk, kk), torch.float32, "dw")
        _need(db, (cout,), torch.float32, "db")
    end = TIMER.bracket(("wgrad", cin_p, cout_p, ks, stride, n, ho, wo)) if TIMER else None
    L.check(L.lib().mil_conv_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), L.ptr(db), workspace.data_ptr(),
                                   workspace.numel() * workspace.element_size(), n, h, w, cin, ho, wo, cout, ks, stride,
                                   pad, 1 if stem else 0, 0 if out is None else 1, L.dt_code(x.dtype, mma=True), L.stream_ptr()),
            "mil_conv_wgrad")
    if end is not None:
        end.record()
    return dw, db


def bwd_fused_workspace_bytes(n, h, w, cout, cin, ks, pad, dtype, dense=False):
    """Slab bytes the fused backward needs, or None when this shape (and gradient layout) has no fused kernel."""
    need = ctypes.c_size_t(0)
    rc = L.lib().mil_conv_bwd_fused_workspace(ctypes.byref(need), n, h, w, cout, cin, ks, pad, L.dt_code(dtype, dense, mma=True))
    if rc == 2:
        return None
    L.check(rc, "mil_conv_bwd_fused_workspace")
    return need.value


def conv_bwd_fused(dz, wpack_dgrad, x, cin, cout, *, addend=None, mask=True, ks=3, pad=1, workspace=None, slope=LEAK,
                   out=None):
    """(dx, dW, db) of a 3x3 stride-1 conv in one pass, or None if unsupported — see mil_conv_bwd_fused.  A dz with exactly
    `cout` (unpadded) channels selects the dense gradient layout (MIL_DT_BF16_DGRAD): addend and dx are then [n,h,w,cin]
    too, x keeps its padded channels."""
    n, h, w, cz = dz.shape
    dense = cz == cout and cpad(cout) != cout
    need = bwd_fused_workspace_bytes(n, h, w, cout, cin, ks, pad, dz.dtype, dense)
    if need is None:
        return None
    _need(dz, (n, h, w, cout if dense else cpad(cout)), dz.dtype, "dz")
    _need(x, (n, h, w, cpad(cin)), dz.dtype, "x")
    _need(addend, (n, h, w, cin if dense else cpad(cin)), dz.dtype, "addend")
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=dz.device)
    dx = torch.empty((n, h, w, cin), dtype=dz.dtype, device=dz.device) if dense else torch.empty_like(x)
    if out is None:
        dw = torch.empty((cout, cin, ks, ks), dtype=torch.float32, device=dz.device)
        db = torch.empty(cout, dtype=torch.float32, device=dz.device)
    else:
        dw, db = out
        _need(dw, (cout, cin, ks, ks), torch.float32, "dw")
        _need(db, (cout,), torch.float32, "db")
    end = TIMER.bracket(("bwd_fused", cpad(cout), cpad(cin), ks, 1, False, n, h, w, addend is not None)) if TIMER else None
    L.check(L.lib().mil_conv_bwd_fused(dz.data_ptr(), wpack_dgrad.data_ptr(), x.data_ptr(), L.ptr(addend), dx.data_ptr(),
                                       dw.data_ptr(), db.data_ptr(), workspace.data_ptr(),
                                       workspace.numel() * workspace.element_size(), n, h, w, cout, cin, ks, pad,
                                       1 if mask else 0, 0 if out is None else 1, slope, L.dt_code(dz.dtype, dense, mma=True), L.stream_ptr()),
            "mil_conv_bwd_fused")
    if end is not None:
        end.record()
    return dx, dw, db


def maxpool_fwd(x):
    n, h, w, cp = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, ho, wo, cp), dtype=x.dtype, device=x.device)
    widx = torch.empty((n, ho, wo, cp), dtype=torch.uint8, device=x.device)
    L.check(L.lib().mil_maxpool_fwd(x.data_ptr(), y.data_ptr(), widx.data_ptr(), n, h, w, cp, L.dt_code(x.dtype),
                                    L.stream_ptr()), "mil_maxpool_fwd")
    return y, widx


def maxpool_bwd(gy, widx, in_hw, lrelu_mask=True, slope=LEAK):
    """Gradient of the pooled tensor scattered back to the [n,H,W,cp] pool input; with lrelu_mask the LeakyReLU
    backward of that input is applied too (from the sign bit recorded in widx)."""
    n, ho, wo, cp = widx.shape
    h, w = in_hw
    if ((h - 1) // 2 + 1, (w - 1) // 2 + 1) != (ho, wo):
        raise ValueError(f"pool input {h}x{w} does not produce {ho}x{wo}")
    _need(gy, widx.shape, gy.dtype, "gy")
    gx = torch.empty((n, h, w, cp), dtype=gy.dtype, device=gy.device)
    L.check(L.lib().mil_maxpool_bwd(gy.data_ptr(), widx.data_ptr(), gx.data_ptr(), n, h, w, cp, 1 if lrelu_mask else 0,
                                    slope, L.dt_code(gy.dtype), L.stream_ptr()), "mil_maxpool_bwd")
    return gx


# channel counts for which the one-pass block forward beats the two persistent launches (measured: 40 channels leave
# room for only one workgroup per CU and lose)
BLOCK_FWD_CHANNELS = (24,)


def conv_block_fwd(x, wpack1, bias1, wpack2, bias2, *, slope=LEAK):
    """(o1, y) of an identity-shortcut residual block in one pass (see mil_conv_block_fwd), or None when the shape/dtype
    has no such kernel."""
    n, h, w, cp = x.shape
    code = L.dt_code(x.dtype, mma=True)
    if code == L.MIL_DT_F32S:                 # fp32 tensors, split products: the 20-channel stage (16 x 8 tiles)
        if cp != 24 or h < 8 or w < 16:
            return None
    elif x.dtype != torch.bfloat16 or cp not in BLOCK_FWD_CHANNELS or h < 16 or w < 16:
        return None
    _need(x, x.shape, x.dtype, "x")
    o1 = torch.empty_like(x)
    y = torch.empty_like(x)
    end = TIMER.bracket(("block_fwd", cp, n, h, w)) if TIMER else None
    rc = L.lib().mil_conv_block_fwd(x.data_ptr(), wpack1.data_ptr(), L.ptr(bias1), wpack2.data_ptr(), L.ptr(bias2),
                                    o1.data_ptr(), y.data_ptr(), n, h, w, cp, slope, code, L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_conv_block_fwd")
    if end is not None:
        end.record()
    return o1, y


RESIDENT_SHAPES = ((80, 8, 8), (64, 16, 16), (80, 10, 10), (64, 19, 19))     # (padded channels, H, W) with a pixel-resident kernel (the last two: the 300x300 driver size)


class _ChainConv(ctypes.Structure):
    _fields_ = [("wpack", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p), ("act", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("lrelu", ctypes.c_int), ("pad_", ctypes.c_int)]


def conv_chain(x, convs, *, slope=LEAK):
    """Outputs of a chain of 3x3 stride-1 convs on LDS-resident whole images, one launch (see mil_conv_chain), or None
    when the shape/dtype has no such kernel.  `convs`: dicts with w, and optionally bias, res, act, lrelu; res / act are
    tensors, or an int k = the output of conv k of this chain (k earlier than the conv that names it)."""
    n, h, w, cp = x.shape
    code = L.dt_code(x.dtype, mma=True)
    if code not in (L.MIL_DT_BF16, L.MIL_DT_F32S) or (cp, h, w) not in RESIDENT_SHAPES or not 1 <= len(convs) <= 6:
        return None
    _need(x, x.shape, x.dtype, "x")
    outs = [torch.empty_like(x) for _ in convs]
    arr = (_ChainConv * len(convs))()
    for k, c in enumerate(convs):
        ops_ = {}
        for name in ("res", "act"):
            t = c.get(name)
            if isinstance(t, int):
                if not 0 <= t < k:
                    raise ValueError(f"conv {k}: {name} refers to conv {t}, which is not earlier in the chain")
                t = outs[t]
            _need(t, x.shape, x.dtype, name)
            ops_[name] = t
        arr[k].wpack = c["w"].data_ptr()
        arr[k].bias = L.ptr(c.get("bias"))
        arr[k].res, arr[k].act = L.ptr(ops_["res"]), L.ptr(ops_["act"])
        arr[k].out = outs[k].data_ptr()
        arr[k].lrelu = 1 if c.get("lrelu") else 0
    end = TIMER.bracket(("chain", cp, n, h, w, len(convs))) if TIMER else None
    rc = L.lib().mil_conv_chain(x.data_ptr(), ctypes.addressof(arr), len(convs), n, h, w, cp, slope, code, L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_conv_chain")
    if end is not None:
        end.record()
    return outs


def conv_pair(x, wA, biasA, wB, biasB, *, resA=None, actA=None, lreluA=False, resB=None, actB=None, lreluB=False, slope=LEAK):
    """(outA, outB) = two 3x3 stride-1 convs back to back on LDS-resident whole images in one launch (conv_chain with two
    convs: 80 channels on 8x8 maps, 64 channels on 16x16 maps), or None when the shape/dtype has no such kernel."""
    outs = conv_chain(x, [dict(w=wA, bias=biasA, res=resA, act=actA, lrelu=lreluA),
                          dict(w=wB, bias=biasB, res=resB, act=actB, lrelu=lreluB)], slope=slope)
    return None if outs is None else (outs[0], outs[1])


def conv_s2_entry(x, wpack3, bias_pad, wpack1, cout_p, *, slope=LEAK):
    """(lrelu(conv3x3_s2(x)+b), conv1x1_s2(x)) in one pass over x (see mil_conv_s2_entry), or None when the shape/dtype
    has no such kernel."""
    n, h, w, cin_p = x.shape
    code = L.dt_code(x.dtype, mma=True)
    if code not in (L.MIL_DT_BF16, L.MIL_DT_F32S):
        return None
    _need(x, x.shape, x.dtype, "x")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y1 = torch.empty((n, ho, wo, cout_p), dtype=x.dtype, device=x.device)
    y2 = torch.empty_like(y1)
    end = TIMER.bracket(("s2_entry", cin_p, cout_p, n, h, w)) if TIMER else None
    rc = L.lib().mil_conv_s2_entry(x.data_ptr(), wpack3.data_ptr(), L.ptr(bias_pad), wpack1.data_ptr(), y1.data_ptr(),
                                   y2.data_ptr(), n, h, w, cin_p, cout_p, slope, code, L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_conv_s2_entry")
    if end is not None:
        end.record()
    return y1, y2


WGRAD_PAIR = True      # False (set from Python): the two stage-entry weight gradients as separate launches (A/B runs)


def conv_wgrad_pair(x, dz1, dz2, cin, cout, *, workspace=None, out=None):
    """(dW3, db3, dW1) of a stage-entry block's 3x3/s2 conv and 1x1/s2 projection from one pass over x (see
    mil_conv_wgrad_pair), or None when the shape/dtype has no such kernel.  out = (dw3, db3, dw1) accumulates in place."""
    n, h, w, _ = x.shape
    _, ho, wo, _ = dz1.shape
    code = L.dt_code(x.dtype, mma=True)
    if code not in (L.MIL_DT_BF16, L.MIL_DT_F32S) or not WGRAD_PAIR:
        return None
    _need(x, (n, h, w, cpad(cin)), x.dtype, "x")
    _need(dz1, (n, ho, wo, cpad(cout)), x.dtype, "dz1")
    _need(dz2, (n, ho, wo, cpad(cout)), x.dtype, "dz2")
    need = ctypes.c_size_t(0)
    rc = L.lib().mil_conv_wgrad_pair_workspace(ctypes.byref(need), n, h, w, cin, ho, wo, cout, code)
    if rc == 2:
        return None
    L.check(rc, "mil_conv_wgrad_pair_workspace")
    if workspace is None or workspace.numel() * workspace.element_size() < need.value:
        workspace = torch.empty((need.value + 3) // 4, dtype=torch.float32, device=x.device)
    if out is None:
        dw3 = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
        db3 = torch.empty(cout, dtype=torch.float32, device=x.device)
        dw1 = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=x.device)
    else:
        dw3, db3, dw1 = out
        _need(dw3, (cout, cin, 3, 3), torch.float32, "dw3")
        _need(db3, (cout,), torch.float32, "db3")
        _need(dw1, (cout, cin, 1, 1), torch.float32, "dw1")
    rc = L.lib().mil_conv_wgrad_pair(x.data_ptr(), dz1.data_ptr(), dz2.data_ptr(), dw3.data_ptr(), db3.data_ptr(), dw1.data_ptr(),
                                     workspace.data_ptr(), workspace.numel() * workspace.element_size(), n, h, w, cin, ho, wo,
                                     cout, 0 if out is None else 1, code, L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_conv_wgrad_pair")
    return dw3, db3, dw1, workspace


def conv_dgrad_s2(dz1, dz2, wpack, cx_p, out_hw, *, act=None, slope=LEAK, dense_cx=None):
    """lrelu'(act) * (conv3x3_s2^T(dz1) + conv1x1_s2^T(dz2)) in one pass (see mil_conv_dgrad_s2), or None when the
    shape/dtype has no such kernel.  dense_cx = the unpadded channel count: the result is [n,H,W,dense_cx] (dense gradient
    layout, MIL_DT_BF16_DGRAD); act keeps its padded channels."""
    n, h, w, cz_p = dz1.shape
    hh, ww = out_hw
    dense = dense_cx is not None
    if dz1.dtype != torch.bfloat16 and L.dt_code(dz1.dtype, mma=True) != L.MIL_DT_F32S:
        return None               # bf16, or fp32 tensors with split-precision products (the 40 -> 24 channel entry)
    _need(dz1, dz1.shape, dz1.dtype, "dz1")
    _need(dz2, dz1.shape, dz1.dtype, "dz2")
    y = torch.empty((n, hh, ww, dense_cx if dense else cx_p), dtype=dz1.dtype, device=dz1.device)
    _need(act, (n, hh, ww, cx_p), dz1.dtype, "act")
    end = TIMER.bracket(("dgrad_s2", cz_p, cx_p, n, hh, ww)) if TIMER else None
    rc = L.lib().mil_conv_dgrad_s2(dz1.data_ptr(), L.ptr(dz2), wpack.data_ptr(), L.ptr(act), y.data_ptr(), n, h, w, cz_p,
                                   hh, ww, cx_p, slope, L.dt_code(dz1.dtype, dense, mma=True), L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_conv_dgrad_s2")
    if end is not None:
        end.record()
    return y


def stem_fwd_fused(x, wpack, bias_pad, cout_p, *, slope=LEAK, dtype=torch.bfloat16, keep_s2d=True):
    """(xs, pool, widx) of the whole stem in one pass over the fp32 NCHW tiles (see mil_stem_fwd_fused), or None when
    the shape/dtype has no fused kernel (the caller then runs stem_s2d / conv / maxpool_fwd).  keep_s2d=False: no
    space-to-depth copy is written (xs is None); the backward then reads x itself (stem_bwd_fused_nchw)."""
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError(f"expected a CUDA fp32 [N,3,H,W] tile stack, got {tuple(x.shape)} {x.dtype} on {x.device}")
    x = x.contiguous()
    n, c, h, w = x.shape
    code = L.dt_code(dtype, mma=True)
    if code == L.MIL_DT_F32S:               # split precision: fp32 pooled map, never an s2d copy (the backward reads x)
        if keep_s2d or cout_p != 24:
            return None
    elif dtype != torch.bfloat16:
        return None
    if h % 2 or w % 4 or cout_p not in (24, 64) or x.data_ptr() % 16:
        return None
    h2, w2 = h // 2, w // 2
    hp, wp = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    xs = torch.empty((n, h2, w2, 16), dtype=dtype, device=x.device) if keep_s2d else None
    pool = torch.empty((n, hp, wp, cout_p), dtype=dtype, device=x.device)
    widx = torch.empty((n, hp, wp, cout_p), dtype=torch.uint8, device=x.device)
    end = TIMER.bracket(("stem_fwd", cout_p, n, h, w)) if TIMER else None
    rc = L.lib().mil_stem_fwd_fused(x.data_ptr(), wpack.data_ptr(), L.ptr(bias_pad), L.ptr(xs), pool.data_ptr(),
                                    widx.data_ptr(), n, h, w, cout_p, slope, code, L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_stem_fwd_fused")
    if end is not None:
        end.record()
    return xs, pool, widx


def stem_fwd_fused_xs(xs, wpack, bias_pad, cout_p, *, slope=LEAK):
    """(pool, widx) of the whole stem in one pass over the bf16 space-to-depth tiles xs [n,H2,W2,16] (see
    mil_stem_fwd_fused_xs), or None when the shape has no fused kernel."""
    if xs.dim() != 4 or xs.shape[3] != 16 or xs.dtype != torch.bfloat16 or not xs.is_cuda or not xs.is_contiguous():
        raise ValueError(f"expected a contiguous CUDA bf16 [N,H/2,W/2,16] tensor, got {tuple(xs.shape)} {xs.dtype}")
    n, h2, w2, _ = xs.shape
    if cout_p not in (24, 64) or xs.data_ptr() % 16:
        return None
    hp, wp = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    pool = torch.empty((n, hp, wp, cout_p), dtype=torch.bfloat16, device=xs.device)
    widx = torch.empty((n, hp, wp, cout_p), dtype=torch.uint8, device=xs.device)
    end = TIMER.bracket(("stem_fwd_xs", cout_p, n, 2 * h2, 2 * w2)) if TIMER else None
    rc = L.lib().mil_stem_fwd_fused_xs(xs.data_ptr(), wpack.data_ptr(), L.ptr(bias_pad), pool.data_ptr(), widx.data_ptr(), n, h2, w2,
                                       cout_p, slope, L.MIL_DT_BF16, L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_stem_fwd_fused_xs")
    if end is not None:
        end.record()
    return pool, widx


def stem_bwd_fused(xs, g_pool, widx, *, workspace=None, out=None, slope=LEAK, ws_alloc=None):
    """(dW [20,3,7,7], db [20]) of the stem from the pooled-output gradient in one pass (see mil_stem_bwd_fused),
    or None when the shape/dtype has no fused kernel."""
    n, h2, w2, c = xs.shape
    dense = g_pool.shape[-1] == 20            # dense gradient layout (MIL_DT_BF16_DGRAD)
    need = ctypes.c_size_t(0)
    rc = L.lib().mil_stem_bwd_fused_workspace(ctypes.byref(need), n, h2, w2, L.dt_code(xs.dtype, dense))
    if rc == 2:
        return None
    L.check(rc, "mil_stem_bwd_fused_workspace")
    hp, wp = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    _need(xs, (n, h2, w2, 16), xs.dtype, "xs")
    _need(g_pool, (n, hp, wp, 20 if dense else 24), xs.dtype, "g_pool")
    _need(widx, (n, hp, wp, 24), torch.uint8, "widx")
    if ws_alloc is not None:                 # deferred reductions: the slab buffer must outlive this call
        workspace = ws_alloc(need.value)
    if workspace is None or workspace.numel() * workspace.element_size() < need.value:
        workspace = torch.empty((need.value + 3) // 4, dtype=torch.float32, device=xs.device)
    if out is None:
        dw = torch.empty((20, 3, 7, 7), dtype=torch.float32, device=xs.device)
        db = torch.empty(20, dtype=torch.float32, device=xs.device)
    else:
        dw, db = out
        _need(dw, (20, 3, 7, 7), torch.float32, "dw")
        _need(db, (20,), torch.float32, "db")
    end = TIMER.bracket(("stem_bwd_xs", n, 2 * h2, 2 * w2)) if TIMER else None
    L.check(L.lib().mil_stem_bwd_fused(xs.data_ptr(), g_pool.data_ptr(), widx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                       workspace.data_ptr(), workspace.numel() * workspace.element_size(), n, h2, w2,
                                       slope, 0 if out is None else 1, L.dt_code(xs.dtype, dense), L.stream_ptr()),
            "mil_stem_bwd_fused")
    if end is not None:
        end.record()
    return dw, db


def stem_bwd_fused_nchw(x, g_pool, widx, *, workspace=None, out=None, slope=LEAK, ws_alloc=None):
    """stem_bwd_fused without a kept space-to-depth copy: reads the fp32 tiles x [n,3,H,W] (see mil_stem_bwd_fused_nchw);
    None when the shape/dtype/alignment has no such kernel."""
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
        return None
    n, _, h, w = x.shape
    dense = g_pool.shape[-1] == 20            # dense gradient layout (MIL_DT_BF16_DGRAD / MIL_DT_F32S_DGRAD)
    need = ctypes.c_size_t(0)
    rc = L.lib().mil_stem_bwd_fused_nchw_workspace(ctypes.byref(need), n, h, w, L.dt_code(g_pool.dtype, dense, mma=True))
    if rc == 2 or x.data_ptr() % 16:
        return None
    L.check(rc, "mil_stem_bwd_fused_nchw_workspace")
    h2, w2 = h // 2, w // 2
    hp, wp = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    _need(g_pool, (n, hp, wp, 20 if dense else 24), g_pool.dtype, "g_pool")
    _need(widx, (n, hp, wp, 24), torch.uint8, "widx")
    if ws_alloc is not None:                 # deferred reductions: the slab buffer must outlive this call
        workspace = ws_alloc(need.value)
    if workspace is None or workspace.numel() * workspace.element_size() < need.value:
        workspace = torch.empty((need.value + 3) // 4, dtype=torch.float32, device=x.device)
    if out is None:
        dw = torch.empty((20, 3, 7, 7), dtype=torch.float32, device=x.device)
        db = torch.empty(20, dtype=torch.float32, device=x.device)
    else:
        dw, db = out
        _need(dw, (20, 3, 7, 7), torch.float32, "dw")
        _need(db, (20,), torch.float32, "db")
    end = TIMER.bracket(("stem_bwd", n, h, w)) if TIMER else None
    rc = L.lib().mil_stem_bwd_fused_nchw(x.data_ptr(), g_pool.data_ptr(), widx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                         workspace.data_ptr(), workspace.numel() * workspace.element_size(), n, h, w,
                                         slope, 0 if out is None else 1, L.dt_code(g_pool.dtype, dense, mma=True), L.stream_ptr())
    if rc == 2:
        return None
    L.check(rc, "mil_stem_bwd_fused_nchw")
    if end is not None:
        end.record()
    return dw, db


def stem_bwd_dense_ok(src, dtype):
    """True when the fused stem backward for this saved input (the fp32 tiles [n,3,H,W] or the s2d copy [n,H2,W2,16])
    exists with the dense pooled-gradient layout."""
    need = ctypes.c_size_t(0)
    if src.dim() == 4 and src.shape[1] == 3 and src.dtype == torch.float32:
        n, _, h, w = src.shape
        if not src.is_contiguous() or src.data_ptr() % 16:
            return False
        return L.lib().mil_stem_bwd_fused_nchw_workspace(ctypes.byref(need), n, h, w, L.dt_code(dtype, True, mma=True)) == 0
    n, h2, w2, _ = src.shape
    return L.lib().mil_stem_bwd_fused_workspace(ctypes.byref(need), n, h2, w2, L.dt_code(dtype, True)) == 0


def avgpool_fc_fwd(x, wfc, c, bias=None):
    n, h, w, cp = x.shape
    nf = wfc.shape[0]
    _need(wfc, (nf, c), torch.float32, "fc.weight")
    _need(bias, (nf,), torch.float32, "fc.bias")
    pooled = torch.empty((n, c), dtype=torch.float32, device=x.device)
    feats = torch.empty((n, nf), dtype=torch.float32, device=x.device)
    L.check(L.lib().mil_avgpool_fc_fwd(x.data_ptr(), wfc.data_ptr(), L.ptr(bias), pooled.data_ptr(), feats.data_ptr(), n,
                                       h * w, cp, c, nf, L.dt_code(x.dtype), L.stream_ptr()), "mil_avgpool_fc_fwd")
    return pooled, feats


def avgpool_fc_bwd(dfeats, wfc, pooled, act, c, slope=LEAK, out=None, want_bias=False, out_bias=None):
    n, h, w, cp = act.shape
    nf = wfc.shape[0]
    _need(dfeats, (n, nf), torch.float32, "dfeats")
    dz = torch.empty_like(act)
    dwfc = torch.empty((nf, c), dtype=torch.float32, device=act.device) if out is None else out
    _need(dwfc, (nf, c), torch.float32, "dwfc")
    dbias = None
    if want_bias:
        dbias = torch.empty(nf, dtype=torch.float32, device=act.device) if out_bias is None else out_bias
    L.check(L.lib().mil_avgpool_fc_bwd(dfeats.data_ptr(), wfc.data_ptr(), pooled.data_ptr(), act.data_ptr(), dz.data_ptr(),
                                       None, None, n, h * w, cp, c, nf, 0, slope, L.dt_code(act.dtype), L.stream_ptr()),
            "mil_avgpool_fc_bwd")
    need = ctypes.c_size_t(0)
    L.check(L.lib().mil_fc_wgrad_workspace(ctypes.byref(need), n, c, nf), "mil_fc_wgrad_workspace")
    ws = torch.empty((need.value + 3) // 4, dtype=torch.float32, device=act.device)
    L.check(L.lib().mil_fc_wgrad(dfeats.data_ptr(), pooled.data_ptr(), dwfc.data_ptr(), L.ptr(dbias), ws.data_ptr(),
                                 ws.numel() * 4, n, c, nf, 0 if out is None else 1, L.stream_ptr()), "mil_fc_wgrad")
    if want_bias:
        return dz, dwfc, dbias
    return dz, dwfc


# ---- wide (multiples of 64 channels) layers: the alt_resnet configuration -----------------------------------
def wide_pack_weights(w, mode, dtype):
    w = w.detach().contiguous()
    cout, cin, ks, _ = w.shape
    elems = ctypes.c_size_t(0)
    L.check(L.lib().mil_wide_packed_elems(ctypes.byref(elems), cout, cin, ks, mode), "mil_wide_packed_elems")
    packed = torch.empty(elems.value, dtype=dtype, device=w.device)
    L.check(L.lib().mil_wide_pack_weights(w.data_ptr(), packed.data_ptr(), cout, cin, ks, mode, L.dt_code(dtype),
                                          L.stream_ptr()), "mil_wide_pack_weights")
    return packed


def wide_conv(x, wpack, cout, *, ks, stride, pad, out_hw=None, res=None, act=None, relu=False, zero_insert=False,
              slope=0.0, bias=None):
    n, h, w, cin = x.shape
    if zero_insert:
        ho, wo = out_hw
    else:
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
    _need(res, y.shape, x.dtype, "res")
    _need(act, y.shape, x.dtype, "act")
    L.check(L.lib().mil_wide_conv(x.data_ptr(), wpack.data_ptr(), L.ptr(bias), L.ptr(res), L.ptr(act), y.data_ptr(), n, h, w,
                                  cin, ho, wo, cout, ks, 1 if zero_insert else stride, pad, 1 if zero_insert else 0,
                                  1 if relu else 0, slope, L.dt_code(x.dtype), L.stream_ptr()), "mil_wide_conv")
    return y


def gconv_supported(cin_x, cout_x, ks, stride):
    """Whether the gather-GEMM kernel runs a conv that contracts `cin_x` channels into `cout_x` (bf16 only)."""
    return bool(L.lib().mil_gconv_supported(cin_x, cout_x, ks, stride))


def gconv_pack_weights(w, mode):
    w = w.detach().contiguous()
    cout, cin, ks, _ = w.shape
    elems = ctypes.c_size_t(0)
    L.check(L.lib().mil_gconv_packed_elems(ctypes.byref(elems), cout, cin, ks, mode), "mil_gconv_packed_elems")
    packed = torch.empty(elems.value, dtype=torch.bfloat16, device=w.device)
    L.check(L.lib().mil_gconv_pack_weights(w.data_ptr(), packed.data_ptr(), cout, cin, ks, mode, L.stream_ptr()),
            "mil_gconv_pack_weights")
    return packed


def gconv(x, wpack, cout, *, ks, stride, pad, transposed=False, out_hw=None, res=None, act=None, relu=False, slope=0.0):
    """y = mask(relu?(conv(x) + res?)) (transposed: the conv's data gradient, x = dz, out_hw = the conv's input extent)."""
    n, h, w, cin = x.shape
    if transposed:
        ho, wo = out_hw if out_hw is not None else (h * stride, w * stride)
    else:
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
    _need(res, y.shape, x.dtype, "res")
    _need(act, y.shape, x.dtype, "act")
    L.check(L.lib().mil_gconv(x.data_ptr(), wpack.data_ptr(), L.ptr(res), L.ptr(act), y.data_ptr(), n, h, w, cin, ho, wo, cout,
                              ks, stride, pad, 1 if transposed else 0, 1 if relu else 0, slope, L.stream_ptr()), "mil_gconv")
    return y


def wide_wgrad(x, dz, cin, cout, *, ks, stride, pad, workspace=None, out=None):
    n, h, w, _ = x.shape
    _, ho, wo, _ = dz.shape
    _need(x, (n, h, w, cin), x.dtype, "x")
    _need(dz, (n, ho, wo, cout), x.dtype, "dz")
    need = ctypes.c_size_t(0)
    L.check(L.lib().mil_wide_wgrad_workspace(ctypes.byref(need), n, h, w, cin, ho, wo, cout, ks, stride, pad,
                                             L.dt_code(x.dtype)), "mil_wide_wgrad_workspace")
    if workspace is None or workspace.numel() * workspace.element_size() < need.value:
        workspace = torch.empty((need.value + 3) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, cin, ks, ks), dtype=torch.float32, device=x.device) if out is None else out
    _need(dw, (cout, cin, ks, ks), torch.float32, "dw")
    L.check(L.lib().mil_wide_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), workspace.data_ptr(),
                                   workspace.numel() * workspace.element_size(), n, h, w, cin, ho, wo, cout, ks, stride, pad,
                                   0 if out is None else 1, L.dt_code(x.dtype), L.stream_ptr()), "mil_wide_wgrad")
    return dw, workspace
