"""ctypes binding of libmil_hip.so (C ABI declared in include/mil_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
"""
import contextlib
import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  (must be imported first: the HIP runtime torch loaded is the one we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIL_LIB_PATH selects another build of the same library (the NaN-poisoned diagnostic build, `make -C csrc POISON=1`)
LIB_PATH = os.environ.get("MIL_LIB_PATH") or os.path.join(_HERE, "libmil_hip.so")

MIL_DT_F32, MIL_DT_BF16, MIL_DT_BF16_DGRAD, MIL_DT_F32S, MIL_DT_F32S_DGRAD = 0, 1, 2, 3, 4
# compute_dtype value of the split-precision path: fp32 tensors (every pointwise kernel of the fp32 path), convolutions
# as three bf16 MFMAs per k-step on hi/lo-split operands (MIL_DT_F32S in include/mil_hip.h)
BF16X3 = "bf16x3"
PACK_FWD, PACK_DGRAD, PACK_STEM, PACK_DGRAD_S2 = 0, 1, 2, 3
_ERR = {1: "invalid argument", 2: "unsupported shape / channel configuration", 3: "kernel launch failed"}


class MilLibraryError(RuntimeError):
    pass


def build_library(verbose=False):
    """Compile every HIP source for gfx950 into libmil_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise MilLibraryError("building libmil_hip.so failed")
    return LIB_PATH


_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t
_SIGS = {
    "mil_abi_version": ([], _i),
    "mil_stream_copy": ([_vp, _vp, _sz, _vp], _i),
    "mil_split_probe": ([_vp, _vp, _vp, _i, _vp], _i),
    "mil_stem_s2d": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mil_packed_weight_elems": ([_c.POINTER(_sz), _i, _i, _i, _i], _i),
    "mil_pack_conv_weights": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "mil_conv_igemm": ([_vp] * 6 + [_i] * 12 + [_f, _i, _vp], _i),
    "mil_conv_wgrad_workspace": ([_c.POINTER(_sz)] + [_i] * 12, _i),
    "mil_conv_wgrad": ([_vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 13 + [_vp], _i),
    "mil_reduce_job_bytes": ([], _i),
    "mil_reduce_defer_begin": ([_vp, _i], _i),
    "mil_reduce_defer_end": ([_c.POINTER(_i)], _i),
    "mil_wgrad_reduce_all": ([_vp, _vp, _i, _vp], _i),
    "mil_conv_bwd_fused_workspace": ([_c.POINTER(_sz)] + [_i] * 8, _i),
    "mil_conv_bwd_fused": ([_vp] * 8 + [_sz] + [_i] * 9 + [_f, _i, _vp], _i),
    "mil_maxpool_fwd": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "mil_maxpool_bwd": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_resize_plan": ([_i, _i, _c.POINTER(_i)], _i),
    "mil_resize_coeffs": ([_i, _i, _vp, _vp], _i),
    "mil_tile_preprocess": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mil_tile_preprocess_s2d": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mil_stem_fwd_fused_xs": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_conv_block_fwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_conv_chain": ([_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_conv_pair": ([_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_conv_s2_entry": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_conv_wgrad_pair_workspace": ([_c.POINTER(_sz), _i, _i, _i, _i, _i, _i, _i, _i], _i),
    "mil_conv_wgrad_pair": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mil_conv_dgrad_s2": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_stem_fwd_fused": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_stem_bwd_fused_workspace": ([_c.POINTER(_sz), _i, _i, _i, _i], _i),
    "mil_stem_bwd_fused": ([_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _i, _i, _vp], _i),
    "mil_stem_bwd_fused_nchw_workspace": ([_c.POINTER(_sz), _i, _i, _i, _i], _i),
    "mil_stem_bwd_fused_nchw": ([_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _i, _i, _vp], _i),
    "mil_avgpool_fc_fwd": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mil_avgpool_fc_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mil_wide_packed_elems": ([_c.POINTER(_sz), _i, _i, _i, _i], _i),
    "mil_wide_pack_weights": ([_vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "mil_wide_conv": ([_vp] * 6 + [_i] * 12 + [_f, _i, _vp], _i),
    "mil_wide_wgrad_workspace": ([_c.POINTER(_sz)] + [_i] * 11, _i),
    "mil_wide_wgrad": ([_vp, _vp, _vp, _vp, _sz] + [_i] * 12 + [_vp], _i),
    "mil_gconv_supported": ([_i, _i, _i, _i], _i),
    "mil_gconv_packed_elems": ([_c.POINTER(_sz), _i, _i, _i, _i], _i),
    "mil_gconv_pack_weights": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mil_gconv": ([_vp] * 5 + [_i] * 12 + [_f, _vp], _i),
    "mil_fc_wgrad_workspace": ([_c.POINTER(_sz), _i, _i, _i], _i),
    "mil_fc_wgrad": ([_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp], _i),
    "mil_adam_step": ([_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _f, _vp], _i),
    "mil_pack_job_bytes": ([], _i),
    "mil_pack_job_fill": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i], _i),
    "mil_pack_all": ([_vp, _i, _vp], _i),
    "mil_head_workspace_floats": ([_c.POINTER(_sz), _i, _i], _i),
    "mil_head_grad_floats": ([], _i),
    "mil_head_rec_floats": ([], _i),
    "mil_head_fwd": ([_vp] * 13 + [_i, _i, _f, _f, _f, _f, _vp], _i),
    "mil_head_bwd": ([_vp] * 12 + [_i, _i, _f, _f, _vp], _i),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """The loaded library; raises MilLibraryError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MilLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C <package>/csrc`). This package has no CPU fallback.")
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise MilLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (args, ret) in _SIGS.items():
            fn = getattr(handle, name, None)
            if fn is None:
                raise MilLibraryError(f"{LIB_PATH} does not export {name}")
            fn.argtypes, fn.restype = args, ret
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise MilLibraryError(f"{what} failed: {_ERR.get(rc, rc)}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


class _MmaState(threading.local):
    """Per THREAD: autograd runs backward on its own thread, and an exact-fp32 model and a BF16X3 model may be driven from
    different threads of one process — a process-global code would let one read the other's (packed [hi|lo] filters
    contracted as plain fp32, or the reverse)."""
    code = MIL_DT_F32


_F32_MMA = _MmaState()


@contextlib.contextmanager
def f32_mma(code):
    """Inside the block (on this thread), the convolution entry points run fp32 tensors with `code` (MIL_DT_F32: exact-f32
    MFMA, MIL_DT_F32S: bf16x3 split products)."""
    prev = _F32_MMA.code
    _F32_MMA.code = code
    try:
        yield
    finally:
        _F32_MMA.code = prev


def storage_dtype(compute_dtype):
    """torch dtype of the activation tensors of a compute mode."""
    return torch.float32 if compute_dtype == BF16X3 else compute_dtype


def mma_code(compute_dtype):
    return MIL_DT_F32S if compute_dtype == BF16X3 else MIL_DT_F32


def dt_code(dtype, dense_grads=False, mma=False):
    """MIL_DT_* code of a compute dtype; dense_grads selects MIL_DT_BF16_DGRAD / MIL_DT_F32S_DGRAD (gradient tensors of the
    20-channel layer at 20 channels per pixel instead of 24; see include/mil_hip.h).  mma=True (the convolution entry points and the filter
    packing): fp32 tensors get the code selected by `f32_mma` (exact or split products)."""
    if dense_grads:
        if dtype == torch.bfloat16:
            return MIL_DT_BF16_DGRAD
        if dtype == torch.float32 and mma and _F32_MMA.code == MIL_DT_F32S:
            return MIL_DT_F32S_DGRAD
        raise ValueError("the dense gradient layout exists for bfloat16 and for the split-precision (bf16x3) mode only")
    if dtype == torch.float32:
        return _F32_MMA.code if mma else MIL_DT_F32
    if dtype == torch.bfloat16:
        return MIL_DT_BF16
    raise ValueError(f"compute dtype must be torch.float32 or torch.bfloat16, got {dtype}")
