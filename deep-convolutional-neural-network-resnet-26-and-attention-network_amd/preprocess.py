"""Device-side tile finalisation: the reference's `RoiBuilder.img_finalize` / `img_finalize_flat`
(RoiBuilder.py:193-210, used by `get_train_data` :222-245 and `get_validation_data` :247-268) on cached uint8 ROIs that
stay resident in HBM, producing the fp32 [T,3,R,R] tile stack `Attention.forward` takes — instead of a per-tile
torchvision/Pillow chain on the host followed by an fp32 upload (the `Tensor.cuda()` time that dominated the reference's
profile).  The arithmetic is Pillow's bilinear resampling, bit-exact (see csrc/preprocess.hip).
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


class S2dTiles:
    """A stack of tiles held as the bf16 space-to-depth tensor the stem kernels read, `xs [T, R/2, R/2, 16]` (channel =
    c*4 + dy*2 + dx of the 2x2 pixel block, 12 real): what `TilePreprocessor(..., out="s2d")` returns and what
    `Attention.forward` / `forward_bags` / `ResNet.forward` accept in place of the fp32 `[T,3,R,R]` stack (bf16 compute mode).
    `shape` is the shape of the fp32 stack it stands for; indexing with a tensor / slice selects tiles."""

    def __init__(self, xs):
        if xs.dim() != 4 or xs.shape[3] != 16 or xs.dtype != torch.bfloat16:
            raise ValueError(f"expected a bf16 [T,H/2,W/2,16] space-to-depth tensor, got {tuple(xs.shape)} {xs.dtype}")
        self.xs = xs

    @property
    def shape(self):
        t, h2, w2, _ = self.xs.shape
        return torch.Size((t, 3, 2 * h2, 2 * w2))

    @property
    def device(self):
        return self.xs.device

    def dim(self):
        return 4

    def detach(self):
        return S2dTiles(self.xs.detach())

    def __len__(self):
        return self.xs.shape[0]

    def __getitem__(self, idx):
        return S2dTiles(self.xs[idx])

    @staticmethod
    def cat(parts):
        return S2dTiles(torch.cat([p.xs for p in parts], dim=0))


class TilePreprocessor:
    """`update_resolution_and_buffer(resolution)` + the two transform chains for ROIs of `roi_size` pixels."""

    def __init__(self, roi_size, resolution, pad=100, device="cuda"):
        self.roi_size, self.resolution, self.pad = int(roi_size), int(resolution), int(pad)
        lib = L.lib()
        ks = ctypes.c_int(0)
        L.check(lib.mil_resize_plan(self.roi_size, self.resolution, ctypes.byref(ks)), "mil_resize_plan")
        self.ksize = ks.value
        self.bounds_host = np.zeros((self.resolution, 2), dtype=np.int32)
        kk = np.zeros((self.resolution, self.ksize), dtype=np.int32)
        L.check(lib.mil_resize_coeffs(self.roi_size, self.resolution, self.bounds_host.ctypes.data, kk.ctypes.data),
                "mil_resize_coeffs")
        self.kk_host = kk
        self.device = torch.device(device)
        self.bounds_dev = self.kk_dev = None

    def draw_params(self, n_tiles, generator=None):
        """Per tile (top, left, hflip, vflip) as the train chain draws them: RandomCrop offsets uniform in [0, 2*pad],
        each flip with probability 0.5 (RoiBuilder.py:197-201).  int32 [n_tiles, 4] on the host."""
        p = torch.empty((n_tiles, 4), dtype=torch.int32)
        p[:, 0] = torch.randint(0, 2 * self.pad + 1, (n_tiles,), generator=generator)
        p[:, 1] = torch.randint(0, 2 * self.pad + 1, (n_tiles,), generator=generator)
        p[:, 2] = (torch.rand(n_tiles, generator=generator) < 0.5).to(torch.int32)
        p[:, 3] = (torch.rand(n_tiles, generator=generator) < 0.5).to(torch.int32)
        return p

    def _tables(self, dev):
        if self.bounds_dev is None or self.bounds_dev.device != dev:
            self.bounds_dev = torch.from_numpy(self.bounds_host).to(dev)
            self.kk_dev = torch.from_numpy(self.kk_host).to(dev)
        return self.bounds_dev, self.kk_dev

    def __call__(self, rois, params=None, out="nchw"):
        """rois: uint8 [T,S,S,3] on the GPU (the cached `data_cache` array).  params: int32 [T,4] from `draw_params`
        (train chain) or None (validation chain).  Returns fp32 [T,3,R,R] in [-1,1] (out="nchw": the reference's tensor), or
        — out="s2d" — the same tiles as `S2dTiles` (bf16 space-to-depth, what the stem kernels read: the fp32 stack is never
        materialised)."""
        if out not in ("nchw", "s2d"):
            raise ValueError("out must be 'nchw' or 's2d'")
        if out == "s2d" and self.resolution % 2:
            raise ValueError("the space-to-depth output needs an even resolution")
        if rois.dtype != torch.uint8 or rois.dim() != 4 or rois.shape[3] != 3 or rois.shape[1] != rois.shape[2]:
            raise ValueError(f"expected uint8 [T,S,S,3] ROIs, got {tuple(rois.shape)} {rois.dtype}")
        if rois.shape[1] != self.roi_size:
            raise ValueError(f"ROI size {rois.shape[1]} != {self.roi_size} this preprocessor was planned for")
        if not rois.is_cuda:
            raise RuntimeError("tile pre-processing runs on an AMD GPU only (no CPU fallback)")
        rois = rois.contiguous()
        t = rois.shape[0]
        b, k = self._tables(rois.device)
        if params is not None:
            params = torch.as_tensor(params, dtype=torch.int32)
            if tuple(params.shape) != (t, 4):
                raise ValueError("params must be int32 [T,4]")
            if int(params[:, :2].min()) < 0 or int(params[:, :2].max()) > 2 * self.pad:
                raise ValueError("crop offsets must lie in [0, 2*pad]")
            params = params.to(rois.device).contiguous()
        r = self.resolution
        if out == "s2d":
            res = torch.empty((t, r // 2, r // 2, 16), dtype=torch.bfloat16, device=rois.device)
            fn, what = L.lib().mil_tile_preprocess_s2d, "mil_tile_preprocess_s2d"
        else:
            res = torch.empty((t, 3, r, r), dtype=torch.float32, device=rois.device)
            fn, what = L.lib().mil_tile_preprocess, "mil_tile_preprocess"
        done = 0
        while done < t:                                      # grid.y limit: 65535 tiles per launch
            n = min(t - done, 65535)
            L.check(fn(rois[done:].data_ptr(), None if params is None else params[done:].data_ptr(),
                       self.bounds_host.ctypes.data, b.data_ptr(), k.data_ptr(), res[done:].data_ptr(),
                       n, self.roi_size, self.pad, r, L.stream_ptr()), what)
            done += n
        return S2dTiles(res) if out == "s2d" else res
