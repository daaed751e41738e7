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

    def __call__(self, rois, params=None):
        """rois: uint8 [T,S,S,3] on the GPU (the cached `data_cache` array).  params: int32 [T,4] from `draw_params`
        (train chain) or None (validation chain).  Returns fp32 [T,3,R,R] in [-1,1]."""
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
        out = torch.empty((t, 3, self.resolution, self.resolution), dtype=torch.float32, device=rois.device)
        done = 0
        while done < t:                                      # grid.y limit: 65535 tiles per launch
            n = min(t - done, 65535)
            L.check(L.lib().mil_tile_preprocess(rois[done:].data_ptr(), None if params is None else params[done:].data_ptr(),
                                                self.bounds_host.ctypes.data, b.data_ptr(), k.data_ptr(), out[done:].data_ptr(),
                                                n, self.roi_size, self.pad, self.resolution, L.stream_ptr()), "mil_tile_preprocess")
            done += n
        return out
