"""Drop-in `Attention` (and friends) with the reference's nn.Module surface on the MI355X kernels.

Boundary: `Attention(n_classes, class_weights=None)`, `forward(full_input[N,3,H,W],
Y[1]) -> dict` with the 13 keys of gbm/model.py:249-264, `.train()/.eval()`, `state_dict()` with the
65 keys of the reference (`cnn.module.*` prefix included), `.weight_mask`, `.cnn.module`, `.context`,
`.attention`, `.buffer`.  Arithmetic runs in libmil_hip.so; there is no torch/CPU fallback.
"""
from collections import OrderedDict

import torch
from torch import nn
from torch.nn import init

from . import hooks
from . import ops
from ._lib import BF16X3
from .encoder import BasicResBlock, ResNet
from .head import DROP_P, SMOOTHING, BagLayout, head_apply
from .preprocess import S2dTiles

SUBSAMPLE = 0.2     # gbm/model.py:193


class CrossEntropyWithProbs(nn.Module):
    """Holder of the label-smoothing / class-weight configuration (nnBlocks.py:47-69).  The loss itself
    is evaluated inside the fused head kernel; calling this module directly is not supported."""

    def __init__(self, classes, smoothing=0.0, weight=None, reduction="mean"):
        super().__init__()
        self.smoothing, self.num_classes, self.weight, self.reduction = smoothing, classes, weight, reduction

    def forward(self, input, target):
        raise RuntimeError("the soft-target cross-entropy is fused into the MIL head kernel")


class ContextLayer(nn.Module):
    """Parameters of gbm/model.py:89-111: BatchNorm1d over the instances of a bag (batch statistics
    always), LeakyReLU(0.1), Dropout(0.25).  Executed inside the fused head kernel."""

    def __init__(self, features):
        super().__init__()
        self.L = features
        self.bn = nn.BatchNorm1d(features, track_running_stats=False)
        self.relu = nn.LeakyReLU(ops.LEAK)
        self.do = nn.Dropout(DROP_P)

    def forward(self, x):
        raise RuntimeError("ContextLayer is fused into the MIL head kernel")


class TileParallel(nn.Module):
    """Stands where the reference puts `nn.DataParallel` (gbm/model.py:132-135): keeps the `module.`
    key prefix and simply runs the wrapped encoder on this process's GPU.  Multi-GPU is one process
    per GPU with bags sharded across ranks (see dist.py), not tile scatter/gather from one process."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return self.module(x)


class BagOutputs(list):
    """The per-bag output dicts of `forward_bags`, plus the un-split loss vector: summing `o["loss"]` over the dicts
    back-propagates through one select per bag, `outs.loss.sum()` through none."""
    loss = None
    l2 = None


class Attention(nn.Module):
    def __init__(self, n_classes, class_weights=None, *, compute_dtype=BF16X3, device="cuda"):
        # The default mode is the one inside the reference tolerance (logits / attention weights within 1e-3 of the fp32 CPU
        # path, gbm/model.py:227-235): fp32 tensors with bf16x3 split products.  `compute_dtype=torch.bfloat16` is the opt-in
        # fast mode (2.2x the throughput; logits 9e-2 off: bf16 storage through 26 layers), `torch.float32` the exact one.
        super().__init__()
        if n_classes != 3:
            # the reference hard-codes three attention maps / classes (gbm/model.py:123,126-130)
            raise ValueError("the reference head is built for 3 classes")
        self.L, self.D, self.O, self.K, self.C = 80, 40, 1, 3, n_classes
        self.loss = CrossEntropyWithProbs(classes=3, weight=class_weights, smoothing=SMOOTHING)
        self.cnn = TileParallel(ResNet(BasicResBlock, (3, 3, 3, 3), num_classes=self.L, compute_dtype=compute_dtype))
        self.context = ContextLayer(self.L)
        self.attention = nn.Sequential(OrderedDict([
            ("lin1", nn.Linear(self.L, self.D)), ("tanh", nn.Tanh()), ("lin2", nn.Linear(self.D, self.K))]))
        self.buffer = nn.Sequential(OrderedDict([
            ("lin1", nn.Linear(self.L, self.D)), ("relu", nn.LeakyReLU(ops.LEAK)),
            ("classifier", nn.Linear(self.D, self.O))]))
        self.weight_mask = nn.Parameter(torch.tensor([0.25, 0.25, 0.25]))
        self.off_diag = 1 - torch.eye(3)
        self.reset_params()
        self.rng_override = None     # tests inject {"indices": LongTensor, "keep_mask": uint8 [n,80]}
        # set by dist.FlatParams: the head backward may add its gradients in place into the flat gradient bucket (plain
        # `loss.backward()` accumulation); False: every gradient is returned to autograd (autograd.grad, hooks)
        self.direct_grad = False
        self.to(device)

    # ---- initialisation (gbm/model.py:161-187) -------------------------------------------------
    @staticmethod
    def weight_init(m, name=""):
        if isinstance(m, nn.Linear):
            if "attention" in name:
                init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="tanh")
            elif "classifier" in name:
                init.xavier_normal_(m.weight)
            else:
                init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="leaky_relu", a=ops.LEAK)
            if m.bias is not None:
                init.zeros_(m.bias)
        if isinstance(m, nn.Conv2d):
            init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu", a=ops.LEAK)
            if m.bias is not None:
                init.zeros_(m.bias)

    def reset_params(self):
        for name, m in self.named_modules():
            self.weight_init(m, name)

    def reset_linear(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="tanh")
                if m.bias is not None:
                    init.zeros_(m.bias)

    @property
    def compute_dtype(self):
        return self.cnn.module.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt):
        self.cnn.module.compute_dtype = dt

    def head_weights(self):
        return [self.context.bn.weight, self.context.bn.bias, self.attention.lin1.weight, self.attention.lin1.bias,
                self.attention.lin2.weight, self.attention.lin2.bias, self.buffer.lin1.weight, self.buffer.lin1.bias,
                self.buffer.classifier.weight, self.buffer.classifier.bias, self.weight_mask]

    # ---- forward hooks on the head's children (see hooks.py) ---------------------------------
    _HEAD_UNMATERIALISED = ("attention.lin1", "attention.tanh", "loss")

    def _hooked_head_modules(self):
        mods = [m for top in (self.context, self.attention, self.buffer, self.loss) for m in top.modules() if hooks.hooked(m)]
        if mods:
            hooks.refuse(self, self._HEAD_UNMATERIALISED)
        return mods

    def _fire_head_hooks(self, H, layout, keep, bterm, internals):
        """One call per bag, as the reference makes one forward per bag.  Every tensor handed to a hook is either what
        the head kernels stored (batch statistics, tanh activations, buffer pre-activations, A_raw, B) or an elementwise
        re-expression of it (the normalised / masked features, which the kernel keeps in registers only)."""
        with torch.no_grad():
            for b in range(layout.nbags):
                n0, n1 = layout.offsets_host[b], layout.offsets_host[b + 1]
                Hb = H[n0:n1]
                mu, rstd = internals["stats"][b, 0], internals["stats"][b, 1]
                hz = (Hb - mu) * rstd * self.context.bn.weight + self.context.bn.bias            # gbm/model.py:109
                act = torch.where(Hb >= 0, Hb, Hb * ops.LEAK)
                hm = act if keep is None else act * keep[n0:n1].to(act.dtype) / (1.0 - DROP_P)   # gbm/model.py:110
                v = internals["v"][n0:n1]
                vr = torch.where(v >= 0, v, v * ops.LEAK)
                araw, bt = internals["araw"][n0:n1], bterm[n0:n1].view(-1, 1)
                for mod, args, out in ((self.context.bn, Hb, hz), (self.context.relu, Hb, act), (self.context.do, act, hm),
                                       (self.context, Hb, (hm, hz)),
                                       (self.attention.lin2, internals["t"][n0:n1], araw), (self.attention, hz, araw),
                                       (self.buffer.lin1, hm, v), (self.buffer.relu, v, vr), (self.buffer.classifier, vr, bt),
                                       (self.buffer, hm, bt)):
                    if hooks.hooked(mod):
                        hooks.fire(mod, args, out)

    # ---- one bag (the reference call, gbm/model.py:189) ------------------------------------------
    def forward(self, full_input, Y=None):
        if Y is None:
            Y = torch.tensor([1])
        outs = self.forward_bags([full_input], Y.reshape(-1)[:1])
        return outs[0]

    # ---- a batch of bags: one encoder pass over all tiles, segmented head --------------------------
    def forward_bags(self, bags, labels):
        """bags: list of [N_b,3,H,W] fp32 tensors (same H,W), or (x_all [sum N_b,3,H,W], [N_b...]) when the
        tiles already sit back to back in one tensor; labels: [n_bags].  Returns one output dict per bag
        (keys/shapes/grad flags of gbm/model.py:249-264); each `loss` back-propagates."""
        dev = self.weight_mask.device
        if dev.type != "cuda":
            raise RuntimeError("Attention runs on an AMD GPU only (module parameters are not on a CUDA/HIP device)")
        if isinstance(bags, tuple):
            x_cat, cat_sizes = bags
            if x_cat.shape[0] != sum(cat_sizes):
                raise ValueError("bag sizes do not add up to the number of tiles")
            if self.training:      # per-bag subsampling needs the bags separately
                if isinstance(x_cat, S2dTiles):
                    bags = [S2dTiles(t) for t in torch.split(x_cat.xs, list(cat_sizes), dim=0)]
                else:
                    bags = list(torch.split(x_cat, list(cat_sizes), dim=0))
            else:
                bags = None
        tiles, sizes, keep = [], [], None
        if bags is None:
            if x_cat.dim() != 4 or x_cat.shape[1] != 3:
                raise ValueError(f"expected [N,3,H,W], got {tuple(x_cat.shape)}")
            tiles, sizes, bags = [x_cat.detach() if isinstance(x_cat, S2dTiles) else x_cat.detach().to(dev, torch.float32)], list(cat_sizes), []
        for b, x in enumerate(bags):
            x = x.detach()
            if x.dim() != 4 or x.shape[1] != 3:
                raise ValueError(f"bag {b}: expected [N,3,H,W], got {tuple(x.shape)}")
            if self.training:
                if self.rng_override is not None and "indices" in self.rng_override:
                    idx = self.rng_override["indices"]
                else:
                    idx = torch.randperm(x.shape[0])[: int(x.shape[0] * SUBSAMPLE)]
                x = x[idx.to(x.device)]
            tiles.append(x if isinstance(x, S2dTiles) else x.to(dev, torch.float32))
            sizes.append(x.shape[0])
        layout = BagLayout.cached(sizes, dev)
        if any(isinstance(t, S2dTiles) for t in tiles):
            if not all(isinstance(t, S2dTiles) for t in tiles):
                raise ValueError("bags of one call must all be fp32 tile stacks or all S2dTiles")
            if any(t.device != dev for t in tiles):
                raise ValueError("S2dTiles must live on the module's device")
            x_all = tiles[0] if len(tiles) == 1 else S2dTiles.cat(tiles)
        else:
            x_all = tiles[0] if len(tiles) == 1 else torch.cat(tiles, dim=0)
        if self.training:
            if self.rng_override is not None and "keep_mask" in self.rng_override:
                keep = self.rng_override["keep_mask"].to(dev, torch.uint8).contiguous()
            else:
                keep = (torch.rand(layout.ntot, self.L, device=dev) >= DROP_P).to(torch.uint8)
        y = labels.to(dev).long().reshape(-1).contiguous()
        if y.numel() != layout.nbags:
            raise ValueError("one label per bag expected")
        cw = self.loss.weight
        if cw is not None:
            cw = torch.as_tensor(cw, dtype=torch.float32, device=dev).contiguous()
        H = self.cnn(x_all)
        return self._finish(H, layout, y, keep, cw)

    def _finish(self, H, layout, y, keep, cw):
        """Segmented head over the features of all bags + the per-bag output dicts (gbm/model.py:198-264)."""
        head_mods = self._hooked_head_modules()
        internals = {} if head_mods else None
        loss, l2, a1, wrois, bterm, kld, rec = head_apply(H, layout, y, keep, cw, self.head_weights(), internals, direct=self.direct_grad)
        Hd = H.detach()
        if head_mods:
            self._fire_head_hooks(Hd, layout, keep, bterm, internals)
        outs = BagOutputs()
        outs.loss, outs.l2 = loss, l2          # [n_bags] / [] with grad: `outs.loss.sum().backward()` is one backward for all bags
        y_hat = rec[:, 16].long()              # one conversion for all bags
        for b in range(layout.nbags):
            n0, n1 = layout.offsets_host[b], layout.offsets_host[b + 1]
            r = rec[b]
            outs.append({
                "Aterm": a1[n0:n1].t(),
                "wROIs": wrois[3 * n0:3 * n1].view(3, n1 - n0),
                "Bterm": bterm[n0:n1].view(-1, 1),
                "Mterm": r[0:3].view(3, 1),
                "Fterm": Hd[n0:n1],
                "Aterm_mu": r[8],
                "Aterm_var": r[9],
                "loss": loss[b],
                "l2": l2,
                "KLD": kld[b],
                "y_pred": r[3:6].view(1, 3),
                "y_pred_hat": y_hat[b],
                "error": r[7].view(1),
            })
        return outs


    # ---- one LARGE bag sharded over ranks (BASELINE config 5; the reference's own multi-GPU mode) ----------------
    def forward_tile_parallel(self, x_slice, Y=None, group=None):
        """Inference of ONE bag whose tiles are split across the ranks of `group`: this rank encodes its slice
        `x_slice [N_r,3,H,W]`, the features are all-gathered in rank order (ragged slices allowed) and every rank runs
        the head on the whole bag, so each rank returns the same output dict as `forward(torch.cat(slices), Y)`.
        This is what the reference's `nn.DataParallel(ResNet, device_ids=[0,1,2,3])` does inside one process
        (scatter tiles -> replicas -> gather features to GPU 0, gbm/model.py:132-135), as one process per GPU over RCCL.
        Forward only: a training step inside one sharded bag would also need the reduce-scatter of dH, which no
        configuration of the reference asks for."""
        from .dist import gather_features
        if self.training:
            raise RuntimeError("forward_tile_parallel is the inference path (attention-map extraction); call .eval() first")
        dev = self.weight_mask.device
        if dev.type != "cuda":
            raise RuntimeError("Attention runs on an AMD GPU only (module parameters are not on a CUDA/HIP device)")
        if x_slice.dim() != 4 or x_slice.shape[1] != 3:
            raise ValueError(f"expected [N,3,H,W], got {tuple(x_slice.shape)}")
        if Y is None:
            Y = torch.tensor([1])
        with torch.no_grad():
            h_local = self.cnn(x_slice.detach().to(dev, torch.float32))
            H = gather_features(h_local, group)
            layout = BagLayout.cached([H.shape[0]], dev)
            y = Y.to(dev).long().reshape(-1)[:1].contiguous()
            cw = self.loss.weight
            if cw is not None:
                cw = torch.as_tensor(cw, dtype=torch.float32, device=dev).contiguous()
            return self._finish(H, layout, y, None, cw)[0]
