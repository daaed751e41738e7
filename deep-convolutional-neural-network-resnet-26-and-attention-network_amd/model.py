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

from . import ops
from .encoder import BasicResBlock, ResNet
from .head import DROP_P, SMOOTHING, BagLayout, head_apply

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
    def __init__(self, n_classes, class_weights=None, *, compute_dtype=torch.bfloat16, device="cuda"):
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
                bags = list(torch.split(x_cat, list(cat_sizes), dim=0))
            else:
                bags = None
        tiles, sizes, keep = [], [], None
        if bags is None:
            if x_cat.dim() != 4 or x_cat.shape[1] != 3:
                raise ValueError(f"expected [N,3,H,W], got {tuple(x_cat.shape)}")
            tiles, sizes, bags = [x_cat.detach().to(dev, torch.float32)], list(cat_sizes), []
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
            tiles.append(x.to(dev, torch.float32))
            sizes.append(x.shape[0])
        layout = BagLayout.cached(sizes, dev)
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
        loss, l2, a1, wrois, bterm, kld, rec = head_apply(H, layout, y, keep, cw, self.head_weights())
        Hd = H.detach()
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
