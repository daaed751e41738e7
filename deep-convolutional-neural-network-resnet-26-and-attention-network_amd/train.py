"""Host-side pieces adjacent to the hot path: the reference's learning-rate / mode schedule,
gradient accumulation over bags with one optimizer step per group, and the attention-map text export.

Reference: `SetStage` (gbm/classify_combined.py:110-138), the 5-bag accumulation loop (:446-454) and `write_map`
(gbm/classify.py:207-225).  Nothing here touches pixels: it drives `Attention`, `FlatParams` and `FlatAdam`.
"""
import torch

BASE_LR = 2e-4                        # gbm/classify_combined.py:111
SCHEDULE = (0, 10, 150, 250, 340)     # gbm/classify_combined.py:112
ACCUM_BAGS = 5                        # gbm/classify_combined.py:450


def stage_for_epoch(epoch, test=False):
    """(stage name, learning rate or None = unchanged, train_mode or None = unchanged, stop) — the arithmetic of
    SetStage without its side effects.  Warm-up divides the base rate by the epochs left until epoch 10."""
    s = SCHEDULE
    if s[0] <= epoch < s[1]:
        return "Warmup", BASE_LR / (s[1] - epoch), True, False
    if s[1] <= epoch < s[2]:
        return "Main", BASE_LR, True, False
    if s[2] <= epoch < s[3]:
        return "Check", BASE_LR / 2.0, not test, False
    if s[3] <= epoch < s[4]:
        return "Freeze", BASE_LR / 10.0, not test, False
    if epoch > s[4]:
        return "Stop", None, None, True
    return "Hold", None, None, False      # epoch == 340: the reference matches no branch and changes nothing


def set_stage(optimizer, model, epoch, test=False):
    """Apply stage_for_epoch to an optimizer (`.lr` attribute as on FlatAdam, or torch param_groups) and a model."""
    name, lr, train_mode, stop = stage_for_epoch(epoch, test)
    if lr is not None:
        if hasattr(optimizer, "param_groups"):
            for group in optimizer.param_groups:
                group["lr"] = lr
        else:
            optimizer.lr = lr
    if train_mode is not None:
        model.train(train_mode)
    return name, stop


class BagTrainer:
    """One call per bag, as the reference loop does; gradients accumulate (un-normalised sum, as in the reference)
    and every `accum_bags` bags the flat gradient bucket is all-reduced over ranks and one Adam step is taken."""

    def __init__(self, model, flat, optimizer, accum_bags=ACCUM_BAGS):
        self.model, self.flat, self.opt, self.accum = model, flat, optimizer, accum_bags
        self.pending = 0
        self.flat.zero_grad()

    def step_bag(self, tiles, label):
        out = self.model(tiles, label)
        out["loss"].backward()
        self.pending += 1
        if self.pending == self.accum:
            self.flush()
        return out

    def flush(self):
        if self.pending:
            self.flat.allreduce_grads()
            self.opt.step()
            self.flat.zero_grad()
            self.pending = 0


def write_attention_map(path, raster, weights, normalise=True):
    """`x y weight` per tile, one line each — the `.dla` overlay format of gbm/classify.py:207-225 (`raster[i]` is
    (row, col); the file lists col first).  With `normalise` the weights are min-max scaled to [0,1] as
    matplotlib's Normalize() does there."""
    w = torch.as_tensor(weights, dtype=torch.float64).flatten().cpu()
    if len(raster) != w.numel():
        raise ValueError("one raster coordinate per weight expected")
    if normalise and w.numel():
        lo, hi = float(w.min()), float(w.max())
        w = (w - lo) / (hi - lo) if hi > lo else torch.zeros_like(w)
    with open(path, "w") as f:
        for (row, col), v in zip(raster, w.tolist()):
            f.write(f"{col} {row} {v}\n")


def save_checkpoint(path, model, optimizer):
    """`{'classifier': state_dict, 'optimizer': Adam state}` exactly as the reference writes it per epoch
    (gbm/classify_combined.py:468-474): its own `torch.load(...)['classifier']` / `optimizer.load_state_dict` read it."""
    opt = optimizer.torch_state_dict() if hasattr(optimizer, "torch_state_dict") else optimizer.state_dict()
    torch.save({"classifier": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "optimizer": opt}, path)


def load_checkpoint(path, model, optimizer=None, transfer=False):
    """gbm/classify_combined.py:521-535: full load (`strict=False`), or with `transfer` only the encoder's conv
    tensors (keys containing both 'cnn' and 'conv').  Returns the (missing, unexpected) key lists."""
    from .encoder import WEIGHT_EPOCH
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt["classifier"]
    if transfer:
        sd = {k: v for k, v in sd.items() if "cnn" in k and "conv" in k}
    with torch.no_grad():
        result = model.load_state_dict(sd, strict=False)
    WEIGHT_EPOCH[0] += 1                     # packed filter copies are stale
    if optimizer is not None and not transfer and "optimizer" in ckpt:
        if hasattr(optimizer, "load_torch_state_dict"):
            optimizer.load_torch_state_dict(ckpt["optimizer"])
        else:
            optimizer.load_state_dict(ckpt["optimizer"])
    return list(result.missing_keys), list(result.unexpected_keys)
