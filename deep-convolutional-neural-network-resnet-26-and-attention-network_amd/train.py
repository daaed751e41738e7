"""Host-side pieces adjacent to the hot path: the reference's learning-rate / mode schedule,
gradient accumulation over bags with one optimizer step per group, the attention-map text export and the tensors
`visualize` derives from an output dict.

Reference: `SetStage` (gbm/classify_combined.py:110-138), the 5-bag accumulation loop (:446-454), `write_map`
(gbm/classify.py:207-225) and `visualize` (gbm/classify_combined.py:142-167).  Nothing here touches pixels: it drives `Attention`, `FlatParams` and `FlatAdam`.
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


def _minmax_f32(values):
    """`matplotlib.colors.Normalize()(x)` with autoscaling, restated.  matplotlib and numpy are third-party code
    outside the reference tree (matplotlib 3.10.8 / numpy 2.2 in this image; the reference pins no versions):
    the array keeps its own dtype (float32 for a float32 tensor); vmin / vmax are the minimum / maximum over ALL
    elements, held as Python floats (the `vmin` setter sanitises them) and re-wrapped as float64 scalars; the result
    is `x -= vmin; x /= (vmax - vmin)` IN PLACE, which under numpy-2 promotion evaluates each step in float64 and
    rounds it back into the float32 array; all zeros when vmin == vmax."""
    import numpy as np
    a = np.array(torch.as_tensor(values).detach().cpu().numpy(), dtype=np.float32, copy=True)
    if a.size == 0:
        return a
    vmin, vmax = np.float64(a.min()), np.float64(a.max())
    if vmin == vmax:
        a.fill(0)
    else:
        a[...] = (a.astype(np.float64) - vmin).astype(np.float32)
        a[...] = (a.astype(np.float64) / (vmax - vmin)).astype(np.float32)
    return a


def _dla_lines(raster, column):
    """`f'{coord[1]} {coord[0]} {value}\n'` (gbm/classify.py:212): the file lists the raster's second coordinate first;
    a float32 element formats as the shortest repr of its value widened to a Python float."""
    return "".join(f"{coord[1]} {coord[0]} {float(v)}\n" for coord, v in zip(raster, column))


def write_map(meta, epoch, raster, attn, activations, output_dir="."):
    """The reference's `.dla` overlay export, gbm/classify.py:207-225: FOUR text files per slide —
    `prediction-AGMIL-ATTN.<name>.dla` with column 0 of the min-max-normalised attention weights (`plt.Normalize()` over
    the whole `attn` array) and `prediction-AGMIL-ACTF{1,2,3}.<name>.dla` with columns 0..2 of `activations`, one
    `x y value` line per tile in raster order.  `attn` is [N, >=1], `activations` [N, >=3] (for this model:
    `out["Aterm"].t()` and `out["wROIs"].t()`); `meta["basename"]` names the slide; `epoch` is unused, as upstream.
    `output_dir` is a module global upstream.  Returns the four paths."""
    import os
    name = meta["basename"]
    attn = torch.as_tensor(attn).detach().float().cpu()
    activations = torch.as_tensor(activations).detach().float().cpu()
    if attn.dim() != 2 or activations.dim() != 2 or activations.shape[1] < 3:
        raise ValueError("attn must be [N,>=1] and activations [N,>=3]")
    if len(raster) > attn.shape[0] or len(raster) > activations.shape[0]:
        raise IndexError("more raster coordinates than tiles")        # upstream: index error while writing
    att = _minmax_f32(attn)
    paths = []
    for tag, col in (("ATTN", att[:, 0]), ("ACTF1", activations[:, 0].numpy()), ("ACTF2", activations[:, 1].numpy()),
                     ("ACTF3", activations[:, 2].numpy())):
        path = os.path.join(output_dir, f"prediction-AGMIL-{tag}.{name}.dla")
        with open(path, "w+") as f:
            f.write(_dla_lines(raster, col))
        paths.append(path)
    return paths


def write_attention_map(path, raster, weights, normalise=True):
    """ONE `.dla` file in the format of gbm/classify.py:209-213 (`x y weight` per tile; `raster[i]` is (row, col), the
    file lists col first) for a single weight vector — e.g. one of the three attention maps `out["Aterm"][k]`.  With
    `normalise` the weights are min-max scaled as `plt.Normalize()` does there (float32 arithmetic)."""
    w = torch.as_tensor(weights).detach().float().flatten().cpu()
    if len(raster) != w.numel():
        raise ValueError("one raster coordinate per weight expected")
    col = _minmax_f32(w) if normalise else w.numpy()
    with open(path, "w") as f:
        f.write(_dla_lines(raster, col))


def visualize_terms(output):
    """The tensors `visualize` (gbm/classify_combined.py:142-167) derives from one forward's output dict before it hands them
    to its plotting helper, restated on CPU tensors exactly as written there:
      angle  degrees(mean over the pairs i<j of arccos(M_i . M_j / (|M_i| |M_j| + 1e-5)))  (:156-160; M = Mterm [3,1])
      A1     (wROIs - min) / (max - min) over the WHOLE [3,N] array                        (:162)
      B1     Fterm viewed [N,8,10]                                                          (:163)
      M1     |Mterm| viewed [3,1,1] and permuted to [1,1,3] (H, W, channel)                 (:164)
    Returns {"angle": float, "A1", "B1", "M1"}.  A constant wROIs array gives NaNs in A1, as upstream (0/0)."""
    import numpy as np
    A = output["wROIs"].detach().float().cpu()
    M = output["Mterm"].detach().float().cpu()
    F = output["Fterm"].detach().float().cpu()
    angles = []
    for m_i, v1 in enumerate(M):
        for m_j, v2 in enumerate(M):
            if m_j > m_i:
                angles.append(np.arccos(v1.dot(v2) / (v1.norm() * v2.norm() + 1e-5)).item())
    angle = float(np.degrees(np.mean(angles)))
    A1 = (A - A.min()) / (A.max() - A.min())
    B1 = F.view(F.shape[0], 8, 10)
    M1 = M.view(3, 1, 1).permute(1, 2, 0).abs()
    return {"angle": angle, "A1": A1, "B1": B1, "M1": M1}


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
