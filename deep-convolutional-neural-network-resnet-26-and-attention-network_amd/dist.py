"""Bag-parallel data parallelism: one process per GPU, bags sharded across ranks, ONE all-reduce of a
flat fp32 gradient bucket per optimizer step over RCCL/xGMI.

Replaces the reference's single-process `nn.DataParallel(ResNet, device_ids=[0,1,2,3])`
(gbm/model.py:132-135), which scatters the tiles of one bag, re-broadcasts all weights every forward
and gathers features to GPU 0.  Here weights stay resident on every rank, every per-bag reduction
(BatchNorm statistics, L1 normalisation, pooling) stays on the rank that owns the bag, and the only
exchange is the 640,967-float gradient sum (2.56 MB) — the reference sums gradients over bags before
an optimizer step (gbm/classify_combined.py:446-454), so the collective is a SUM, not a mean.
"""
import torch
import torch.distributed as dist


def shard_bags(n_bags, rank, world):
    """Indices of the bags rank `rank` owns: g, g+G, g+2G, ...."""
    return list(range(rank, n_bags, world))


class FlatParams:
    """Re-points every parameter (and its .grad) of `module` into one contiguous fp32 buffer each, so the
    gradient exchange is a single collective and an optimizer can sweep one array.

    Also the OPT-IN to in-place gradient accumulation (`direct_grad` on the encoder and on `Attention`): the hand-written
    backward passes then add their gradients straight into `flat_grad` and hand autograd None for those parameters.  That is
    valid for plain `loss.backward()` (what the reference's loop does, gbm/classify_combined.py:446-447) and nothing else:
    `torch.autograd.grad`, `backward(inputs=...)` and per-parameter gradient hooks need the gradients returned — use the module
    without a FlatParams (or set `direct_grad = False` on `net` and `net.cnn.module`) for those."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                off += n
        self.numel = total
        # gradients now have a fixed home: let the encoder's backward accumulate into it directly (saves one
        # torch add launch and one temporary per parameter per backward)
        for m in module.modules():
            if hasattr(m, "direct_grad"):
                m.direct_grad = True

    def zero_grad(self):
        self.flat_grad.zero_()
        off = 0
        for p in self.params:       # re-attach in case something replaced .grad (e.g. set_to_none)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[off:off + n].data_ptr():
                p.grad = self.flat_grad[off:off + n].view(p.shape)
            off += n

    def allreduce_grads(self, group=None, force=False):
        """Sum the flat gradient bucket over ranks (no-op for a single process; force=True issues the collective even in a
        one-rank group — the RCCL smoke test and the bench's collective timing use it)."""
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
        return self.flat_grad

    def broadcast_params(self, src=0, group=None, force=False):
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.broadcast(self.flat, src=src, group=group)


def gather_features(feats, group=None, force=False):
    """Tile-parallel inference of ONE large bag (BASELINE config 5): every rank encodes its slice of the
    tiles, then all ranks gather H [N/G,80] and run the (tiny) head redundantly."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return feats
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=feats.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([feats.shape[0]], dtype=torch.int64, device=feats.device), group=group)
    sizes = [int(s) for s in sizes]
    mx = max(sizes)
    padded = torch.zeros((mx, feats.shape[1]), dtype=feats.dtype, device=feats.device)
    padded[:feats.shape[0]] = feats
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


class FlatAdam:
    """torch.optim.Adam semantics (the reference's optimizer, gbm/classify_combined.py:519) as ONE HIP launch over
    the flat parameter / gradient buckets of a FlatParams (mil_adam_step in include/mil_hip.h)."""

    def __init__(self, flat, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.t = 0

    def step(self, grad_scale=1.0):
        from . import _lib as L
        from .encoder import WEIGHT_EPOCH
        if not self.flat.flat.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only")
        self.t += 1
        L.check(L.lib().mil_adam_step(self.flat.flat.data_ptr(), self.flat.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.flat.numel, self.lr, self.betas[0], self.betas[1],
                                      self.eps, self.weight_decay, self.t, grad_scale, L.stream_ptr()), "mil_adam_step")
        WEIGHT_EPOCH[0] += 1          # packed filter copies are stale now

    def state_dict(self):
        return {"t": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def torch_state_dict(self):
        """The same state in `torch.optim.Adam.state_dict()` form (what the reference stores under 'optimizer',
        gbm/classify_combined.py:468-474): per-parameter step / exp_avg / exp_avg_sq, one param group."""
        state, off = {}, 0
        for i, p in enumerate(self.flat.params):
            n = p.numel()
            if self.t > 0:
                state[i] = {"step": torch.tensor(float(self.t)),
                            "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(self.flat.params)))}
        return {"state": state, "param_groups": [group]}

    def load_torch_state_dict(self, sd):
        """Inverse of torch_state_dict(); accepts a checkpoint written by the reference's torch.optim.Adam over the same
        parameter list (parameters without state yet start from zero moments)."""
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.flat.params):
            raise ValueError(f"optimizer state has {len(group['params'])} parameters, the model has {len(self.flat.params)}")
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        self.weight_decay = float(group.get("weight_decay", 0.0))
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        steps, off = set(), 0
        for i, p in enumerate(self.flat.params):
            n = p.numel()
            st = sd["state"].get(group["params"][i])
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            off += n
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state the fused step can continue")
        self.t = steps.pop() if steps else 0

    def load_state_dict(self, sd):
        self.t = int(sd["t"]); self.lr = float(sd.get("lr", self.lr))
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
