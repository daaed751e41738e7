"""-m gpu: forward hooks on the children of the fused modules fire with the reference's tensors (SURVEY.md §8b:
`prime_activation_summary` attaches hooks, gbm/classify_combined.py:418).  Hooked values are checked against the
activations the golden generator captured with the very same hooks on the reference (tests/golden/make_golden.py
`stage_hooks`: stem LeakyReLU, max-pool, layer1..4) and against the oracle for the head's children."""
import os

import numpy as np
import pytest
import torch

from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(golden_dir, dtype):
    import mil_amd
    w = np.load(os.path.join(golden_dir, "weights.npz"))
    net = mil_amd.Attention(3, compute_dtype=dtype)
    net.load_state_dict({k: torch.tensor(w[k]) for k in w.keys()}, strict=True)
    return net.eval(), w


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", ["eval_n8_64", "eval_n5_50x70"])
def test_encoder_child_hooks_see_reference_activations(golden_dir, name, mode):
    import mil_amd
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    net, _w = _model(golden_dir, torch.float32 if mode == "fp32" else mil_amd.BF16X3)     # split precision: the same fp32 tensors
    cnn = net.cnn.module
    x, y = torch.tensor(g["x"]).cuda(), torch.tensor(g["y"]).cuda()
    plain = net(x, y)
    seen, calls, handles = {}, [], []

    def keep(key):
        def hook(mod, inp, out):
            assert isinstance(inp, tuple)
            seen[key] = (inp[0].detach().cpu().numpy(), out.detach().cpu().numpy())
            calls.append(key)
        return hook
    handles.append(cnn.relu.register_forward_hook(keep("stem")))
    handles.append(cnn.maxpool.register_forward_hook(keep("pool")))
    for i in (1, 2, 3, 4):
        handles.append(getattr(cnn, f"layer{i}").register_forward_hook(keep(f"layer{i}")))
    handles.append(cnn.layer2[0].register_forward_hook(keep("layer2.0")))
    handles.append(cnn.fc.register_forward_hook(keep("fc")))
    handles.append(cnn.avgpool.register_forward_hook(keep("avgpool")))
    handles.append(net.cnn.register_forward_hook(keep("cnn")))             # the wrapper itself: an ordinary module call
    relu_calls = []
    handles.append(cnn.layer1[1].relu.register_forward_hook(lambda m, i, o: relu_calls.append(tuple(o.shape))))
    out = net(x, y)
    for k in ("stem", "pool", "layer1", "layer2", "layer3", "layer4"):
        assert seen[k][1].shape == g["act." + k].shape and seen[k][1].dtype == np.float32, k
        assert _rel(seen[k][1], g["act." + k]) < (2e-5 if mode == "fp32" else 1e-4), k
    assert np.array_equal(seen["pool"][0], seen["stem"][1])                 # max-pool's input is the stem activation
    assert np.array_equal(seen["layer1"][0], seen["pool"][1])               # stage input = previous stage output
    assert np.array_equal(seen["layer2"][0], seen["layer1"][1]) and np.array_equal(seen["layer2.0"][0], seen["layer1"][1])
    assert seen["layer2.0"][1].shape == seen["layer2"][1].shape
    assert seen["avgpool"][1].shape == (x.shape[0], 80, 1, 1) and np.array_equal(seen["avgpool"][0], seen["layer4"][1])
    assert _rel(seen["fc"][1], g["out.Fterm"]) < (5e-5 if mode == "fp32" else 2e-4) and np.array_equal(seen["cnn"][1], seen["fc"][1])
    assert relu_calls == [seen["layer1"][1].shape] * 2                      # the block's LeakyReLU runs twice (nnBlocks.py:180,187)
    assert calls.index("stem") < calls.index("pool") < calls.index("layer1") < calls.index("layer2.0") < calls.index("layer2") \
        < calls.index("layer4") < calls.index("avgpool") < calls.index("fc") < calls.index("cnn")
    # hooks observe; they change nothing (same kernels for the blocks; the stem ran un-fused, fp32: identical arithmetic)
    # (split precision: the hooked run takes the un-fused stem — another product order, 16 significant bits per operand)
    rt, at = (1e-6, 1e-8) if mode == "fp32" else (1e-3, 1e-3)
    for k in ("Aterm", "Mterm", "loss", "Fterm"):
        assert torch.allclose(out[k], plain[k], rtol=rt, atol=at), k
    for h in handles:
        h.remove()
    calls.clear()
    net(x, y)
    assert calls == []                                                       # removed hooks no longer fire
    h = cnn.layer3[1].conv2.register_forward_hook(lambda m, i, o: None)      # never materialised: refuses loudly
    with pytest.raises(RuntimeError, match="never"):
        net(x, y)
    h.remove()


def test_head_child_hooks_and_bf16_fused_path(golden_dir):
    """Head children (context / attention / buffer) per bag against the oracle; and in bf16 the hooked encoder returns
    the same bits as the un-hooked one wherever the same kernels ran (block hooks do not change the kernel choice)."""
    net, w = _model(golden_dir, torch.float32)
    sd = orc.load_state(w)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(11, 3, 64, 64, generator=gen).clamp_(-1, 1)
    sizes, labels = [5, 6], torch.tensor([0, 2])
    got = {"context": [], "attention": [], "buffer": [], "bn": [], "lin2": []}
    hs = [net.context.register_forward_hook(lambda m, i, o: got["context"].append((i[0].cpu(), o[0].cpu(), o[1].cpu()))),
          net.context.bn.register_forward_hook(lambda m, i, o: got["bn"].append(o.cpu())),
          net.attention.register_forward_hook(lambda m, i, o: got["attention"].append((i[0].cpu(), o.cpu()))),
          net.attention.lin2.register_forward_hook(lambda m, i, o: got["lin2"].append((i[0].cpu(), o.cpu()))),
          net.buffer.register_forward_hook(lambda m, i, o: got["buffer"].append((i[0].cpu(), o.cpu())))]
    outs = net.forward_bags((x.cuda(), sizes), labels)
    assert all(len(v) == 2 for v in got.values())                            # one call per bag, as the reference's loop
    off = 0
    for b, n in enumerate(sizes):
        with torch.no_grad():
            feats = orc.backbone(sd, x[off:off + n])
            mean, var = feats.mean(0, keepdim=True), feats.var(0, unbiased=False, keepdim=True)
            hz = (feats - mean) / torch.sqrt(var + 1e-5) * sd["context.bn.weight"] + sd["context.bn.bias"]
            hm = torch.nn.functional.leaky_relu(feats, 0.1)
            t = torch.tanh(hz @ sd["attention.lin1.weight"].t() + sd["attention.lin1.bias"])
            araw = t @ sd["attention.lin2.weight"].t() + sd["attention.lin2.bias"]
            ref = orc.mil_head(sd, feats, labels[b:b + 1])
        h_in, hm_got, hz_got = got["context"][b]
        assert _rel(h_in.numpy(), feats.numpy()) < 5e-5 and _rel(hm_got.numpy(), hm.numpy()) < 5e-5
        assert _rel(hz_got.numpy(), hz.numpy()) < 2e-3 and torch.equal(got["bn"][b], hz_got)      # 5-6 instance batch-norm
        assert _rel(got["attention"][b][1].numpy(), araw.numpy()) < 2e-3 and torch.equal(got["attention"][b][0], hz_got)
        assert _rel(got["lin2"][b][0].numpy(), t.numpy()) < 2e-3 and torch.equal(got["lin2"][b][1], got["attention"][b][1])
        assert _rel(got["buffer"][b][1].numpy(), ref["Bterm"].numpy()) < 1e-4 and got["buffer"][b][1].shape == (n, 1)
        assert torch.equal(got["buffer"][b][1], outs[b]["Bterm"].cpu())
        off += n
    for h in hs:
        h.remove()
    net16, _ = _model(golden_dir, torch.bfloat16)
    xg = x.cuda()
    plain = net16.cnn(xg)
    rec = []
    h = net16.cnn.module.layer1[0].register_forward_hook(lambda m, i, o: rec.append((i[0].dtype, o.shape)))
    hooked = net16.cnn(xg)
    h.remove()
    assert rec == [(torch.float32, (11, 20, 16, 16))] and torch.equal(plain, hooked)
