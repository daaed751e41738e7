"""CPU gate on the BUILT gfx950 code objects: no shipped kernel may spill vector registers to scratch unless it is listed
here with the reason it is tolerated.  Reads the AMDGPU metadata notes of csrc/build/*.o through
tools/kernel_resources.py (llvm-objdump --offloading + llvm-readelf --notes): what is checked is what the library was
linked from.  SGPR spills are not gated: they go to VGPR lanes (v_writelane), not to memory."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name substring -> (max spilled VGPRs tolerated, reason).  Every entry is a kernel NO BASELINE configuration and not
# the 300x300 live-driver size launches on its hot path, or one whose spill was measured not to matter; the list only shrinks.
ALLOWED = {
    "pack_all_kernel": (0, "round-3 state: on the round-4 work list"),
    "pack_weights_kernel": (0, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 24, 2, 3, false, false, 8>": (18, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 24, 2, 3, false, true, 8>": (18, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 24, 2, 3, true, false, 8>": (26, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 24, 2, 3, true, true, 8>": (26, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 64, 4, 3, false, true, 8>": (2, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 64, 4, 3, true, false, 8>": (15, "round-3 state: on the round-4 work list"),
    "conv_bwd_fused_kernel<BF16, 64, 4, 3, true, true, 8>": (18, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 24, 2, 3, 4, 2, 4, 2>": (13, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 24, 2, 3, 4, 5, 4, 2>": (13, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 40, 3, 3, 2, -1, 8, 3>": (25, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 40, 3, 3, 2, 2, 8, 3>": (12, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 40, 3, 3, 2, 3, 8, 3>": (27, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 40, 3, 3, 2, 5, 8, 3>": (6, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<BF16, 40, 3, 3, 4, -1, 4, 3>": (5, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<F32S, 24, 2, 3, 4, -1, 4, 2>": (10, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<F32S, 24, 2, 3, 4, 3, 4, 2>": (9, "round-3 state: on the round-4 work list"),
    "conv_igemm_pf_kernel<F32S, 80, 4, 3, 1, -1, 8, 4>": (1, "round-3 state: on the round-4 work list"),
    "wgrad_kernel<BF16, 3, 64, 5, 2, true, 8, true>": (59, "round-3 state: on the round-4 work list"),
    "wgrad_kernel<F32S, 3, 24, 2, 1, true, 8, false>": (22, "round-3 state: on the round-4 work list"),
    "wgrad_kernel<F32S, 3, 40, 4, 1, true, 8, false>": (30, "round-3 state: on the round-4 work list"),
}


def _table():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.isdir(mod.BUILD) or not os.path.exists(os.path.join(mod.LLVM, "llvm-readelf")):
        pytest.skip("no build directory / LLVM tools here")
    return mod.kernel_table()


def test_no_unlisted_register_spills_in_shipped_kernels():
    rows = _table()
    assert len(rows) > 200                       # the gate must not pass vacuously
    offenders, stale = [], dict(ALLOWED)
    for r in rows:
        spill, scratch = r["vgpr_spill_count"], r["private_segment_fixed_size"]
        if not spill and not scratch:
            continue
        key = next((k for k in ALLOWED if k in r["name"]), None)
        if key is None:
            offenders.append((r["name"], spill, scratch))
            continue
        stale.pop(key, None)
        if spill > ALLOWED[key][0]:
            offenders.append((r["name"], spill, scratch, f"listed for <= {ALLOWED[key][0]}"))
    assert not offenders, "kernels spilling to scratch:\n" + "\n".join(map(str, offenders))
    assert not stale, f"whitelist entries that no longer spill (remove them): {sorted(stale)}"
