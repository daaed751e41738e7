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
    # measured round 4 (tools/ab_libs.sh, MI355X): compiled for 3 waves per SIMD (168 VGPRs, no spill) this forward conv runs
    # 139 us per launch against 113 us with the 6 spills at 4 waves per SIMD (256x256 tiles; 179 vs 136 us at 300x300): the
    # second resident workgroup is worth more than the spill costs
    "conv_igemm_pf_kernel<BF16, 40, 3, 3, 2, 5, 8, 3>": (6, "faster than the spill-free 3-waves-per-SIMD build: 113 vs 139 us"),
    # 2 VGPRs = the lanes that hold its 28-32 spilled SGPRs.  Round 4 moved the addend fetch behind the data-gradient loop:
    # 18 -> 2 spilled VGPRs, 118.6 -> 104.2 us per launch (addend variants)
    "conv_bwd_fused_kernel<BF16, 64, 4, 3, false, true, 8>": (2, "SGPR-spill lanes; 18 -> 2 after the late addend fetch"),
    "conv_bwd_fused_kernel<BF16, 64, 4, 3, true, false, 8>": (2, "SGPR-spill lanes; 18 -> 2 after the late addend fetch"),
    "conv_bwd_fused_kernel<BF16, 64, 4, 3, true, true, 8>": (2, "SGPR-spill lanes; 18 -> 2 after the late addend fetch"),
    # split-precision generic forms that no BASELINE configuration and not the 300x300 driver size launches any more: round 4's
    # conv_block_fwd_x3_kernel / conv_bwd_fused16x3_kernel take every 20-channel launch on maps of 16 pixels and more, and the
    # 80 -> 64 channel zero-insert form runs on the generic kernel (conv_igemm.hip: launch_conv_pf, SPLIT cases)
    "conv_igemm_pf_kernel<F32S, 24, 2, 3, 4, -1, 4, 2>": (10, "cold since round 4 (maps below 16 pixels only)"),
    "conv_igemm_pf_kernel<F32S, 24, 2, 3, 4, 3, 4, 2>": (9, "cold since round 4 (maps below 16 pixels only)"),
    "conv_igemm_pf_kernel<F32S, 80, 4, 3, 1, -1, 8, 4>": (1, "one register at the 256-VGPR cap; 128-pixel-tile zero-insert form"),
    # 16 bytes of private segment for the by-value PackJob argument's indexed fields, no register spill; one ~9 us launch per step
    "pack_all_kernel": (0, "argument copy, no spill"),
    "pack_weights_kernel": (0, "argument copy, no spill"),
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


# ---- occupancy gate ------------------------------------------------------------------------------------
# Every kernel that a BASELINE configuration launches (the names in the committed rocprofv3 kernel-stats summaries of the
# benchmark command, bf16 and split precision) and that needs more than 256 VGPRs + AGPRs (ONE wave per SIMD) or spills more
# than 32 SGPRs must be listed here with its share of the step and the reason it is left that way.  Shares are of the
# config-2 step in profiles/r04_* / r05_* (bf16 8.5 ms, split precision 18.9 ms).
LARGE_ALLOWED = {
    "wgrad_kernel<BF16, 3, 24, 3, 1, true, 4, true>": "2.1 % of the bf16 step (layer-2 entry: 3x3/s2 + 1x1/s2 weight gradients from one pass over x: two accumulator sets, 388 registers); the 128-pixel-tile two-workgroup form measured 219 -> 187 us in round 2 and is what ships",
    "wgrad_kernel<BF16, 3, 40, 4, 1, true, 4, true>": "1.2 % of the bf16 step (layer-3 entry pair: 23 row tiles x 4 column tiles x two filters = 646 registers, 52 SGPRs spilled to lanes); one 106 us launch per step",
    "conv_dgrad_s2_kernel<BF16, 64, 3, false, 4>": "1.2 % of the bf16 step (60 -> 40 parity-class entry gradient, 272 registers: four class accumulator sets)",
    "conv_dgrad_s2_kernel<BF16, 80, 4, false, 4>": "0.5 % of the bf16 step (80 -> 60 entry gradient, one 45 us launch)",
    "conv_s2_entry_kernel<40, 4, 2>": "0.8 % of the bf16 step (40 -> 60 entry forward pair: conv + projection accumulators)",
    "conv_s2_entry_kernel<64, 5, 1>": "0.4 % of the bf16 step (60 -> 80 entry forward pair on 64-pixel tiles: 100 KB of filters leave one workgroup per CU anyway)",
    "head_inst_fwd_kernel": "0.3 % of the step: one hidden unit per lane with its two weight rows (160 values) in registers, by design (DESIGN.md section 3.18)",
    "head_inst_bwd_kernel": "0.35 % of the step: one feature column per lane with both matrix columns in registers, by design",
    "conv_dgrad_s2_kernel<F32S, 80, 4, true, 4>": "0.7 % of the split-precision step: the streamed 80 -> 60 entry gradient runs ONE wave per SIMD by design (round-4: 58-211 spilled VGPRs at two; 0.34 + 0.09 -> 0.13 ms)",
    "conv_s2_entry_x3_kernel<64, 5, 64>": "0.5 % of the split-precision step (60 -> 80 entry pair: 83 KB of halo planes leave one workgroup per CU anyway)",
    "stem_fwd_fused_kernel<4, 4, false, false>": "alt_resnet's 64-channel stem only (4 % of ITS step): 36 row tiles x 4 column tiles of accumulators",
    "wgrad_kernel<BF16, 4, 16, 4, 1, true, 4, false>": "alt_resnet's stem weight gradient only (2.7 % of its step)",
}


def _hot_kernel_names():
    """Kernel names of the committed kernel-stats summaries (latest round present) of the benchmark command."""
    import csv
    import glob
    names = set()
    prof = os.path.join(ROOT, "profiles")
    rounds = sorted({os.path.basename(p)[:3] for p in glob.glob(os.path.join(prof, "r??_*kernel_stats.csv"))})
    assert rounds, "no committed kernel-stats summaries"
    for p in glob.glob(os.path.join(prof, rounds[-1] + "_*kernel_stats.csv")):
        with open(p) as f:
            for row in csv.DictReader(f):
                names.add(row["Name"].split("(")[0].replace("void ", "").strip())
    return names


def test_hot_path_kernels_above_one_wave_per_simd_are_listed():
    rows = _table()
    hot = _hot_kernel_names()
    assert len(hot) > 30
    offenders, seen = [], set()
    for r in rows:
        short = r["name"].split("(")[0].replace("void ", "").strip()
        if short not in hot:
            continue
        regs = r["vgpr_count"] + r["agpr_count"]
        if regs <= 256 and r["sgpr_spill_count"] <= 32:
            continue
        key = next((k for k in LARGE_ALLOWED if k in r["name"]), None)
        if key is None:
            offenders.append((short, regs, r["sgpr_spill_count"]))
        else:
            seen.add(key)
    assert not offenders, "hot-path kernels at one wave per SIMD / > 32 spilled SGPRs without a written reason:\n" + "\n".join(map(str, offenders))
