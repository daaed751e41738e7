"""Registers, spills and scratch of every kernel in the BUILT code objects (what actually ships), from the AMDGPU metadata
notes of the gfx950 code object bundled in each `csrc/build/*.o`:

    python tools/kernel_resources.py                 # every kernel with spills or scratch
    python tools/kernel_resources.py --all [substr]  # every kernel (optionally: demangled name contains substr)

`kernel_table()` is what tests/test_cpu_kernel_resources.py gates on: a kernel that spills registers to scratch is waiting on
memory in its inner loop (`wgrad_kernel<BF16,3,64,5,2,true,8,true>` with 59 spilled VGPRs ran at 0.18 of the HBM
peak), so a spill in a shipped kernel must either be removed or be listed, with its measured reason, in the test's whitelist.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "deep-convolutional-neural-network-resnet-26-and-attention-network_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
_FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
           "group_segment_fixed_size", "max_flat_workgroup_size")


def _demangle(names):
    if not names:
        return {}
    res = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True)
    return dict(zip(names, res.stdout.splitlines()))


def kernel_table(build_dir=BUILD):
    """[{name (demangled), object, vgpr_count, vgpr_spill_count, private_segment_fixed_size, ...}] for every kernel of every
    object under build_dir.  Raises when an object carries no gfx950 code object (the gate must not pass vacuously)."""
    rows = []
    objs = sorted(glob.glob(os.path.join(build_dir, "*.o")))
    if not objs:
        raise RuntimeError(f"no objects under {build_dir}: build the library first")
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)                       # llvm-objdump extracts next to its input
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, check=True)
            cos = glob.glob(local + ".*gfx950")
            if not cos:
                if os.path.basename(obj) == "capi.o":     # host-only translation unit
                    continue
                raise RuntimeError(f"{obj}: no gfx950 code object found")
            for co in cos:
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE,
                                       text=True, check=True).stdout
                # kernels are the list items of `amdhsa.kernels:`; every item starts with "  - ." at list indentation
                body = notes.split("amdhsa.kernels:", 1)[1].split("amdhsa.target:", 1)[0] if "amdhsa.kernels:" in notes else ""
                for item in re.split(r"\n  - ", "\n" + body)[1:]:
                    m = re.search(r"\.name:\s+(\S+)", item)
                    if not m:
                        continue
                    row = {"mangled": m.group(1), "object": os.path.basename(obj)}
                    for f in _FIELDS:
                        mm = re.search(r"\." + f + r":\s+(\d+)", item)
                        row[f] = int(mm.group(1)) if mm else 0
                    rows.append(row)
    dem = _demangle(sorted({r["mangled"] for r in rows}))
    for r in rows:
        r["name"] = dem.get(r["mangled"], r["mangled"])
    return rows


def spilling(rows):
    return [r for r in rows if r["vgpr_spill_count"] or r["sgpr_spill_count"] or r["private_segment_fixed_size"]]


def main(argv):
    rows = kernel_table()
    show_all = "--all" in argv
    want = [a for a in argv if not a.startswith("--")]
    sel = rows if show_all else spilling(rows)
    sel = [r for r in sel if all(w in r["name"] for w in want)]
    for r in sorted(sel, key=lambda r: (-r["vgpr_spill_count"], r["name"])):
        print(f"VGPR {r['vgpr_count']:>4} AGPR {r['agpr_count']:>3} vspill {r['vgpr_spill_count']:>3} sspill {r['sgpr_spill_count']:>3} "
              f"scratch {r['private_segment_fixed_size']:>5} B  wg<= {r['max_flat_workgroup_size']:>4}  {r['object']:<18} {r['name'][:150]}")
    print(f"{len(rows)} kernels, {len(spilling(rows))} with spills or scratch")


if __name__ == "__main__":
    main(sys.argv[1:])
