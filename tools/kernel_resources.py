"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (VGPRs, spills, occupancy).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c <file.hip> -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> /tmp/x.rpass
    python tools/kernel_resources.py /tmp/x.rpass [substring ...]
"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
seen = set()
for blk in re.split(r"remark: Function Name: ", txt)[1:]:
    name = blk.split(" ", 1)[0]
    if name in seen:
        continue
    seen.add(name)
    dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    if want and not all(w in dem for w in want):
        continue
    def g(key):
        m = re.search(key + r": (\d+)", blk)
        return m.group(1) if m else "?"
    print(f"VGPR {g('  VGPRs'):>4} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} occ {g('Occupancy .waves/SIMD.')} SGPR {g('TotalSGPRs'):>3}  {dem[:120]}")
