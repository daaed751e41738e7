#!/usr/bin/env python3
"""Per-kernel SQ counter ratios from two rocprofv3 --pmc passes of the same bench.py command (tools/profile_round.sh):

    pass 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
    pass 2: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM

    python profiles/summarise_sq_counters.py <dir pass 1> <dir pass 2> profiles/sq_counters.json "<label>"  > profiles/rNN_sq_counters.txt

wait_any / wait_inst / active are fractions of SQ_WAVE_CYCLES (disjoint, sum ~1: MI355X_MICROARCH.md, PMC slots).
mfma_busy_frac_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): the MFMA-busy counter sums the four SIMDs
of a CU, SQ_BUSY_CU_CYCLES counts cycles a CU has work — the matrix pipe's utilisation while the kernel runs.
bench.py copies these fractions into `roofline.sq_counters` for the dominant kernel."""
import collections
import csv
import glob
import json
import sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
    return acc, cnt


def library_sha16():
    """sha256 (first 16 hex digits) of the libmil_hip.so these counters were collected on: bench.py reports whether the
    summary it quotes belongs to the library it is running."""
    import glob as _g
    import hashlib
    import os
    hits = _g.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "*_amd", "libmil_hip.so"))
    return hashlib.sha256(open(hits[0], "rb").read()).hexdigest()[:16] if hits else None


def main():
    a, c = load(sys.argv[1])
    b, _ = load(sys.argv[2])
    out = {"_how": __doc__.strip().replace("\n", " "), "_source": sys.argv[4] if len(sys.argv) > 4 else "", "kernels": {}}
    out["_library_sha16"] = library_sha16()
    print("# " + out["_source"])
    print("# fractions of SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); insts per launch")
    for k in sorted(a, key=lambda k: -a[k]["SQ_WAVE_CYCLES"]):
        if len(k) > 300 or "at::native" in k or c[k] == 0:
            continue
        v, n = a[k], c[k]
        wc = v["SQ_WAVE_CYCLES"] or 1.0
        rec = {"launches": n, "wait_any_frac": v["SQ_WAIT_ANY"] / wc, "wait_inst_frac": v["SQ_WAIT_INST_ANY"] / wc,
               "active_inst_frac": v["SQ_ACTIVE_INST_ANY"] / wc, "valu_frac": v["SQ_ACTIVE_INST_VALU"] / wc,
               "lds_frac": v["SQ_ACTIVE_INST_LDS"] / wc, "wait_lds_frac": v["SQ_WAIT_INST_LDS"] / wc}
        line = (f"{k[:70]:70s} n={n:3d} wait_any={rec['wait_any_frac']:5.2f} wait_inst={rec['wait_inst_frac']:5.2f} "
                f"active={rec['active_inst_frac']:5.2f} valu={rec['valu_frac']:5.2f} lds={rec['lds_frac']:5.2f} wait_lds={rec['wait_lds_frac']:5.2f}")
        if k in b:
            w = b[k]
            busy = w["SQ_BUSY_CU_CYCLES"] or 1.0
            rec.update({"mfma_busy_frac_per_simd": w["SQ_VALU_MFMA_BUSY_CYCLES"] / busy / 4.0,
                        "insts_valu": w["SQ_INSTS_VALU"] / n, "insts_lds": w["SQ_INSTS_LDS"] / n,
                        "insts_salu": w["SQ_INSTS_SALU"] / n, "insts_vmem": w["SQ_INSTS_VMEM"] / n})
            line += (f" | mfma={rec['mfma_busy_frac_per_simd']:5.2f} insts valu={rec['insts_valu']/1e6:6.1f}M lds={rec['insts_lds']/1e6:6.1f}M "
                     f"salu={rec['insts_salu']/1e6:6.1f}M vmem={rec['insts_vmem']/1e6:6.1f}M")
        out["kernels"][k] = rec
        print(line)
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
