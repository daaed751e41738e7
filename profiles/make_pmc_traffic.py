#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, --kernel-trace only) of
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer` into profiles/pmc_traffic.json:
HBM-side bytes per launch for every kernel, averaged over its launches.

    python profiles/make_pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> profiles/pmc_traffic.json

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: counters are in KB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-byte stores.
"""
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            key = (r.get("Dispatch_Id"), k)
            acc.setdefault(k, {}).setdefault(key, 0.0)
            acc[k][key] += float(r["Counter_Value"])          # one row per XCD/instance: sum per dispatch
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in acc.items()}


def library_sha16():
    """sha256 (first 16 hex digits) of the libmil_hip.so these counters were collected on: bench.py reports whether the
    summary it quotes belongs to the library it is running."""
    import glob as _g
    import hashlib
    import os
    hits = _g.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "*_amd", "libmil_hip.so"))
    return hashlib.sha256(open(hits[0], "rb").read()).hexdigest()[:16] if hits else None


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_how": __doc__.strip().replace("\n", " "), "kernels": {}}
    out["_library_sha16"] = library_sha16()
    for k in sorted(set(fetch) | set(write)):
        if len(k) > 300:                                        # torch's templated elementwise kernels: not ours, skip
            continue
        n, fkb = fetch.get(k, (0, 0.0))
        n2, wkb = write.get(k, (n, 0.0))
        fb, wb = 2.0 * fkb * 1024.0, wkb * 1024.0
        out["kernels"][k] = {"launches": n or n2, "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:14]:
        print(f"{k[:90]:90s} n={v['launches']:4d} fetch={v['fetch_bytes']/1e6:9.1f} MB write={v['write_bytes']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
