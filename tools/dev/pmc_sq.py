import csv, glob, sys, collections
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); cnt[k] += 1
    return acc, cnt
a, c = load(sys.argv[1])
b, _ = load(sys.argv[2]) if len(sys.argv) > 2 else ({}, None)
names = [k for k in a if any(t in k for t in ("conv_bwd_fused", "conv_block_fwd", "stem_bwd", "stem_fwd", "pf_kernel<40, 3, 3, 4, 2", "wgrad_kernel<BF16, 3, 40", "dgrad_s2_kernel<40"))]
for k in names:
    v = a[k]; n = c[k]
    wc = v["SQ_WAVE_CYCLES"] or 1
    line = f"{k[:60]:60s} n={n:3d} wait_any={v['SQ_WAIT_ANY']/wc:5.2f} wait_inst={v['SQ_WAIT_INST_ANY']/wc:5.2f} active={v['SQ_ACTIVE_INST_ANY']/wc:5.2f} valu={v['SQ_ACTIVE_INST_VALU']/wc:5.2f} lds={v['SQ_ACTIVE_INST_LDS']/wc:5.2f} wait_lds={v['SQ_WAIT_INST_LDS']/wc:5.2f}"
    if k in b:
        w = b[k]
        busy = w["SQ_BUSY_CU_CYCLES"] or 1
        line += f" | mfma_busy/cu_busy={w['SQ_VALU_MFMA_BUSY_CYCLES']/busy:5.2f} vmem={w['SQ_ACTIVE_INST_VMEM']/wc:5.2f} sca={w['SQ_ACTIVE_INST_SCA']/wc:5.2f} insts valu={w['SQ_INSTS_VALU']/n/1e6:6.1f}M lds={w['SQ_INSTS_LDS']/n/1e6:6.1f}M salu={w['SQ_INSTS_SALU']/n/1e6:6.1f}M vmem={w['SQ_INSTS_VMEM']/n/1e6:6.1f}M"
    print(line)
