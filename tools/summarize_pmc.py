#!/usr/bin/env python3
"""Per kernel and run of tools/gpu_r4_cfg4_pmc.sh: mean counter values per launch (rocprofv3
counter_collection.csv files under <dir>/<label>_<pass>/), for the kernels with >= 20 launches."""
import collections, csv, glob, os, sys
root = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))   # (label, kernel) -> counter -> values
for d in sorted(glob.glob(os.path.join(root, "*_[0-9]"))):
    label = os.path.basename(d).rsplit("_", 1)[0]
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            tab[(label, k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (label, k), c in sorted(tab.items()):
    n = max(len(v) for v in c.values())
    if n < 20 or not ("spmv" in k or "update" in k or "k_pcg" in k):
        continue
    m = {name: sum(sorted(v)[len(v) // 4: len(v) - len(v) // 4 or None]) / max(1, len(sorted(v)[len(v) // 4: len(v) - len(v) // 4 or None])) for name, v in c.items()}
    print("%-8s %-46s launches %4d" % (label, k[:46], n))
    for name in sorted(m):
        print("      %-44s %16.0f" % (name, m[name]))
    rd, wr = m.get("TCC_EA0_RDREQ_sum"), m.get("TCC_EA0_WRREQ_sum")
    if rd and m.get("TCC_EA0_RDREQ_LEVEL_sum"):
        print("      -> mean fabric read latency  %8.0f L2 cycles; DRAM share of reads %.2f" % (
            m["TCC_EA0_RDREQ_LEVEL_sum"] / rd, m.get("TCC_EA0_RDREQ_DRAM_sum", 0) / rd))
    if wr and m.get("TCC_EA0_WRREQ_LEVEL_sum"):
        print("      -> mean fabric write latency %8.0f L2 cycles; write stall cycles per request %.2f" % (
            m["TCC_EA0_WRREQ_LEVEL_sum"] / wr, m.get("TCC_EA0_WRREQ_STALL_sum", 0) / wr))
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        print("      -> of the wave cycles: parked (waitcnt) %.2f, issue-stalled %.2f, issuing %.2f" % (
            m.get("SQ_WAIT_ANY", 0) / w, m.get("SQ_WAIT_INST_ANY", 0) / w, m.get("SQ_ACTIVE_INST_ANY", 0) / w))
