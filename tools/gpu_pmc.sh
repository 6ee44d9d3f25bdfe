#!/bin/bash
# PMC passes over the SpMV lab (each pass its own run; kernel-trace only).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- tools/spmv_lab.bin lap2d 3162 1 > $OUT/$name.log 2>&1
  local rc=$?; echo "$name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
for p in ("sq1", "sq2", "ta", "tcp", "grbm"):
    fs = glob.glob(sys.argv[1] + "/%s/*/*counter_collection.csv" % p)
    if not fs:
        print(p, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "fill" in k: continue
        print(p, k, " ".join("%s=%.3g" % (c.replace("SQ_", "").replace("_sum", ""), sum(v) / len(v)) for c, v in sorted(d.items())))
PY
