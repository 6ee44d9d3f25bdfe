#!/bin/bash
# round 4, closing call: the whole GPU suite, the smoke entry, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_final}; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -n 4 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
T0=$(date +%s); timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; rc=$?
echo "bench rc=$rc wall=$(( $(date +%s) - T0 )) s"
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]; print("headline", d["value"], r["launch_ms"], r["frac"], r["traffic"], r["frac_fabric"], r["traffic_source"][:60])
        for k in ("cfg4", "general_values"):
            q = d[k]["spmv"]; print(k, d[k]["value"], q["launch_ms"], q["frac"], q["traffic"], q["frac_fabric"], q["spmv_flags"], q.get("xcd_period_slices"))
        for k, v in d["csr_kernel"]["kernels"].items():
            print("csr", k, v["launch_us"], v["frac"], v["traffic"], v["frac_fabric"])
        print("cfg2", {k: v["value"] for k, v in d["cfg2"]["solvers"].items()}, d["cfg2"]["cpu_direct_baseline"]["value"])
        print("cfg5", d["cfg5"]["value"], d["cfg5"]["spmv"]["frac"], d["cfg5"]["spmv"]["traffic"])
PY
exit $rc
