#!/bin/bash
# slice templates: parity tests, the timing pass's lines on configs 3 and 4, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_tmpl}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_sell.py tests/test_precond.py tests/test_mixed.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -n 5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --verbose 2 --steps 3 > $OUT/bench.log 2> $OUT/bench.err; rc=$?
grep "spmv tune" $OUT/bench.err | tail -n 60
python - "$OUT/bench.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r = d["roofline"]
print("cfg3", d["value"], "solves/s", d["ms_per_step"], "ms; spmv", r["launch_ms"], "b2b", r["back_to_back_launch_ms"], "frac", r["frac"], "bytes", r["algorithmic_bytes"], "flags", r["spmv_flags"])
c = d["cfg4"]; print("cfg4", c["value"], c["spmv"]["launch_ms"], c["spmv"]["frac"], c["spmv"]["spmv_flags"], c["spmv"]["xcd_period_slices"], c["spmv"]["algorithmic_bytes"])
g = d["general_values"]; print("general", g["value"], g["spmv"]["launch_ms"], g["spmv"]["frac"])
PY
exit $rc
