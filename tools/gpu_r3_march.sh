#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_march}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "march or plane_periodic or constant_slots" > $OUT/pytest.log 2>&1; rc=$?
tail -n 5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --workload lap3d --verbose 2 --steps 2 --cpu-seconds 0 > $OUT/bench3d.log 2> $OUT/bench3d.err; rc=$?
grep "spmv tune form=5 flags=\(70\|198\)" $OUT/bench3d.err
python - "$OUT/bench3d.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); r = d["roofline"]
print("cfg4", d["value"], "solves/s; spmv", r["launch_ms"], "b2b", r["back_to_back_launch_ms"], "frac", r["frac"], "bytes", r["algorithmic_bytes"], "flags", r["spmv_flags"], "period", r["xcd_period_slices"])
PY
exit $rc
