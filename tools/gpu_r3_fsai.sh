#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_fsai}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_precond.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -n 5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
F=file:tests/golden/matrices/xn3b_A_18.txt.gz
C2="--workload $F --tol 1e-12 --steps 100 --warmup 5 --cfg4 0 --verify 0 --cpu-seconds 0 --persistent 0"
for k in 1 2 3; do
  timeout -k 10 200 python bench.py $C2 --precond fsai --fsai-power $k --verbose 1 > $OUT/cfg2_fsai$k.log 2> $OUT/cfg2_fsai$k.err || exit 1
  grep FSAI $OUT/cfg2_fsai$k.err | tail -n 1
  python - "$OUT/cfg2_fsai$k.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1], "%.1f solves/s" % d["value"], d["config"]["iterations_per_solve"], "its", "%.2f us/it" % (d["ms_per_step"] / d["config"]["iterations_per_solve"] * 1e3), "setup %.2f s" % d["setup_seconds"])
PY
done
timeout -k 10 200 python bench.py $C2 > $OUT/cfg2_jacobi.log 2>&1
python - "$OUT/cfg2_jacobi.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1], "%.1f solves/s" % d["value"], d["config"]["iterations_per_solve"], "its", "%.2f us/it" % (d["ms_per_step"] / d["config"]["iterations_per_solve"] * 1e3))
PY
