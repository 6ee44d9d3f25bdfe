#!/bin/bash
# round 4: the GPU suite on the current tree (optionally only the tests matching $2), then the GMRES lines
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_suite}; mkdir -p $OUT
if [ -n "$2" ]; then timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "$2" > $OUT/pytest.log 2>&1; rc=$?
else timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; fi
tail -n 6 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --csr-kernel 0 --cfg2 0 --cfg5 0"
F=file:tests/golden/matrices/xn3b_A_18.txt.gz
timeout -k 10 300 python bench.py --workload lap2d_coef --krylov gmres --restart 30 --fixed-iters 300 --steps 2 --warmup 1 $Q > $OUT/gmres_coef.log 2>$OUT/gmres_coef.err; echo "gmres_coef rc=$?"
timeout -k 10 300 python bench.py --workload $F --operator raw --krylov gmres --restart 30 --tol 1e-10 --steps 20 --warmup 2 --verify 0 $Q > $OUT/gmres_xn3b_raw.log 2>$OUT/gmres_xn3b_raw.err; echo "gmres_xn3b rc=$?"
python - "$OUT" <<'PY'
import json, sys
for n in ("gmres_coef", "gmres_xn3b_raw"):
    for l in open("%s/%s.log" % (sys.argv[1], n)):
        if l.startswith("{"):
            d = json.loads(l); print(n, d["value"], d["config"]["iterations_per_solve"], d["iteration"])
PY
