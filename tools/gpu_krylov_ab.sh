#!/bin/bash
# classic against single-reduction PCG through bench.py (real stop test, verified residual), one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-krylov_ab}; mkdir -p $OUT
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --steps 2"
for rep in a b; do
for w in lap2d lap3d; do
  for k in cg cg1; do
    timeout -k 10 300 python bench.py --workload $w --krylov $k $Q > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
    python3 - "$w" "$k" $OUT/run.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][0])
print("%-6s %-4s %.4f solves/s  %d iterations  %.1f us per iteration  SpMV in the solve %.1f us  mask %s  %s" % (sys.argv[1], sys.argv[2], d["value"], d["config"]["iterations_per_solve"], d["iteration"]["us"] if d.get("iteration") else -1, d["roofline"]["launch_ms"] * 1e3, d["blas1_nt_mask"], d["config"]["solver"]), flush=True)
PY
  done
done
done
