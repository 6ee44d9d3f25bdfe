#!/bin/bash
# generic preconditioners (Chebyshev, block-Jacobi) on config 3: the sweeps' nontemporal mask
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-nt_generic}; mkdir -p $OUT
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --steps 2"
for m in 63 47 41 37 0; do
  for p in "cheb --cheb-degree 4" "cheb --cheb-degree 16" "bj --block-size 8"; do
    LSBENCH_HIP_BLAS1_NT=$m timeout -k 10 300 python bench.py --precond $p $Q > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
    python3 - "$m" "$p" $OUT/run.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][0])
print("mask %2s %-22s %.4f solves/s  outer SpMV %.1f us  mask used %s" % (sys.argv[1], sys.argv[2], d["value"], d["roofline"]["launch_ms"] * 1e3, d["blas1_nt_mask"]), flush=True)
PY
  done
done
