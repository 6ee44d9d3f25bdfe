#!/bin/bash
# which operands of the two BLAS-1 sweeps should be loaded nontemporal?  LSBENCH_HIP_BLAS1_NT masks
# (bit 0 x, 1 p and q, 2 r in k_pcg_update_xr; 3 r, 4 p in k_pcg_update_p), config 3 (and config 4
# with "lap3d" as the second argument), microseconds per iteration over fixed iterations
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-nt_masks}; mkdir -p $OUT
WL=${2:-lap2d}; IT=${3:-1500}
for m in ${MASKS:-31 0 1 3 5 7 9 17 25 15 23 24 8 16}; do
  LSBENCH_HIP_BLAS1_NT=$m timeout -k 10 300 python bench.py --workload $WL --fixed-iters $IT --steps 2 --warmup 1 --cpu-seconds 0 --cfg4 0 --general-values 0 > $OUT/m$m.log 2>&1 || exit 1
  python - "$OUT/m$m.log" $m $IT <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); r = d["roofline"]
print("mask %2s: %7.2f us/iteration, spmv %.1f us in the solve" % (sys.argv[2], d["ms_per_step"] / int(sys.argv[3]) * 1e3, r["launch_ms"] * 1e3), flush=True)
PY
done
