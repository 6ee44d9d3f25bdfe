#!/bin/bash
# classic against single-reduction PCG on ONE GPU, configs 3 and 4, through bench.py (round 4: with the vector slab)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_krylov}; mkdir -p $out
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --csr-kernel 0 --cfg2 0 --cfg5 0"
for rep in 1 2; do for k in cg cg1; do
  timeout -k 10 300 python bench.py --krylov $k --steps 3 $Q > $out/c3_${k}_$rep.log 2>/dev/null
  timeout -k 10 300 python bench.py --workload lap3d --krylov $k --steps 2 $Q > $out/c4_${k}_$rep.log 2>/dev/null
done; done
python - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], "%.4f solves/s" % d["value"], d["config"]["iterations_per_solve"], "its", "%.1f us/iter" % (1e6 / d["iterations_per_sec"]), d["config"]["solver"])
PY
