#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 2 3; do for rep in 1 2; do
python3 bench.py --workload file:tests/golden/matrices/xn3b_A_18.txt.gz --tol 1e-12 --steps 200 --warmup 20 --cpu-seconds 0 --spmv $v 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg2 spmv_variant=$v solves/s=%.1f ms/solve=%.3f its=%d spmv_launch=%.2f us' % (d['value'], d['ms_per_step'], d['config']['iterations_per_solve'], d['roofline']['launch_ms']*1e3))
"
done; done
