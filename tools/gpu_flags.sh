#!/bin/bash
# in-solve SpMV time for every adaptive-kernel flavour (fixed 600 iterations)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-flags}; mkdir -p $OUT
for wl in lap2d lap3d; do
for f in -1 0 1 2 3; do
  timeout -k 10 300 python bench.py --workload $wl --spmv-tune $f --fixed-iters 600 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$wl tune=$f flags=%s spmv=%.1f us (%.1f%%) iter=%.1f us' % (r['spmv_flags'], r['launch_ms']*1e3, r['frac']*100, d['ms_per_step']*1e3/600))
" | tee -a $OUT/flags.log
  rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then exit $rc; fi
done; done
