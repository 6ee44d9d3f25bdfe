#!/bin/bash
# in-solve A/B of environment switches: tools/gpu_ab.sh OUT "ENV=VAL ..." "ENV=VAL ..." ...
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-ab}; shift; mkdir -p $OUT
for wl in lap2d lap3d; do
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python bench.py --workload $wl --fixed-iters 600 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$wl [$cfg] flags=%s spmv=%.1f us (%.1f%%) iter=%.1f us' % (r['spmv_flags'], r['launch_ms']*1e3, r['frac']*100, d['ms_per_step']*1e3/600))
" | tee -a $OUT/ab.log
  rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then exit $rc; fi
done; done; done
