#!/bin/bash
# constant-slot SpMV lab (tools/stencil_lab.hip): config 3's operator under kernel variants
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-stencil_lab}; mkdir -p $OUT
timeout -k 10 ${2:-300} tools/stencil_lab.bin ${3:-3162} ${4:-3162} ${5:-200} > $OUT/lab.log 2>&1; rc=$?
cat $OUT/lab.log
exit $rc
