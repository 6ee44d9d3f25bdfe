#!/bin/bash
# round 3, first call: the GPU suite on the changed code, then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_first}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -n 5 $OUT/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 500 python bench.py > $OUT/bench.log 2> $OUT/bench.err; rc=$?
tail -c 6000 $OUT/bench.log; tail -n 5 $OUT/bench.err
exit $rc
