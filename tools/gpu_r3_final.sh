#!/bin/bash
# round-3 closing call: the GPU suite, smoke, then the profile bundle
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_final}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -n 4 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?
tail -n 2 $OUT/smoke.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/gpu_profiles.sh ${1:-r3_final}/profiles
