#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_fuse}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_sell.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -n 5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --steps 3"
for nf in 0 1; do
  LSBENCH_HIP_NO_FUSE_P_TMPL=$nf timeout -k 10 300 python bench.py $Q > $OUT/bench_nofuse$nf.log 2>&1 || exit 1
  python - "$OUT/bench_nofuse$nf.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); r = d["roofline"]
print(sys.argv[1], d["value"], "solves/s", d["config"]["iterations_per_solve"], "its", d["ms_per_step"]/d["config"]["iterations_per_solve"]*1e3, "us/it; kernel", r["kernel"], r["launch_ms"], "ms frac", r["frac"], "bytes", r["algorithmic_bytes"])
PY
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --cpu-seconds 0 --cfg4 0 --general-values 0 --fixed-iters 400 --steps 2 --warmup 0 > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -n 1); cut -d, -f1-4 $f | head -n 6 | cut -c1-120
find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
