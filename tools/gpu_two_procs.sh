#!/bin/bash
# Two PROCESSES on the one GPU (fake RCCL for set-up, real HIP IPC + direct path for
# the iterations): upper bound of what the direct path adds to an iteration.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/two; mkdir -p $OUT
make -s -C tests/fake_rccl || exit 1
W=${1:-lap2d}
python bench.py --workload $W --fixed-iters 600 --steps 2 --warmup 1 --cpu-seconds 0 --krylov cg1 > $OUT/n1.log 2>&1 || exit 1
for c in p2p; do
LD_PRELOAD=$PWD/tests/fake_rccl/libfake_rccl.so LSB_BENCH_BACKEND=gloo LSB_BENCH_ONE_GPU=1 \
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29677 \
  bench.py --gpus 2 --workload $W --fixed-iters 600 --steps 2 --warmup 1 --cpu-seconds 0 --comm $c > $OUT/n2_$c.log 2>&1 || exit 1
done
python3 - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "n_gpus", d["n_gpus"], "us/iter %.1f" % (d["ms_per_step"] * 1e3 / d["config"]["iterations_per_solve"]),
                  "comm", d["comm"]["mode"], "selftest us", d["comm"]["selftest_direct_us"], d["comm"]["selftest_rccl_us"], "spmv us %.1f" % (d["roofline"]["launch_ms"] * 1e3))
PY
