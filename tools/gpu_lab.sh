#!/bin/bash
# One gpurun call for kernel experiments: CPU-share probe, SpMV lab, PMC passes.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-lab}
mkdir -p "$OUT"
export TMPDIR=/tmp
{ echo "nproc: $(nproc)"; python3 -c "import os;print('affinity',len(os.sched_getaffinity(0)))";
  cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null;
  lscpu | head -20; free -g | head -2; } > $OUT/cpu_probe.log 2>&1
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "rc=$rc"; if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
step lab_lap2d 300 tools/spmv_lab lap2d 3162 15
step lab_lap3d 300 tools/spmv_lab lap3d 256 9
step lab_powerlaw 300 tools/spmv_lab powerlaw 2000000 9
step pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- tools/spmv_lab lap2d 3162 1
step pmc_write 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- tools/spmv_lab lap2d 3162 1
step pmc_l2 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_l2" -- tools/spmv_lab lap2d 3162 1
step pytest_rest 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py::test_lap3d_64m_rows
cat $OUT/lab_lap2d.log; tail -3 $OUT/pytest_rest.log
