#!/bin/bash
# round 4: where the vectors land -- one slab (default) against separate allocations, with and
# without round 3's lottery; 8 fresh solvers each, config 3, its general-values twin, Chebyshev 4
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_place}; mkdir -p $out
run() { label=$1; shift; echo "== $label" >> $out/log.txt; env "$@" >> $out/log.txt 2>$out/err_$label.txt; rc=$?; tail -3 $out/log.txt; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
P="timeout -k 10 400 python tools/gpu_placement_probe.py"
run slab            $P 8
run pieces          LSBENCH_HIP_NO_SLAB=1 $P 8
run pieces_lottery  LSBENCH_HIP_NO_SLAB=1 LSBENCH_HIP_PLACEMENT_LOTTERY=1 $P 8
run slab_coef       $P 6 lap2d:nx=3162,ny=3162,coef=1
run pieces_coef     LSBENCH_HIP_NO_SLAB=1 $P 6 lap2d:nx=3162,ny=3162,coef=1
run slab_cheb       $P 6 lap2d:nx=3162,ny=3162 cheb
run pieces_cheb     LSBENCH_HIP_NO_SLAB=1 $P 6 lap2d:nx=3162,ny=3162 cheb
export TMPDIR=/tmp; timeout -k 10 120 rocprofv3 -L > $out/counters.txt 2>&1; echo "counter list: $(wc -l < $out/counters.txt) lines"
cat $out/log.txt
