#!/bin/bash
# kernel trace of the small (latency-bound) config 2 solve + power-law variants
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-small}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --workload file:tests/golden/matrices/xn3b_A_18.txt.gz --tol 1e-12 --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/cfg2.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
seg = rows[len(rows)//2: len(rows)//2 + 3000]
durs = collections.defaultdict(list); gaps = collections.defaultdict(list)
for a, b in zip(seg, seg[1:]):
    ka, kb = a["Kernel_Name"][5:24], b["Kernel_Name"][5:24]
    durs[ka].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gaps[(ka, kb)].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in durs.items():
    v.sort(); print("dur  %-22s median %6.2f us  n=%d" % (k, v[len(v)//2]/1e3, len(v)))
for k, v in gaps.items():
    v.sort(); print("gap  %-44s median %6.2f us  n=%d" % (k, v[len(v)//2]/1e3, len(v)))
PY
