#!/usr/bin/env python3
"""Condense gpurun_out/<dir> (tools/gpu_profiles.sh) into profiles/<round>_*:
kernel-trace stats, PMC traffic per launch (FETCH_SIZE corrected x2 for gfx950
as MI355X_MICROARCH.md section HBM prescribes; WRITE_SIZE as read), and
profiles/pmc_traffic.json that bench.py reads for roofline.traffic."""
import collections, csv, glob, json, os, shutil, sys

src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return g[-1] if g else None  # newest run wins


for tag in ("trace", "trace_lap3d", "trace_cfg2"):
    f = one(tag + "/*/*kernel_stats.csv")
    if f:
        shutil.copyfile(f, os.path.join(out, "%s_%s_kernel_stats.csv" % (rnd, tag)))

pmc = {}
for cname, tag in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = one(tag + "/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort()
        pmc.setdefault(k, {})[cname + "_KB_median"] = v[len(v) // 2]
        pmc[k]["launches_" + cname] = len(v)

lines = ["kernel,launches,FETCH_SIZE_KB(median),fetch_bytes_corrected_x2,WRITE_SIZE_KB(median),write_bytes,hbm_bytes_per_launch"]
traffic = {}
for k, d in sorted(pmc.items()):
    if "FETCH_SIZE_KB_median" not in d or "WRITE_SIZE_KB_median" not in d:
        continue
    fb = 2 * d["FETCH_SIZE_KB_median"] * 1024  # gfx950: FETCH_SIZE reads half the bytes
    wb = d["WRITE_SIZE_KB_median"] * 1024
    lines.append("%s,%d,%.1f,%.0f,%.1f,%.0f,%.0f" % (k, d["launches_FETCH_SIZE"], d["FETCH_SIZE_KB_median"], fb,
                                                     d["WRITE_SIZE_KB_median"], wb, fb + wb))
    if "k_spmv_" in k and d["launches_FETCH_SIZE"] > 20:
        # the SpMV kernel the solve ran (the timing pass at setup launches the
        # other forms a few times each)
        name = k.split("<")[0].split()[-1]
        if d["launches_FETCH_SIZE"] > traffic.get("lap2d", {}).get("launches", 0):
            traffic["lap2d"] = {"bytes": fb + wb, "kernel": name, "launches": d["launches_FETCH_SIZE"]}
open(os.path.join(out, "%s_pmc_traffic_lap2d.csv" % rnd), "w").write("\n".join(lines) + "\n")
if traffic:
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
for tag in ("bench",):
    f = os.path.join(src, tag + ".log")
    if os.path.exists(f):
        with open(f) as fi, open(os.path.join(out, "%s_%s.jsonl" % (rnd, tag)), "w") as fo:
            fo.writelines(l for l in fi if l.startswith("{"))
print("\n".join(lines))
print(traffic)
