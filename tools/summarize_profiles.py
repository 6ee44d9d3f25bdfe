#!/usr/bin/env python3
"""Condense gpurun_out/<dir> (tools/gpu_profiles.sh) into profiles/<round>_*:
kernel-trace stats, PMC traffic per launch (FETCH_SIZE corrected x2 for gfx950
as MI355X_MICROARCH.md section HBM prescribes; WRITE_SIZE as read), the bench
lines, and profiles/pmc_traffic.json that bench.py reads for roofline.traffic
(bytes per launch of the dominant kernel, per workload)."""
import collections, csv, glob, json, os, shutil, sys

src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return g[-1] if g else None  # newest run wins


for tag in ("trace", "trace_coef", "trace_lap3d", "trace_cfg2", "trace_cfg2_fsai", "trace_powerlaw", "trace_csr",
            "trace_gmres", "trace_coef_fp32"):
    f = one(tag + "/*/*kernel_stats.csv")
    if f:
        shutil.copyfile(f, os.path.join(out, "%s_%s_kernel_stats.csv" % (rnd, tag)))

traffic = {}
def bench_line_of(tag):
    """The JSON line the profiled command itself printed (its roofline record names the
    kernel sources' hash and the layout's kept / all value slots the counters belong to)."""
    f = os.path.join(src, tag + ".log")
    if os.path.exists(f):
        for l in open(f):
            if l.startswith("{"):
                return json.loads(l)
    return None


# (workload, directory suffix): the un-forced pass of every workload, then passes with the SpMV flavour FORCED
# (tools/gpu_profiles.sh part D) -- the timing pass picks per box, and bench.py quotes the entry of the
# flavour its own run picked
VARIANTS = [("lap2d", "", ""), ("lap2d_coef", "_coef", ""), ("lap3d", "_lap3d", ""), ("powerlaw", "_powerlaw", ""),
            ("lap2d", "_f70", "_f70"), ("lap2d", "_f198", "_f198"), ("lap3d", "_lap3d_f198", "_f198"),
            ("lap3d", "_lap3d_f198p", "_f198p"), ("lap3d", "_lap3d_f70", "_f70"), ("lap3d", "_lap3d_f70p", "_f70p"),
            ("lap3d", "_lap3d_f326", "_f326"), ("lap2d_coef", "_coef_f6", "_f6")]
for wl, suffix, vtag in VARIANTS:
    pmc = {}
    for cname, tag in (("FETCH_SIZE", "pmc_fetch" + suffix), ("WRITE_SIZE", "pmc_write" + suffix)):
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
            pmc[k][cname + "_KB_sum"] = sum(v)
            pmc[k]["launches_" + cname] = len(v)
    if not pmc:
        continue
    lines = ["kernel,launches,FETCH_SIZE_KB(median),fetch_bytes_corrected_x2,WRITE_SIZE_KB(median),write_bytes,hbm_bytes_per_launch"]
    best = also = None
    for k, d in sorted(pmc.items()):
        if "FETCH_SIZE_KB_median" not in d or "WRITE_SIZE_KB_median" not in d:
            continue
        fb = 2 * d["FETCH_SIZE_KB_median"] * 1024  # gfx950: FETCH_SIZE reads half the bytes
        wb = d["WRITE_SIZE_KB_median"] * 1024
        lines.append("%s,%d,%.1f,%.0f,%.1f,%.0f,%.0f" % (k, d["launches_FETCH_SIZE"], d["FETCH_SIZE_KB_median"], fb,
                                                         d["WRITE_SIZE_KB_median"], wb, fb + wb))
        if ("k_spmv_" in k or "k_pb_" in k or "k_pcg_col_px" in k) and d["launches_FETCH_SIZE"] > (10 if "k_pcg_col_px" in k else 20):
            name = k.split("<")[0].split()[-1]
            if best is None or d["launches_FETCH_SIZE"] > best[1]:
                best = (name, d["launches_FETCH_SIZE"], fb + wb)
            if name == "k_pcg_col_px":   # the launch that carries the SpMV in the two-launch iteration: an entry of its own,
                # the mean over its two instantiations (with / without the x update of every second iteration)
                n0, b0 = (also[1], also[2]) if also else (0, 0.0)
                n1 = d["launches_FETCH_SIZE"]
                also = (name, n0 + n1, (b0 * n0 + (fb + wb) * n1) / (n0 + n1))
    if wl == "powerlaw":
        # one SpMV of the two-phase form = k_pb_products + k_pb_reduce: per-launch means of the two
        # (the FETCH and WRITE passes are separate runs)
        tp = {}
        for k, d in pmc.items():
            for name in ("k_pb_products", "k_pb_reduce"):
                if name in k and "FETCH_SIZE_KB_sum" in d and "WRITE_SIZE_KB_sum" in d:
                    tp[name] = (2 * d["FETCH_SIZE_KB_sum"] / d["launches_FETCH_SIZE"] +
                                d["WRITE_SIZE_KB_sum"] / d["launches_WRITE_SIZE"]) * 1024
                    tp[name + "_n"] = d["launches_FETCH_SIZE"]
        if "k_pb_products" in tp and "k_pb_reduce" in tp:
            tot = tp["k_pb_products"] + tp["k_pb_reduce"]
            best = ("k_pb_products + k_pb_reduce", tp["k_pb_products_n"], tot)
            lines.append("# two-phase SpMV = k_pb_products (%.0f bytes) + k_pb_reduce (%.0f bytes): %.0f bytes"
                         % (tp["k_pb_products"], tp["k_pb_reduce"], tot))
        # one SpMV of the binned form = one launch per window of x: all launches of the kernel
        # summed, divided by the number of SpMVs (launches / windows; 8 M columns, 524288 per window)
        nbins = -(-8000000 // 524288)
        for k, d in pmc.items():
            if "k_spmv_binned<2" in k and "FETCH_SIZE_KB_sum" in d and "WRITE_SIZE_KB_sum" in d:
                # (the two passes are separate runs and may hold different numbers of launches of
                # this flavour -- the start-up timing pass picks it or its plain-load twin)
                nf, nw = d["launches_FETCH_SIZE"], d["launches_WRITE_SIZE"]
                tot = (2 * d["FETCH_SIZE_KB_sum"] / nf + d["WRITE_SIZE_KB_sum"] / nw) * 1024 * nbins
                if not best or not best[0].startswith("k_pb_"):
                    best = ("k_spmv_binned", nf, tot)
                lines.append("# k_spmv_binned per SpMV = %d windows x mean launch (%d launches in the FETCH "
                             "pass, %d in the WRITE pass): %.0f bytes" % (nbins, nf, nw, tot))
    csvname = "%s_pmc_traffic_%s%s.csv" % (rnd, wl, vtag)
    open(os.path.join(out, csvname), "w").write("\n".join(lines) + "\n")
    if best:
        d = bench_line_of("pmc_fetch" + suffix) or {}
        roof = d.get("roofline", {})
        vs = roof.get("value_slots", {})
        if not roof.get("kernels_sha16"):  # (a line of an older bench.py: the same call, the same sources)
            roof["kernels_sha16"] = (bench_line_of("pmc_fetch") or {}).get("roofline", {}).get("kernels_sha16")
        entry = {"bytes": best[2], "kernel": best[0], "launches": best[1],
                       "kernels_sha16": roof.get("kernels_sha16"),
                       "value_slots": [vs.get("kept", 0), vs.get("all", 0)],
                       # the flavour the timing pass of the profiled run picked: bench.py quotes the
                       # figure only for a run that picked the same
                       "spmv_flags": roof.get("spmv_flags"), "xcd_period_slices": roof.get("xcd_period_slices"),
                       "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                 "separate passes of this command; FETCH_SIZE x 2)" % csvname}
        have = traffic.setdefault(wl, [])
        for b in (best, also):
            if not b:
                continue
            e2 = dict(entry, bytes=b[2], kernel=b[0], launches=b[1])
            if not any((e["spmv_flags"], e["xcd_period_slices"], e["kernel"]) ==
                       (e2["spmv_flags"], e2["xcd_period_slices"], e2["kernel"]) for e in have):
                have.append(e2)
    print(wl, suffix)
    print("\n".join(lines))
# csr_kernel (bench.py --only csr_kernel): the two kernels that stream 12 B per non-zero
pmc = {}
for cname, tag in (("FETCH_SIZE", "pmc_fetch_csr"), ("WRITE_SIZE", "pmc_write_csr")):
    f = one(tag + "/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort()
        pmc.setdefault(k, {})[cname] = (v[len(v) // 2], len(v))
line = (bench_line_of("pmc_fetch_csr") or {}).get("csr_kernel")
if pmc and line:
    lines = ["kernel,launches,FETCH_SIZE_KB(median),fetch_bytes_corrected_x2,WRITE_SIZE_KB(median),write_bytes,hbm_bytes_per_launch"]
    for short in ("k_spmv_adaptive", "k_spmv_sell"):
        best = None
        for k, d in pmc.items():
            if k.split("<")[0].split()[-1] == short and "FETCH_SIZE" in d and "WRITE_SIZE" in d and d["FETCH_SIZE"][1] > 20:
                if best is None or d["FETCH_SIZE"][1] > best[1]["FETCH_SIZE"][1]:
                    best = (k, d)
        if not best:
            continue
        fb, wb = 2 * best[1]["FETCH_SIZE"][0] * 1024, best[1]["WRITE_SIZE"][0] * 1024
        lines.append("%s,%d,%.1f,%.0f,%.1f,%.0f,%.0f" % (best[0], best[1]["FETCH_SIZE"][1], best[1]["FETCH_SIZE"][0], fb,
                                                         best[1]["WRITE_SIZE"][0], wb, fb + wb))
        kk = line["kernels"][short]
        traffic["csr_kernel:" + short] = [{
            "bytes": fb + wb, "kernel": short, "launches": best[1]["FETCH_SIZE"][1],
            "kernels_sha16": line.get("kernels_sha16"), "value_slots": [0, 0],
            "spmv_flags": kk["spmv_flags"], "xcd_period_slices": 0,
            "source": "profiles/%s_pmc_traffic_csr_kernel.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes "
                      "of `bench.py --only csr_kernel`; FETCH_SIZE x 2)" % rnd}]
    open(os.path.join(out, "%s_pmc_traffic_csr_kernel.csv" % rnd), "w").write("\n".join(lines) + "\n")
    print("csr_kernel")
    print("\n".join(lines))
if traffic:
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
# bench lines
names = ["bench", "bench_driver", "bench_driver_final", "bench_coef", "bench_coef_fp32", "gmres_coef", "gmres_xn3b_raw", "only_subrecords", "bench_powerlaw", "bench_powerlaw_v7", "bench_powerlaw_v6", "bench_powerlaw_v1", "cfg5_spd_cg", "cfg2_launches", "cfg2_persistent",
         "cfg2_dense_inverse", "cfg2_cheb4", "cfg2_fsai2", "cfg2_fsai3", "cfg2_fsai3_six_launches", "cfg3_no_templates", "cfg3_fp32", "cfg3_cheb4", "cfg3_cheb16", "cfg3_bj8"]
with open(os.path.join(out, "%s_bench.jsonl" % rnd), "w") as fo:
    for tag in names:
        f = os.path.join(src, tag + ".log")
        if os.path.exists(f):
            for l in open(f):
                if l.startswith("{"):
                    d = json.loads(l)
                    d["_run"] = tag
                    fo.write(json.dumps(d) + "\n")
print(traffic)
