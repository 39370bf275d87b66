#!/usr/bin/env python3
"""Condense a tools/profile_rNN.sh run (gpurun_out/prof_rNN_*) into profiles/rNN_* files that are small
enough to commit: the rocprofv3 --stats table, per-kernel PMC averages and the derived figures
(HBM traffic per launch with the gfx950 FETCH_SIZE correction, LDS conflict rate, VALU issue share)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s_" % tag
os.makedirs("profiles", exist_ok=True)


def one(pat):
    f = glob.glob(pat)
    return f[0] if f else None


stats = one(src + "trace/vm/*_kernel_stats.csv")
shutil.copy(stats, "profiles/%s_kernel_stats.csv" % tag)
bench = json.load(open(src + "trace.json"))
kernels = collections.OrderedDict()
for r in csv.DictReader(open(stats)):
    if "fwht_pass_kernel" in r["Name"]:
        kernels[r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                              "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
pmc = collections.defaultdict(dict)
for name in ("fetch", "write", "sq", "tcc", "grbm", "wrreq", "rdreq", "sq2", "tcc2", "lat"):
    f = one(src + name + "/vm/*_counter_collection.csv")
    if not f:
        continue
    # a kernel's launches inside the steps all have one grid size; the same kernel is also launched on other grids outside
    # them (the placement search rates candidates with one full-size gather pass even where a step runs it chunk by chunk):
    # only the launches of the most common grid size count
    rows = [r for r in csv.DictReader(open(f)) if "fwht_pass_kernel" in r["Kernel_Name"]]
    grids = collections.defaultdict(collections.Counter)
    for r in rows:
        grids[r["Kernel_Name"]][r["Grid_Size"]] += 1
    modal = {k: c.most_common(1)[0][0] for k, c in grids.items()}
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Grid_Size"] == modal[r["Kernel_Name"]]:
            a = acc[(r["Kernel_Name"], r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for (k, c), (n, s) in acc.items():
        pmc[k][c] = s / n
out = {"tag": tag, "bench_line_of_the_traced_run": bench, "kernels": []}
for k, st in kernels.items():
    c = pmc.get(k, {})
    d = dict(name=k, **st, counters=c)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane reads
        d["hbm_read_bytes_per_launch_corrected"] = 2 * c["FETCH_SIZE"] * 1024
        d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
        d["hbm_traffic_bytes_per_launch"] = d["hbm_read_bytes_per_launch_corrected"] + d["hbm_write_bytes_per_launch"]
        d["hbm_traffic_GBps"] = d["hbm_traffic_bytes_per_launch"] / (st["avg_ms"] * 1e-3) / 1e9
    if "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_bank_conflict_rate"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
    if "SQ_WAVE_CYCLES" in c:
        d["valu_issue_share_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
        d["wait_share_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "GRBM_GUI_ACTIVE" in c:
        d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (st["avg_ms"] * 1e-3) / 1e9
    out["kernels"].append(d)
# one step = launches_per_step launches of each pass kernel (the cache-resident schedule runs a step chunk by chunk); the kind
# of a kernel is in its template arguments: <T, K, LOGV, RB, LU, LLDS, in mode, out mode, single> with mode 0 = workspace
lps = bench.get("roofline", {}).get("launches_per_step")
if lps and all("hbm_traffic_bytes_per_launch" in d for d in out["kernels"]):
    total = 0.0
    for d in out["kernels"]:
        args = d["name"][d["name"].index("<") + 1:d["name"].rindex(">")].split(",")
        inm, outm = int(args[6]), int(args[7])
        kind = "first_pass" if inm != 0 else "middle_passes" if outm == 0 else "last_pass"
        d["kind"], d["launches_per_step"] = kind, lps[kind]
        total += d["hbm_traffic_bytes_per_launch"] * lps[kind]
    out["hbm_traffic_bytes_per_step"] = total
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
if "hbm_traffic_bytes_per_step" in out:
    print("HBM traffic per step: %.2f GB" % (out["hbm_traffic_bytes_per_step"] / 1e9))
for d in out["kernels"]:
    print(d["name"][:90])
    print("   avg %.3f ms  traffic %.2f GB (%.0f GB/s)  lds conflicts %.4f  valu %.3f  wait %.3f  l2hit %.3f  clk %.2f GHz" % (
        d["avg_ms"], d.get("hbm_traffic_bytes_per_launch", 0) / 1e9, d.get("hbm_traffic_GBps", 0),
        d.get("lds_bank_conflict_rate", -1), d.get("valu_issue_share_of_wave_cycles", -1),
        d.get("wait_share_of_wave_cycles", -1), d.get("l2_hit_rate", -1), d.get("effective_clock_GHz", -1)))
