#!/usr/bin/env python3
"""Turn rocprofv3 CSV output under gpurun_out/ into the small, committed summaries under profiles/.
usage: tools/summarize_profiles.py <tag> <bench_trace_dir> [<pmc_dir>]"""
import collections, csv, glob, json, os, sys
tag, trace_dir = sys.argv[1], sys.argv[2]
pmc_dir = sys.argv[3] if len(sys.argv) > 3 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles"); os.makedirs(out, exist_ok=True)
stats = list(csv.DictReader(open(glob.glob(os.path.join(trace_dir, "*", "*kernel_stats.csv"))[0])))
with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in stats:
        w.writerow([r["Name"].split("(")[0], r["Calls"], f"{int(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.2f}",
                    f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}", r["Percentage"]])
print("wrote", f"{tag}_bench_kernel_stats.csv", len(stats), "kernels")
if pmc_dir:
    def load(sub):
        return list(csv.DictReader(open(glob.glob(os.path.join(pmc_dir, sub, "*", "*counter_collection.csv"))[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write", "sq", "mfma"):
        try:
            for r in load(sub):
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        except IndexError:
            pass
    tr = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(glob.glob(os.path.join(pmc_dir, "trace", "*", "*kernel_stats.csv"))[0]))}
    rows = []
    for k, c in agg.items():
        if k.startswith("__amd"):
            continue
        row = {"kernel": k, "launches": len(next(iter(c.values()))), "avg_us(kernel-trace pass)": round(tr.get(k, float("nan")), 2)}
        for name, v in c.items():
            row[name + "_avg"] = round(sum(v) / len(v), 1)
        if "FETCH_SIZE_avg" in row:
            # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced
            # stream (uncalibrated for narrow accesses) -> give both the raw figure and the x2 upper estimate.
            row["hbm_read_KB_raw"] = row["FETCH_SIZE_avg"]; row["hbm_read_KB_x2"] = round(2 * row["FETCH_SIZE_avg"], 1)
        rows.append(row)
    with open(os.path.join(out, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print("wrote", f"{tag}_pmc_summary.json", len(rows), "kernels")
