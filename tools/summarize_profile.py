"""Condenses rocprofv3 output directories into the small files committed under profiles/.

usage: summarize_profile.py <round-tag> <stats_dir> [<fetch_pmc_dir> <write_pmc_dir> [<sq_pmc_dir>]]
"""
import csv, glob, json, os, sys, collections

tag, stats_dir = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(root, exist_ok=True)

def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None

ks = find(stats_dir, "kernel_stats.csv")
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        name = r["Name"]
        name = name if len(name) < 120 else name[:117] + "..."
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print(open(os.path.join(root, f"{tag}_kernel_stats.csv")).read()[:1500])

if len(sys.argv) >= 5:
    out = {}
    for label, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
        cc = find(d, "counter_collection.csv")
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] == label:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {})[label + "_KB_mean"] = sum(v) / len(v)
            out[k]["launches"] = len(v)
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per launch, KB. "
                   "gfx950 correction: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads, "
                   "so read bytes = 2 * FETCH_SIZE * 1024; other access widths are uncalibrated.",
           "kernels": out}
    for k, v in out.items():
        if "FETCH_SIZE_KB_mean" in v and "WRITE_SIZE_KB_mean" in v:
            v["hbm_bytes_per_launch"] = 2 * v["FETCH_SIZE_KB_mean"] * 1024 + v["WRITE_SIZE_KB_mean"] * 1024
    for k, v in out.items():
        for kern in ("k_search_fused_wave", "k_search_wave_injected", "k_search_wave_aos21"):
            if k.startswith("void " + kern) or k.startswith(kern):
                res[kern + "_bytes_per_launch"] = v.get("hbm_bytes_per_launch")
    json.dump(res, open(os.path.join(root, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])

if len(sys.argv) >= 6:
    # SQ / GRBM pass: matrix-pipe busy cycles against the kernel's active cycles.  rocprofv3 reports
    # GRBM_GUI_ACTIVE summed over the 8 XCDs; the chip has 256 CUs x 4 SIMDs.
    cc = find(sys.argv[5], "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_search" in k or "k_forward" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "mean per launch; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves", "kernels": {}}
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if m.get("GRBM_GUI_ACTIVE"):
            m["mfma_util"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if m.get("SQ_WAVE_CYCLES"):
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if n in m:
                    m[n + "_share_of_wave_cycles"] = m[n] / m["SQ_WAVE_CYCLES"]
        out["kernels"][k] = m
    json.dump(out, open(os.path.join(root, f"{tag}_sq_counters.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])
