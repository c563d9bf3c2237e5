"""Turns the raw rocprofv3 output of tools/profile_bench.sh into the committed summaries:

    python tools/summarize_profile.py <round-tag> [workload]

writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats table), profiles/<tag>_pmc_summary.txt
(per-kernel counter means per launch) and updates profiles/pmc_traffic.json (HBM-side bytes per
launch of the dominant kernels: 2 x FETCH_SIZE + WRITE_SIZE, both in KiB in the raw files; the
factor 2 is the gfx950 correction for FETCH_SIZE, which tallies 128-byte requests at 64 bytes).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "<true" in name:   # counting instantiations (bench.py's one statistics launch): not the timed kernels
        return None
    for k in ("wf_extend_retry", "wf_extend", "wf_shade", "wf_generate", "wf_accumulate", "wf_bin_hist", "wf_bin_scan",
              "wf_bin_scatter", "k_render_mega", "k_primary", "k_intersect", "k_tonemap"):
        if k in name:
            return k
    return None


def main():
    tag = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "spheres1m"
    raw = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = glob.glob(os.path.join(raw, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(prof, tag + "_kernel_stats.csv"))
    log = os.path.join(raw, "stats.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(prof, tag + "_bench_under_rocprof.json"), "w").write(lines[-1])
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(raw, "pmc*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(prof, tag + "_pmc_summary.txt"), "w") as out:
        out.write("# rocprofv3 --pmc, separate passes; mean per launch (launches counted)\n")
        for k in sorted(per):
            for c in sorted(per[k]):
                v = per[k][c]
                out.write("%-16s %-36s %16.6g  (%d launches)\n" % (k, c, sum(v) / len(v), len(v)))
    tpath = os.path.join(prof, "pmc_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for k in per:
        if "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k]:
            f = sum(per[k]["FETCH_SIZE"]) / len(per[k]["FETCH_SIZE"])
            w = sum(per[k]["WRITE_SIZE"]) / len(per[k]["WRITE_SIZE"])
            traffic["%s/%s" % (workload, k)] = {
                "bytes_per_launch": round((2.0 * f + w) * 1024.0), "fetch_kib_raw": round(f, 1), "write_kib": round(w, 1),
                "launches": len(per[k]["FETCH_SIZE"]), "source": tag + "_pmc_summary.txt"}
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    # Per-launch counter means the bench line's ceilings are computed from (bench.py divides them by the
    # launch time it measures itself), and the gather rate of the vector-memory path from the
    # microbenchmark output committed next to them.
    rates = {"source": tag + "_pmc_summary.txt", "workload": workload, "kernels": {}}
    # ray segments per wf_extend launch of the PMC passes (their bench line): bench.py scales the per-launch
    # counters to the launch size it measures itself
    plog = os.path.join(raw, "pmc1.log")
    if os.path.exists(plog):
        lines = [l for l in open(plog) if l.startswith("{")]
        if lines:
            rates["segments_per_launch"] = json.loads(lines[-1])["roofline"]["algorithmic"]["segments_per_launch"]
    want = {"TCP_TOTAL_CACHE_ACCESSES_sum": "l1_accesses", "SQ_INSTS_VALU": "valu_wave_insts",
            "SQ_INSTS_VMEM_RD": "vmem_rd_wave_insts", "SQ_INSTS_VMEM_WR": "vmem_wr_wave_insts",
            "GRBM_GUI_ACTIVE": "gui_active_cycles_sum_xcd", "TCP_TCC_READ_REQ_sum": "l1_misses_to_l2",
            "TCC_HIT_sum": "l2_hits", "TCC_MISS_sum": "l2_misses"}
    for k in per:
        rec = {}
        for c, name in want.items():
            if c in per[k]:
                rec[name] = sum(per[k][c]) / len(per[k][c])
                rec["launches"] = len(per[k][c])
        if "%s/%s" % (workload, k) in traffic:
            rec["fabric_bytes"] = traffic["%s/%s" % (workload, k)]["bytes_per_launch"]
        rates["kernels"][k] = rec
    mb = os.path.join(prof, tag[:3] + "_microbench_node_fetch.txt")
    if os.path.exists(mb):
        # first table = 16 KB of records (resident in every L1); A32 = two global_load_dwordx4 per lane on a
        # random 32-byte record, next address depends on the data: what a BVH node visit does
        for line in open(mb):
            f = line.split()
            if f and f[0] == "A32":
                rates["vmem_gather_peak"] = {"lane_loads_per_s": float(f[3]) * 1e9 * 2.0,
                                             "source": os.path.basename(mb) + ": A32, 16 KB table, 2 x 16-byte loads per lane and record"}
                break
    json.dump(rates, open(os.path.join(prof, "pmc_rates.json"), "w"), indent=1, sort_keys=True)
    print(open(os.path.join(prof, tag + "_pmc_summary.txt")).read()[:3000])
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
