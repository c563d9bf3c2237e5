"""Turns the raw rocprofv3 output of tools/profile_bench.sh into the committed summaries:

    python tools/summarize_profile.py <round-tag> [workload]

writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --stats table of the bench command),
profiles/<tag>_pmc_summary.txt (counters per kernel, summed over the launches of ONE pass of the wavefront
pipeline, for two passes of the same run: the first -- the sort tuner has measured nothing yet, every bounce is
traced unsorted -- and the last -- the tuner has settled, which is the state the timed steps of bench.py run
in), and profiles/pmc_rates.json, which bench.py reads for the ceilings of its roofline object.

Every PMC pass is a run of `bench.py --steps 1 --warmup 1 --spp 512`: eight passes of 128 samples.  The rows of a
counter file carry the dispatch order and each dispatch's start and end time, so the launches of a pass are
the n-th group of (launches of that kernel / launches of wf_generate) rows, and a kernel's duration in that
pass comes from the same rows as its counters.

HBM-side bytes: 2 x FETCH_SIZE + WRITE_SIZE, both in KiB in the raw files; the factor 2 is the gfx950 correction
for FETCH_SIZE, which tallies 128-byte requests at 64 bytes (Infinity Cache hits are in it: an upper bound on
HBM proper).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNELS = ("wf_extend_retry", "wf_extend", "wf_shade", "wf_generate", "wf_accumulate", "wf_bin_hist", "wf_bin_scan",
           "wf_bin_scatter", "k_render_mega", "k_primary", "k_intersect", "k_tonemap")


def short(name):
    if "<true" in name:   # counting instantiations (bench.py's one statistics launch): not the timed kernels
        return None
    for k in KERNELS:
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

    # per kernel and counter: [(dispatch id, value, duration ns)] in dispatch order
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(raw, "pmc*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                rows[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                   float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    passes = 0
    for c, v in rows.get("wf_generate", {}).items():
        passes = len(v)   # one wf_generate per pass
        break

    def of_pass(k, c, which):
        """Sum of counter c (and of the durations) over kernel k's launches of pass `which` (0 = first, -1 = last)."""
        v = sorted(rows[k][c])
        per = len(v) // max(passes, 1) if passes and len(v) % max(passes, 1) == 0 else len(v)
        if per == 0:
            return None
        part = v[:per] if which == 0 else v[len(v) - per:]
        return sum(x[1] for x in part), sum(x[2] for x in part) * 1e-6, per

    want = {"TCP_TOTAL_CACHE_ACCESSES_sum": "l1_accesses", "SQ_INSTS_VALU": "valu_wave_insts",
            "SQ_INSTS_VMEM_RD": "vmem_rd_wave_insts", "SQ_INSTS_VMEM_WR": "vmem_wr_wave_insts",
            "GRBM_GUI_ACTIVE": "gui_active_cycles_sum_xcd", "TCP_TCC_READ_REQ_sum": "l1_misses_to_l2",
            "TCC_HIT_sum": "l2_hits", "TCC_MISS_sum": "l2_misses", "TA_BUSY_avr": "ta_busy_cycles_avr",
            "TD_TD_BUSY_sum": "td_busy_cycles_sum", "SQ_WAVE_CYCLES": "wave_cycles", "SQ_BUSY_CYCLES": "sq_busy_cycles"}
    modes = {"first_pass": 0, "settled": -1}
    rates = {"source": tag + "_pmc_summary.txt", "workload": workload, "passes_in_run": passes,
             "what": "per kernel: counters and duration summed over the launches of one pass of 128 samples per pixel; "
                     "first_pass = no bounce sorted yet, settled = the last pass of the run (the sort tuner has decided)"}
    with open(os.path.join(prof, tag + "_pmc_summary.txt"), "w") as out:
        out.write("# rocprofv3 --pmc, separate passes of `bench.py --steps 1 --warmup 1 --spp 512` (%d wavefront passes each);\n"
                  "# per kernel and counter: the sum over that kernel's launches of ONE wavefront pass, and the summed duration of\n"
                  "# those same launches (ms, from the dispatch timestamps of the same rows)\n" % passes)
        for mode, which in modes.items():
            out.write("\n## %s\n" % mode)
            rec_all = {}
            for k in sorted(rows):
                rec = {}
                durs = []
                for c in sorted(rows[k]):
                    got = of_pass(k, c, which)
                    if not got:
                        continue
                    total, ms, n = got
                    out.write("%-16s %-36s %18.7g  %10.4f ms  (%d launches)\n" % (k, c, total, ms, n))
                    durs.append(ms)
                    rec["launches"] = n
                    if c in want:
                        rec[want[c]] = total
                    if c == "FETCH_SIZE":
                        rec["fetch_kib_raw"] = total
                    if c == "WRITE_SIZE":
                        rec["write_kib"] = total
                if "fetch_kib_raw" in rec and "write_kib" in rec:
                    rec["fabric_bytes"] = round((2.0 * rec["fetch_kib_raw"] + rec["write_kib"]) * 1024.0)
                if durs:
                    rec["duration_ms"] = sum(durs) / len(durs)   # mean over the PMC passes (each timed its own launches)
                rec_all[k] = rec
            rates[mode] = {"kernels": rec_all}
    # ray segments per pass of the PMC runs (their bench line)
    plog = os.path.join(raw, "pmc1.log")
    if os.path.exists(plog):
        lines = [l for l in open(plog) if l.startswith("{")]
        if lines:
            rates["segments_per_pass"] = json.loads(lines[-1])["roofline"]["segments_per_pass"]
    mb = os.path.join(prof, "r02_microbench_node_fetch.txt")
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
    print(open(os.path.join(prof, tag + "_pmc_summary.txt")).read()[:6000])


if __name__ == "__main__":
    main()
