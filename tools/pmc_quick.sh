#!/bin/bash
# On the GPU box, from the repo root:  bash tools/pmc_quick.sh <variant|-> "<counters>" <prof_run.py args...>
# One rocprofv3 --pmc pass of tools/prof_run.py with the named library variant ("-" = the in-tree build);
# prints the per-kernel totals of every counter over all launches of the run.
v=$1; ctrs=$2; shift 2
root=$PWD
[ "$v" != "-" ] && export RT_AMD_LIB=$root/raytracing_amd/lib/variants/$v/librt_amd.so
out=/tmp/pmcq_$$; rm -rf $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $root/tools/prof_run.py "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out <<'EOF'
import collections, csv, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for s in ("wf_extend", "wf_shade", "wf_generate", "wf_accumulate", "k_render_mega"):
            if s in k:
                k = s + ("<true" in k and "_count" or "")
                break
        else:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], ) not in seen:
            seen.add((r["Dispatch_Id"], ))
            n[k] += 1
for k in sorted(tot):
    print(k, "launches", n[k], " ".join("%s=%.4g" % (c, v) for c, v in sorted(tot[k].items())))
EOF
