#!/bin/bash
# On the GPU box, from the repo root:  bash tools/pmc_by_launch.sh <variant|-> <kernel substring> "<counters>" <prof_run.py args...>
# One rocprofv3 --pmc pass of tools/prof_run.py; prints the counters of every launch of the named kernel in the
# LAST render, in launch order (for the wavefront kernels: one line per bounce).
v=$1; kern=$2; ctrs=$3; shift 3
root=$PWD
[ "$v" != "-" ] && export RT_AMD_LIB=$root/raytracing_amd/lib/variants/$v/librt_amd.so
out=/tmp/pmcl_$$; rm -rf $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $root/tools/prof_run.py "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out "$kern" <<'EOF'
import collections, csv, glob, sys
rows = collections.OrderedDict()
gen = []
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d = int(r["Dispatch_Id"])
        if "wf_generate" in r["Kernel_Name"]:
            gen.append(d)
        if sys.argv[2] in r["Kernel_Name"] and "retry" not in r["Kernel_Name"]:
            rows.setdefault(d, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
last = max(gen) if gen else 0
for b, d in enumerate(sorted(k for k in rows if k > last)):
    print("launch %d" % b, " ".join("%s=%.4g" % (c, x) for c, x in sorted(rows[d].items())))
EOF
