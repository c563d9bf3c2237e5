#!/bin/bash
# On the GPU box, from the repo root:  bash tools/stats_quick.sh <variant|-> <prof_run.py args...>
# rocprofv3 --kernel-trace --stats of one tools/prof_run.py run; prints the kernel table (name cut short).
v=$1; shift
root=$PWD
[ "$v" != "-" ] && export RT_AMD_LIB=$root/raytracing_amd/lib/variants/$v/librt_amd.so
out=/tmp/stats_$$; rm -rf $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/prof_run.py "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep ms/launch $out.log | cut -c1-200
python3 - $out <<'PY'
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"\(.*", "", n)[:40]
        print("%-40s calls %5s total %9.3f ms avg %9.1f us  %5.1f%%" % (n, r["Calls"], float(r["TotalDurationNs"]) * 1e-6, float(r["AverageNs"]) * 1e-3, float(r["Percentage"])))
PY
