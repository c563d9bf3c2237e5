#!/bin/bash
# On the GPU box, from the repo root:  bash tools/trace_launches.sh <variant|-> <prof_run.py args...>
# rocprofv3 --kernel-trace of one tools/prof_run.py run; prints the duration of every wavefront launch of the
# LAST render in launch order (one line per bounce: extend ms, shade ms).
v=$1; shift
root=$PWD
[ "$v" != "-" ] && export RT_AMD_LIB=$root/raytracing_amd/lib/variants/$v/librt_amd.so
out=/tmp/trace_$$; rm -rf $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/prof_run.py "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for s in ("wf_extend", "wf_shade", "wf_generate", "wf_accumulate", "wf_bin_scatter", "wf_bin_hist"):
            if s in n and "retry" not in n:
                rows.append((int(r["Start_Timestamp"]), s, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
rows.sort()
# last render = after the last wf_generate
last = max(i for i, r in enumerate(rows) if r[1] == "wf_generate")
ext = [r[2] for r in rows[last:] if r[1] == "wf_extend"]
shd = [r[2] for r in rows[last:] if r[1] == "wf_shade"]
srt = [r[2] for r in rows[last:] if r[1] == "wf_bin_scatter"]
hst = [r[2] for r in rows[last:] if r[1] == "wf_bin_hist"]
for b, (e, s) in enumerate(zip(ext, shd)):
    print("bounce %d extend %.3f ms shade %.3f ms" % (b, e, s))
if srt:
    print("sort launches (histogram + scatter, ms):", " ".join("%.3f" % (a + c) for a, c in zip(hst, srt)))
print("sum extend %.3f shade %.3f" % (sum(ext), sum(shd)))
PY
