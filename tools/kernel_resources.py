"""Register use of every kernel of rt_kernels.hip (compiler remarks, no GPU needed):
    python tools/kernel_resources.py [filter] [extra hipcc flags...]
Prints: kernel, VGPRs, spilled VGPRs, scratch bytes per lane, waves per SIMD."""
import re
import subprocess
import sys

flt = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-c", "raytracing_amd/csrc/rt_kernels.hip", "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for name, r in rows.items():
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", short.replace("(anonymous namespace)::", "").replace("void ", ""))
    if flt in short:
        print("%-52s vgpr %3d spill %3d scratch %5d lds %6d waves/SIMD %d" % (short, r.get("vgpr", -1), r.get("spill", -1),
                                                                              r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
