#!/bin/bash
# On the GPU box: the bench frame and the 4K glass frame with ray sorting off and in the default (adaptive) mode.
for w in spheres1m glass4k; do
  for s in 0 d 0 d; do
    if [ $s = d ]; then unset RT_WF_SORT; else export RT_WF_SORT=$s; fi
    timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-primary > gpurun_out/ab_${w}_$s.json 2>/dev/null || exit 1
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_${w}_$s.json"))
print("$w sort=$s", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
PY
  done
done
