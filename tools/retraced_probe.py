"""How many rays does the wavefront trace kernel hand to its second kernel, and what does that cost?
    python tools/retraced_probe.py [workload ...]      (GPU box; RT_AMD_LIB selects another build)"""
import os, sys, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracing_amd as rt
from raytracing_amd import scenes
sys.path.insert(0, ROOT)
from bench import WORKLOADS

for w in sys.argv[1:] or ["cornell", "emissive"]:
    gen, kw, W, H, spp, depth = WORKLOADS[w]
    path = scenes.generate(gen, os.path.join(tempfile.mkdtemp(), gen + ".gltf"), **kw)
    scene = rt.Scene(rt.load_gltf(path, W, H), device=0)
    buf = torch.empty(W * H * 3, dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream()
    spp = min(spp, 32)
    for counting in (False, True):
        scene.render_dev(W, H, spp, buf.data_ptr(), max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=1, pipeline=rt.PIPE_WAVEFRONT, stream=st)
        torch.cuda.synchronize()
        scene.reset_counters(st)
        scene.set_counting(counting)
        scene.set_timing(True)
        scene.render_dev(W, H, spp, buf.data_ptr(), max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=1, pipeline=rt.PIPE_WAVEFRONT, stream=st)
        torch.cuda.synchronize()
        kt = scene.kernel_times()
        c = scene.counters() if counting else {}
        scene.set_timing(False)
        scene.set_counting(False)
        print(json.dumps(dict(workload=w, counting=counting, hit_order=scene.stats()["hit_order"], extend_ms=round(kt["extend"][0], 3), extend_launches=kt["extend"][1],
                              shade_ms=round(kt["shade"][0], 3), segments=c.get("segments"), retraced=c.get("retraced"))), flush=True)
