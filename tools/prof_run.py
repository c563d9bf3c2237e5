"""Small driver for profiling runs: builds one workload and launches the chosen kernels a few times.

    python tools/prof_run.py [--workload spheres1m|cornell|hf100k] [--spp N] [--depth D]
                             [--pipeline mega|wavefront|primary] [--reps R] [--width W --height H]
Prints wall-clock per launch and Mrays/s; meant to sit behind `rocprofv3 ... -- python tools/prof_run.py`.
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="spheres1m")
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--pipeline", default="mega")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--count", action="store_true")
    ap.add_argument("--grow-spp", type=int, default=0, help="render this many samples once first (sizes the path buffers)")
    a = ap.parse_args()

    import torch
    import raytracing_amd as rt
    from raytracing_amd import scenes

    gens = {"spheres1m": ("spheres1m", {}), "cornell": ("cornell", {}), "hf100k": ("heightfield100k", {}),
            "glass": ("glass", {"grid": 6, "subdiv": 5}), "emissive": ("emissive", {"subdiv": 4, "lamps": 3})}
    gen, kw = gens[a.workload]
    tmp = tempfile.mkdtemp(prefix="rt_prof_")
    path = scenes.generate(gen, os.path.join(tmp, gen + ".gltf"), **kw)
    W, H = a.width, a.height
    scene = rt.Scene(rt.load_gltf(path, W, H), device=0)
    print("scene", scene.stats())
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    buf = torch.empty(W * H * 3, dtype=torch.float32, device=dev)
    ids = torch.empty(W * H, dtype=torch.int32, device=dev)
    scene.set_counting(a.count)

    def launch():
        if a.pipeline == "primary":
            scene.trace_primary_dev(W, H, ids.data_ptr(), buf.data_ptr(), stream=stream)
        else:
            scene.render_dev(W, H, a.spp, buf.data_ptr(), max_depth=a.depth, rng_mode=rt.RNG_COUNTER, seed=1,
                             pipeline=rt.PIPE_MEGAKERNEL if a.pipeline == "mega" else rt.PIPE_WAVEFRONT, stream=stream)

    if a.grow_spp:
        spp, a.spp = a.spp, a.grow_spp
        launch()
        torch.cuda.synchronize()
        a.spp = spp
    launch()
    torch.cuda.synchronize()
    scene.reset_counters(stream)
    scene.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    c = scene.counters()
    kt = {k: round(v[0] / a.reps, 2) for k, v in scene.kernel_times().items() if v[1]}
    print("ms/launch %.3f  Mrays/s %.1f  kernels(ms) %s  counters %s" % (dt * 1e3, c["segments"] / a.reps / dt * 1e-6, kt, c))


if __name__ == "__main__":
    main()
