"""Per-bounce cost of the wavefront kernels on one workload.

    python tools/bounce_profile.py [--workload spheres1m] [--spp 64] [--depth 8]

Renders the frame once per max_depth 1..D to get the number of ray segments of every bounce (the paths of
bounce b are the same for every max_depth > b + 1: same random numbers), then once more at depth D with an
event pair around every launch, and prints per bounce: segments, wf_extend ms, wf_shade ms and the rates.
"""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="spheres1m")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()

    import torch
    import raytracing_amd as rt
    from raytracing_amd import scenes

    gens = {"spheres1m": ("spheres1m", {}), "cornell": ("cornell", {}), "glass": ("glass", {"grid": 6, "subdiv": 5})}
    gen, kw = gens[a.workload]
    tmp = tempfile.mkdtemp(prefix="rt_prof_")
    path = scenes.generate(gen, os.path.join(tmp, gen + ".gltf"), **kw)
    W, H = a.width, a.height
    scene = rt.Scene(rt.load_gltf(path, W, H), device=0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    buf = torch.empty(W * H * 3, dtype=torch.float32, device=dev)

    def render(depth):
        scene.render_dev(W, H, a.spp, buf.data_ptr(), max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=1,
                         pipeline=rt.PIPE_WAVEFRONT, stream=stream)
        torch.cuda.synchronize()

    render(a.depth)
    total, nodes, tris, nslots, tslots = [], [], [], [], []
    scene.set_counting(True)
    for d in range(1, a.depth + 1):
        scene.reset_counters(stream)
        render(d)
        c = scene.counters()
        total.append(c["segments"])
        nodes.append(c["nodes"])
        tris.append(c["tri_tests"])
        nslots.append(c["node_lane_slots"])
        tslots.append(c["tri_lane_slots"])
    scene.set_counting(False)
    diff = lambda v: [v[0]] + [v[i] - v[i - 1] for i in range(1, len(v))]
    per, pn, pt, pns, pts = diff(total), diff(nodes), diff(tris), diff(nslots), diff(tslots)
    print("bounce  nodes/seg  tris/seg  node-phase lane occupancy  triangle-phase lane occupancy")
    for b in range(a.depth):
        print("%4d %9.2f %9.2f %12.3f %12.3f" % (b, pn[b] / per[b], pt[b] / per[b], pn[b] / max(pns[b], 1), pt[b] / max(pts[b], 1)))
    # per-launch times: one render per bounce count, timing on, difference of the kernel sums
    ext, shd = [], []
    for d in range(1, a.depth + 1):
        best = None
        for _ in range(3):
            scene.set_timing(True)
            render(d)
            kt = scene.kernel_times()
            cur = (kt["extend"][0], kt["shade"][0])
            best = cur if best is None or cur[0] + cur[1] < best[0] + best[1] else best
        ext.append(best[0])
        shd.append(best[1])
    print("bounce  segments   extend_ms  Grays/s   shade_ms  Gvert/s")
    for b in range(a.depth):
        e = ext[b] - (ext[b - 1] if b else 0.0)
        s = shd[b] - (shd[b - 1] if b else 0.0)
        print("%4d %11d %9.3f %8.2f %9.3f %8.2f" % (b, per[b], e, per[b] / e * 1e-6 if e > 0 else 0, s, per[b] / s * 1e-6 if s > 0 else 0))
    print("total %d segments, extend %.2f ms, shade %.2f ms" % (total[-1], ext[-1], shd[-1]))


if __name__ == "__main__":
    main()
