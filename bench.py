#!/usr/bin/env python
"""bench.py -- throughput of the path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--pipeline mega|wavefront]

Workload (default `spheres1m`): the million-triangle mixed-material scene at 1920x1080, path depth 8,
counter RNG, 1024 samples per pixel IN TOTAL -- the frame BASELINE.json's metric is quoted on -- split
evenly over the GPUs (strong scaling: 1024 on one GPU, 128 each on eight).  One step = one whole frame:
every rank renders its sample range of every pixel into its radiance-sum buffer, the buffers are summed
onto rank 0 (RCCL reduce; nothing for N=1) and rank 0 tone-maps the frame.  The scene is resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

For N>1 launch with torch.distributed.run (one process per GPU); RANK/LOCAL_RANK/WORLD_SIZE come from
the environment.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (scene generator name, generator kwargs, W, H, spp of the whole frame, depth)
    "spheres1m": ("spheres1m", {}, 1920, 1080, 1024, 8),
    "cornell": ("cornell", {}, 1920, 1080, 256, 8),
    "glass4k": ("glass", {"grid": 6, "subdiv": 5}, 3840, 2160, 64, 16),
    "emissive": ("emissive", {"subdiv": 4, "lamps": 3}, 1920, 1080, 16, 8),   # 15 360 light triangles
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="spheres1m", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", default="wavefront", choices=["mega", "wavefront"])
    ap.add_argument("--spp", type=int, default=0, help="override the workload's samples per pixel of the whole frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-primary", action="store_true")
    args = ap.parse_args()

    # Standard output carries the one JSON line and nothing else: libraries that write to file descriptor 1
    # themselves (RCCL prints a version banner there when its communicator opens) are sent to standard error.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) does not match WORLD_SIZE (%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1 runs the collective path (RCCL init, reduce, barrier) even with one rank,
    # to rehearse it on a single-GPU box.
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    gen, gen_kw, W, H, spp_total, depth = WORKLOADS[args.workload]
    if args.spp > 0:
        spp_total = args.spp
    if spp_total % world:
        raise SystemExit("the frame's %d samples per pixel do not divide over %d GPUs" % (spp_total, world))
    spp_gpu = spp_total // world
    pipeline = rt.PIPE_MEGAKERNEL if args.pipeline == "mega" else rt.PIPE_WAVEFRONT
    seed = 0x5EED
    npix = W * H

    tmp = tempfile.mkdtemp(prefix="rt_bench_%d_" % rank)
    path = scenes.generate(gen, os.path.join(tmp, gen + ".gltf"), **gen_kw)
    host = rt.load_gltf(path, W, H)
    scene = rt.Scene(host, device=local_rank)
    stats = scene.stats()
    scene.reserve(W, H, spp_gpu, max_depth=depth, rng_mode=rt.RNG_COUNTER, pipeline=pipeline)  # render_dev only enqueues

    stream = torch.cuda.current_stream()
    buf = torch.empty(npix * 3, dtype=torch.float32, device=dev)
    rgb = torch.empty(npix * 3, dtype=torch.uint8, device=dev)

    def step(k=None):
        scene.render_dev(W, H, spp_gpu, buf.data_ptr(), max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=seed,
                         spp_begin=rank * spp_gpu, pipeline=pipeline, stream=stream)
        if use_dist:
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            scene.tonemap_dev(buf.data_ptr(), npix, spp_total, out_rgb_ptr=rgb.data_ptr(), stream=stream)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    scene.reset_counters(stream)
    scene.set_timing(True)  # HIP events around every kernel, on the launch stream
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0

    segs = scene.counters()["segments"]
    ktimes = scene.kernel_times()
    scene.set_timing(False)
    t = torch.tensor([elapsed, float(segs)], dtype=torch.float64, device=dev)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_segs = float(tmax[0]), float(tsum[1])
    else:
        total_segs = float(segs)

    if rank != 0:
        dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel -----------------------------------------------------------
    # Dominant kernel: wf_extend (closest hit of a ray queue) for the wavefront pipeline, k_render_mega
    # for the megakernel.  Three ceilings are priced and the one the kernel sits closest to is reported
    # as roofline.bound / achieved / peak / frac:
    #   vmem  the CU's vector-memory path: 64-byte L1 accesses per second (TCP_TOTAL_CACHE_ACCESSES of the
    #         committed rocprofv3 --pmc pass / the launch time measured here) against one access per cycle and
    #         CU at the 2.4 GHz peak clock (a pure dependent-gather microbenchmark with its table in L1,
    #         tools/microbench/node_fetch A32, sustains 0.85 per cycle; the sorted, partly coherent rays of this
    #         kernel get closer to one than that)
    #   valu  SQ_INSTS_VALU x 2 cycles (CDNA4's SIMDs are 32 lanes wide: a wave64 instruction issues over 2 cycles)
    #         against 256 CUs x 4 SIMDs x the launch's cycles
    #   hbm   measured L2<->fabric bytes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction) per second
    #         against the 8 TB/s HBM peak (an upper bound on HBM proper: Infinity Cache hits are in it)
    # The ALGORITHMIC bytes (48 B per four-child node visited, 48 B per ray/triangle test, 44 B of queue traffic
    # per segment; visit counts from the counting build of the same kernels on the same workload, one
    # extra launch outside the timed region) are reported beside them; they are served by L1/L2/Infinity
    # Cache -- the 83 MB scene stays on chip -- so their rate is not bounded by HBM.
    # Duration: HIP events recorded around every launch of that kernel on its stream during the timed steps.
    scene.set_counting(True)
    scene.reset_counters(stream)
    count_spp = min(spp_gpu, 4)
    scene.render_dev(W, H, count_spp, buf.data_ptr(), max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=seed,
                     spp_begin=0, pipeline=pipeline, stream=stream)
    c = scene.counters()
    scene.set_counting(False)
    n_node = c["nodes"] / c["segments"]
    n_tri = c["tri_tests"] / c["segments"]
    n_light = c["light_tests"] / c["segments"]
    seg_per_sample = c["segments"] / float(npix * count_spp)
    if pipeline == rt.PIPE_MEGAKERNEL:
        kname, kind = "k_render_mega", "mega"
        bytes_per_seg = 32.0 * n_node + 48.0 * n_tri + 32.0 + 64.0 * n_light
    else:
        kname, kind = "wf_extend", "extend"
        # wf_extend walks the four-child tree (48-byte nodes) unless RT_WF_WIDE=0 (32-byte two-child nodes);
        # per segment it also reads a 4-byte queue entry and the 32-byte ray and writes the 8-byte hit
        node_bytes = 32.0 if os.environ.get("RT_WF_WIDE") == "0" else 48.0
        bytes_per_seg = node_bytes * n_node + 48.0 * n_tri + 44.0
    k_ms, k_launches = ktimes[kind]
    launch_ms = k_ms / max(k_launches, 1)
    launch_s = launch_ms * 1e-3
    segs_per_launch = segs / float(max(k_launches, 1))
    algorithmic = segs_per_launch * bytes_per_seg / launch_s / 1e9
    ceilings, traffic, pmc_src = {}, None, None
    rpath = os.path.join(ROOT, "profiles", "pmc_rates.json")
    if os.path.exists(rpath):
        rates = json.load(open(rpath))
        rec = rates["kernels"].get(kname) if rates.get("workload") == args.workload else None
        if rec:
            pmc_src = rates["source"]
            # the counters are per launch of the PMC pass: scale them to this run's launch size
            scale = segs_per_launch / rates["segments_per_launch"] if rates.get("segments_per_launch") else 1.0
            rec = {k: (v * scale if k != "launches" else v) for k, v in rec.items()}
            traffic = rec.get("fabric_bytes")
            if traffic:
                ceilings["hbm"] = {"achieved": round(traffic / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            peak = rates.get("vmem_gather_peak")
            if "l1_accesses" in rec:
                ceilings["vmem"] = {"achieved": round(rec["l1_accesses"] / launch_s / 1e9, 1),
                                    "peak": round(256 * 2.4, 1), "unit": "G L1 accesses/s",
                                    "peak_source": "256 CUs x 2.4 GHz x one 64-byte L1 access per cycle",
                                    "gather_microbenchmark": round(peak["lane_loads_per_s"] / 1e9, 1) if peak else None}
            if "valu_wave_insts" in rec and "gui_active_cycles_sum_xcd" in rec:
                cycles = rec["gui_active_cycles_sum_xcd"] / 8.0     # the counter sums the 8 XCDs
                ceilings["valu"] = {"achieved": round(rec["valu_wave_insts"] * 2.0 / cycles, 1), "peak": 256 * 4.0,
                                    "unit": "busy SIMDs (wave64 VALU instruction = 2 cycles on a SIMD-32; cycles from the PMC pass)"}
    for v in ceilings.values():
        v["frac"] = round(v["achieved"] / v["peak"], 4)
    if ceilings:
        bound = max(ceilings, key=lambda k: ceilings[k]["frac"])
        top = ceilings[bound]
    else:   # no committed PMC pass for this workload: only the algorithmic rate is known
        bound, top = "hbm", {"achieved": round(algorithmic, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(algorithmic / HBM_PEAK_GBS, 5)}
    roofline = {"bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
                "traffic": traffic, "ceilings": ceilings, "pmc_source": pmc_src,
                "kernel": kname, "launch_ms": round(launch_ms, 4), "launches": k_launches,
                "algorithmic": {"GB/s": round(algorithmic, 1), "of_hbm_peak": round(algorithmic / HBM_PEAK_GBS, 4),
                                "bytes_per_segment": round(bytes_per_seg, 1), "segments_per_launch": round(segs_per_launch, 1),
                                "served_by": "L1 / L2 / Infinity Cache (the scene stays on chip): not an HBM rate"},
                "nodes_per_segment": round(n_node, 2), "tri_tests_per_segment": round(n_tri, 2),
                "light_tests_per_segment": round(n_light, 2), "segments_per_sample": round(seg_per_sample, 3),
                "lane_occupancy": {"node_phase": round(c["nodes"] / max(c["node_lane_slots"], 1), 3),
                                   "triangle_phase": round(c["tri_tests"] / max(c["tri_lane_slots"], 1), 3)},
                "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in ktimes.items() if v[1]}}

    # ---- primary-ray closest hit on the 100k-triangle mesh (informational) ----------------------
    extra = {}
    if not args.no_primary and world == 1:
        p2 = scenes.generate("heightfield100k", os.path.join(tmp, "hf100k.gltf"))
        s2 = rt.Scene(rt.load_gltf(p2, 1920, 1080), device=local_rank)
        ids = torch.empty(1920 * 1080, dtype=torch.int32, device=dev)
        ts = torch.empty(1920 * 1080, dtype=torch.float32, device=dev)
        for _ in range(3):
            s2.trace_primary_dev(1920, 1080, ids.data_ptr(), ts.data_ptr(), stream=stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        reps = 20
        for _ in range(reps):
            s2.trace_primary_dev(1920, 1080, ids.data_ptr(), ts.data_ptr(), stream=stream)
        b.record(stream)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        extra["primary_100k_tris_1080p"] = {"ms_per_frame": round(ms, 4), "Mrays_per_s": round(1920 * 1080 / ms * 1e-3, 1),
                                            "ntri": s2.stats()["ntri"]}
        s2.close()

    # ---- CPU baseline: the oracle (a port of the reference's CPU path) on this host's cores ------
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # reported once, by the single-GPU run
        from oracle import oracle
        cw, ch, cspp = W // 2, H // 2, 12   # about 10 s of wall time on 16 host threads for the default workload
        host_small = rt.load_gltf(path, cw, ch).as_dict()
        tb = time.perf_counter()
        orc = oracle.OracleScene(host_small["tri_verts"], host_small["tri_material"], host_small["materials"],
                                 host_small["camera"])
        build_s = time.perf_counter() - tb
        tb = time.perf_counter()
        _, cc = orc.render(cw, ch, cspp, max_depth=depth, rng_mode=1, seed=seed)
        render_s = time.perf_counter() - tb
        cpu = {"value": round(cc["segments"] / render_s * 1e-6, 4), "unit": "Mrays/s", "cores": oracle.num_threads(),
               "kind": "port",
               "sample": "%dx%d, %d spp of the same scene, depth and RNG (render only; oracle BVH build %.1f s)"
                         % (cw, ch, cspp, build_s)}
        orc.close()

    value = total_segs / elapsed * 1e-6
    out = {
        "metric": "Mrays/s (all bounces)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d triangles, %dx%d, %d spp per frame (%d per GPU), depth %d, counter RNG, %s pipeline"
                               % (args.workload, stats["ntri"], W, H, spp_total, spp_gpu, depth, args.pipeline),
                   "parallelism": "spp-split x%d + RCCL reduce" % world if world > 1 else "single GPU",
                   "ms_per_frame": round(elapsed / args.steps * 1e3, 3),
                   "bvh": {"nodes": stats["nnodes"], "height": stats["tree_height"],
                           "height_capped_splits": stats["height_capped_splits"],
                           "build_s": round(stats["build_seconds"], 3)}},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    out.update(extra)
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    scene.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
