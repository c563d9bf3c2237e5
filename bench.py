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
    ap.add_argument("--multi", type=int, default=0, metavar="N",
                    help="single-process run only: also time the same frame through rt_multi_render (one process, N GPUs, "
                         "RCCL reduce inside the library) and report it beside the per-process path")
    ap.add_argument("--no-primary", action="store_true")
    ap.add_argument("--no-counting", action="store_true",
                    help="skip the counting launch behind the timed region (profiling runs: every launch is then one of the timed ones)")
    args = ap.parse_args()

    # Standard output carries the one JSON line and nothing else: libraries that write to file descriptor 1
    # themselves (RCCL prints a version banner there when its communicator opens) are sent to standard error.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Every rank builds the scene's BVH with OpenMP on the host before the timed region: the ranks of one node share
    # its cores (set before the library and its OpenMP runtime load).
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), (os.cpu_count() or 16) // max(world, 1)))
    os.environ["OMP_NUM_THREADS"] = str(threads)

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd import scenes

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

    # The scene file is written once, by rank 0; the others read it after a barrier.
    if world > 1:
        tmp = os.path.join(tempfile.gettempdir(), "rt_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run")))
        if rank == 0:
            os.makedirs(tmp, exist_ok=True)
            scenes.generate(gen, os.path.join(tmp, gen + ".gltf"), **gen_kw)
        dist.barrier()
        path = os.path.join(tmp, gen + ".gltf")
    else:
        tmp = tempfile.mkdtemp(prefix="rt_bench_")
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

    # ---- roofline ------------------------------------------------------------------------------------
    # Dominant kernel: wf_extend (closest hit of a ray queue) for the wavefront pipeline, k_render_mega for the
    # megakernel; wf_shade is priced beside it.  Per kernel three ceilings, and the one the kernel sits closest
    # to is its bound:
    #   vmem  the CU's vector-memory path: 64-byte L1 accesses per second (TCP_TOTAL_CACHE_ACCESSES) against one
    #         access per cycle and CU at the 2.4 GHz peak clock; the rate a pure dependent-gather microbenchmark
    #         sustains (tools/microbench/node_fetch, table resident in L1) is printed next to it
    #   valu  SQ_INSTS_VALU x 2 cycles (CDNA4's SIMDs are 32 lanes wide: a wave64 instruction issues over 2 cycles)
    #         against 256 CUs x 4 SIMDs x the cycles of the same launches (GRBM_GUI_ACTIVE)
    #   hbm   L2<->fabric bytes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction) per second against the 8 TB/s
    #         HBM peak (an upper bound on HBM proper: Infinity Cache hits are in it)
    # Counters: the committed rocprofv3 --pmc passes of this workload (profiles/pmc_rates.json), summed over the
    # launches of ONE wavefront pass -- the last pass of a run whose sort tuner had settled, i.e. the state the
    # timed steps here run in -- and scaled by the ray segments of a pass; durations: HIP events recorded here
    # around every launch on its stream, summed per pass.  Without a PMC record of the workload the ceilings stay
    # empty and `frac` is null.
    # The ALGORITHMIC bytes (wf_extend: 48 B per four-child node visited, 48 B per ray/triangle test, 44 B of queue
    # traffic per segment; visit counts from the counting build of the same kernels on the same workload, one
    # extra launch outside the timed region) are reported beside them; they are served by L1/L2/Infinity Cache --
    # the 83 MB scene stays on chip -- so their rate is not bounded by HBM.
    count_spp = min(spp_gpu, 4)
    if args.no_counting:   # visits per segment unknown: priced as zero below
        c = {"nodes": 0, "tri_tests": 0, "light_tests": 0, "segments": segs / float(args.steps), "node_lane_slots": 0, "tri_lane_slots": 0}
        count_spp = spp_gpu
    else:
        scene.set_counting(True)
        scene.reset_counters(stream)
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
        passes = max(ktimes["mega"][1], 1)
    else:
        kname, kind = "wf_extend", "extend"
        # wf_extend walks the four-child tree (48-byte nodes) unless RT_WF_WIDE=0 (32-byte two-child nodes);
        # per segment it also reads a 4-byte queue entry and the 32-byte ray and writes the 8-byte hit
        node_bytes = 32.0 if os.environ.get("RT_WF_WIDE") == "0" else 48.0
        bytes_per_seg = node_bytes * n_node + 48.0 * n_tri + 44.0
        passes = max(ktimes["generate"][1], 1)          # one wf_generate per pass of up to 128 frames of samples
    k_ms, k_launches = ktimes[kind]
    launch_ms = k_ms / max(k_launches, 1)
    launch_s = launch_ms * 1e-3
    segs_per_launch = segs / float(max(k_launches, 1))
    segs_per_pass = segs / float(passes)
    algorithmic = segs_per_launch * bytes_per_seg / launch_s / 1e9
    # wf_shade, per path vertex (= per segment: every traced ray is shaded once): 4 B queue entry + 8 B hit + 48 B
    # record read, 32 B of the hit triangle gathered, 52 B written for the paths that go on (all but one vertex per sample)
    shade_bytes = 60.0 + 32.0 + 52.0 * (1.0 - 1.0 / seg_per_sample)

    def ceilings_of(rec, scale, seconds):
        out = {}
        if rec.get("fabric_bytes"):
            out["hbm"] = {"achieved": round(rec["fabric_bytes"] * scale / seconds / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        if "l1_accesses" in rec:
            out["vmem"] = {"achieved": round(rec["l1_accesses"] * scale / seconds / 1e9, 1), "peak": round(256 * 2.4, 1),
                           "unit": "G L1 accesses/s", "peak_source": "theoretical: 256 CUs x 2.4 GHz x one 64-byte L1 access per cycle"}
        if "l1_accesses" in rec and "gui_active_cycles_sum_xcd" in rec and rec.get("duration_ms"):
            # per cycle the chip actually ran (GRBM_GUI_ACTIVE of the same launches: the clock under this kernel is
            # below the 2.4 GHz of the peak), next to what the gather microbenchmark sustains per cycle
            cyc = rec["gui_active_cycles_sum_xcd"] / 8.0
            out["vmem"]["clock_GHz_in_the_pmc_launches"] = round(cyc / (rec["duration_ms"] * 1e-3) / 1e9, 3)
            out["vmem"]["accesses_per_cycle_and_cu"] = round(rec["l1_accesses"] / (256.0 * cyc), 3)
        if "valu_wave_insts" in rec and "gui_active_cycles_sum_xcd" in rec:
            cycles = rec["gui_active_cycles_sum_xcd"] / 8.0     # the counter sums the 8 XCDs
            out["valu"] = {"achieved": round(rec["valu_wave_insts"] * 2.0 / cycles, 1), "peak": 256 * 4.0,
                           "unit": "busy SIMDs (wave64 VALU instruction = 2 cycles on a SIMD-32; instructions and cycles of the same PMC launches)"}
        for v in out.values():
            v["frac"] = round(v["achieved"] / v["peak"], 4)
        return out


    kernels, traffic, pmc_src, gather = {}, None, None, None
    rpath = os.path.join(ROOT, "profiles", "pmc_rates.json")
    if os.path.exists(rpath):
        rates = json.load(open(rpath))
        settled = rates.get("settled", {}).get("kernels", {}) if rates.get("workload") == args.workload else {}
        if settled and rates.get("segments_per_pass"):
            pmc_src = rates["source"] + " (last pass of the run: sort tuner settled)"
            scale = segs_per_pass / rates["segments_per_pass"]
            peak = rates.get("vmem_gather_peak")
            gather = round(peak["lane_loads_per_s"] / 1e9, 1) if peak else None
            for kn, kd, alg_bytes in ((kname, kind, bytes_per_seg), ("wf_shade", "shade", shade_bytes)):
                rec = settled.get(kn)
                if not rec or not ktimes[kd][1]:
                    continue
                pass_s = ktimes[kd][0] / passes * 1e-3
                ce = ceilings_of(rec, scale, pass_s)
                if "vmem" in ce and gather:
                    ce["vmem"]["measured_gather_peak"] = gather
                    ce["vmem"]["frac_of_measured_gather_peak"] = round(ce["vmem"]["achieved"] / gather, 4)
                bnd = max(ce, key=lambda k: ce[k]["frac"]) if ce else None
                kernels[kn] = {"bound": bnd, "frac": ce[bnd]["frac"] if bnd else None, "ceilings": ce,
                               "ms_per_pass": round(pass_s * 1e3, 3), "pmc_ms_per_pass": round(rec.get("duration_ms", 0.0), 3),
                               "traffic_per_pass": round(rec["fabric_bytes"] * scale) if rec.get("fabric_bytes") else None,
                               "algorithmic_bytes_per_segment": round(alg_bytes, 1),
                               "algorithmic_GB/s": round(segs_per_pass * alg_bytes / pass_s / 1e9, 1),
                               "l1_accesses_per_segment": round(rec["l1_accesses"] / rates["segments_per_pass"], 2) if "l1_accesses" in rec else None,
                               "l2_misses_per_segment": round(rec["l2_misses"] / rates["segments_per_pass"], 3) if "l2_misses" in rec else None}
            if kname in kernels and kernels[kname]["traffic_per_pass"]:
                traffic = kernels[kname]["traffic_per_pass"] * passes / float(max(k_launches, 1))    # per launch, like `achieved`
    dom = kernels.get(kname)
    if dom and dom["bound"]:
        top = dom["ceilings"][dom["bound"]]
        bound, achieved, peak_v, unit, frac = dom["bound"], top["achieved"], top["peak"], top["unit"], top["frac"]
    else:   # no committed PMC pass for this workload: only the algorithmic rate is known, which is not an HBM rate
        bound, achieved, peak_v, unit, frac = "hbm", None, HBM_PEAK_GBS, "GB/s", None
    model = None
    mpath = os.path.join(ROOT, "profiles", "r03_bvh_sim.json")
    if args.workload == "spheres1m" and os.path.exists(mpath):
        m = json.load(open(mpath))
        tree = m["four_child_tree_8bit_boxes" if pipeline == rt.PIPE_WAVEFRONT and os.environ.get("RT_WF_WIDE") != "0" else "two_child_tree"]
        model = {"source": "profiles/r03_bvh_sim.json: " + m["tool"],
                 "per_bounce": [{"bounce": r["bounce"], "nodes": r["nodes_per_ray"], "tri_tests": r["tri_tests_per_ray"]} for r in tree]}
    roofline = {"bound": bound, "achieved": achieved, "peak": peak_v, "unit": unit, "frac": frac,
                "traffic": traffic, "pmc_source": pmc_src, "kernel": kname, "kernels": kernels,
                "launch_ms": round(launch_ms, 4), "launches": k_launches, "passes": passes,
                "segments_per_pass": round(segs_per_pass, 1),
                "algorithmic": {"GB/s": round(algorithmic, 1), "of_hbm_peak": round(algorithmic / HBM_PEAK_GBS, 4),
                                "bytes_per_segment": round(bytes_per_seg, 1), "segments_per_launch": round(segs_per_launch, 1),
                                "served_by": "L1 / L2 / Infinity Cache (the scene stays on chip): not an HBM rate"},
                "nodes_per_segment": round(n_node, 2), "tri_tests_per_segment": round(n_tri, 2),
                "cpu_model_of_the_walk": model,
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

    # ---- the same frame through the one-process entry point (rt_multi_*: one host BVH build, RCCL inside) --------
    if args.multi > 0 and world == 1:
        scene.close()
        with rt.MultiScene(host, list(range(args.multi))) as multi:
            frame = None
            for _ in range(max(args.warmup, 1)):
                frame = multi.render(W, H, spp_total, max_depth=depth, seed=seed, pipeline=pipeline)
            tb = time.perf_counter()
            for _ in range(args.steps):
                frame = multi.render(W, H, spp_total, max_depth=depth, seed=seed, pipeline=pipeline)
            multi_s = (time.perf_counter() - tb) / args.steps
            msegs = sum(multi.scene(g).counters()["segments"] for g in range(multi.ndev)) / float(max(args.warmup, 1) + args.steps)
            extra["multi_one_process"] = {
                "n_gpus": multi.ndev, "ms_per_frame": round(multi_s * 1e3, 3), "Mrays_per_s": round(msegs / multi_s * 1e-6, 1),
                "note": "rt_multi_render: samples split over the GPUs, one ncclReduce, result copied to the host (included); "
                        "no tone mapping in the timed region"}

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
    if not (args.multi > 0 and world == 1):
        scene.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
