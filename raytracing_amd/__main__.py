"""python -m raytracing_amd <scene.gltf> <width> <height> <samples> <output.ppm> [options]

The reference's command line (main.cpp:1244-1259, run.sh:1) on top of the GPU library, with the
samples split over all ranks when launched under torch.distributed.run (one process per GPU):

    python -m torch.distributed.run --nproc-per-node 8 -m raytracing_amd scene.gltf 1920 1080 1024 out.ppm

Options: --depth N (default 6), --rng reference|counter (default reference on one GPU; several
GPUs need counter), --pipeline mega|wavefront, --seed S, --hit-order leaf|reference|walk (how hits that depend
on the reference's traversal order are resolved; default: reference where the scene is one the reference can hold).
"""
import argparse
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raytracing_amd")
    ap.add_argument("scene")
    ap.add_argument("width", type=int)
    ap.add_argument("height", type=int)
    ap.add_argument("samples", type=int)
    ap.add_argument("output")
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--rng", choices=["reference", "counter"], default=None)
    ap.add_argument("--pipeline", choices=["mega", "wavefront"], default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hit-order", choices=["leaf", "reference", "walk"], default=None)
    a = ap.parse_args(argv)
    start = time.perf_counter()

    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.dist import render_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rng = a.rng or ("counter" if world > 1 else "reference")
    if world > 1 and rng != "counter":
        raise SystemExit("several GPUs need --rng counter (the reference stream is sequential per pixel)")
    pipeline = a.pipeline or ("wavefront" if rng == "counter" else "mega")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    scene = rt.Scene(rt.load_gltf(a.scene, a.width, a.height), device=local_rank)
    if a.hit_order:
        scene.set_hit_order({"leaf": rt.HITS_LEAF_ORDER, "reference": rt.HITS_REFERENCE_ORDER,
                             "walk": rt.HITS_REFERENCE_WALK}[a.hit_order])
    if rank == 0:
        print("Processed input after %gs" % (time.perf_counter() - start))
    npix = a.width * a.height
    buf = torch.empty(npix * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def render_range(begin, count, out):
        scene.render_dev(a.width, a.height, count, out.data_ptr(), max_depth=a.depth,
                         rng_mode=rt.RNG_COUNTER if rng == "counter" else rt.RNG_REFERENCE, seed=a.seed,
                         spp_begin=begin, pipeline=rt.PIPE_WAVEFRONT if pipeline == "wavefront" else rt.PIPE_MEGAKERNEL,
                         stream=stream)

    render_frame(render_range, buf, a.samples)
    if rank == 0:
        rgb = torch.empty(npix * 3, dtype=torch.uint8, device=dev)
        scene.tonemap_dev(buf.data_ptr(), npix, a.samples, out_rgb_ptr=rgb.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        img = rt.SceneFloat(a.width, a.height, None, rgb.cpu().numpy().reshape(-1, 3))
        rt.print_image(img, a.output)
        print("Finished after %gs" % (time.perf_counter() - start), end="")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
