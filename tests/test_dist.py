"""World-size-2 test of the multi-GPU path on CPU (gloo): sample split, per-rank partial radiance
sums, sum-reduce onto rank 0.  The per-rank renderer here is the CPU oracle in counter-RNG mode
(test-only stand-in for the GPU launch, which needs a device); the partition and the collective are
the product's (raytracing_amd.dist)."""
import os
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, scene_path, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import gltf_ref, oracle
    from raytracing_amd.dist import render_frame
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    W, H, spp = 32, 24, 5
    sc = gltf_ref.parse_gltf(scene_path, W, H)
    orc = oracle.OracleScene(sc["tri_verts"], sc["tri_material"], sc["materials"], sc["camera"])
    buf = torch.zeros(W * H * 3, dtype=torch.float32)

    def render_range(begin, count, out):
        part, _ = orc.render(W, H, count, max_depth=8, rng_mode=1, seed=17, spp_begin=begin, nthreads=2)
        out.copy_(torch.from_numpy(part.reshape(-1)))

    rng = render_frame(render_range, buf, spp)
    if rank == 0:
        np.save(out_path, buf.numpy())
    assert rng == ((0, 3) if rank == 0 else (3, 2))
    dist.barrier()
    dist.destroy_process_group()


def test_spp_split_and_reduce_world2(oracle_mod, scene_dir):
    import torch.multiprocessing as mp
    from conftest import make_scene
    from oracle import gltf_ref
    path = make_scene(scene_dir, "cornell")
    out = os.path.join(tempfile.mkdtemp(), "sum.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, path, out), nprocs=2, join=True)
    got = np.load(out).reshape(-1, 3)
    sc = gltf_ref.parse_gltf(path, 32, 24)
    orc = oracle_mod.OracleScene(sc["tri_verts"], sc["tri_material"], sc["materials"], sc["camera"])
    full, _ = orc.render(32, 24, 5, max_depth=8, rng_mode=1, seed=17)
    assert np.allclose(got, full, rtol=1e-5, atol=1e-6)
    assert got.sum() > 0


def _gpu_worker(rank, world, port, scene_path, out_path):
    """One rank of the device test: the product's renderer on the GPU (both ranks share the box's one GPU),
    the partial sums brought together over gloo from host copies -- RCCL refuses two ranks on one device."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import raytracing_amd as rt
    from raytracing_amd.dist import render_frame
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    W, H, spp = 96, 72, 7
    scene = rt.Scene(rt.load_gltf(scene_path, W, H), device=0)
    dev = torch.device("cuda", 0)
    dbuf = torch.empty(W * H * 3, dtype=torch.float32, device=dev)
    buf = torch.zeros(W * H * 3, dtype=torch.float32)

    def render_range(begin, count, out):
        scene.render_dev(W, H, count, dbuf.data_ptr(), max_depth=8, rng_mode=rt.RNG_COUNTER, seed=17, spp_begin=begin,
                         pipeline=rt.PIPE_WAVEFRONT, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        out.copy_(dbuf.cpu())

    rng = render_frame(render_range, buf, spp)
    assert rng == ((0, 4) if rank == 0 else (4, 3))
    if rank == 0:
        np.save(out_path, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()
    scene.close()


@pytest.mark.gpu
def test_spp_split_on_the_gpu_world2_is_independent_of_n(scene_dir):
    """Two ranks render 4 + 3 samples of every pixel on the GPU; their reduced sum equals one rank's 7-sample
    render up to the order of the additions: a relative 2 * 2^-24 of a pixel's sum (stated bound: N * 2^-24)."""
    import torch.multiprocessing as mp
    from conftest import make_scene
    import raytracing_amd as rt
    path = make_scene(scene_dir, "cornell")
    out = os.path.join(tempfile.mkdtemp(), "sum_gpu.npy")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(2, port, path, out), nprocs=2, join=True)
    got = np.load(out).reshape(-1, 3)
    scene = rt.Scene(rt.load_gltf(path, 96, 72), device=0)
    one = scene.render(96, 72, 7, max_depth=8, rng_mode=rt.RNG_COUNTER, seed=17, pipeline=rt.PIPE_WAVEFRONT)
    scene.close()
    ok = np.isfinite(one).all(axis=1) & np.isfinite(got).all(axis=1)
    assert ok.mean() > 0.99
    assert np.array_equal(np.isfinite(one), np.isfinite(got))
    # each of the 7 terms is non-negative, so |difference| <= 2 * 2^-24 * sum; 4x margin for the per-sample rounding
    assert np.all(np.abs(got[ok] - one[ok]) <= 8 * 2.0 ** -24 * np.abs(one[ok]) + 1e-30)
    assert got[ok].sum() > 0
