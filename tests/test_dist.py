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
