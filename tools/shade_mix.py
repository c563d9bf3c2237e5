"""What does the mix of material types inside one wave cost the shading kernel?

    python tools/shade_mix.py            (on a GPU box)

Renders the million-triangle sphere field four times -- the bench's mix (a third each of diffuse, metallic and
dielectric spheres on a diffuse ground), and the same geometry with every surface diffuse, every surface
metallic, every surface dielectric -- and prints the device time of wf_shade and wf_extend per ray segment.
In a single-type scene every lane of a wave takes the same branch of the material switch; the difference
between the mixed scene's time per segment and the mean of the single-type scenes weighted by the mixed scene's
own type counts is what waves sorted by material type could save at most.
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import raytracing_amd as rt  # noqa: E402
from raytracing_amd import scenes  # noqa: E402

W, H, SPP, DEPTH = 1920, 1080, 256, 8
CASES = [
    ("mixed (bench scene)", {}),
    ("all diffuse", dict(glass_fraction=0.0, metal_fraction=0.0, ground_kind=scenes.DIFFUSER)),
    ("all metallic", dict(glass_fraction=0.0, metal_fraction=1.0, ground_kind=scenes.METALLIC)),
    ("all dielectric", dict(glass_fraction=1.0, metal_fraction=0.0, ground_kind=scenes.DIELECTRIC)),
]


def main():
    tmp = tempfile.mkdtemp(prefix="rt_mix_")
    dev = torch.device("cuda", 0)
    buf = torch.empty(W * H * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    rows = []
    for name, kw in CASES:
        path = scenes.generate("spheres1m", os.path.join(tmp, "s.gltf"), **kw)
        scene = rt.Scene(rt.load_gltf(path, W, H), device=0)
        scene.reserve(W, H, SPP, max_depth=DEPTH, rng_mode=rt.RNG_COUNTER, pipeline=rt.PIPE_WAVEFRONT)

        def render():
            scene.render_dev(W, H, SPP, buf.data_ptr(), max_depth=DEPTH, rng_mode=rt.RNG_COUNTER, seed=7,
                             pipeline=rt.PIPE_WAVEFRONT, stream=stream)
        render()
        render()   # four passes behind it: the ray-sorting tuner has decided
        torch.cuda.synchronize()
        scene.reset_counters(stream)
        scene.set_timing(True)
        render()
        torch.cuda.synchronize()
        kt = scene.kernel_times()
        segs = scene.counters()["segments"]
        scene.set_timing(False)
        row = dict(case=name, segments=segs, shade_ms=kt["shade"][0], extend_ms=kt["extend"][0],
                   sort_ms=kt["sort"][0], shade_ns_per_segment=kt["shade"][0] * 1e6 / segs,
                   extend_ns_per_segment=(kt["extend"][0]) * 1e6 / segs, segments_per_sample=segs / float(W * H * SPP))
        rows.append(row)
        print(json.dumps(row), flush=True)
        del scene


if __name__ == "__main__":
    main()
