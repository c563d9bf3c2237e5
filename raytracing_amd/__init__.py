"""raytracing_amd -- MI355X (gfx950) implementation of the Monte Carlo path-tracing loop of
sergeyrid/raytracing behind a C ABI (include/rt_amd.h); this package is the Python host side."""
from .api import (PRIMITIVE_DTYPE, PRIM_PLANE, PRIM_ELLIPSOID, PRIM_BOX, build_flat, HostScene, InputData, MultiScene, multi_sample_range, Scene, SceneFloat, counter_seed, device_count, generate_scene, load_gltf,
                  parse_gltf, print_image, selftest_rng, PIPE_MEGAKERNEL, PIPE_WAVEFRONT, RNG_COUNTER, RNG_REFERENCE, MISS_ID,
                  INF, RtError, HITS_LEAF_ORDER, HITS_REFERENCE_ORDER, HITS_REFERENCE_WALK)

__all__ = ["PRIMITIVE_DTYPE", "PRIM_PLANE", "PRIM_ELLIPSOID", "PRIM_BOX", "build_flat", "HostScene", "InputData", "MultiScene", "multi_sample_range", "Scene", "SceneFloat", "counter_seed", "device_count", "generate_scene",
           "load_gltf", "parse_gltf", "print_image", "selftest_rng", "PIPE_MEGAKERNEL", "PIPE_WAVEFRONT", "RNG_COUNTER",
           "RNG_REFERENCE", "MISS_ID", "INF", "RtError", "HITS_LEAF_ORDER", "HITS_REFERENCE_ORDER", "HITS_REFERENCE_WALK"]
