// raytracing_cli.cpp -- the command-line host.  Keeps the reference's contract
// (main.cpp:1244-1259, run.sh:1):
//
//     raytracing <scene.gltf> <width> <height> <samples> <output.ppm> [options]
//
// writes a binary PPM (P6) and prints the two timing lines of the reference
// ("Processed input after ...s", "Finished after ...s").  Options after the five positionals:
//     --depth N        maximum path depth (default 6, the reference's rayDepth, main.cpp:692)
//     --rng MODE       reference | counter   (default reference: one minstd_rand stream per pixel)
//     --pipeline P     mega | wavefront      (default mega; wavefront needs --rng counter)
//     --seed S         seed of the counter RNG (default 0)
//     --device D       GPU index (default 0)
//     --gpus N         split the samples over GPUs 0..N-1 of this node and sum-reduce over RCCL (needs
//                      --rng counter; N = 1 goes through the same path)
//     --dump-hits F    also write the primary-ray closest hits (uint32 id, float t per pixel) to F
//     --gltf-spec      read the scene by the glTF 2.0 rules the reference does not follow (RT_GLTF_SPEC:
//                      byteStride, index offsets, data URIs, node hierarchy, .glb, KHR_materials_ior)
//     --gltf-background  take the background colour from scenes[scene].extras.backgroundColor
//     --hit-order O    leaf | reference | walk: how hits that depend on the reference's traversal order (equal
//                      distances above all) are resolved (RT_HITS_*; default: reference for scenes the reference
//                      itself can hold, leaf beyond)
// All rendering runs on the GPU through the C ABI of librt_amd.so; there is no CPU fallback.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int die(const char *what) {
    std::fprintf(stderr, "raytracing: %s: %s\n", what, rt_last_error());
    return 1;
}

int main(int argc, char *argv[]) {
    const double start = now();
    if (argc < 6) {
        std::fprintf(stderr,
                     "usage: %s <scene.gltf> <width> <height> <samples> <output.ppm> [--depth N] [--rng reference|counter]"
                     " [--pipeline mega|wavefront] [--seed S] [--device D] [--gpus N] [--dump-hits FILE] [--gltf-spec] [--gltf-background]"
                     " [--hit-order leaf|reference|walk]\n",
                     argv[0]);
        return 2;
    }
    const std::string inputPath = argv[1], outputPath = argv[5];
    const long width = std::atol(argv[2]), height = std::atol(argv[3]), samples = std::atol(argv[4]);
    if (width <= 0 || height <= 0 || samples <= 0 || width > 65535 || height > 65535) {
        std::fprintf(stderr, "raytracing: width, height and samples must be positive (width/height at most 65535)\n");
        return 2;
    }
    uint32_t depth = 6, seed = 0, gltfFlags = 0;
    int rng = RT_RNG_REFERENCE, pipeline = RT_PIPE_MEGAKERNEL, device = 0, gpus = 0, hitOrder = -1;
    std::string dumpHits;
    for (int i = 6; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "raytracing: %s needs a value\n", name);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--depth") depth = (uint32_t)std::atol(need("--depth"));
        else if (a == "--seed") seed = (uint32_t)std::strtoul(need("--seed"), nullptr, 0);
        else if (a == "--device") device = std::atoi(need("--device"));
        else if (a == "--gpus") gpus = std::atoi(need("--gpus"));
        else if (a == "--dump-hits") dumpHits = need("--dump-hits");
        else if (a == "--gltf-spec") gltfFlags |= RT_GLTF_SPEC;
        else if (a == "--gltf-background") gltfFlags |= RT_GLTF_BACKGROUND;
        else if (a == "--rng") {
            std::string v = need("--rng");
            if (v == "reference") rng = RT_RNG_REFERENCE;
            else if (v == "counter") rng = RT_RNG_COUNTER;
            else {
                std::fprintf(stderr, "raytracing: --rng must be reference or counter\n");
                return 2;
            }
        } else if (a == "--hit-order") {
            std::string v = need("--hit-order");
            if (v == "leaf") hitOrder = RT_HITS_LEAF_ORDER;
            else if (v == "reference") hitOrder = RT_HITS_REFERENCE_ORDER;
            else if (v == "walk") hitOrder = RT_HITS_REFERENCE_WALK;
            else {
                std::fprintf(stderr, "raytracing: --hit-order must be leaf, reference or walk\n");
                return 2;
            }
        } else if (a == "--pipeline") {
            std::string v = need("--pipeline");
            if (v == "mega") pipeline = RT_PIPE_MEGAKERNEL;
            else if (v == "wavefront") pipeline = RT_PIPE_WAVEFRONT;
            else {
                std::fprintf(stderr, "raytracing: --pipeline must be mega or wavefront\n");
                return 2;
            }
        } else {
            std::fprintf(stderr, "raytracing: unknown option %s\n", a.c_str());
            return 2;
        }
    }

    rt_host_scene *hs = nullptr;
    if (rt_gltf_load_ex(inputPath.c_str(), (uint32_t)width, (uint32_t)height, gltfFlags, &hs) != RT_OK)
        return die("loading the scene");
    rt_scene_desc desc;
    rt_host_scene_get_desc(hs, &desc);
    rt_scene *scene = nullptr;
    rt_multi *multi = nullptr;
    if (gpus > 0) {
        if (rng != RT_RNG_COUNTER) {
            std::fprintf(stderr, "raytracing: --gpus needs --rng counter (the reference's stream is sequential per pixel)\n");
            return 2;
        }
        if (rt_multi_create(&desc, nullptr, gpus, &multi) != RT_OK) return die("uploading the scene to the GPUs");
        scene = rt_multi_scene(multi, 0);
    } else if (rt_scene_create(&desc, device, &scene) != RT_OK) {
        return die("uploading the scene");
    }
    rt_host_scene_free(hs);
    if (hitOrder >= 0)
        for (int g = 0; g < (multi ? rt_multi_device_count(multi) : 1); ++g)
            if (rt_scene_set_hit_order(multi ? rt_multi_scene(multi, g) : scene, hitOrder) != RT_OK) return die("--hit-order");
    std::printf("Processed input after %gs\n", now() - start);

    const size_t npix = (size_t)width * height;
    if (!dumpHits.empty()) {
        std::vector<uint32_t> ids(npix);
        std::vector<float> ts(npix);
        if (rt_trace_primary(scene, (uint32_t)width, (uint32_t)height, ids.data(), ts.data()) != RT_OK) return die("tracing primary rays");
        FILE *f = std::fopen(dumpHits.c_str(), "wb");
        if (!f) {
            std::fprintf(stderr, "raytracing: cannot open %s\n", dumpHits.c_str());
            return 1;
        }
        std::fwrite(ids.data(), 4, npix, f);
        std::fwrite(ts.data(), 4, npix, f);
        std::fclose(f);
    }

    std::vector<float> sum(npix * 3);
    const double t0 = now();
    if (multi) {
        if (rt_multi_render(multi, (uint32_t)width, (uint32_t)height, 0, (uint32_t)samples, depth, seed, pipeline, sum.data()) != RT_OK)
            return die("rendering");
    } else if (rt_render(scene, (uint32_t)width, (uint32_t)height, 0, (uint32_t)samples, depth, rng, seed, pipeline, sum.data()) != RT_OK) {
        return die("rendering");
    }
    const double renderSeconds = now() - t0;
    std::vector<uint8_t> rgb(npix * 3);
    if (rt_tonemap(scene, sum.data(), npix, (uint32_t)samples, nullptr, rgb.data()) != RT_OK) return die("tone mapping");
    if (rt_write_ppm(outputPath.c_str(), (uint32_t)width, (uint32_t)height, rgb.data()) != RT_OK) return die("writing the image");
    unsigned long long segments = 0;
    for (int g = 0; g < (multi ? rt_multi_device_count(multi) : 1); ++g) {
        rt_counters c;
        if (rt_counters_read(multi ? rt_multi_scene(multi, g) : scene, &c) == RT_OK) segments += c.segments;
    }
    if (renderSeconds > 0)
        std::printf("Traced %llu ray segments in %gs (%.1f Mrays/s)\n", segments, renderSeconds,
                    (double)segments / renderSeconds * 1e-6);
    if (multi) rt_multi_destroy(multi);
    else rt_scene_destroy(scene);
    std::printf("Finished after %gs", now() - start);
    return 0;
}
