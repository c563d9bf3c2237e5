// rt_scene.h -- the device-resident scene behind the opaque rt_scene handle, shared by the translation
// units that implement the C ABI (rt_abi.hip: one GPU; rt_multi.hip: several GPUs in one process).
#pragma once
#include <memory>

#include "rt_internal.h"
#include "rt_launch.h"

struct rt_scene {
    int device = 0;
    std::shared_ptr<const rt::FlatScene> flat;  // host copy of the flattened structure (shared by the scenes of an rt_multi)
    rtd::DevScene view{};
    uint4 *d_nodes = nullptr;
    uint4 *d_wide = nullptr;
    float4 *d_tris = nullptr;
    uint32_t *d_lights = nullptr;
    uint4 *d_lnodes = nullptr;
    float4 *d_lrecs = nullptr;
    rtd::DevMaterial *d_mats = nullptr;
    float4 *d_prims = nullptr;  // analytic primitives, 3 x float4 each
    float4 *d_rnodes = nullptr;   // the reference's own tree (FlatScene::ref), when the scene is within its limits
    uint32_t *d_ritems = nullptr;
    float4 *d_rreach = nullptr;   // RefTree::reach
    int hit_order = RT_HITS_LEAF_ORDER;  // RT_HITS_*: how order-dependent hits are resolved (rt_scene_set_hit_order)
    int test_walk = 0;  // RT_TEST_INTERSECT_WALK (tests): 2 pruned, 3 ties
    unsigned long long *d_counters = nullptr;
    rt::WavefrontBuffers wf;
    uint32_t *d_replay = nullptr;  // vertex records of the replay kernel, grown to the largest request
    uint64_t replay_words = 0;
    uint64_t device_bytes = 0;
    bool counting = false;
    bool wide_tree = true;   // wavefront pipeline: wf_extend walks the four-child tree
    int stack_limit = rtd::kStack;
    int sort_rays = 0;       // wavefront pipeline: regroup the rays of the later bounces by cell (0 never, 1 always, 2 adaptive)
    uint32_t sort_min_rays = 1u << 20;
    uint32_t nmat = 0;
    rt::KernelTimer timer;
    uint64_t host_segments = 0;  // segments of launches whose ray count the host knows (one per ray)
};

namespace rt {
// Uploads an already built flattened scene to `device`.  `flat` may be shared between scenes (one host
// BVH build for all GPUs of an rt_multi).  Sets the thread's error message on failure.
int scene_create_from_flat(std::shared_ptr<const FlatScene> flat, const rt_scene_desc &desc, int device, rt_scene **out);
// Host side of rt_scene_create: argument checks + BVH build; maps builder exceptions to RT_ERR_*.
int build_flat_checked(const rt_scene_desc *desc, std::shared_ptr<const FlatScene> *out, const char *who);
}  // namespace rt
