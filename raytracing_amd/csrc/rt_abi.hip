// rt_abi.hip -- the extern "C" boundary of librt_amd.so (declared in include/rt_amd.h):
// scene upload, argument checks, launches, host-pointer convenience wrappers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>

#include "rt_internal.h"
#include "rt_launch.h"
#include "rt_scene.h"

namespace rt {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
}  // namespace rt

// ---- static scratch of rt_render_noalloc_dev ---------------------------------------------------
// Path records, index queues, queue counters and work cursors of the wavefront pipeline for
// kNoAllocPaths paths in flight, laid out back to back in one array that is part of the library's code
// object: every GPU the library runs on holds its own copy from the first launch on that GPU until the
// process ends (206 MB: 90 bytes per path, and 2^20 frames of the reference-order walker).  Host code must not use the array's name as an address: the address on a
// given GPU comes from hipGetSymbolAddress with that GPU current (scratch_of below).
namespace {
constexpr uint32_t kNoAllocPaths = RT_NOALLOC_MAX_PIXELS;  // one path per pixel at least
constexpr size_t kScratchRay = 0;                                             // float4[2 * paths]
constexpr size_t kScratchLum = kScratchRay + (size_t)kNoAllocPaths * 32;      // float4[paths]
constexpr size_t kScratchC = kScratchLum + (size_t)kNoAllocPaths * 16;        // float4[paths]
constexpr size_t kScratchHit = kScratchC + (size_t)kNoAllocPaths * 16;        // float2[paths]
constexpr size_t kScratchQueue0 = kScratchHit + (size_t)kNoAllocPaths * 8;    // uint32_t[paths]
constexpr size_t kScratchQueue1 = kScratchQueue0 + (size_t)kNoAllocPaths * 4; // uint32_t[paths]
constexpr size_t kScratchRetry = kScratchQueue1 + (size_t)kNoAllocPaths * 4; // uint32_t[2 * paths]
constexpr size_t kScratchBin = kScratchRetry + (size_t)kNoAllocPaths * 8;     // uint16_t[paths]
constexpr size_t kScratchQcount = kScratchBin + (size_t)kNoAllocPaths * 2;    // uint32_t[kWfCountWords] = 256 B
constexpr size_t kScratchCursors = kScratchQcount + 256;                      // uint32_t[kWfCursorWords]
constexpr size_t kScratchBinHist = kScratchCursors + rt::kWfCursorWords * sizeof(uint32_t);  // uint32_t[kWfBinWords]
constexpr size_t kScratchTuner = kScratchBinHist + rt::kWfBinWords * sizeof(uint32_t);             // uint32_t[kWfTuneWords]
constexpr size_t kScratchRefFrames = kScratchTuner + rt::kWfTuneWords * sizeof(uint32_t);  // RefFrame[kNoAllocRefFrames]
constexpr size_t kNoAllocRefFrames = (size_t)1 << 20;                                       // 16 MB: (tree depth + 1) x lanes
constexpr size_t kScratchBytes = kScratchRefFrames + kNoAllocRefFrames * sizeof(rtd::RefFrame);
}  // namespace
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {nullptr};

// Address of `scratch` on `device`, resolved once per device with that device current (the caller
// holds a DeviceGuard).  Called from rt_scene_create, the per-device set-up call.
hipError_t resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    return hipSuccess;
}

unsigned char *scratch_of(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[device];
}

struct DeviceGuard {  // makes the scene's GPU current for the call, restores the caller's on exit
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

int fail(int code, const std::string &msg) {
    rt::set_error(msg);
    return code;
}

int hip_fail(const char *what, hipError_t e) {
    return fail(RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define RT_HIP(call)                                  \
    do {                                              \
        hipError_t e__ = (call);                      \
        if (e__ != hipSuccess) return hip_fail(#call, e__); \
    } while (0)

template <typename T>
hipError_t dev_alloc(rt_scene *s, T **p, size_t count) {
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e == hipSuccess) s->device_bytes += bytes;
    return e;
}

template <typename T>
void dev_free(rt_scene *s, T *&p, size_t count) {
    if (!p) return;
    (void)hipFree(p);
    p = nullptr;
    s->device_bytes -= (count ? count : 1) * sizeof(T);
}

void free_wavefront(rt_scene *s) {
    rt::WavefrontBuffers &w = s->wf;
    const size_t cap = w.capacity;
    dev_free(s, w.ray, cap * 2);
    dev_free(s, w.c, cap);
    dev_free(s, w.lum, cap);
    dev_free(s, w.hit, cap);
    dev_free(s, w.queue[0], cap);
    dev_free(s, w.queue[1], cap);
    dev_free(s, w.bin, cap);
    dev_free(s, w.retry, cap * 2);
    dev_free(s, w.qcount, rt::kWfCountWords);
    dev_free(s, w.cursors, rt::kWfCursorWords);
    dev_free(s, w.tuner, rt::kWfTuneWords);
    dev_free(s, w.binhist, rt::kWfBinWords);
    dev_free(s, w.ref_frames, (size_t)w.ref_lanes * (w.ref_depth + 1));
    w = rt::WavefrontBuffers();
}

// Path records in flight per wavefront pass: whole frames of samples, up to 128 of them or about 256M
// paths (90 B each with the queues: 24 GB of the 288 GB), never more than the call renders.  The
// deeper bounces of a pass keep only a few percent of its paths alive, and every launch pays a ramp
// and a drain of about one ray's duration, so fewer, larger launches win: 16M -> 3424, 32M -> 3656,
// 64M -> 3763, 128M -> 3827 Mrays/s on the million-triangle 1080p depth-8 workload with unsorted rays; with
// sorted rays 64M -> 5110, 128M -> 5229, 256M -> 5328, 512M -> 5342.  RT_WF_PATHS overrides the path budget.
uint32_t wavefront_slots(uint32_t npix, uint32_t spp_count) {
    uint64_t target = 256u << 20;
    if (const char *e = std::getenv("RT_WF_PATHS")) target = std::strtoull(e, nullptr, 0);
    uint64_t slots = target / npix;
    if (slots > 128) slots = 128;
    if (slots > spp_count) slots = spp_count;
    if (slots < 1) slots = 1;
    return (uint32_t)slots;
}

// Makes the scene's path-record buffers hold at least `slots` frames of npix paths.  The buffers only
// ever grow: a smaller request keeps what is there.  Growing waits for the device (the old buffers may
// still be in use by enqueued launches) and allocates.
int ensure_wavefront(rt_scene *s, uint32_t npix, uint32_t slots) {
    const uint64_t cap64 = (uint64_t)slots * npix;
    if (cap64 > 0x7FFFFFFFull) return fail(RT_ERR_LIMIT, "wavefront pipeline: image too large");
    const uint32_t cap = (uint32_t)cap64;
    if (s->wf.capacity >= cap) return RT_OK;
    RT_HIP(hipDeviceSynchronize());
    free_wavefront(s);
    rt::WavefrontBuffers &w = s->wf;
    // capacity first: if an allocation below fails, free_wavefront releases what was allocated with the sizes it
    // was allocated with (and accounted under), and leaves the scene without path records
    w.capacity = cap;
    hipError_t e = hipSuccess;
#define RT_WF(call) \
    if (e == hipSuccess) e = (call)
    RT_WF(dev_alloc(s, &w.ray, (size_t)cap * 2));
    RT_WF(dev_alloc(s, &w.c, (size_t)cap));
    RT_WF(dev_alloc(s, &w.lum, (size_t)cap));
    RT_WF(dev_alloc(s, &w.hit, (size_t)cap));
    RT_WF(dev_alloc(s, &w.queue[0], (size_t)cap));
    RT_WF(dev_alloc(s, &w.queue[1], (size_t)cap));
    RT_WF(dev_alloc(s, &w.bin, (size_t)cap));
    RT_WF(dev_alloc(s, &w.retry, (size_t)cap * 2));
    RT_WF(dev_alloc(s, &w.binhist, rt::kWfBinWords));
    RT_WF(hipMemset(w.binhist, 0, rt::kWfBinWords * sizeof(uint32_t)));
    RT_WF(dev_alloc(s, &w.qcount, rt::kWfCountWords));
    RT_WF(hipMemset(w.qcount, 0, rt::kWfCountWords * sizeof(uint32_t)));
    RT_WF(dev_alloc(s, &w.cursors, rt::kWfCursorWords));
    RT_WF(hipMemset(w.cursors, 0, rt::kWfCursorWords * sizeof(uint32_t)));
    RT_WF(dev_alloc(s, &w.tuner, rt::kWfTuneWords));
    RT_WF(hipMemset(w.tuner, 0, rt::kWfTuneWords * sizeof(uint32_t)));
    if (s->flat->ref.available) {  // frames of the reference-order walk of wf_extend_retry_ref
        w.ref_depth = s->flat->ref.depth;
        w.ref_lanes = rt::kWfRefLanes;
        RT_WF(dev_alloc(s, &w.ref_frames, (size_t)w.ref_lanes * (w.ref_depth + 1)));
    }
#undef RT_WF
    if (e != hipSuccess) {
        free_wavefront(s);
        return hip_fail("wavefront pipeline: allocating the path records", e);
    }
    w.capacity = cap;
    w.sort = s->sort_rays;
    w.sort_min_rays = s->sort_min_rays;
    w.wide = s->wide_tree;
    w.ties = s->hit_order != RT_HITS_LEAF_ORDER;
    w.stack_limit = s->stack_limit;
    return RT_OK;
}

int ensure_replay(rt_scene *s, const rt::RenderParams &rp) {
    const uint64_t words = rt::replay_words(rp);
    if (words <= s->replay_words) return RT_OK;
    RT_HIP(hipDeviceSynchronize());
    dev_free(s, s->d_replay, (size_t)s->replay_words);
    s->replay_words = 0;
    RT_HIP(dev_alloc(s, &s->d_replay, (size_t)words));
    s->replay_words = words;
    return RT_OK;
}

constexpr size_t kCounterWords = (size_t)rtd::kCounterShards * rtd::kCounterStride;

int check_image(uint32_t W, uint32_t H) {
    if (W == 0 || H == 0) return fail(RT_ERR_INVALID, "image width and height must be positive");
    if ((uint64_t)W * H > (1ull << 28)) return fail(RT_ERR_LIMIT, "image has more than 2^28 pixels");
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return rt::g_error.c_str(); }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint32_t rt_counter_seed(uint32_t pixel, uint32_t sample, uint32_t seed) { return rtd::counter_seed(pixel, sample, seed); }

int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out) {
    if (!desc || !out) return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    *out = nullptr;
    int ndev = rt_device_count();
    if (ndev <= 0) return fail(RT_ERR_HIP, "rt_scene_create: no HIP device is visible");
    if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID, "rt_scene_create: device index out of range");
    std::shared_ptr<const rt::FlatScene> flat;
    int rc = rt::build_flat_checked(desc, &flat, "rt_scene_create");
    if (rc != RT_OK) return rc;
    return rt::scene_create_from_flat(flat, *desc, device, out);
}

}  // extern "C"

namespace rt {

int build_flat_checked(const rt_scene_desc *desc, std::shared_ptr<const FlatScene> *out, const char *who) {
    auto flat = std::make_shared<FlatScene>();
    try {
        build_flat_scene(*desc, *flat);
    } catch (const std::exception &e) {
        std::string msg = e.what();
        return fail(msg.find("too many") != std::string::npos ? RT_ERR_LIMIT : RT_ERR_INVALID, std::string(who) + ": " + msg);
    }
    if (flat->tree_height > kStackEntries)
        return fail(RT_ERR_LIMIT, std::string(who) + ": BVH height exceeds the kernels' traversal stack");
    *out = flat;
    return RT_OK;
}

int scene_create_from_flat(std::shared_ptr<const FlatScene> flat, const rt_scene_desc &descRef, int device, rt_scene **out) {
    const rt_scene_desc *desc = &descRef;
    *out = nullptr;
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_create: out of host memory");
    s->device = device;
    s->flat = flat;
    DeviceGuard g(device);
    if (!g.ok) {
        delete s;
        return fail(RT_ERR_HIP, "rt_scene_create: hipSetDevice failed");
    }
    {
        hipError_t es = resolve_scratch(device);  // the static scratch of rt_render_noalloc_dev, once per GPU
        if (es != hipSuccess) {
            delete s;
            return hip_fail("rt_scene_create: hipGetSymbolAddress(scratch)", es);
        }
    }
    const rt::FlatScene &f = *s->flat;
    std::vector<rtd::DevMaterial> mats(desc->nmat ? desc->nmat : 1);
    for (uint32_t i = 0; i < desc->nmat; ++i) {
        const rt_material &m = desc->materials[i];
        if (m.type < 0 || m.type > 2) {
            delete s;
            return fail(RT_ERR_INVALID, "rt_scene_create: material type must be 0, 1 or 2");
        }
        float tbits;
        int32_t ty = m.type;
        std::memcpy(&tbits, &ty, 4);
        mats[i].a = make_float4(m.color[0], m.color[1], m.color[2], tbits);
        mats[i].b = make_float4(m.emission[0], m.emission[1], m.emission[2], m.ior);
    }
    s->nmat = desc->nmat;
    hipError_t e;
#define RT_TRY(call)                       \
    if ((e = (call)) != hipSuccess) {      \
        int rc = hip_fail(#call, e);       \
        rt_scene_destroy(s);               \
        return rc;                         \
    }
    RT_TRY(dev_alloc(s, &s->d_nodes, f.nodes.size() / 4));
    RT_TRY(dev_alloc(s, &s->d_wide, f.wide.size() / 4));
    RT_TRY(hipMemcpy(s->d_wide, f.wide.data(), f.wide.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_TRY(dev_alloc(s, &s->d_tris, f.tris.size() / 4));
    RT_TRY(dev_alloc(s, &s->d_lights, f.lights.size()));
    RT_TRY(dev_alloc(s, &s->d_lnodes, f.light_nodes.size() / 4));
    RT_TRY(dev_alloc(s, &s->d_lrecs, f.light_recs.size() / 4));
    RT_TRY(dev_alloc(s, &s->d_mats, mats.size()));
    RT_TRY(dev_alloc(s, &s->d_prims, f.prims.size() / 4));
    RT_TRY(hipMemcpy(s->d_prims, f.prims.data(), f.prims.size() * sizeof(float), hipMemcpyHostToDevice));
    if (f.ref.available) {
        RT_TRY(dev_alloc(s, &s->d_rnodes, f.ref.nodes.size() / 4));
        RT_TRY(hipMemcpy(s->d_rnodes, f.ref.nodes.data(), f.ref.nodes.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        RT_TRY(dev_alloc(s, &s->d_ritems, f.ref.items.size()));
        RT_TRY(hipMemcpy(s->d_ritems, f.ref.items.data(), f.ref.items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        RT_TRY(dev_alloc(s, &s->d_rreach, f.ref.reach.size() / 4));
        RT_TRY(hipMemcpy(s->d_rreach, f.ref.reach.data(), f.ref.reach.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    RT_TRY(dev_alloc(s, &s->d_counters, kCounterWords));
    RT_TRY(hipMemcpy(s->d_nodes, f.nodes.data(), f.nodes.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(s->d_tris, f.tris.data(), f.tris.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!f.lights.empty())
        RT_TRY(hipMemcpy(s->d_lights, f.lights.data(), f.lights.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(s->d_lnodes, f.light_nodes.data(), f.light_nodes.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(s->d_lrecs, f.light_recs.data(), f.light_recs.size() * sizeof(float), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(s->d_mats, mats.data(), mats.size() * sizeof(rtd::DevMaterial), hipMemcpyHostToDevice));
    RT_TRY(hipMemset(s->d_counters, 0, kCounterWords * sizeof(unsigned long long)));
#undef RT_TRY
    rtd::DevScene &v = s->view;
    v.nodes = s->d_nodes;
    v.tris = s->d_tris;
    v.wide = s->d_wide;
    v.lights = s->d_lights;
    v.lnodes = s->d_lnodes;
    v.lrecs = s->d_lrecs;
    {
        // LightSurface::pdf loops over the lights (cached in LDS) below this count and walks the light tree from
        // it on.  Measured crossover (emissive icospheres, 1080p): 20 lights 26.8 vs 32.2 ms, 40: 39.7 vs 39.5,
        // 60: 48.3 vs 43.4, 240: 108 vs 54, 960: 321 vs 65 (loop vs tree).
        const char *e = std::getenv("RT_LIGHT_TREE_MIN");
        const long minLights = e ? std::atol(e) : 48;
        v.light_tree = (long)f.ntrilights >= minLights && f.ntrilights ? std::max<uint32_t>(f.light_height, 1u) : 0u;
        // wf_shade has an instantiation with a short stack for light trees of at most 16 levels;
        // RT_LIGHT_STACK_FULL=1 (tests) reports a deep tree and so selects the general one
        if (v.light_tree && std::getenv("RT_LIGHT_STACK_FULL")) v.light_tree = std::max<uint32_t>(v.light_tree, rtd::kStack);
    }
    v.mats = s->d_mats;
    v.prims = s->d_prims;
    v.rnodes = s->d_rnodes;
    v.ritems = s->d_ritems;
    v.rreach = s->d_rreach;
    {
        // Results that depend on the reference's traversal order (equal distances above all: coincident faces, a
        // box standing on a floor) are resolved as the reference resolves them whenever the scene is one the
        // reference itself can hold; RT_HIT_ORDER=leaf keeps the kernels' own rule (rt_scene_set_hit_order).
        s->hit_order = f.ref.available ? RT_HITS_REFERENCE_ORDER : RT_HITS_LEAF_ORDER;
        if (const char *e = std::getenv("RT_HIT_ORDER"))
            if (std::strcmp(e, "leaf") == 0) s->hit_order = RT_HITS_LEAF_ORDER;
        // tests: rt_intersect_rays in reference order answers with the pruned form of the reference's walk
        // ("pruned"), or the way the throughput kernels do: fast walk, pruned walk where hits lay close ("ties")
        if (const char *e = std::getenv("RT_TEST_INTERSECT_WALK"))
            s->test_walk = std::strcmp(e, "pruned") == 0 ? 2 : std::strcmp(e, "ties") == 0 ? 3 : 0;
    }
    v.ntri = f.nslots;
    v.npad = f.nslots - f.ntri;
    v.nprims = f.nprims;
    v.counters = s->d_counters;
    v.nlights = (uint32_t)f.lights.size();
    v.ntrilights = f.ntrilights;
    v.nmats = (uint32_t)mats.size();
    for (int k = 0; k < 3; ++k) {
        v.grid_origin[k] = f.grid_origin[k];
        v.grid_cell[k] = f.grid_cell[k];
        v.lgrid_origin[k] = f.light_grid_origin[k];
        v.lgrid_cell[k] = f.light_grid_cell[k];
        v.cam_pos[k] = desc->camera.position[k];
        v.cam_right[k] = desc->camera.right[k];
        v.cam_up[k] = desc->camera.up[k];
        v.cam_fwd[k] = desc->camera.forward[k];
    }
    v.tan_x = desc->camera.tan_x;
    v.tan_y = desc->camera.tan_y;
    for (int k = 0; k < 3; ++k) v.bin_scale[k] = (float)(1u << rtd::kBinBits) / (65000.f * f.grid_cell[k]);
    {
        // Regrouping a bounce's rays by the cell they start in pays where two things hold: the tree and its
        // triangles do not fit the L2 caches (32 MB over the eight XCDs), and the rays have lost the coherence of
        // the pixel tiles they started in.  The first is known here; the second differs from scene to scene and
        // bounce to bounce (from the fourth bounce on the million-triangle scene, from the seventh on the glass
        // scene) and is found out by measuring on the device (mode 2: the tuner of rt_kernels.hip).
        // RT_WF_SORT=0 / 1 / 2 forces never / every bounce / tuned; RT_WF_SORT_MIN_RAYS lowers the queue length
        // below which mode 2 never sorts (tests).
        const size_t tree_bytes = (size_t)f.nwide * 48u + (size_t)f.nslots * 64u;
        s->sort_rays = tree_bytes >= ((size_t)32 << 20) ? 2 : 0;
        if (const char *e = std::getenv("RT_WF_SORT")) {
            const int m = std::atoi(e);
            s->sort_rays = m < 0 ? 0 : (m > 2 ? 2 : m);
        }
        if (const char *e = std::getenv("RT_WF_SORT_MIN_RAYS")) s->sort_min_rays = (uint32_t)std::strtoul(e, nullptr, 10);
        // RT_WF_WIDE=0: wf_extend walks the two-child tree; RT_WF_STACK_LIMIT=n (tests): the wide walk gives up
        // beyond n stack entries, which sends rays through the retry kernel
        if (const char *e = std::getenv("RT_WF_WIDE")) s->wide_tree = std::atoi(e) != 0;
        if (const char *e = std::getenv("RT_WF_STACK_LIMIT")) {
            const int n = std::atoi(e);
            s->stack_limit = n < 0 ? 0 : (n > rtd::kStack ? rtd::kStack : n);
        }
    }
    // no walk visits a node or a leaf twice: nodes + leaves bounds the rounds and visits of one ray's walk (the wide
    // walk's retry included); four times that is the counting launches' alarm (TraceStats::runaway)
    v.far_c[0] = v.far_c[1] = v.far_c[2] = 0.f;
    v.far_r = 3.0e38f;
    if (f.ref.available && f.ref.nnodes) {   // root of RefTree::reach: everything finite in the scene
        const float *q = f.ref.reach.data();
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) {
            if (!(q[a] <= q[3 + a])) continue;   // nothing finite on this axis (a scene of planes)
            v.far_c[a] = 0.5f * (q[a] + q[3 + a]);
            ext = std::max(ext, q[3 + a] - q[a]);
        }
        if (ext > 0.f) v.far_r = 4.f * ext;
    }
    v.step_bound = 4u * (f.nnodes + f.nslots + f.light_nnodes) + 64u;
    if (const char *e = std::getenv("RT_TEST_STEP_BOUND")) v.step_bound = (uint32_t)std::strtoul(e, nullptr, 10);  // tests: trip the alarm
    v.has_background = 0u;
    for (int k = 0; k < 3; ++k) {
        v.background[k] = desc->background[k];
        if (desc->background[k] != 0.f) v.has_background = 1u;
    }
    *out = s;
    return RT_OK;
}

}  // namespace rt

extern "C" {

void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    DeviceGuard g(s->device);
    free_wavefront(s);
    dev_free(s, s->d_replay, (size_t)s->replay_words);
    s->timer.release();
    (void)hipFree(s->d_nodes);
    (void)hipFree(s->d_tris);
    (void)hipFree(s->d_wide);
    (void)hipFree(s->d_lights);
    (void)hipFree(s->d_lnodes);
    (void)hipFree(s->d_lrecs);
    (void)hipFree(s->d_mats);
    (void)hipFree(s->d_prims);
    (void)hipFree(s->d_rnodes);
    (void)hipFree(s->d_ritems);
    (void)hipFree(s->d_rreach);
    (void)hipFree(s->d_counters);
    delete s;
}

int rt_scene_get_stats(const rt_scene *s, rt_scene_stats *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_get_stats: null argument");
    out->ntri = s->flat->ntri;
    out->ntri_records = s->flat->nslots;
    out->nlights = (uint32_t)s->flat->lights.size();
    out->nnodes = s->flat->nnodes;
    out->tree_height = s->flat->tree_height;
    out->build_seconds = s->flat->build_seconds;
    out->device_bytes = s->device_bytes;
    out->height_capped_splits = s->flat->capped_splits;
    out->nprims = s->flat->nprims;
    out->hit_order = (uint32_t)s->hit_order;
    out->ref_tree_nodes = s->flat->ref.available ? s->flat->ref.nnodes : 0u;
    out->ref_tree_depth = s->flat->ref.available ? s->flat->ref.depth : 0u;
    return RT_OK;
}

int rt_scene_set_hit_order(rt_scene *s, int order) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_hit_order: null scene");
    if (order != RT_HITS_LEAF_ORDER && order != RT_HITS_REFERENCE_ORDER && order != RT_HITS_REFERENCE_WALK)
        return fail(RT_ERR_INVALID, "rt_scene_set_hit_order: order must be one of RT_HITS_*");
    if (order != RT_HITS_LEAF_ORDER && !s->flat->ref.available)
        return fail(RT_ERR_LIMIT, "rt_scene_set_hit_order: the reference's own tree is not available for this scene: " +
                                      s->flat->ref.reason);
    s->hit_order = order;
    s->wf.ties = order != RT_HITS_LEAF_ORDER;
    return RT_OK;
}

int rt_scene_get_flat(const rt_scene *s, float grid[6], uint32_t *nodes, float *tris) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_get_flat: null scene");
    if (grid)
        for (int a = 0; a < 3; ++a) {
            grid[a] = s->flat->grid_origin[a];
            grid[3 + a] = s->flat->grid_cell[a];
        }
    if (nodes) std::memcpy(nodes, s->flat->nodes.data(), (size_t)s->flat->nnodes * 8 * sizeof(uint32_t));
    if (tris) std::memcpy(tris, s->flat->tris.data(), (size_t)s->flat->nslots * 16 * sizeof(float));
    return RT_OK;
}

int rt_intersect_rays_dev(rt_scene *s, uint64_t n, const float *d_origins, const float *d_directions,
                          uint32_t *d_prim_id, float *d_t, void *stream) {
    if (!s) return fail(RT_ERR_INVALID, "rt_intersect_rays_dev: null scene");
    if (n == 0) return RT_OK;
    if (!d_origins || !d_directions || !d_prim_id || !d_t) return fail(RT_ERR_INVALID, "rt_intersect_rays_dev: null buffer");
    if (n > (1ull << 31)) return fail(RT_ERR_LIMIT, "rt_intersect_rays_dev: more than 2^31 rays in one call");
    DeviceGuard g(s->device);
    RT_HIP(rt::launch_intersect(s->view, n, d_origins, d_directions, d_prim_id, d_t, s->counting,
                                s->hit_order == RT_HITS_LEAF_ORDER ? 0 : (s->test_walk ? s->test_walk : 1), (hipStream_t)stream, s->timer));
    s->host_segments += n;
    return RT_OK;
}

int rt_intersect_rays(rt_scene *s, uint64_t n, const float *origins, const float *directions, uint32_t *prim_id,
                      float *t) {
    if (!s) return fail(RT_ERR_INVALID, "rt_intersect_rays: null scene");
    if (n == 0) return RT_OK;
    if (!origins || !directions || !prim_id || !t) return fail(RT_ERR_INVALID, "rt_intersect_rays: null buffer");
    DeviceGuard g(s->device);
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    uint32_t *d_id = nullptr;
    int rc = RT_OK;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc((void **)&d_o, n * 12)) == hipSuccess && (e = hipMalloc((void **)&d_d, n * 12)) == hipSuccess &&
        (e = hipMalloc((void **)&d_t, n * 4)) == hipSuccess && (e = hipMalloc((void **)&d_id, n * 4)) == hipSuccess &&
        (e = hipMemcpy(d_o, origins, n * 12, hipMemcpyHostToDevice)) == hipSuccess &&
        (e = hipMemcpy(d_d, directions, n * 12, hipMemcpyHostToDevice)) == hipSuccess) {
        rc = rt_intersect_rays_dev(s, n, d_o, d_d, d_id, d_t, nullptr);
        if (rc == RT_OK && (e = hipDeviceSynchronize()) == hipSuccess &&
            (e = hipMemcpy(prim_id, d_id, n * 4, hipMemcpyDeviceToHost)) == hipSuccess)
            e = hipMemcpy(t, d_t, n * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_o);
    (void)hipFree(d_d);
    (void)hipFree(d_t);
    (void)hipFree(d_id);
    if (e != hipSuccess) return hip_fail("rt_intersect_rays", e);
    return rc;
}

int rt_trace_primary_dev(rt_scene *s, uint32_t width, uint32_t height, uint32_t *d_prim_id, float *d_t, void *stream) {
    if (!s || !d_prim_id || !d_t) return fail(RT_ERR_INVALID, "rt_trace_primary_dev: null argument");
    int rc = check_image(width, height);
    if (rc != RT_OK) return rc;
    DeviceGuard g(s->device);
    RT_HIP(rt::launch_primary(s->view, width, height, d_prim_id, d_t, s->counting, s->hit_order != RT_HITS_LEAF_ORDER,
                              (hipStream_t)stream, s->timer));
    s->host_segments += (uint64_t)width * height;
    return RT_OK;
}

int rt_trace_primary(rt_scene *s, uint32_t width, uint32_t height, uint32_t *prim_id, float *t) {
    if (!s || !prim_id || !t) return fail(RT_ERR_INVALID, "rt_trace_primary: null argument");
    int rc = check_image(width, height);
    if (rc != RT_OK) return rc;
    DeviceGuard g(s->device);
    const size_t n = (size_t)width * height;
    uint32_t *d_id = nullptr;
    float *d_t = nullptr;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc((void **)&d_id, n * 4)) == hipSuccess && (e = hipMalloc((void **)&d_t, n * 4)) == hipSuccess) {
        rc = rt_trace_primary_dev(s, width, height, d_id, d_t, nullptr);
        if (rc == RT_OK && (e = hipDeviceSynchronize()) == hipSuccess &&
            (e = hipMemcpy(prim_id, d_id, n * 4, hipMemcpyDeviceToHost)) == hipSuccess)
            e = hipMemcpy(t, d_t, n * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_id);
    (void)hipFree(d_t);
    if (e != hipSuccess) return hip_fail("rt_trace_primary", e);
    return rc;
}

int rt_render_dev(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count,
                  uint32_t max_depth, int rng_mode, uint32_t seed, int pipeline, float *d_radiance_sum, void *stream) {
    if (!s || !d_radiance_sum) return fail(RT_ERR_INVALID, "rt_render_dev: null argument");
    int rc = check_image(width, height);
    if (rc != RT_OK) return rc;
    if (rng_mode != RT_RNG_REFERENCE && rng_mode != RT_RNG_COUNTER) return fail(RT_ERR_INVALID, "rt_render_dev: unknown rng_mode");
    if (rng_mode == RT_RNG_REFERENCE && spp_begin != 0)
        return fail(RT_ERR_INVALID, "rt_render_dev: the reference RNG stream is carried across a pixel's samples, spp_begin must be 0");
    if (pipeline != RT_PIPE_MEGAKERNEL && pipeline != RT_PIPE_WAVEFRONT) return fail(RT_ERR_INVALID, "rt_render_dev: unknown pipeline");
    if (pipeline == RT_PIPE_WAVEFRONT && rng_mode != RT_RNG_COUNTER)
        return fail(RT_ERR_INVALID, "rt_render_dev: the wavefront pipeline needs RT_RNG_COUNTER (samples of a pixel run concurrently)");
    if (max_depth > 65535u) return fail(RT_ERR_LIMIT, "rt_render_dev: max_depth above 65535");
    if ((uint64_t)spp_begin + spp_count > 0xFFFFFFFFull) return fail(RT_ERR_LIMIT, "rt_render_dev: sample range overflows 32 bits");
    DeviceGuard g(s->device);
    rt::RenderParams rp{width, height, spp_begin, spp_count, max_depth, rng_mode, seed};
    const bool ref_order = s->hit_order != RT_HITS_LEAF_ORDER;
    if (pipeline == RT_PIPE_MEGAKERNEL && rng_mode == RT_RNG_REFERENCE) {
        // the reference's stream in the reference's evaluation order (bit-exact validation path)
        if (max_depth > 64) return fail(RT_ERR_LIMIT, "rt_render_dev: RT_RNG_REFERENCE supports max_depth <= 64");
        rc = ensure_replay(s, rp);
        if (rc != RT_OK) return rc;
        RT_HIP(rt::launch_render_replay(s->view, rp, d_radiance_sum, s->d_replay, s->counting, ref_order, (hipStream_t)stream, s->timer));
    } else if (pipeline == RT_PIPE_MEGAKERNEL) {
        const int order = s->hit_order == RT_HITS_REFERENCE_WALK    ? rtd::kOrderRefFull
                          : s->hit_order == RT_HITS_REFERENCE_ORDER ? rtd::kOrderRefTies
                                                                    : rtd::kOrderLeaf;
        RT_HIP(rt::launch_render_mega(s->view, rp, d_radiance_sum, s->counting, order, (hipStream_t)stream, s->timer));
    } else {
        const uint32_t slots = wavefront_slots(width * height, spp_count);
        rc = ensure_wavefront(s, width * height, slots);
        if (rc != RT_OK) return rc;
        s->wf.ties = ref_order;
        s->wf.ref_all = s->hit_order == RT_HITS_REFERENCE_WALK;
        RT_HIP(rt::launch_render_wavefront(s->view, rp, d_radiance_sum, s->wf, slots, s->counting, (hipStream_t)stream, s->timer));
    }
    return RT_OK;
}

int rt_render_noalloc_dev(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count,
                          uint32_t max_depth, uint32_t seed, float *d_radiance_sum, void *stream) {
    if (!s || !d_radiance_sum) return fail(RT_ERR_INVALID, "rt_render_noalloc_dev: null argument");
    int rc = check_image(width, height);
    if (rc != RT_OK) return rc;
    const uint64_t npix64 = (uint64_t)width * height;
    if (npix64 > kNoAllocPaths)
        return fail(RT_ERR_LIMIT, "rt_render_noalloc_dev: image has " + std::to_string(npix64) +
                                      " pixels, the static scratch holds path records for at most " +
                                      std::to_string(kNoAllocPaths) + " (use rt_render_dev)");
    if (max_depth > 65535u) return fail(RT_ERR_LIMIT, "rt_render_noalloc_dev: max_depth above 65535");
    if ((uint64_t)spp_begin + spp_count > 0xFFFFFFFFull)
        return fail(RT_ERR_LIMIT, "rt_render_noalloc_dev: sample range overflows 32 bits");
    unsigned char *base = scratch_of(s->device);
    if (!base) return fail(RT_ERR_HIP, "rt_render_noalloc_dev: the static scratch was not resolved on this device");
    DeviceGuard g(s->device);
    const uint32_t npix = (uint32_t)npix64;
    uint32_t slots = kNoAllocPaths / npix;
    if (slots > 64) slots = 64;
    if (slots > spp_count) slots = spp_count;
    if (slots < 1) slots = 1;
    rt::WavefrontBuffers wb;
    wb.capacity = kNoAllocPaths;
    wb.ray = reinterpret_cast<float4 *>(base + kScratchRay);
    wb.c = reinterpret_cast<float4 *>(base + kScratchC);
    wb.lum = reinterpret_cast<float4 *>(base + kScratchLum);
    wb.hit = reinterpret_cast<float2 *>(base + kScratchHit);
    wb.queue[0] = reinterpret_cast<uint32_t *>(base + kScratchQueue0);
    wb.queue[1] = reinterpret_cast<uint32_t *>(base + kScratchQueue1);
    wb.bin = reinterpret_cast<uint16_t *>(base + kScratchBin);
    wb.retry = reinterpret_cast<uint32_t *>(base + kScratchRetry);
    wb.binhist = reinterpret_cast<uint32_t *>(base + kScratchBinHist);
    wb.sort = s->sort_rays;
    wb.sort_min_rays = s->sort_min_rays;
    wb.wide = s->wide_tree;
    wb.ties = s->hit_order != RT_HITS_LEAF_ORDER;
    wb.ref_all = s->hit_order == RT_HITS_REFERENCE_WALK;
    wb.ref_depth = s->flat->ref.depth;
    wb.ref_frames = reinterpret_cast<rtd::RefFrame *>(base + kScratchRefFrames);
    wb.ref_lanes = (uint32_t)(kNoAllocRefFrames / (wb.ref_depth + 1) / rtd::kBlock) * rtd::kBlock;   // at least 64 blocks: depth <= 64
    if (wb.ref_lanes > rt::kWfRefLanes) wb.ref_lanes = rt::kWfRefLanes;
    wb.stack_limit = s->stack_limit;
    wb.qcount = reinterpret_cast<uint32_t *>(base + kScratchQcount);
    wb.cursors = reinterpret_cast<uint32_t *>(base + kScratchCursors);
    wb.tuner = reinterpret_cast<uint32_t *>(base + kScratchTuner);
    // the array keeps the previous call's contents: counters, cursors and the cell histogram start from zero, the
    // records and queues are rewritten by wf_generate / wf_shade before they are read.  The sort tuner's books
    // (zero when the code object is loaded) are kept from call to call like a scene's own are, so that a loop of
    // short calls reaches a decision; they are per GPU here, not per scene: after a change of scene the running
    // means take a few passes to follow (a matter of speed only, the sums do not depend on what is sorted).
    RT_HIP(hipMemsetAsync(base + kScratchQcount, 0, kScratchTuner - kScratchQcount, (hipStream_t)stream));
    rt::RenderParams rp{width, height, spp_begin, spp_count, max_depth, RT_RNG_COUNTER, seed};
    rt::KernelTimer untimed;  // event pairs of rt_set_timing would create events: not here
    RT_HIP(rt::launch_render_wavefront(s->view, rp, d_radiance_sum, wb, slots, s->counting, (hipStream_t)stream, untimed));
    return RT_OK;
}

int rt_scene_reserve(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_count, uint32_t max_depth, int rng_mode,
                     int pipeline) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_reserve: null scene");
    int rc = check_image(width, height);
    if (rc != RT_OK) return rc;
    DeviceGuard g(s->device);
    if (pipeline == RT_PIPE_WAVEFRONT) return ensure_wavefront(s, width * height, wavefront_slots(width * height, spp_count));
    if (pipeline == RT_PIPE_MEGAKERNEL && rng_mode == RT_RNG_REFERENCE) {
        if (max_depth > 64) return fail(RT_ERR_LIMIT, "rt_scene_reserve: RT_RNG_REFERENCE supports max_depth <= 64");
        rt::RenderParams rp{width, height, 0u, spp_count, max_depth, rng_mode, 0u};
        return ensure_replay(s, rp);
    }
    return RT_OK;  // the counter-RNG megakernel needs no scratch
}

int rt_render(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
              int rng_mode, uint32_t seed, int pipeline, float *radiance_sum) {
    if (!s || !radiance_sum) return fail(RT_ERR_INVALID, "rt_render: null argument");
    int rc = check_image(width, height);
    if (rc != RT_OK) return rc;
    DeviceGuard g(s->device);
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    float *d_sum = nullptr;
    hipError_t e = hipMalloc((void **)&d_sum, bytes);
    if (e != hipSuccess) return hip_fail("rt_render: hipMalloc", e);
    rc = rt_render_dev(s, width, height, spp_begin, spp_count, max_depth, rng_mode, seed, pipeline, d_sum, nullptr);
    if (rc == RT_OK && (e = hipDeviceSynchronize()) == hipSuccess) e = hipMemcpy(radiance_sum, d_sum, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_sum);
    if (e != hipSuccess) return hip_fail("rt_render", e);
    return rc;
}

int rt_tonemap_dev(rt_scene *s, const float *d_radiance_sum, uint64_t npixels, uint32_t spp, float *d_out_float,
                   uint8_t *d_out_rgb, void *stream) {
    if (!s || !d_radiance_sum) return fail(RT_ERR_INVALID, "rt_tonemap_dev: null argument");
    if (spp == 0) return fail(RT_ERR_INVALID, "rt_tonemap_dev: spp must be positive");
    DeviceGuard g(s->device);
    RT_HIP(rt::launch_tonemap(d_radiance_sum, npixels, spp, d_out_float, d_out_rgb, (hipStream_t)stream, s->timer));
    return RT_OK;
}

int rt_tonemap(rt_scene *s, const float *radiance_sum, uint64_t npixels, uint32_t spp, float *out_float, uint8_t *out_rgb) {
    if (!s || !radiance_sum) return fail(RT_ERR_INVALID, "rt_tonemap: null argument");
    if (npixels == 0) return RT_OK;
    DeviceGuard g(s->device);
    const size_t n3 = (size_t)npixels * 3;
    float *d_sum = nullptr, *d_f = nullptr;
    uint8_t *d_b = nullptr;
    int rc = RT_OK;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc((void **)&d_sum, n3 * 4)) == hipSuccess && (e = hipMalloc((void **)&d_f, n3 * 4)) == hipSuccess &&
        (e = hipMalloc((void **)&d_b, n3)) == hipSuccess &&
        (e = hipMemcpy(d_sum, radiance_sum, n3 * 4, hipMemcpyHostToDevice)) == hipSuccess) {
        rc = rt_tonemap_dev(s, d_sum, npixels, spp, d_f, d_b, nullptr);
        if (rc == RT_OK && (e = hipDeviceSynchronize()) == hipSuccess) {
            if (out_float) e = hipMemcpy(out_float, d_f, n3 * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess && out_rgb) e = hipMemcpy(out_rgb, d_b, n3, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d_sum);
    (void)hipFree(d_f);
    (void)hipFree(d_b);
    if (e != hipSuccess) return hip_fail("rt_tonemap", e);
    return rc;
}

// printImage, main.cpp:856-864: "P6\n<W> <H>\n255\n" followed by W*H RGB bytes, top row first.
int rt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb) {
    if (!path || !rgb) return fail(RT_ERR_INVALID, "rt_write_ppm: null argument");
    std::ofstream f(path, std::ios::binary);
    if (!f) return fail(RT_ERR_IO, std::string("rt_write_ppm: cannot open ") + path);
    f << "P6" << '\n' << width << " " << height << '\n' << 255 << '\n';
    f.write(reinterpret_cast<const char *>(rgb), (std::streamsize)((size_t)width * height * 3));
    f.close();
    if (!f) return fail(RT_ERR_IO, std::string("rt_write_ppm: write failed for ") + path);
    return RT_OK;
}

int rt_selftest_rng(int device, const uint32_t *seeds, uint32_t nseeds, uint32_t n, uint32_t nlights, int what, float *uni,
                    float *nrm, uint32_t *ui) {
    if (!seeds || !uni || !nrm || !ui) return fail(RT_ERR_INVALID, "rt_selftest_rng: null argument");
    if (what < 0 || what > 4) return fail(RT_ERR_INVALID, "rt_selftest_rng: what must be 0..4");
    if (nlights == 0 || nlights > 65536u) return fail(RT_ERR_INVALID, "rt_selftest_rng: nlights must be in 1..65536");
    if ((uint64_t)nseeds * n > (1ull << 26)) return fail(RT_ERR_LIMIT, "rt_selftest_rng: more than 2^26 values");
    if (device < 0 || device >= rt_device_count()) return fail(RT_ERR_INVALID, "rt_selftest_rng: device index out of range");
    DeviceGuard g(device);
    const size_t total = (size_t)nseeds * n;
    if (total == 0) return RT_OK;
    uint32_t *d_seeds = nullptr, *d_ui = nullptr;
    float *d_uni = nullptr, *d_nrm = nullptr;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc((void **)&d_seeds, (size_t)nseeds * 4)) == hipSuccess &&
        (e = hipMalloc((void **)&d_uni, total * 4)) == hipSuccess && (e = hipMalloc((void **)&d_nrm, total * 4)) == hipSuccess &&
        (e = hipMalloc((void **)&d_ui, total * 4)) == hipSuccess &&
        (e = hipMemcpy(d_seeds, seeds, (size_t)nseeds * 4, hipMemcpyHostToDevice)) == hipSuccess &&
        (e = hipMemset(d_uni, 0, total * 4)) == hipSuccess && (e = hipMemset(d_nrm, 0, total * 4)) == hipSuccess &&
        (e = hipMemset(d_ui, 0, total * 4)) == hipSuccess &&
        (e = rt::launch_rng_selftest(d_seeds, nseeds, n, nlights, what, d_uni, d_nrm, d_ui, nullptr)) == hipSuccess &&
        (e = hipDeviceSynchronize()) == hipSuccess && (e = hipMemcpy(uni, d_uni, total * 4, hipMemcpyDeviceToHost)) == hipSuccess &&
        (e = hipMemcpy(nrm, d_nrm, total * 4, hipMemcpyDeviceToHost)) == hipSuccess)
        e = hipMemcpy(ui, d_ui, total * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_seeds);
    (void)hipFree(d_uni);
    (void)hipFree(d_nrm);
    (void)hipFree(d_ui);
    if (e != hipSuccess) return hip_fail("rt_selftest_rng", e);
    return RT_OK;
}

int rt_set_counting(rt_scene *s, int enable) {
    if (!s) return fail(RT_ERR_INVALID, "rt_set_counting: null scene");
    s->counting = enable != 0;
    return RT_OK;
}

int rt_set_timing(rt_scene *s, int enable) {
    if (!s) return fail(RT_ERR_INVALID, "rt_set_timing: null scene");
    DeviceGuard g(s->device);
    RT_HIP(hipDeviceSynchronize());
    s->timer.reset();
    s->timer.enabled = enable != 0;
    return RT_OK;
}

int rt_timing_read(rt_scene *s, rt_kernel_times *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_timing_read: null argument");
    DeviceGuard g(s->device);
    RT_HIP(hipDeviceSynchronize());
    s->timer.collect();
    for (int k = 0; k < RT_KERNEL_KINDS; ++k) {
        out->ms[k] = s->timer.ms[k];
        out->launches[k] = s->timer.launches[k];
    }
    return RT_OK;
}

int rt_counters_reset(rt_scene *s, void *stream) {
    if (!s) return fail(RT_ERR_INVALID, "rt_counters_reset: null scene");
    DeviceGuard g(s->device);
    RT_HIP(hipMemsetAsync(s->d_counters, 0, kCounterWords * sizeof(unsigned long long), (hipStream_t)stream));
    s->host_segments = 0;
    return RT_OK;
}

int rt_counters_read(rt_scene *s, rt_counters *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_counters_read: null argument");
    DeviceGuard g(s->device);
    std::vector<unsigned long long> raw(kCounterWords);
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(raw.data(), s->d_counters, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sh = 0; sh < rtd::kCounterShards; ++sh)
        for (int k = 0; k < 8; ++k) h[k] += raw[(size_t)sh * rtd::kCounterStride + k];
    out->segments = h[0] + s->host_segments;
    out->nodes = h[1];
    out->tri_tests = h[2];
    out->light_tests = h[3];
    out->node_lane_slots = h[4];
    out->tri_lane_slots = h[5];
    out->retraced = h[7];
    if (h[6])
        return fail(RT_ERR_INVALID, "rt_counters_read: " + std::to_string(h[6]) +
                                        " workgroup(s) of a counting launch cut a ray's walk short at the step bound: the "
                                        "traversal state machine or the tree is broken");
    return RT_OK;
}

}  // extern "C"
