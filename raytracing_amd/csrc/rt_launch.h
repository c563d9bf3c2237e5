// rt_launch.h -- host-callable launchers of the gfx950 kernels (defined in rt_kernels.hip).
// All launchers enqueue on `stream` and return the launch status; none synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "rt_device.h"

namespace rt {

// Optional per-kernel timing with HIP events recorded on the launch stream (rt_set_timing).
// Spans are collected after a synchronisation; disabled, begin/end cost nothing.
enum KernelKind { kKernExtend = 0, kKernShade, kKernGenerate, kKernAccumulate, kKernMega, kKernClosestHit, kKernTonemap, kKernSort, kKernKinds };

struct KernelTimer {
    bool enabled = false;
    std::vector<hipEvent_t> pool;
    struct Span {
        int kind;
        hipEvent_t a, b;
    };
    std::vector<Span> spans;
    double ms[kKernKinds] = {0};
    unsigned long long launches[kKernKinds] = {0};

    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(int kind, hipStream_t stream) {
        if (!enabled) return;
        Span s{kind, get(), get()};
        (void)hipEventRecord(s.a, stream);
        spans.push_back(s);
    }
    void end(hipStream_t stream) {
        if (!enabled || spans.empty()) return;
        (void)hipEventRecord(spans.back().b, stream);
    }
    void collect() {  // caller has synchronised the device
        for (const Span &s : spans) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) {
                ms[s.kind] += t;
                ++launches[s.kind];
            }
            pool.push_back(s.a);
            pool.push_back(s.b);
        }
        spans.clear();
    }
    void reset() {
        collect();
        for (int k = 0; k < kKernKinds; ++k) {
            ms[k] = 0;
            launches[k] = 0;
        }
    }
    void release() {
        collect();
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

struct RenderParams {
    uint32_t W, H;
    uint32_t spp_begin, spp_count;
    uint32_t max_depth;
    int rng_mode;
    uint32_t seed;
};

// Scratch of the wavefront pipeline for `capacity` paths in flight, 90 bytes per path: a 48-byte record
// (32 contiguous bytes of ray + 16 apart; three 16-byte accesses per lane: the vector-memory path charges per
// lane and instruction, the memory system per sector), the radiance gathered so far, the hit of the bounce
// being traced, two index queues, the retry list of the wide walk and the ray-sorting cells; plus queue
// counters, work cursors and the cell histogram.  Owned by the scene, grown on demand.
constexpr size_t kWfCountWords = 64;  // queue lengths [0], [1], retry count [2]
constexpr size_t kWfTuneWords = 128;  // the sort tuner's books (rt_kernels.hip: SortChoice)
constexpr size_t kWfCursorWords = 8 * 1024;  // 8 cursors, 4 KB apart (rt_kernels.hip: kShards x kCursorStride)
constexpr size_t kWfBinWords = 2 * 4096;     // histogram + cursors of the ray-sorting cells (rt_device.h: kBins)
constexpr uint32_t kWfRefLanes = 256 * 256;  // lanes of the reference-order retry kernel (scene-owned scratch)
constexpr size_t kWfBytesPerPath = 16 + 16 + 16 + 16 + 8 + 4 + 4 + 8 + 2;
struct WavefrontBuffers {
    uint32_t capacity = 0;
    float4 *ray = nullptr;    // 2 x capacity: [2i] = (o.xyz, d.x)  [2i+1] = (d.y, d.z, throughput.x, throughput.y)
    float4 *c = nullptr;      // capacity: (throughput.z, bits(rng), bits(home slot), 0)
    float4 *lum = nullptr;    // capacity: (radiance gathered so far .xyz, 0)
    float2 *hit = nullptr;    // capacity: (hit t, bits(hit triangle)) of each path's current ray
    uint32_t *queue[2] = {nullptr, nullptr};  // path indices alive in the current / next bounce, in append order or regrouped by cell
    uint32_t *retry = nullptr;  // 2 x capacity: (slot, hit position) of the rays whose walk of the four-child tree ran out of stack
    uint16_t *bin = nullptr;  // capacity: ray-sorting cell of each entry of the queue being filled
    uint32_t *binhist = nullptr;  // kWfBinWords: histogram of the cells, then their cursors
    int sort = 0;             // regroup the rays of the later bounces by cell: 0 never, 1 every bounce, 2 the bounces for
                              // which it was measured to pay (decided on the device, see SortChoice in rt_kernels.hip)
    uint32_t sort_min_rays = 1u << 20;  // mode 2: shorter queues are not worth three launches
    uint32_t *tuner = nullptr;  // kWfTuneWords words, zero when allocated and then the tuner's own (mode 2)
    bool wide = true;         // wf_extend walks the four-child tree
    bool ties = false;        // rays that met two primitives at the best distance are traced again in the reference's order
    bool ref_all = false;     // ... and so is every other ray, by the unpruned walk (RT_HITS_REFERENCE_WALK; implies ties)
    uint32_t ref_depth = 0;   // ... by a walk over the reference's own tree: its deepest level,
    rtd::RefFrame *ref_frames = nullptr;  // (ref_depth + 1) x ref_lanes frames of that walk,
    uint32_t ref_lanes = 0;   // and the lanes (a multiple of 256) the frames are laid out for
    int stack_limit = 24;     // entries of the traversal stack the wide walk may use (the kernel caps it at its own 20; less only in tests)
    uint32_t *qcount = nullptr; // kWfCountWords words: queue lengths, retry count (device)
    uint32_t *cursors = nullptr; // work cursors of the extend kernel, kWfCursorWords words
};

hipError_t launch_intersect(const rtd::DevScene &S, uint64_t n, const float *o, const float *d, uint32_t *prim_id,
                            float *t, bool count, int ref_order, hipStream_t stream, KernelTimer &timer);
hipError_t launch_primary(const rtd::DevScene &S, uint32_t W, uint32_t H, uint32_t *prim_id, float *t, bool count,
                          bool ref_order, hipStream_t stream, KernelTimer &timer);
// ref_order (only for scenes that hold the reference's tree, S.rnodes): results that depend on the reference's
// traversal order are resolved as the reference resolves them, by its own walk (ref_closest_hit) for every ray.
// The megakernel takes a rtd::HitOrder: own tree / own tree + the reference's walk for rays that met two
// primitives at the best distance / the reference's walk for every ray.
hipError_t launch_render_mega(const rtd::DevScene &S, const RenderParams &rp, float *radiance_sum, bool count, int order,
                              hipStream_t stream, KernelTimer &timer);
// Replay kernel (reference RNG stream, backward evaluation): words of per-lane vertex records it needs.
uint64_t replay_words(const RenderParams &rp);
hipError_t launch_render_replay(const rtd::DevScene &S, const RenderParams &rp, float *radiance_sum, uint32_t *rec,
                                bool count, bool ref_order, hipStream_t stream, KernelTimer &timer);
hipError_t launch_render_wavefront(const rtd::DevScene &S, const RenderParams &rp, float *radiance_sum,
                                   const WavefrontBuffers &wb, uint32_t slots, bool count, hipStream_t stream,
                                   KernelTimer &timer);
hipError_t launch_rng_selftest(const uint32_t *seeds, uint32_t nseeds, uint32_t n, uint32_t L, int what, float *uni,
                               float *nrm, uint32_t *ui, hipStream_t stream);
hipError_t launch_tonemap(const float *radiance_sum, uint64_t npixels, uint32_t spp, float *out_float,
                          uint8_t *out_rgb, hipStream_t stream, KernelTimer &timer);

}  // namespace rt
