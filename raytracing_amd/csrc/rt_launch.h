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
enum KernelKind { kKernExtend = 0, kKernShade, kKernGenerate, kKernAccumulate, kKernMega, kKernClosestHit, kKernTonemap, kKernKinds };

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

// Scratch of the wavefront pipeline: path records for `capacity` paths (64 B each, as four 16-byte
// quarters so that a lane moves a record with four loads or stores: the vector-memory path charges
// per lane and instruction, not per byte), two index queues, queue counters.  Owned by the scene,
// sized on first use.
// Record layout: RT_REC_STRIDE 2 = two arrays (ray halves, state halves); 4 = one array of 64-byte
// records, ray half then state half (state = ray + 2), so that a path touched through an index queue
// costs one 64-byte sector instead of two half-used ones.
#ifndef RT_REC_STRIDE
#define RT_REC_STRIDE 2
#endif
constexpr size_t kWfRecStride = RT_REC_STRIDE;
static_assert(kWfRecStride == 2 || kWfRecStride == 4, "record stride");
constexpr size_t kWfCursorWords = 8 * 1024;  // 8 cursors, 4 KB apart (rt_kernels.hip: kShards x kCursorStride)
struct WavefrontBuffers {
    uint32_t capacity = 0;
    float4 *rec = nullptr;    // the allocation behind ray and state: 4 x capacity float4
    float4 *ray = nullptr;    // 2 x capacity: [2i] = (o.xyz, d.x)  [2i+1] = (d.y, d.z, hit t, bits(hit triangle))
    float4 *state = nullptr;  // 2 x capacity: [2i] = (throughput.xyz, bits(rng))  [2i+1] = (radiance.xyz, bits(depth left))
    uint32_t *queue[2] = {nullptr, nullptr};  // path indices alive in the current / next bounce
    uint32_t *qcount = nullptr; // [0], [1]: queue lengths (device)
    uint32_t *cursors = nullptr; // work cursors of the extend kernel, kWfCursorWords words

};

hipError_t launch_intersect(const rtd::DevScene &S, uint64_t n, const float *o, const float *d, uint32_t *prim_id,
                            float *t, bool count, hipStream_t stream, KernelTimer &timer);
hipError_t launch_primary(const rtd::DevScene &S, uint32_t W, uint32_t H, uint32_t *prim_id, float *t, bool count,
                          hipStream_t stream, KernelTimer &timer);
hipError_t launch_render_mega(const rtd::DevScene &S, const RenderParams &rp, float *radiance_sum, bool count,
                              hipStream_t stream, KernelTimer &timer);
// Replay kernel (reference RNG stream, backward evaluation): words of per-lane vertex records it needs.
uint64_t replay_words(const RenderParams &rp);
hipError_t launch_render_replay(const rtd::DevScene &S, const RenderParams &rp, float *radiance_sum, uint32_t *rec,
                                bool count, hipStream_t stream, KernelTimer &timer);
hipError_t launch_render_wavefront(const rtd::DevScene &S, const RenderParams &rp, float *radiance_sum,
                                   const WavefrontBuffers &wb, uint32_t slots, bool count, hipStream_t stream,
                                   KernelTimer &timer);
hipError_t launch_rng_selftest(const uint32_t *seeds, uint32_t nseeds, uint32_t n, uint32_t L, int what, float *uni,
                               float *nrm, uint32_t *ui, hipStream_t stream);
hipError_t launch_tonemap(const float *radiance_sum, uint64_t npixels, uint32_t spp, float *out_float,
                          uint8_t *out_rgb, hipStream_t stream, KernelTimer &timer);

}  // namespace rt
