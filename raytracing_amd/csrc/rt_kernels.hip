// rt_kernels.hip -- gfx950 kernels of the path tracer and their launchers.
//
// Kernels (all 256-thread workgroups = 4 wave64s, per-lane traversal stack in LDS):
//   k_intersect      closest hit for a caller-supplied ray batch      (intersectScene, main.cpp:1113)
//   k_primary        camera ray of sample 0 per pixel + closest hit   (main.cpp:1212-1216, 1221)
//   k_render_mega    persistent lane per pixel, path regeneration     (generatePixel's loop, main.cpp:1220-1225)
//   k_render_replay  the reference's RNG stream and recursion order, bit-exact validation kernel
//   wf_generate / wf_extend (+ wf_extend_retry) / wf_shade / wf_accumulate
//                    the same estimator as queue-per-bounce kernels over 48-byte path records; wf_extend
//                    walks the four-child tree, everything else the two-child one
//   wf_bin_hist / wf_bin_scan / wf_bin_scatter
//                    regrouping of a ray queue by the cell of the ray origin where its paths have lost their coherence (RT_WF_SORT)
//   k_tonemap        mean, ACES, gamma, byte conversion               (main.cpp:1226-1231, 840-846)
//
// Pixel kernels map a wave to an 8x8 pixel tile (coherent primary rays) and a workgroup to a
// 16x16 tile; workgroup ids are re-dealt so that the workgroups of one XCD cover a contiguous
// band of the image and its L2 keeps the part of the BVH that band looks at.
#include "rt_launch.h"

using namespace rtd;

namespace {

// blockIdx.x -> logical tile index; workgroups b and b+8 share an XCD (round-robin dispatch), so
// XCD k is given the k-th contiguous slice of the tile list.  Bijective for any block count.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks) {
    const uint32_t xcd = b & 7u, j = b >> 3;
    const uint32_t q = nblocks >> 3, r = nblocks & 7u;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + j;
}

// Pixel of this lane: 16x16 tile per workgroup, 8x8 sub-tile per wave.
__device__ __forceinline__ bool tile_pixel(uint32_t W, uint32_t H, uint32_t &px, uint32_t &py) {
    const uint32_t tilesX = (W + 15u) >> 4;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    px = (tx << 4) + ((wave & 1u) << 3) + (lane & 7u);
    py = (ty << 4) + ((wave >> 1) << 3) + (lane >> 3);
    return px < W && py < H;
}

// Statistics counters are sharded: workgroup b adds into shard b % kCounterShards (one 128-byte
// line each), because every wave adding to ONE word serialises at the memory side (a single
// address takes about 88 atomics per microsecond, which alone cost a millisecond per 1080p
// launch).  The host sums the shards when it reads them.
__device__ __forceinline__ void add_counters(const DevScene &S, uint32_t segs, const TraceStats &st, uint32_t lights,
                                             bool count) {
    unsigned long long *c = S.counters + (size_t)(blockIdx.x & (rtd::kCounterShards - 1)) * rtd::kCounterStride;
    if (segs) atomicAdd(&c[0], (unsigned long long)segs);
    if (count) {
        atomicAdd(&c[1], (unsigned long long)st.nodes);
        atomicAdd(&c[2], (unsigned long long)st.tris);
        atomicAdd(&c[4], (unsigned long long)st.nodeWave);
        atomicAdd(&c[5], (unsigned long long)st.triWave);
        if (st.runaway) atomicAdd(&c[6], 1ull);
    }
    if (lights) atomicAdd(&c[3], (unsigned long long)lights);
}

// Primitive id reported to the caller: the loader-order id of a triangle, ntri + k for analytic primitive k.
__device__ __forceinline__ uint32_t tri_loader_id(const DevScene &S, int tri) {
    if ((uint32_t)tri >= S.ntri) return (uint32_t)tri - S.npad;
    return __float_as_uint(S.tris[4 * (size_t)tri + 3].x);
}

// REF 1: every ray is answered by the reference's own walk over the reference's own tree (ref_closest_hit);
// REF 2 (tests): by that walk in the pruned form the throughput kernels use for the rays they hand over;
// REF 3 (tests): as the megakernel answers in reference order -- the fast walk, and that pruned walk for the rays
// on which two hits lay close together.
template <bool COUNT, int REF>
__global__ __launch_bounds__(kBlock) void k_intersect(DevScene S, uint64_t n, const float *__restrict__ o,
                                                       const float *__restrict__ d, uint32_t *__restrict__ prim_id,
                                                       float *__restrict__ tout) {
    __shared__ int s_stack[kStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const v3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    const v3 rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    float t;
    int tri;
    TraceStats st{0, 0, 0, 0, 0};
    if (REF == 2) ref_closest_hit<COUNT, true>(S, ro, rd, t, tri, st);
    else if (REF == 3) closest_hit<COUNT, kOrderRefTies>(S, ro, rd, s_stack + threadIdx.x, t, tri, st);
    else closest_hit<COUNT, REF ? kOrderRefFull : kOrderLeaf>(S, ro, rd, s_stack + threadIdx.x, t, tri, st);
    prim_id[i] = tri >= 0 ? tri_loader_id(S, tri) : 0xFFFFFFFFu;
    tout[i] = t;
    if (COUNT) add_counters(S, 0u, st, 0u, true);  // the host adds one segment per ray
}

template <bool COUNT, bool REF>
__global__ __launch_bounds__(kBlock) void k_primary(DevScene S, uint32_t W, uint32_t H, uint32_t *__restrict__ prim_id,
                                                     float *__restrict__ tout) {
    __shared__ int s_stack[kStack * kBlock];
    uint32_t px, py;
    if (!tile_pixel(W, H, px, py)) return;
    const uint32_t pix = py * W + px;
    uint32_t rng = rng_seed(pix);  // minstd_rand{py * W + px}, main.cpp:1221
    const v3 rd = primary_direction(S, px, py, W, H, rng);
    const v3 ro = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
    float t;
    int tri;
    TraceStats st{0, 0, 0, 0, 0};
    closest_hit<COUNT, REF ? kOrderRefFull : kOrderLeaf>(S, ro, rd, s_stack + threadIdx.x, t, tri, st);
    prim_id[pix] = tri >= 0 ? tri_loader_id(S, tri) : 0xFFFFFFFFu;
    tout[pix] = t;
    if (COUNT) add_counters(S, 0u, st, 0u, true);  // the host adds one segment per pixel
}

// One lane owns one pixel and runs its samples back to back; when a path ends the lane starts
// the next sample in the same loop iteration ("path regeneration"), so lanes of a wave stay
// busy although their paths have different lengths.
// ORDER (HitOrder): kOrderRefTies: a ray that met two primitives at the best distance is traced again in the
// reference's order; kOrderRefFull: every ray is (the reference's NaN and beyond-INF hits and its subtree cuts too).
// ALIGHT: the light list holds emissive analytic primitives; their sampling and density code sits in functions
// that are not inlined, and a kernel that may call one pays for the callee's registers and a call stack whether
// a scene has such lights or not (146 instead of 82 VGPRs here, 3 instead of 5 waves per SIMD, -34 % rays/s).
// kOrderRefTies at four waves per SIMD: the call to the reference-order walker for the few rays that need it
// makes the compiler budget 144 registers (three waves); held to 128 it spills 19 of them and the Cornell box
// renders at 9.95 instead of 8.67 Grays/s (leaf order, which has no such call: 10.86; five waves, 52 spills: 8.88).
template <bool COUNT, bool TREE, int ORDER, bool ALIGHT>
__global__ __launch_bounds__(kBlock, ORDER == kOrderRefTies ? 4 : 1) void k_render_mega(DevScene S, rt::RenderParams rp, float *__restrict__ sum) {
    __shared__ int s_stack[kStack * kBlock];
    __shared__ LightCacheStorage s_lights;
    const LightCache LC = load_light_cache(S, s_lights, s_stack + threadIdx.x);
    uint32_t px, py;
    if (!tile_pixel(rp.W, rp.H, px, py)) return;
    const uint32_t pix = py * rp.W + px;
    const v3 camPos = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);

    PathState P;
    P.rng = rng_seed(pix);  // reference stream; overwritten per sample in counter mode
    P.o = camPos;
    P.d = mk(0.f, 0.f, -1.f);
    P.T = mk(1.f, 1.f, 1.f);
    P.L = mk(0.f, 0.f, 0.f);
    P.depth = 0;
    v3 acc = mk(0.f, 0.f, 0.f);
    uint32_t s = 0, segs = 0, lightTests = 0;
    bool active = false;
    TraceStats st{0, 0, 0, 0, 0};
    for (;;) {
        if (!active) {
            if (s >= rp.spp_count) break;
            if (rp.rng_mode == 1) P.rng = rng_seed(counter_seed(pix, rp.spp_begin + s, rp.seed));
            P.d = primary_direction(S, px, py, rp.W, rp.H, P.rng);
            P.o = camPos;
            P.T = mk(1.f, 1.f, 1.f);
            P.L = mk(0.f, 0.f, 0.f);
            P.depth = rp.max_depth;
            active = true;
            ++s;
        }
        float t;
        int tri;
        closest_hit<COUNT, ORDER>(S, P.o, P.d, s_stack + threadIdx.x, t, tri, st);
        ++segs;
        active = shade_hit<TREE, ALIGHT>(S, LC, P, t, tri, lightTests);
        if (!active) acc = acc + P.L;  // color += generateSample(...), main.cpp:1224
    }
    sum[3 * (size_t)pix] = acc.x;
    sum[3 * (size_t)pix + 1] = acc.y;
    sum[3 * (size_t)pix + 2] = acc.z;
    add_counters(S, segs, st, lightTests, COUNT);
}

// The reference's own evaluation order, for its own random stream: one lane per pixel, samples in
// order with the engine carried across them (main.cpp:1221-1225), and each sample's chain of
// vertices recorded and then folded BACKWARDS exactly as the recursion of applyLight* returns
// (L = k * L_next + e, innermost first).  With the bit-exact closest hit and the restated
// libstdc++ / glibc sampling code this reproduces the CPU path's radiance sums bit for bit; it is
// the validation kernel, not the fast one.  rec: [depth][3][stride] words (op | material << 3,
// wn, pdf), stride = threads of the launch.
// REF: every ray through ref_closest_hit, so that the order-dependent results are the reference's too.
template <bool COUNT, bool TREE, bool REF, bool ALIGHT>
__global__ __launch_bounds__(kBlock) void k_render_replay(DevScene S, rt::RenderParams rp, float *__restrict__ sum,
                                                          uint32_t *__restrict__ rec, uint32_t stride) {
    __shared__ int s_stack[kStack * kBlock];
    __shared__ LightCacheStorage s_lights;
    const LightCache LC = load_light_cache(S, s_lights, s_stack + threadIdx.x);
    uint32_t px, py;
    if (!tile_pixel(rp.W, rp.H, px, py)) return;
    const uint32_t pix = py * rp.W + px;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    const v3 camPos = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
    PathState P;
    P.rng = rng_seed(pix);
    P.T = mk(1.f, 1.f, 1.f);
    P.L = mk(0.f, 0.f, 0.f);
    v3 acc = mk(0.f, 0.f, 0.f);
    uint32_t segs = 0, lightTests = 0;
    TraceStats st{0, 0, 0, 0, 0};
    for (uint32_t s = 0; s < rp.spp_count; ++s) {
        if (rp.rng_mode == 1) P.rng = rng_seed(counter_seed(pix, rp.spp_begin + s, rp.seed));
        P.d = primary_direction(S, px, py, rp.W, rp.H, P.rng);
        P.o = camPos;
        P.depth = rp.max_depth;
        uint32_t nrec = 0;
        for (;;) {
            float t;
            int tri;
            closest_hit<COUNT, REF ? kOrderRefFull : kOrderLeaf>(S, P.o, P.d, s_stack + threadIdx.x, t, tri, st);
            ++segs;
            Vertex V;
            const bool alive = sample_vertex<TREE, ALIGHT>(S, LC, P, t, tri, V, lightTests);
            uint32_t *r = rec + (size_t)nrec * 3u * stride + g;
            r[0] = (uint32_t)V.op | (V.mat << 3);
            r[stride] = __float_as_uint(V.wn);
            r[2 * (size_t)stride] = __float_as_uint(V.pdf);
            ++nrec;
            if (!alive) break;
        }
        v3 l = mk(0.f, 0.f, 0.f);
        for (uint32_t k = nrec; k-- > 0;) {
            const uint32_t *r = rec + (size_t)k * 3u * stride + g;
            const uint32_t w0 = r[0];
            l = fold_backward(S, LC, (int)(w0 & 7u), w0 >> 3, __uint_as_float(r[stride]), __uint_as_float(r[2 * (size_t)stride]), l);
        }
        acc = acc + l;  // color += generateSample(...), main.cpp:1224
    }
    sum[3 * (size_t)pix] = acc.x;
    sum[3 * (size_t)pix + 1] = acc.y;
    sum[3 * (size_t)pix + 2] = acc.z;
    add_counters(S, segs, st, lightTests, COUNT);
}

// ---------------------------------------------------------------------------------------------
// Wavefront pipeline.  Path record i lives at slot i of the record arrays for its whole life;
// queues hold the slot indices still alive.  Slot i renders pixel frame_pixel(i % npix), sample
// spp0 + i / npix.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kShards = 8;           // work cursors of wf_extend (see there)
constexpr uint32_t kCursorStride = 1024;  // words between cursors
static_assert(kShards * kCursorStride == rt::kWfCursorWords, "cursor buffer size");
static_assert(2 * kBins == rt::kWfBinWords, "histogram + cursor buffer size");
static_assert(kBins % kBlock == 0 && kBins <= 65536, "cells per thread in wf_bin_scan; cells are stored as 16 bits");

// The record in slot i is 48 bytes: 32 contiguous ones that hold the ray (what wf_extend reads) and 16 apart; a
// lane moves it with three 16-byte accesses.  It holds what the next bounce needs and nothing else.  The
// radiance a path has gathered lives apart, at the slot the path started in (its home slot: that is the pixel
// and sample it belongs to), and changes only when the path meets an emitter; the depth left is the same for
// every path of a bounce; so neither travels through the shading kernel's loads and stores.  Records do not
// keep their slots: wf_shade moves the survivors of every wave together (see there).  Before the first bounce
// a path is (d.xyz, bits(rng)) in lum[i] and nothing else: see wf_generate.
struct WfView {
    float4 *ray;    // [2i] = (o.xyz, d.x)  [2i+1] = (d.y, d.z, throughput.x, throughput.y)
    float4 *c;      // [i] = (throughput.z, bits(rng), bits(home slot), 0)
    float4 *lum;    // [home slot] = (radiance gathered so far .xyz, 0)
    float2 *hit;    // (hit t, bits(hit triangle)) of a path's current ray: at its queue position, or at its slot when
                    // the queue wf_extend traces is a regrouped copy of the one wf_shade walks (hit_by_slot)
    uint16_t *bin;  // [j] = ray-sorting cell of the ray at position j of the queue wf_shade is filling
    uint32_t cap;
};

// Is the queue of a bounce traced in cell order?  mode 0: no; 1: yes; 2: as the tuner decided (*decision).
//
// The tuner (mode 2).  Whether regrouping a bounce's rays pays differs from scene to scene and from bounce to
// bounce -- on the million-triangle scene from the fourth bounce on (a third of neighbouring lanes still start
// in the same cell there), on the glass scene only from the seventh (a sixth), and measures of the rays'
// coherence do not tell the two apart: what differs is what incoherence costs.  So the cost itself is
// measured.  Every pass of a frame traces statistically the same rays, bounce by bounce; the last workgroup
// of every wf_shade launch leaves the time, the last workgroup of the next wf_extend launch takes the
// difference -- the sort launches, if any, and the trace -- per ray, files it under "sorted" or "unsorted" of
// its bounce (a running mean) and decides the next bounce: unsorted until that has been measured, then sorted
// until that has, then whichever was cheaper, with the other one tried again every 64th decision.  Everything
// lives in `tune` (device memory of the scene, kept from pass to pass and call to call); nothing returns to
// the host, and the kernels that need a decision (wf_shade for the cells, the sort kernels, wf_extend,
// the next wf_shade for where the hits lie) read the same word.
//   tune[0..1] time mark   [2] finished workgroups of the launch at hand   [3] decisions taken
//   tune[8 + 4 b ..]: bounce b: cost unsorted, cost sorted (2^-20 ticks of 10 ns per ray), decision, unused
constexpr uint32_t kTuneBounces = (rt::kWfTuneWords - 8u) / 4u;
struct SortChoice {
    uint32_t mode;
    const uint32_t *decision;
};
__device__ __forceinline__ bool sorted_queue(const SortChoice c) {
    return c.mode == 1u || (c.mode == 2u && *c.decision != 0u);
}
__device__ __forceinline__ uint32_t *tune_bounce(uint32_t *tune, uint32_t b) {
    return tune + 8u + 4u * (b < kTuneBounces ? b : kTuneBounces - 1u);
}
// True in exactly one thread of the launch: thread 0 of the workgroup that finishes last.  Every thread of
// every workgroup must call it.
__device__ __forceinline__ bool last_workgroup(uint32_t *tune) {
    __syncthreads();
    if (threadIdx.x != 0u) return false;
    __threadfence();
    if (atomicAdd(tune + 2, 1u) != gridDim.x - 1u) return false;
    tune[2] = 0u;
    return true;
}

// Record j of a frame of samples renders pixel frame_pixel(j): 8x8 pixel tiles when the image divides
// into them, so that the 64 rays a wave works on start (and first land) close together; row order
// otherwise (+2.6 % on the million-triangle scene, +1.7 % on the Cornell box over row order).
__device__ __forceinline__ uint32_t frame_pixel(uint32_t j, uint32_t W, uint32_t H) {
    if ((W | H) & 7u) return j;
    const uint32_t tile = j >> 6, in = j & 63u, tilesX = W >> 3;
    return ((tile / tilesX) * 8u + (in >> 3)) * W + (tile % tilesX) * 8u + (in & 7u);
}

__global__ __launch_bounds__(kBlock) void wf_generate(DevScene S, rt::RenderParams rp, WfView V, uint32_t npaths,
                                                       uint32_t spp0, uint32_t *__restrict__ qcount,
                                                       uint32_t *__restrict__ cursors) {
    const uint32_t npix = rp.W * rp.H;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npaths; i += gridDim.x * kBlock) {
        const uint32_t pix = frame_pixel(i % npix, rp.W, rp.H), sample = spp0 + i / npix;
        const uint32_t px = pix % rp.W, py = pix / rp.W;
        uint32_t rng = rng_seed(counter_seed(pix, sample, rp.seed));
        const v3 d = primary_direction(S, px, py, rp.W, rp.H, rng);
        // A camera ray's record is its direction and engine state: the origin is the camera's and the throughput
        // is 1.  It lies where the path's radiance will: 16 bytes to write here instead of 56 (a record that is
        // not a whole number of 32-byte sectors, or leaves gaps between records, costs more than it saves: 16 of
        // every 32 bytes measured 5.1 ms where the 56 took 3.2).  The kernels of the first bounce know -- they are
        // the ones launched without a queue -- and wf_shade then stores the radiance and the full record.
        V.lum[i] = make_float4(d.x, d.y, d.z, __uint_as_float(rng));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        qcount[0] = npaths;
        qcount[1] = 0u;
        qcount[2] = 0u;  // wf_extend's retry list
    }
    if (blockIdx.x == 0 && threadIdx.x < kShards) cursors[threadIdx.x * kCursorStride] = 0u;  // first extend pass
}

// Traces the rays of the input queue (null queue = identity: slots 0..count-1) and stores the hit.
// Persistent waves with per-lane dynamic ray fetch: a lane whose ray has finished takes the next
// ray of the queue instead of idling until the slowest lane of its wave is done.  A wave grabs
// kChunk consecutive queue entries with one atomic on a work cursor and deals them to its idle
// lanes (ballot + prefix count) whenever at least kRefill lanes are idle.
// One word of global memory takes only ~88 atomics per microsecond for the whole chip, and the chunk
// size trades that against the end of the launch, where every wave still works through its last chunk
// (512 rays = 8 per lane took 0.7 ms on the million-triangle scene: the waves' finishing times spread
// over 17 % of a 33 M-ray launch).  So the queue is cut into kShards equal ranges with a cursor each,
// 4 KB apart; a workgroup starts on shard blockIdx % kShards (with round-robin dispatch, its XCD's)
// and moves on to the next shard when one runs dry.  Eight cursors take eight times the atomics, which
// lets the chunk be 128 rays.
constexpr uint32_t kChunk = 128;  // 64: -1 %, 256: -2 %, 512: -5 % (million-triangle scene); Cornell flat from 128 up
constexpr int kRefill = 20;  // 4: -3 %, 12: -1 %, 16 to 24: within 0.5 %, 32: -2 %

// PRIMS: the scene holds analytic primitives, folded into every finished walk.  A separate instantiation:
// with the fold compiled into the one kernel, scenes without primitives traced 12 % slower.
// WIDE: walks the four-child tree; a ray whose walk ran out of stack (Traversal::overflow) gets no hit here,
// its slot goes to the `retry` list and wf_extend_retry traces it on the two-child tree.
// The wide walk has no stack bound to honour (a walk that runs out is repeated on the two-child tree), so its
// stack is sized for occupancy: 20 entries = 22 KB of LDS per workgroup = seven workgroups per CU, at 72 VGPRs
// (seven waves per SIMD, nothing spilled).  Against six with 24 entries: -2.7 % on the million-triangle frame,
// -5 % on the Cornell box and -4 % on the 100 k-triangle height field (the later bounces wait for memory and
// gain most), +-0 on the glass scene, where 61 instead of 4 us per launch go into repeated walks.  Eight waves
// (17 entries, 64 VGPRs with 15 spilled): +15 %.
// TIES: a ray whose walk met two primitives at the best distance goes to the `retry` list as well, and
// wf_extend_retry<REF> answers it in the reference's own order.
constexpr int kWideStack = 20;
constexpr int kWideWaves = 7;
template <bool COUNT, bool PRIMS, bool WIDE, bool TIES>
__global__ __launch_bounds__(kBlock, WIDE ? kWideWaves : 1) void wf_extend(DevScene S, WfView V, const uint32_t *__restrict__ queue_shaded,
                                                     const uint32_t *__restrict__ queue_sorted, SortChoice choice,
                                                     const uint32_t *__restrict__ qcount_in,
                                                     uint32_t *__restrict__ qcount_zero, uint32_t *__restrict__ cursors,
                                                     uint32_t *__restrict__ retry, uint32_t *__restrict__ retry_count,
                                                     int stack_limit, uint32_t *__restrict__ tune, uint32_t bounce,
                                                     uint32_t sort_min_rays) {
    // the queue in the order wf_shade filled it, or its copy regrouped by cell; hits then go to the paths' slots,
    // because wf_shade walks the queue it filled
    const bool hit_by_slot = queue_shaded && sorted_queue(choice);
    const uint32_t *__restrict__ queue = hit_by_slot ? queue_sorted : queue_shaded;
    __shared__ int s_stack[(WIDE ? kWideStack : kStack) * kBlock];
    __shared__ uint32_t s_xchg[4][128];
    Traversal<COUNT, kExitLanes, true, WIDE, TIES> T;
    T.limit = WIDE && stack_limit > kWideStack ? kWideStack : stack_limit;
    T.xchg = s_xchg[threadIdx.x >> 6];
    const uint32_t count = *qcount_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *qcount_zero = 0u;  // the queue the next shade pass fills
    const uint32_t lane = threadIdx.x & 63u;
    T.idle();
    T.bind(s_stack + threadIdx.x);
    uint32_t slot = 0, hitAt = 0, segs = 0;  // path slot of this lane's ray, and where its hit goes
    uint32_t rounds = 0;                      // counting launches: rounds this lane's ray has been walking
    bool hasRay = false;
    uint32_t chunkNext = 0, chunkEnd = 0;  // wave-uniform
    uint32_t shard = blockIdx.x % kShards, dry = 0;  // wave-uniform: current shard, shards found empty in a row
    bool exhausted = false;                // wave-uniform: every shard ran dry
    TraceStats st{0, 0, 0, 0, 0};
    for (;;) {
        if (hasRay && T.done()) {
            bool again = WIDE && T.overflow;
            if (!again) {
                bool tie = T.tied();
                T.triBest = T.best();
                if (PRIMS) tie = (analytic_closest(S, T.o, T.d, T.tBest, T.triBest) || tie) && TIES;
                again = tie || (TIES && far_origin(S, T.o));
            }
            if (again) {
                const uint32_t at = atomicAdd(retry_count, 1u);  // (slot, where its hit goes)
                retry[2 * (size_t)at] = slot;
                retry[2 * (size_t)at + 1] = hitAt;
            } else {
                V.hit[hitAt] = make_float2(T.tBest, __int_as_float(T.triBest));
                ++segs;
            }
            hasRay = false;
        }
        const unsigned long long idleMask = __ballot(!hasRay);
        const int nIdle = __builtin_popcountll(idleMask);
        if (nIdle == 64 && exhausted) break;
        if (!exhausted && nIdle >= kRefill) {
            const uint32_t rank = (uint32_t)__builtin_popcountll(idleMask & ((1ull << lane) - 1ull));
            uint32_t given = 0;
            while (given < (uint32_t)nIdle) {
                if (chunkNext == chunkEnd) {
                    // shard s covers [count * s / kShards, count * (s + 1) / kShards)
                    const uint32_t lo = (uint32_t)((uint64_t)count * shard / kShards);
                    const uint32_t hi = (uint32_t)((uint64_t)count * (shard + 1u) / kShards);
                    uint32_t off = 0;
                    if (lane == 0) off = atomicAdd(cursors + shard * kCursorStride, kChunk);
                    off = __shfl(off, 0);
                    if (off >= hi - lo) {
                        shard = (shard + 1u) % kShards;
                        if (++dry == kShards) {
                            exhausted = true;
                            break;
                        }
                        continue;
                    }
                    dry = 0;
                    chunkNext = lo + off;
                    chunkEnd = (hi - chunkNext < kChunk) ? hi : chunkNext + kChunk;
                }
                const uint32_t avail = chunkEnd - chunkNext;
                const uint32_t take = avail < (uint32_t)nIdle - given ? avail : (uint32_t)nIdle - given;
                if (!hasRay && rank >= given && rank < given + take) {
                    const uint32_t qpos = chunkNext + (rank - given);
                    slot = queue ? queue[qpos] : qpos;
                    hitAt = hit_by_slot ? slot : qpos;  // dense 8-byte stores when wf_shade walks this same queue
                    const float4 r0 = queue ? V.ray[2 * (size_t)slot] : V.lum[slot];
                    if (queue) {
                        const float4 r1 = V.ray[2 * (size_t)slot + 1];
                        T.start(S, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y));
                    } else {  // camera rays: (direction, engine state), see wf_generate
                        T.start(S, mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]), mk(r0.x, r0.y, r0.z));
                    }
                    hasRay = true;
                    rounds = 0;
                }
                chunkNext += take;
                given += take;
            }
        }
        T.step(S, st);
        if (COUNT && hasRay && ++rounds > S.step_bound) T.abandon(st);
    }
    if (segs || (COUNT && st.runaway)) add_counters(S, segs, st, 0u, COUNT);
    if (tune && last_workgroup(tune)) {  // the tuner: file this bounce's cost, decide the next bounce
        const unsigned long long now = wall_clock64(), mark = *reinterpret_cast<const unsigned long long *>(tune);
        if (!COUNT && bounce >= 2u && bounce < kTuneBounces && count) {  // (the counting instantiation is slower: not filed)
            uint32_t *cost = tune_bounce(tune, bounce) + (hit_by_slot ? 1 : 0);
            const unsigned long long m = ((now - mark) << 20) / count;
            const uint32_t v = m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (m ? (uint32_t)m : 1u), old = *cost;
            *cost = old ? (uint32_t)((3ull * old + v) >> 2) : v;
        }
        const uint32_t next = bounce + 1u;
        uint32_t d = 0u;
        if (next >= 2u && next < kTuneBounces && count >= sort_min_rays) {
            const uint32_t *t = tune_bounce(tune, next);
            if (t[0] && !t[1]) d = 1u;
            else if (t[0] && t[1]) {
                d = t[1] < t[0] ? 1u : 0u;
                if ((tune[3]++ & 63u) == 63u) d ^= 1u;
            }
        }
        // Only a bounce with books of its own gets a decision.  (tune_bounce() clamps: the write for bounce
        // kTuneBounces would land in the last entry, whose decision the kernels of THAT bounce are still reading.)
        if (next < kTuneBounces) tune_bounce(tune, next)[2] = d;
    }
}

// The rays wf_extend gave up on, one lane each, on the two-child tree (whose height bounds the stack).
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void wf_extend_retry(DevScene S, WfView V, const uint32_t *__restrict__ retry,
                                                           const uint32_t *__restrict__ retry_count, uint32_t camera_rays) {
    __shared__ int s_stack[kStack * kBlock];
    const uint32_t count = *retry_count;
    uint32_t segs = 0;
    TraceStats st{0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < count; k += gridDim.x * kBlock) {
        const uint32_t slot = retry[2 * (size_t)k], hitAt = retry[2 * (size_t)k + 1];
        const float4 r0 = camera_rays ? V.lum[slot] : V.ray[2 * (size_t)slot];
        v3 o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]), d = mk(r0.x, r0.y, r0.z);  // see wf_generate
        if (!camera_rays) {
            const float4 r1 = V.ray[2 * (size_t)slot + 1];
            o = mk(r0.x, r0.y, r0.z);
            d = mk(r0.w, r1.x, r1.y);
        }
        float t;
        int tri;
        closest_hit<COUNT>(S, o, d, s_stack + threadIdx.x, t, tri, st);
        V.hit[hitAt] = make_float2(t, __int_as_float(tri));
        ++segs;
    }
    if (segs) {
        add_counters(S, segs, st, 0u, COUNT);
        atomicAdd(&S.counters[(size_t)(blockIdx.x & (rtd::kCounterShards - 1)) * rtd::kCounterStride + 7], (unsigned long long)segs);
    }
}

// RT_HITS_REFERENCE_WALK: no fast walk at all.  Stands in for wf_extend: every ray of the bounce's queue goes onto
// the list wf_extend_retry_ref works off, and the queue the next shading pass fills starts empty.  A validation
// mode (one lane walks each ray through the reference's tree of double-size boxes): never sorted, never timed.
__global__ __launch_bounds__(kBlock) void wf_list_all(const uint32_t *__restrict__ queue, const uint32_t *__restrict__ qcount_in,
                                                       uint32_t *__restrict__ qcount_zero, uint32_t *__restrict__ retry,
                                                       uint32_t *__restrict__ retry_count) {
    const uint32_t count = *qcount_in;
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < count; q += gridDim.x * kBlock) {
        retry[2 * (size_t)q] = queue ? queue[q] : q;  // the path's slot
        retry[2 * (size_t)q + 1] = q;                 // its hit goes to the queue position
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *retry_count = count;
        *qcount_zero = 0u;
    }
}

// The same list in scenes whose hits are resolved in the reference's order: every ray of it (the rays that met two
// primitives at the best distance, and those the wide walk ran out of stack on) by the reference's own walk.  The
// walk's frames lie in global memory, `frames[level * lanes + lane]` (coalesced), lanes = threads of the launch.
// The kernel is launched once per bounce and finds a handful of rays (on a scene with emissive meshes: the light
// samples that land exactly on an edge, a few in 10^7); it lasts as long as the slowest of their walks, each node
// visit two or three dependent trips to memory.  Hence the pruned walk (ref_walk<PRUNE>): 17 930 triangles, 33 M
// paths per bounce: 1.1 ms per launch unpruned, 0.3 to 0.8 pruned (4.6 % of that frame; 0.4 % of the Cornell frame).
constexpr uint32_t kRefRetryBlocks = 256;  // x 256 lanes, each with (tree depth + 1) frames of 16 bytes
// PRUNE = false: the reference's walk as it is, for RT_HITS_REFERENCE_WALK, where wf_list_all hands over every ray.
template <bool COUNT, bool PRUNE>
__global__ __launch_bounds__(kBlock) void wf_extend_retry_ref(DevScene S, WfView V, const uint32_t *__restrict__ retry,
                                                               const uint32_t *__restrict__ retry_count, uint32_t camera_rays,
                                                               RefFrame *__restrict__ frames, int depth) {
    const uint32_t count = *retry_count;
    const uint32_t lanes = gridDim.x * kBlock, me = blockIdx.x * kBlock + threadIdx.x;
    uint32_t segs = 0;
    TraceStats st{0, 0, 0, 0, 0};
    for (uint32_t k = me; k < count; k += lanes) {
        const uint32_t slot = retry[2 * (size_t)k], hitAt = retry[2 * (size_t)k + 1];
        const float4 r0 = camera_rays ? V.lum[slot] : V.ray[2 * (size_t)slot];
        v3 o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]), d = mk(r0.x, r0.y, r0.z);  // see wf_generate
        if (!camera_rays) {
            const float4 r1 = V.ray[2 * (size_t)slot + 1];
            o = mk(r0.x, r0.y, r0.z);
            d = mk(r0.w, r1.x, r1.y);
        }
        float t;
        int tri;
        ref_walk<COUNT, PRUNE>(S, o, d, frames + me, (int)lanes, depth, t, tri, st);
        V.hit[hitAt] = make_float2(t, __int_as_float(tri));
        ++segs;
    }
    if (segs) {
        add_counters(S, segs, st, 0u, COUNT);
        atomicAdd(&S.counters[(size_t)(blockIdx.x & (rtd::kCounterShards - 1)) * rtd::kCounterStride + 7], (unsigned long long)segs);
    }
}

// Shades the hits of the input queue and appends the surviving paths to the output queue.
// Appending is hierarchical, because one word of global memory takes only about 88 atomic adds per
// microsecond however many CUs issue them: with one global atomic per wave and round this kernel
// (and through its work cursor, wf_extend) ran at exactly that rate -- 1.65 M atomics = 17 ms per
// 16-spp pass of the 1080p Cornell frame.  Survivors are first collected in an LDS buffer of the
// workgroup (one LDS atomic per wave and round, lanes write at their ballot-prefix offset); every few
// rounds, and after the last one, the workgroup reserves space in the global queue with ONE atomic and
// copies the buffer out with coalesced stores.
constexpr uint32_t kAppendBuf = 3072;        // LDS entries per workgroup (12 KB + 6 KB of cells): six workgroups per CU
constexpr uint32_t kAppendBufCamera = 2048;  // the camera-ray instantiation runs eight workgroups per CU
// The light-tree variant keeps a 24 KB traversal stack in LDS as well: with half the buffer four of its
// workgroups still fit a CU (with the full one only three did: 89.8 -> 109 ms on the 15 360-light frame).
constexpr uint32_t kAppendBufTree = 1024;

// Six waves per SIMD (80 VGPRs, no spills).  Left alone the compiler takes 83 registers and five waves (+2 %).
// At eight waves (64 VGPRs) the kernel spills 16 registers since it carries the home slot and moves the survivors:
// 29.7 instead of 26.0 ms per 128-sample pass of the million-triangle frame (before the survivors moved, eight
// waves with ten spills had been the best: 27.5 ms).
// (Not the light-tree variant: its stack in LDS bounds it to fewer waves anyway.)
constexpr int kShadeWaves = 6;
// The instantiation for the camera rays (CAMERA: no queue, 16-byte records, the home slot is the slot) needs
// fewer registers and runs at eight waves with two spills: 5.5 instead of 6.0 ms for the first launch of that pass.
constexpr int kShadeWavesCamera = 8;
// The light-tree instantiations hold a traversal stack in LDS: LSTACK entries, 24 in general and 16 for light
// trees of at most 16 levels (the walk pushes at most one reference per level), which lets five instead of four
// workgroups share a CU (96 VGPRs): -3.7 % on the 15 360-light frame.
constexpr int kLightStackShort = 16;
// ALIGHT: the light list holds emissive analytic primitives too (their sampling and density code is kept out of
// the instantiations for scenes without them).
template <bool TREE, bool CAMERA, int LSTACK = kStack, bool ALIGHT = false>
__global__ __launch_bounds__(kBlock, TREE ? (LSTACK == kLightStackShort ? 5 : 1) : (CAMERA ? kShadeWavesCamera : kShadeWaves)) void wf_shade(DevScene S, WfView V, const uint32_t *__restrict__ queue_,
                                                    const uint32_t *__restrict__ qcount_in,
                                                    uint32_t *__restrict__ queue_out, uint32_t *__restrict__ qcount_out,
                                                    uint32_t *__restrict__ cursors, uint32_t depth, SortChoice next,
                                                    uint32_t *__restrict__ retry_count, SortChoice traced,
                                                    uint32_t *__restrict__ tune) {
    // `traced`: how wf_extend chose the queue it traced; if a regrouped copy, the hits lie at the paths' slots.
    // `next`: whether the queue filled here will be regrouped, i.e. whether its cells are wanted.
    const bool hit_by_slot = !CAMERA && sorted_queue(traced);
    const bool want_bins = sorted_queue(next);
    constexpr uint32_t kBuf = TREE ? kAppendBufTree : CAMERA ? kAppendBufCamera : kAppendBuf;
    const uint32_t *__restrict__ queue = CAMERA ? nullptr : queue_;  // camera rays: the identity queue
    __shared__ uint32_t s_buf[kBuf];
    __shared__ uint16_t s_bin[kBuf];
    __shared__ uint32_t s_count, s_base;
    __shared__ LightCacheStorage s_lights;
    __shared__ int s_stack[TREE ? LSTACK * kBlock : 1];  // light_pdf's walk of the light tree
    const LightCache LC = load_light_cache(S, s_lights, s_stack + (TREE ? threadIdx.x : 0));
    const uint32_t count = *qcount_in;
    if (blockIdx.x == 0 && threadIdx.x < kShards) cursors[threadIdx.x * kCursorStride] = 0u;  // next extend pass
    if (blockIdx.x == 0 && threadIdx.x == 0) *retry_count = 0u;
    if (threadIdx.x == 0) s_count = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lightTests = 0;
    // every thread of the workgroup runs the same number of rounds: the barriers below are uniform
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t rounds = (count + stride - 1u) / stride;
    // The queue entry of a round is fetched one round ahead; the round then waits for one trip to memory:
    // the path's record and its hit side by side.  `queue` is the order in which the previous shading pass
    // appended the survivors -- ascending slots within each run -- so the records stream instead of being
    // gathered.  When wf_extend traced a copy of that queue regrouped by cell, it left the hits at the paths'
    // slots (hit_by_slot); otherwise they lie at the queue positions, written and read densely.
    uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    uint32_t iNext = 0;
    if (j < count) iNext = queue ? queue[j] : j;
    for (uint32_t r = 0; r < rounds; ++r, j += stride) {
        bool alive = false;
        const uint32_t i = iNext;
        uint32_t cell = 0, home = i;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        if (j < count) {
            const float2 h = V.hit[hit_by_slot ? i : j];
            r0 = queue ? V.ray[2 * (size_t)i] : V.lum[i];
            const uint32_t jn = j + stride;
            PathState P;
            if (queue) {
                r1 = V.ray[2 * (size_t)i + 1];
                r2 = V.c[i];
                if (jn < count) iNext = queue[jn];
                P.o = mk(r0.x, r0.y, r0.z);
                P.d = mk(r0.w, r1.x, r1.y);
                P.T = mk(r1.z, r1.w, r2.x);
                P.rng = __float_as_uint(r2.y);
                home = __float_as_uint(r2.z);
            } else {  // camera rays: (direction, engine state), see wf_generate
                if (jn < count) iNext = jn;
                P.o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
                P.d = mk(r0.x, r0.y, r0.z);
                P.T = mk(1.f, 1.f, 1.f);
                P.rng = __float_as_uint(r0.w);
            }
            P.L = mk(0.f, 0.f, 0.f);
            P.depth = depth;  // rayDepth of every hit of this bounce (main.cpp:1188-1199)
            alive = shade_hit<TREE, ALIGHT>(S, LC, P, h.x, __float_as_int(h.y), lightTests);
            if (!queue) {  // first bounce: the camera record makes way for the radiance
                V.lum[home] = make_float4(P.L.x, P.L.y, P.L.z, 0.f);
            } else if (P.L.x != 0.f || P.L.y != 0.f || P.L.z != 0.f) {  // the path met an emitter (or went NaN): 0 + x == x, so
                const float4 l = V.lum[home];                     // the stored sum is what a carried one would be
                V.lum[home] = make_float4(l.x + P.L.x, l.y + P.L.y, l.z + P.L.z, 0.f);
            }
            if (alive) {
                r0 = make_float4(P.o.x, P.o.y, P.o.z, P.d.x);
                r1 = make_float4(P.d.y, P.d.z, P.T.x, P.T.y);
                r2 = make_float4(P.T.z, __uint_as_float(P.rng), __uint_as_float(home), 0.f);
                if (want_bins) cell = ray_bin(S, P.o);
            }
        }
        // The survivors of a wave's 64 paths move to the first of the wave's 64 slots (a wave has read all its
        // records before it stores any): the paths still alive after some bounces would otherwise lie scattered
        // over the slots they started in, every record costing whole lines to read and masked lines to
        // write.  Runs of slots stay runs: the next bounce's waves read the queue in the order written here.
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const int leader = __builtin_ctzll(mask);
            uint32_t base = 0;
            if (lane == (uint32_t)leader) base = atomicAdd(&s_count, (uint32_t)__builtin_popcountll(mask));
            base = __shfl(base, leader);
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            // lane `rank` holds a path of this round: survivors <= paths, and the paths fill the low lanes
            const uint32_t to = __shfl(i, (int)rank);
            if (alive) {
                const uint32_t at = base + rank;
                V.ray[2 * (size_t)to] = r0;
                V.ray[2 * (size_t)to + 1] = r1;
                V.c[to] = r2;
                s_buf[at] = to;
                s_bin[at] = (uint16_t)cell;
            }
        }
        // The buffer holds kFlushRounds rounds whatever survives, so the rounds between two flushes need no
        // barrier and the workgroup's waves drift apart by as much (a barrier per round, to decide the flush by
        // the buffer's fill, cost 2.4 % of the kernel).
        constexpr uint32_t kFlushRounds = kBuf / kBlock - 1u;
        if (r % kFlushRounds == kFlushRounds - 1u || r + 1 == rounds) {
            __syncthreads();
            const uint32_t n = s_count;
            if (threadIdx.x == 0) s_base = atomicAdd(qcount_out, n);
            __syncthreads();
            const uint32_t gbase = s_base;
            for (uint32_t k = threadIdx.x; k < n; k += kBlock) queue_out[gbase + k] = s_buf[k];
            if (want_bins)
                for (uint32_t k = threadIdx.x; k < n; k += kBlock) V.bin[gbase + k] = s_bin[k];
            __syncthreads();
            if (threadIdx.x == 0) s_count = 0u;
            __syncthreads();
        }
    }
    if (lightTests) add_counters(S, 0u, TraceStats{0, 0, 0, 0, 0}, lightTests, false);
    if (tune && last_workgroup(tune)) *reinterpret_cast<unsigned long long *>(tune) = wall_clock64();  // the tuner's time mark
}

// ---------------------------------------------------------------------------------------------
// Ray sorting.  After a few bounces the survivors of a pass are scattered over the whole scene, and rays
// that are traced at the same time touch unrelated parts of a tree that is far larger than an XCD's L2:
// the same node visits per ray cost 2.3 times as much at the last bounce as at the second.  A queue whose
// paths have lost their coherence (SortChoice) is therefore regrouped by the cell of the ray origin
// (ray_bin: 4096 cells in Morton order) before it is traced -- a counting sort in three small kernels,
// entirely on the device (the queue length never visits the host): histogram of the cells wf_shade wrote
// next to the queue, exclusive scan, scatter.  Only path INDICES move here.  The order inside a cell is
// whatever the atomics make it, which no result depends on: every path's arithmetic is its own and sums
// are taken per home slot.
// Measured on the million-triangle frame (64 samples per pass): wf_extend 70.2 -> 55.2 ms with 4096
// cells (512: 57.1, 64: 60.6, 8: 65.9; 2^24 cells: 52.7); sorting only inside blocks of 2K to 128K
// entries gains a fifth of that, so it is the working set per XCD that matters, not the wave.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSortGrid = 2048;      // workgroups of the histogram and scatter kernels (1024: +7 % sort time, 4096: the same)

constexpr uint32_t kSortFromBounce = 1;   // the rays leaving the first hits still follow the pixel tiles: never sorted
constexpr uint32_t kSortTile = kBins;     // entries a workgroup places at a time (4 x: +2 %, 16 x: +5 % at this grid)

__global__ __launch_bounds__(kBlock) void wf_bin_hist(const uint16_t *__restrict__ bin, const uint32_t *__restrict__ qcount,
                                                       uint32_t *__restrict__ hist, SortChoice choice) {
    __shared__ uint32_t s_hist[kBins];
    if (!sorted_queue(choice)) return;
    const uint32_t count = *qcount;
    if (blockIdx.x * kBlock >= count) return;  // uniform per workgroup
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) s_hist[b] = 0u;
    __syncthreads();
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < count; j += gridDim.x * kBlock) atomicAdd(&s_hist[bin[j]], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) {
        const uint32_t c = s_hist[b];
        if (c) atomicAdd(&hist[b], c);
    }
}

// One workgroup: cursor[b] = number of entries in cells below b; clears the histogram for its next use.
__global__ __launch_bounds__(kBlock) void wf_bin_scan(uint32_t *__restrict__ hist, uint32_t *__restrict__ cursor, SortChoice choice) {
    __shared__ uint32_t s_part[kBlock];
    if (!sorted_queue(choice)) return;
    constexpr uint32_t per = kBins / kBlock;
    uint32_t v[per], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) {
        v[k] = hist[threadIdx.x * per + k];
        hist[threadIdx.x * per + k] = 0u;
        sum += v[k];
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < kBlock; off <<= 1) {  // inclusive scan of the per-thread sums
        const uint32_t add = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) {
        cursor[threadIdx.x * per + k] = run;
        run += v[k];
    }
}

// queue_out = queue_in regrouped by cell.  A workgroup takes a tile of entries, counts them per cell in LDS,
// reserves the tile's share of every cell it met with one global atomic per cell, then reads the tile again
// and hands every entry the next free position of its cell.  (Tiles of 2048 entries with the ranks kept in
// registers took the same time: neither the global atomics nor the second read is what this kernel waits
// for, the scattered 4-byte stores are.)
__global__ __launch_bounds__(kBlock) void wf_bin_scatter(const uint32_t *__restrict__ queue_in, const uint16_t *__restrict__ bin,
                                                          const uint32_t *__restrict__ qcount, uint32_t *__restrict__ cursor,
                                                          uint32_t *__restrict__ queue_out, SortChoice choice) {
    __shared__ uint32_t s_cell[kBins];  // entries of the tile per cell, then the next free position of the cell
    if (!sorted_queue(choice)) return;
    const uint32_t count = *qcount;
    for (uint32_t t0 = blockIdx.x * kSortTile; t0 < count; t0 += gridDim.x * kSortTile) {  // uniform per workgroup
        const uint32_t t1 = (count - t0 < kSortTile) ? count : t0 + kSortTile;
        for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) s_cell[b] = 0u;
        __syncthreads();
        for (uint32_t j = t0 + threadIdx.x; j < t1; j += kBlock) atomicAdd(&s_cell[bin[j]], 1u);
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) {
            const uint32_t c = s_cell[b];
            if (c) s_cell[b] = atomicAdd(&cursor[b], c);
        }
        __syncthreads();
        for (uint32_t j = t0 + threadIdx.x; j < t1; j += kBlock) queue_out[atomicAdd(&s_cell[bin[j]], 1u)] = queue_in[j];
        __syncthreads();
    }
}

// radiance_sum[pix] (+)= sum over the batch's sample slots, in slot order (deterministic).
__global__ __launch_bounds__(kBlock) void wf_accumulate(WfView V, uint32_t W, uint32_t H, uint32_t nslots, int first,
                                                         float *__restrict__ sum) {
    const uint32_t npix = W * H;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < npix; j += gridDim.x * kBlock) {
        const uint32_t pix = frame_pixel(j, W, H);
        v3 acc = first ? mk(0.f, 0.f, 0.f) : mk(sum[3 * (size_t)pix], sum[3 * (size_t)pix + 1], sum[3 * (size_t)pix + 2]);
        for (uint32_t s = 0; s < nslots; ++s) {
            const size_t i = (size_t)s * npix + j;
            const float4 l = V.lum[i];
            acc = acc + mk(l.x, l.y, l.z);
        }
        sum[3 * (size_t)pix] = acc.x;
        sum[3 * (size_t)pix + 1] = acc.y;
        sum[3 * (size_t)pix + 2] = acc.z;
    }
}

__global__ __launch_bounds__(kBlock) void k_tonemap(const float *__restrict__ sum, uint64_t n3, float spp,
                                                     float *__restrict__ out_float, uint8_t *__restrict__ out_rgb) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (uint64_t)gridDim.x * kBlock) {
        const float c = tonemap_channel(sum[i], spp);
        if (out_float) out_float[i] = c;
        if (out_rgb) out_rgb[i] = (uint8_t)(c * 255.f);  // truncation, colorFloatToInt main.cpp:840-846
    }
}

// Self-test of the RNG restatements: thread k seeds an engine with seeds[k] (as generatePixel does,
// main.cpp:1221) and writes n values per output array at [k * n + i].  what 0: per i one uniform, one
// normal, one uint in [0, L) in that order from ONE engine (the interleaving a path produces);
// 1 / 2 / 3: only uniforms / normals / uints; 4: raw engine outputs (into ui).
__global__ void k_rng_selftest(const uint32_t *__restrict__ seeds, uint32_t nseeds, uint32_t n, uint32_t L, int what,
                               float *__restrict__ uni, float *__restrict__ nrm, uint32_t *__restrict__ ui) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nseeds) return;
    uint32_t x = rng_seed(seeds[k]);
    for (uint32_t i = 0; i < n; ++i) {
        const size_t o = (size_t)k * n + i;
        if (what == 0 || what == 1) uni[o] = rng_uniform(x);
        if (what == 0 || what == 2) nrm[o] = rng_normal(x);
        if (what == 0 || what == 3) ui[o] = rng_uint(x, L);
        if (what == 4) ui[o] = rng_next(x);
    }
}

inline uint32_t pixel_grid(uint32_t W, uint32_t H) { return ((W + 15u) / 16u) * ((H + 15u) / 16u); }
inline uint32_t ceil_blocks(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }
constexpr uint32_t kPersistentBlocks = 256u * 8u;  // 256 CUs x 8 workgroups for grid-stride kernels
constexpr uint32_t kExtendBlocks = 256u * 6u;      // 256 CUs x 6 resident workgroups (24 KB of LDS each); the wide walk: kWideWaves
constexpr uint32_t kRetryGrid = 64;                // the retry list is empty or nearly so

}  // namespace

namespace rt {

static uint32_t *tune_bounce_host(uint32_t *tune, uint32_t b) { return tune + 8u + 4u * (b < kTuneBounces ? b : kTuneBounces - 1u); }

hipError_t launch_intersect(const DevScene &S, uint64_t n, const float *o, const float *d, uint32_t *prim_id, float *t,
                            bool count, int ref_order, hipStream_t stream, KernelTimer &timer) {
    if (n == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)((n + kBlock - 1) / kBlock);
    timer.begin(kKernClosestHit, stream);
    auto kern = ref_order == 3 ? (count ? k_intersect<true, 3> : k_intersect<false, 3>)
                : ref_order == 2 ? (count ? k_intersect<true, 2> : k_intersect<false, 2>)
                : ref_order    ? (count ? k_intersect<true, 1> : k_intersect<false, 1>)
                               : (count ? k_intersect<true, 0> : k_intersect<false, 0>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, stream, S, n, o, d, prim_id, t);
    timer.end(stream);
    return hipGetLastError();
}

hipError_t launch_primary(const DevScene &S, uint32_t W, uint32_t H, uint32_t *prim_id, float *t, bool count,
                          bool ref_order, hipStream_t stream, KernelTimer &timer) {
    const uint32_t grid = pixel_grid(W, H);
    timer.begin(kKernClosestHit, stream);
    auto kern = ref_order ? (count ? k_primary<true, true> : k_primary<false, true>)
                          : (count ? k_primary<true, false> : k_primary<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, stream, S, W, H, prim_id, t);
    timer.end(stream);
    return hipGetLastError();
}

hipError_t launch_render_mega(const DevScene &S, const RenderParams &rp, float *radiance_sum, bool count, int order,
                              hipStream_t stream, KernelTimer &timer) {
    const uint32_t grid = pixel_grid(rp.W, rp.H);
    timer.begin(kKernMega, stream);
#define RT_MEGA2(O_, A_)                                                                                 \
    (S.light_tree ? (count ? k_render_mega<true, true, O_, A_> : k_render_mega<false, true, O_, A_>)     \
                  : (count ? k_render_mega<true, false, O_, A_> : k_render_mega<false, false, O_, A_>))
#define RT_MEGA(O_) (S.nlights > S.ntrilights ? RT_MEGA2(O_, true) : RT_MEGA2(O_, false))
    auto kern = order == kOrderRefFull ? RT_MEGA(kOrderRefFull) : order == kOrderRefTies ? RT_MEGA(kOrderRefTies) : RT_MEGA(kOrderLeaf);
#undef RT_MEGA
#undef RT_MEGA2
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, stream, S, rp, radiance_sum);
    timer.end(stream);
    return hipGetLastError();
}

uint64_t replay_words(const RenderParams &rp) {
    const uint64_t depth = rp.max_depth ? rp.max_depth : 1u;
    return depth * 3u * (uint64_t)pixel_grid(rp.W, rp.H) * kBlock;
}

hipError_t launch_render_replay(const DevScene &S, const RenderParams &rp, float *radiance_sum, uint32_t *rec, bool count,
                                bool ref_order, hipStream_t stream, KernelTimer &timer) {
    const uint32_t grid = pixel_grid(rp.W, rp.H);
    timer.begin(kKernMega, stream);
#define RT_REPLAY2(R_, A_)                                                                                        \
    (S.light_tree ? (count ? k_render_replay<true, true, R_, A_> : k_render_replay<false, true, R_, A_>)          \
                  : (count ? k_render_replay<true, false, R_, A_> : k_render_replay<false, false, R_, A_>))
#define RT_REPLAY(R_) (S.nlights > S.ntrilights ? RT_REPLAY2(R_, true) : RT_REPLAY2(R_, false))
    auto kern = ref_order ? RT_REPLAY(true) : RT_REPLAY(false);
#undef RT_REPLAY
#undef RT_REPLAY2
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, stream, S, rp, radiance_sum, rec, grid * kBlock);
    timer.end(stream);
    return hipGetLastError();
}

hipError_t launch_render_wavefront(const DevScene &S, const RenderParams &rp, float *radiance_sum,
                                   const WavefrontBuffers &wb, uint32_t slots, bool count, hipStream_t stream,
                                   KernelTimer &timer) {
    const uint32_t npix = rp.W * rp.H;
    // `slots` = whole frames of samples in flight per pass; the host guarantees 1 <= slots and
    // slots * npix <= wb.capacity.  A pixel's samples are added in sample order whatever the pass
    // boundaries are (wf_accumulate continues from the stored sum), so the result does not depend on slots.
    WfView V{wb.ray, wb.c, wb.lum, wb.hit, wb.bin, wb.capacity};
    uint32_t done = 0;
    bool first = true;
    while (done < rp.spp_count) {
        const uint32_t ns = (rp.spp_count - done < slots) ? rp.spp_count - done : slots;
        const uint32_t npaths = ns * npix;
        const uint32_t grid = (npaths + kBlock - 1) / kBlock < kPersistentBlocks ? (npaths + kBlock - 1) / kBlock
                                                                                 : kPersistentBlocks;
        // extend runs persistent waves: as many workgroups as stay resident
        const uint32_t eblocks = wb.wide ? 256u * (uint32_t)kWideWaves : kExtendBlocks;
        const uint32_t egrid = grid < eblocks ? grid : eblocks;
        timer.begin(kKernGenerate, stream);
        hipLaunchKernelGGL(wf_generate, dim3(grid), dim3(kBlock), 0, stream, S, rp, V, npaths, rp.spp_begin + done,
                           wb.qcount, wb.cursors);
        timer.end(stream);
        const uint32_t bounces = rp.max_depth ? rp.max_depth : 1u;  // depth 0 still traces the camera ray
        // Two queue buffers.  qu = the one holding this bounce's paths in the order wf_shade appended them
        // (-1: the identity queue of the camera rays), which is the order wf_shade walks them in; qs = the one
        // wf_extend traces, the same paths regrouped by cell when the previous bounce sorted.  wf_shade fills
        // the buffer that is not qu (wf_extend is done with it by then); a sort then writes into qu's buffer,
        // which nothing reads any more.
        int qu = -1, qs = -1;
        uint32_t *tune = wb.sort == 2 ? wb.tuner : nullptr;
        SortChoice traced{0u, wb.tuner + 10};  // how the queue of the bounce at hand was chosen
        for (uint32_t b = 0; b < bounces; ++b) {
            const uint32_t in = b & 1u, out = in ^ 1u;
            const uint32_t *qsorted = (qs < 0) ? nullptr : wb.queue[qs];
            const uint32_t *qshade = (qu < 0) ? nullptr : wb.queue[qu];
            const int fill = (qu == 0) ? 1 : 0;
            // may the queue this bounce fills be regrouped?  (the tuner keeps books for kTuneBounces bounces)
            const bool sort = wb.sort && !wb.ref_all && b >= kSortFromBounce && b + 1 < bounces && (wb.sort == 1 || b + 1 < kTuneBounces);
            const SortChoice next{sort ? (uint32_t)wb.sort : 0u, tune_bounce_host(wb.tuner, b + 1) + 2};
            timer.begin(kKernExtend, stream);
            uint32_t *retry = wb.retry, *retryCount = wb.qcount + 2;
#define RT_EXT(W_, T_)                                                                                     \
    (S.nprims ? (count ? wf_extend<true, true, W_, T_> : wf_extend<false, true, W_, T_>)                   \
              : (count ? wf_extend<true, false, W_, T_> : wf_extend<false, false, W_, T_>))
            auto ext = wb.wide ? (wb.ties ? RT_EXT(true, true) : RT_EXT(true, false))
                               : (wb.ties ? RT_EXT(false, true) : RT_EXT(false, false));
#undef RT_EXT
            if (wb.ref_all)
                hipLaunchKernelGGL(wf_list_all, dim3(grid), dim3(kBlock), 0, stream, qshade, wb.qcount + in, wb.qcount + out, retry,
                                   retryCount);
            else
                hipLaunchKernelGGL(ext, dim3(egrid), dim3(kBlock), 0, stream, S, V, qshade, qsorted, traced, wb.qcount + in,
                                   wb.qcount + out, wb.cursors, retry, retryCount, wb.stack_limit, tune, b, wb.sort_min_rays);
            if (wb.ties) {
                const int depth = (int)wb.ref_depth + 1;
                const uint32_t rblocks = wb.ref_lanes / kBlock;   // the host sized the frame buffer for this many lanes
                auto rref = wb.ref_all ? (count ? wf_extend_retry_ref<true, false> : wf_extend_retry_ref<false, false>)
                                       : (count ? wf_extend_retry_ref<true, true> : wf_extend_retry_ref<false, true>);
                hipLaunchKernelGGL(rref, dim3(rblocks), dim3(kBlock), 0, stream, S, V, retry, retryCount, b == 0 ? 1u : 0u,
                                   wb.ref_frames, depth);
            } else if (wb.wide) {
                hipLaunchKernelGGL(count ? wf_extend_retry<true> : wf_extend_retry<false>, dim3(kRetryGrid), dim3(kBlock), 0, stream, S, V,
                                   retry, retryCount, b == 0 ? 1u : 0u);
            }
            timer.end(stream);
            timer.begin(kKernShade, stream);
#define RT_SHADE(A_)                                                                                                  \
    (!S.light_tree ? (qshade ? wf_shade<false, false, kStack, A_> : wf_shade<false, true, kStack, A_>)                    \
     : S.light_tree <= (uint32_t)kLightStackShort                                                                         \
         ? (qshade ? wf_shade<true, false, kLightStackShort, A_> : wf_shade<true, true, kLightStackShort, A_>)            \
         : (qshade ? wf_shade<true, false, kStack, A_> : wf_shade<true, true, kStack, A_>))
            auto shade = S.nlights > S.ntrilights ? RT_SHADE(true) : RT_SHADE(false);
#undef RT_SHADE
            // as many workgroups as stay resident (2048 at six per CU ran the last quarter of them after the rest:
            // 25.4 instead of 24.7 ms per pass); the light-tree variant balances better with more, smaller shares
            const uint32_t sblocks = 256u * (!S.light_tree && qshade ? (uint32_t)kShadeWaves : 8u);
            hipLaunchKernelGGL(shade, dim3(grid < sblocks ? grid : sblocks), dim3(kBlock), 0, stream, S, V,
                               qshade, wb.qcount + in, wb.queue[fill], wb.qcount + out, wb.cursors,
                               rp.max_depth - (rp.max_depth ? b : 0u), next, wb.qcount + 2, traced, tune);
            timer.end(stream);
            qu = qs = fill;
            traced = next;
            if (sort) {
                const uint32_t sgrid = grid < kSortGrid ? grid : kSortGrid;
                timer.begin(kKernSort, stream);
                hipLaunchKernelGGL(wf_bin_hist, dim3(sgrid), dim3(kBlock), 0, stream, wb.bin, wb.qcount + out, wb.binhist, traced);
                hipLaunchKernelGGL(wf_bin_scan, dim3(1), dim3(kBlock), 0, stream, wb.binhist, wb.binhist + kBins, traced);
                hipLaunchKernelGGL(wf_bin_scatter, dim3(sgrid), dim3(kBlock), 0, stream, wb.queue[fill], wb.bin, wb.qcount + out,
                                   wb.binhist + kBins, wb.queue[1 - fill], traced);
                timer.end(stream);
                qs = 1 - fill;
            }
        }
        const uint32_t agrid = (npix + kBlock - 1) / kBlock < kPersistentBlocks ? (npix + kBlock - 1) / kBlock
                                                                                : kPersistentBlocks;
        timer.begin(kKernAccumulate, stream);
        hipLaunchKernelGGL(wf_accumulate, dim3(agrid), dim3(kBlock), 0, stream, V, rp.W, rp.H, ns, first ? 1 : 0, radiance_sum);
        timer.end(stream);
        first = false;
        done += ns;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (rp.spp_count == 0) return hipMemsetAsync(radiance_sum, 0, (size_t)npix * 3 * sizeof(float), stream);
    return hipGetLastError();
}

hipError_t launch_rng_selftest(const uint32_t *seeds, uint32_t nseeds, uint32_t n, uint32_t L, int what, float *uni,
                               float *nrm, uint32_t *ui, hipStream_t stream) {
    if (nseeds == 0 || n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rng_selftest, dim3((nseeds + 63u) / 64u), dim3(64), 0, stream, seeds, nseeds, n, L, what, uni, nrm,
                       ui);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *radiance_sum, uint64_t npixels, uint32_t spp, float *out_float, uint8_t *out_rgb,
                          hipStream_t stream, KernelTimer &timer) {
    const uint64_t n3 = npixels * 3;
    if (n3 == 0) return hipSuccess;
    uint64_t grid = (n3 + kBlock - 1) / kBlock;
    if (grid > kPersistentBlocks) grid = kPersistentBlocks;
    timer.begin(kKernTonemap, stream);
    hipLaunchKernelGGL(k_tonemap, dim3((uint32_t)grid), dim3(kBlock), 0, stream, radiance_sum, n3, (float)spp, out_float,
                       out_rgb);
    timer.end(stream);
    return hipGetLastError();
}

}  // namespace rt
