// rt_device.h -- device-side building blocks of the gfx950 path tracer: fp32 vector math in
// the reference's operation order, the minstd_rand / libstdc++ distribution restatements, the
// exact-arithmetic ray/triangle test, the conservative ray/box slab test, stack traversal of
// the 32-byte two-child BVH nodes, and one shading step of the reference's estimator.
//
// Everything here is compiled with -ffp-contract=off: results that decide which triangle a ray
// hits must not depend on fused multiply-adds, because the reference is built for x86-64
// without FMA (CMakeLists.txt:4, build.sh:3) and the closest-hit ids are compared bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_libm.h"

namespace rtd {

constexpr float kEps = 0.0001f;    // EPS, main.cpp:22
constexpr float kInf = 10000.f;    // INF, main.cpp:23
constexpr float kOneOverPi = 0.318309886183790671537767526745028724f;  // glm::one_over_pi<float>()
constexpr int kBlock = 256;        // threads per workgroup of every traversal kernel (4 waves)
constexpr int kStack = 24;         // traversal stack entries per lane, all in LDS (host checks tree height)
constexpr int kExitLanes = 32;     // wf_extend: node phase ends when this few walking lanes still lack a leaf
constexpr int kCounterShards = 256;   // statistics counters: shards (power of two) ...
constexpr int kCounterStride = 16;    // ... of 16 x u64 = one 128-byte line each

// ---------------------------------------------------------------------------------------------
// vec3 in glm 0.9.9.8 operation order (one rounding per operation, sums left to right)
// ---------------------------------------------------------------------------------------------
struct v3 {
    float x, y, z;
};
__device__ __forceinline__ v3 mk(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 normalize(v3 v) { return v * (1.f / __builtin_sqrtf(dot(v, v))); }

// ---------------------------------------------------------------------------------------------
// RNG: std::minstd_rand and the libstdc++ 11 distributions the reference instantiates per call
// (main.cpp:98-111).  Engine: x <- 48271 x mod (2^31 - 1); seed s -> s mod m, 0 -> 1.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rng_seed(uint32_t s) {
    uint32_t x = s % 2147483647u;
    return x == 0u ? 1u : x;
}
__device__ __forceinline__ uint32_t rng_next(uint32_t &x) {
    uint64_t p = (uint64_t)x * 48271u;
    uint32_t r = (uint32_t)(p & 0x7FFFFFFFu) + (uint32_t)(p >> 31);
    if (r >= 2147483647u) r -= 2147483647u;
    x = r;
    return r;
}
// uniform_real_distribution<float>(0,1): generate_canonical<float,24> draws once;
// float(x - 1) / 2^31 (the engine range 2147483646 rounds to 2^31 in fp32), clamped below 1.
__device__ __forceinline__ float rng_uniform(uint32_t &x) {
    float u = (float)(rng_next(x) - 1u) * 4.656612873077392578125e-10f;  // exact scaling by 2^-31
    if (u >= 1.f) u = 0.99999994f;                                      // nextafter(1, 0)
    return u;
}
// uniform_int_distribution<uint16_t>(0, L-1) over an engine range of 2147483645: the generic
// down-scaling branch (two divisions, rejection above L * scaling).
__device__ __forceinline__ uint32_t rng_uint(uint32_t &x, uint32_t L) {
    const uint32_t scaling = 2147483645u / L;
    const uint32_t past = L * scaling;
    uint32_t r;
    do {
        r = rng_next(x) - 1u;
    } while (r >= past);
    return r / scaling;
}

// The polar method below takes std::log of a float = glibc's logf; rt_logf (rt_libm.h) restates that
// algorithm so that the Gaussian samples -- and with them every later branch of a path -- come out
// bit-identical to the CPU's.
// normal_distribution<float>(0,1), Marsaglia polar; the distribution object is a temporary in
// the reference, so its cached second variate is discarded and every call runs the loop and
// returns y * mult.
__device__ __forceinline__ float rng_normal(uint32_t &x) {
    float a, b, r2;
    do {
        a = 2.0f * rng_uniform(x) - 1.0f;
        b = 2.0f * rng_uniform(x) - 1.0f;
        r2 = a * a + b * b;
    } while (r2 > 1.0f || r2 == 0.0f);
    const float mult = __builtin_sqrtf(-2.f * rt_logf(r2) / r2);
    return b * mult;
}

// Seed of the per-(pixel, sample) stream in counter mode (no reference counterpart).
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t counter_seed(uint32_t pixel, uint32_t sample, uint32_t seed) {
    uint32_t h = fmix32(pixel ^ 0x9E3779B9u);
    h = fmix32(h ^ (sample * 0x85EBCA6Bu + 0x7F4A7C15u));
    h = fmix32(h ^ seed);
    return h;
}

// ---------------------------------------------------------------------------------------------
// Scene view passed by value to every kernel
// ---------------------------------------------------------------------------------------------
struct DevMaterial {
    float4 a;  // color.xyz, bits(type)
    float4 b;  // emission.xyz, ior
};

struct DevScene {
    const uint4 *nodes;       // 2 x uint4 per node: 16-bit grid boxes of both children + two references
    const uint4 *wide;        // 3 x uint4 per node: the same tree with up to four children per node, 8-bit boxes (wf_extend walks it)
    const float4 *tris;       // 4 x float4 per triangle, leaf order
    const uint4 *lnodes;      // tree over the light triangles only (same node format, own grid)
    const float4 *lrecs;      // 4 x float4 per light in that tree's leaf order; q3 = (bits(light index), bits(leaf position), pdfConst, 0)
    const uint32_t *lights;   // the light list as record indices: the emissive triangles (leaf-order positions), then ntri + k for
                              // every emissive analytic primitive k
    const DevMaterial *mats;
    const float4 *prims;      // 3 x float4 per analytic primitive: (position.xyz, bits(type)), rotation xyzw, (param.xyz, bits(material))
    const float4 *rnodes;     // 4 x float4 per node of the tree the reference itself would build (ref_closest_hit); null when the scene is beyond its limits
    const uint32_t *ritems;   // the reference's re-ordered primitive list as record indices (triangle record, or ntri + k)
    const float4 *rreach;     // 2 x float4 per node of that tree: the box that holds everything a ray can hit below it
    unsigned long long *counters;  // kCounterShards x kCounterStride; per shard [0] segments [1] nodes [2] tri tests [3] light tests
    uint32_t nlights, nmats;             // nlights: all lights; the first ntrilights of them are triangles
    uint32_t ntrilights;
    uint32_t ntri, nprims;               // ntri = triangle RECORDS (pads included): a hit index >= ntri names analytic primitive (index - ntri)
    uint32_t npad;                       // pad records among them: the caller sees primitive k as (loader triangle count) + k
    uint32_t light_tree;                 // nonzero: the launchers pick the kernels whose light_pdf walks lnodes; the value is that tree's height
    float grid_origin[3], grid_cell[3];  // node-box plane = origin + q * cell, q a 16-bit integer
    float lgrid_origin[3], lgrid_cell[3];  // the same for lnodes
    float cam_pos[3], cam_right[3], cam_up[3], cam_fwd[3];
    float tan_x, tan_y;
    float background[3];  // InputData::backgroundColor (main.cpp:694): radiance of a ray that hits nothing
    uint32_t has_background;  // any component nonzero
    float bin_scale[3];   // (position - grid_origin) * bin_scale = coordinate in units of one ray-sorting cell
    // Reference order in the throughput kernels: a ray that starts farther than far_r from far_c (the middle of
    // everything finite in the scene; far_r = four times its extent) is answered by the reference's walk: see
    // far_origin().
    float far_c[3], far_r;
    uint32_t step_bound;  // counting launches: no ray's walk can legitimately take more rounds / node visits than this
};

// Ray-sorting cells: the node grid's extent cut into 2^kBinBits cells per axis, numbered in Morton order.
// The wavefront pipeline groups the rays of a bounce by the cell of their origin, so that the rays in
// flight at one time walk the same part of the tree and that part stays in L2 (see wf_bin_scatter).
constexpr uint32_t kBinBits = 4;
constexpr uint32_t kBins = 1u << (3u * kBinBits);
__device__ __forceinline__ uint32_t bin_spread(uint32_t v) {  // 4 bits -> every third bit
    v = (v | (v << 4)) & 0x0C3u;
    v = (v | (v << 2)) & 0x249u;
    return v;
}
static_assert(kBinBits <= 4, "bin_spread handles 4 bits per axis");
__device__ __forceinline__ uint32_t ray_bin(const DevScene &S, v3 o) {
    const float m = (float)((1u << kBinBits) - 1u);
    // fmaxf(NaN, 0) = 0: a path whose origin went NaN lands in cell 0
    const uint32_t qx = (uint32_t)__builtin_fminf(__builtin_fmaxf((o.x - S.grid_origin[0]) * S.bin_scale[0], 0.f), m);
    const uint32_t qy = (uint32_t)__builtin_fminf(__builtin_fmaxf((o.y - S.grid_origin[1]) * S.bin_scale[1], 0.f), m);
    const uint32_t qz = (uint32_t)__builtin_fminf(__builtin_fmaxf((o.z - S.grid_origin[2]) * S.bin_scale[2], 0.f), m);
    return bin_spread(qx) | (bin_spread(qy) << 1) | (bin_spread(qz) << 2);
}

struct TraceStats {
    uint32_t nodes, tris;
    // lane occupancy of the two traversal phases (COUNT builds only): iterations this lane was active
    // in, and iterations its wave ran (the maximum over the wave's lanes), summed over phases
    uint32_t nodeWave, triWave;
    // counting launches only: a walk exceeded DevScene::step_bound and was cut short.  The walk ends when a lane
    // holds neither a node nor a leaf; that rests on one invariant -- a leaf reference (negative) may sit in
    // `cur` only while `leaf` holds another one -- and a state that breaks it would spin forever.
    uint32_t runaway;
};

// Value of lane `src` (0..63) of the wave: ds_bpermute with the byte address of that lane.  __shfl() adds the
// caller's own lane number masked by the width to the index, which for the full width of 64 adds nothing and
// keeps a register alive across the whole traversal loop.
__device__ __forceinline__ int lane_value(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ float lane_value(float v, int src) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m);
        v = v > o ? v : o;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// Ray / triangle: uvt = inverse(mat3(AB, AC, -d)) * (o - A) evaluated exactly as glm does
// (Triangle::intersectPrimitive, main.cpp:478-498).  N = cross(AB, AC) is stored with the
// triangle; its components are bit-identical to the three cofactors of the inverse's last row.
// Returns true and the distance when  !(u < 0 || v < 0 || t < 0 || u + v > 1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tri_test(float4 q0, float4 q1, float4 q2, v3 o, v3 d, float &tOut) {
    const float m00 = q0.w, m01 = q1.x, m02 = q1.y;     // AB
    const float m10 = q1.z, m11 = q1.w, m12 = q2.x;     // AC
    const float m20 = -d.x, m21 = -d.y, m22 = -d.z;     // -d
    const float wx = o.x - q0.x, wy = o.y - q0.y, wz = o.z - q0.z;
    const float ca = m11 * m22 - m21 * m12;
    const float cb = m01 * m22 - m21 * m02;
    const float cc = q2.y;  // m01*m12 - m11*m02
    const float ood = 1.f / ((m00 * ca - m10 * cb) + m20 * cc);
    const float i00 = ca * ood;
    const float i10 = -(m10 * m22 - m20 * m12) * ood;
    const float i20 = (m10 * m21 - m20 * m11) * ood;
    const float i01 = -cb * ood;
    const float i11 = (m00 * m22 - m20 * m02) * ood;
    const float i21 = -(m00 * m21 - m20 * m01) * ood;
    const float i02 = cc * ood;
    const float i12 = q2.z * ood;  // -(m00*m12 - m10*m02)
    const float i22 = q2.w * ood;  // m00*m11 - m10*m01
    const float u = (i00 * wx + i10 * wy) + i20 * wz;
    const float v = (i01 * wx + i11 * wy) + i21 * wz;
    const float t = (i02 * wx + i12 * wy) + i22 * wz;
    tOut = t;
    return !(u < 0.f || v < 0.f || t < 0.f || u + v > 1.f);
}

// Closest hit (BVH::intersect, main.cpp:560-605): the minimum-t triangle among all triangles the
// ray hits.  A hit needs t < INF (1e4): the reference prunes every subtree at that distance until a
// first hit is found.
//
// The traversal is a per-lane state machine in "while-while" form: a node phase that walks inner
// nodes and only REMEMBERS the leaves it reaches (one postponed in `leaf`, a second one left in
// `cur`), then a leaf phase in which the lanes of the wave run their ray/triangle tests together.
// Testing a leaf the moment one lane reaches it would make all 64 lanes step through the
// triangle code on almost every iteration.  The stack holds inner-node and leaf references alike and
// lives in LDS as [kStack][kBlock]: lane l touches bank l % 32 only, and 24 KB per workgroup admits
// six workgroups (24 waves) per CU.  Measured alternatives on the million-triangle scene: 32 entries
// (5 workgroups per CU) ran 3-15 % slower; 16 or 24 LDS entries with a spill of deeper entries to
// private or global memory also ran 4-20 % slower (extra registers and branches in the hot loop).
// A pending entry exists only for an inner node on the current root-to-leaf path, so the builder's
// height bound (kStack inner levels) is what keeps the stack from overflowing.
constexpr int kSentinel = 0x7FFFFFFF;  // "stack is empty": never a node index, never a leaf reference
constexpr uint32_t kHelperMiss = 0xFFFFFFFFu;  // a helper lane's "no hit" (a NaN with every payload bit set)

constexpr float kTieRelative = 2.0e-5f;  // see Traversal::consider
constexpr float kTieAbsolute = 1.0e-7f;

// TIES: the walk only notes (kTieBit of triBest) that a second primitive came back at exactly the best distance;
// the caller then asks the reference's own walk, which answers in the reference's order.
template <bool COUNT, int EXIT_LANES = 0, bool HELP = false, bool WIDE = false, bool TIES = false>
struct Traversal {
    struct f2 {
        float x, y;
    };
    v3 o, d;       // origin, direction
    v3 ga, gn, gf; // plane distance t = q * ga + g: ga = cell/d, g = (grid origin - o)/d, moved towards the
                   // ray origin for near planes (gn) and away from it for far planes (gf) by the rounding bound
    uint32_t rx, ry, rz;  // 16 where the direction component is negative (the near plane is the high one), else 0
    float tBest;
    int triBest;
    int cur;   // >= 0 inner node to visit, < 0 leaf reference, kSentinel = nothing left
    int leaf;  // postponed leaf reference (< 0) or 0 = none
    int sp;
    int *stack;   // this lane's column of the LDS stack
    uint32_t *xchg;  // HELP: 128 words of LDS owned by this wave
    // WIDE: a visit pushes up to three references, so the walk is not bounded by the tree height; a push
    // beyond `limit` entries is dropped and `overflow` set: the caller discards the walk and traces the ray
    // again on the two-child tree
    int limit;
    bool overflow;

    // lane_stack: &lds[threadIdx.x]
    __device__ __forceinline__ void bind(int *lane_stack) { stack = lane_stack; }

    // 1/v with |v| clamped to >= 1e-30: tilts an axis-parallel ray by less than any box padding over
    // the scene's extent and keeps every plane distance finite
    static __device__ __forceinline__ float recip_clamped(float v) {
        if (__builtin_fabsf(v) < 1.0e-30f) v = __builtin_copysignf(1.0e-30f, v);
        return 1.f / v;
    }
    __device__ __forceinline__ void start(const DevScene &S, v3 ro, v3 rd) {
        setup(S.grid_origin, S.grid_cell, ro, rd);
    }
    __device__ __forceinline__ void setup(const float *origin, const float *cell, v3 ro, v3 rd) {
        o = ro;
        d = rd;
        const v3 r = mk(recip_clamped(rd.x), recip_clamped(rd.y), recip_clamped(rd.z));
        ga = mk(cell[0] * r.x, cell[1] * r.y, cell[2] * r.z);
        const v3 go = mk(origin[0] - ro.x, origin[1] - ro.y, origin[2] - ro.z);
        const v3 gb = mk(go.x * r.x, go.y * r.y, go.z * r.z);
        // Rounding of go, gb, ga and of the FMA itself displaces a plane by at most
        // |r| (1.5 |go| + extent) 2^-23 in t; four times that is taken off near planes and added to far
        // ones, so a far-away ray origin (large |go|) cannot produce a false miss.  65000 cells = extent.
        const float k = 4.76837158e-7f;  // 2^-21
        const v3 dl = mk(__builtin_fabsf(r.x) * (__builtin_fabsf(go.x) + 65000.f * cell[0]) * k,
                         __builtin_fabsf(r.y) * (__builtin_fabsf(go.y) + 65000.f * cell[1]) * k,
                         __builtin_fabsf(r.z) * (__builtin_fabsf(go.z) + 65000.f * cell[2]) * k);
        gn = gb - dl;
        gf = gb + dl;
        rx = (__float_as_uint(ga.x) >> 31) << 4;
        ry = (__float_as_uint(ga.y) >> 31) << 4;
        rz = (__float_as_uint(ga.z) >> 31) << 4;
        tBest = kInf;
        triBest = -1;
        // a NaN direction hits nothing (every comparison of main.cpp:1158 fails); here it must not walk
        // the tree either, because max(NaN, 0) <= min(NaN, tBest) would hold for every box
        cur = (rd.x + rd.y + rd.z) == (rd.x + rd.y + rd.z) ? 0 : kSentinel;
        leaf = 0;
        sp = 0;
        overflow = false;
    }
    __device__ __forceinline__ void idle() {  // a lane with no ray: steps are no-ops
        cur = kSentinel;
        leaf = 0;
        sp = 0;
    }
    __device__ __forceinline__ bool done() const { return cur == kSentinel && leaf == 0; }
    // Counting launches: gives the walk up and says so (TraceStats::runaway)
    __device__ __forceinline__ void abandon(TraceStats &st) {
        idle();
        st.runaway = 1u;
    }
    // Strict minimum (main.cpp:563-571).  Two triangles at exactly the same distance (a ray through a shared
    // edge, two coincident faces): the reference keeps whichever its own tree meets first.  Without TIES the one
    // earlier in leaf order wins, whatever tree is walked in whatever order, so that every kernel names the same
    // triangle; with TIES the event is noted and the caller has the ray traced in the reference's order.
    // (The mark rides in a spare bit of triBest -- record indices stay below 2^26 -- so that a strictly nearer hit
    // clears it by being assigned, and the hot loop carries no extra register.)
    // "Exactly the same distance" is too narrow a net.  Two coincident faces with different vertices get distances
    // an ulp apart, and which of them the reference returns is then decided by the rounding of ITS box test: a flat
    // node's slab is the plane itself, (plane - o) / d rounds differently from the triangle's 3x3 inverse, and
    // `distance < minDistance` (main.cpp:587, 596) cuts the subtree that holds the hit an ulp nearer.  So the mark
    // is set whenever two hits lie within tieTol of each other; the reference's own walk then answers, and for
    // rays with one clear nearest hit it answers the same as this walk, only slower.
    static constexpr int kTieBit = 1 << 30;
    // The tolerance: the reference's slab distance (plane - o) / d carries 2 ulp(coordinate) / (distance travelled
    // along that axis), 1.2e-5 of t for a ray that crosses a hundredth of the scene on it.  Far wider nets catch
    // honest geometry -- a lamp 2 mm under a ceiling -- and send a tenth of a Cornell box's rays the slow way
    // (1e-3: the frame three times as slow) for answers the fast walk gets right.
    // (Constants: a per-scene absolute term held in a register cost this kernel five spilled registers and 15 %.)
    static constexpr float kTieRel = kTieRelative;
    __device__ __forceinline__ void consider(float t, int tri) {
        if (TIES) {
            // (nothing hit yet: tBest = INF is not within the tolerance of any hit that counts, and -1 | mark = -1)
            const float gap = t - tBest;
            // (written as "not clearly apart" so that a NaN distance -- a ray in the triangle's plane -- marks too)
            const int mark = __builtin_fabsf(gap) > __builtin_fmaf(kTieRel, t, kTieAbsolute) ? 0 : kTieBit;
            if (gap < 0.f) {
                tBest = t;
                triBest = tri | mark;
            } else {
                triBest |= mark;
            }
        } else if (t < tBest) {
            tBest = t;
            triBest = tri;
        } else if (t == tBest && tri < triBest) {
            triBest = tri;
        }
    }
    __device__ __forceinline__ bool tied() const { return TIES && triBest >= 0 && (triBest & kTieBit) != 0; }
    __device__ __forceinline__ int best() const { return TIES && triBest >= 0 ? (triBest & ~kTieBit) : triBest; }
    __device__ __forceinline__ int pop() {
        if (sp == 0) return kSentinel;
        --sp;
        return stack[sp * kBlock];
    }
    __device__ __forceinline__ void push(int ref) {
        if (WIDE && sp >= limit) {
            overflow = true;
            return;
        }
        stack[sp * kBlock] = ref;
        ++sp;
    }
    // (near, far) distances of one child box given its three words, far already cut at the best hit
    __device__ __forceinline__ f2 box_span(uint32_t wx, uint32_t wy, uint32_t wz) const {
        const f2 ax = slab(wx, rx, ga.x, gn.x, gf.x), ay = slab(wy, ry, ga.y, gn.y, gf.y), az = slab(wz, rz, ga.z, gn.z, gf.z);
        return f2{__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(ax.x, ay.x), az.x), 0.f),
                  __builtin_fminf(__builtin_fminf(__builtin_fminf(ax.y, ay.y), az.y), tBest)};
    }
    static __device__ __forceinline__ void order(float &ka, int &ra, float &kb, int &rb) {  // nearer one first
        if (kb < ka) {
            const float k = ka;
            const int r = ra;
            ka = kb, ra = rb, kb = k, rb = r;
        }
    }
    // One axis of one child box, word = lo | hi << 16: rotating the word by 16 bits for a negative
    // direction puts the near plane in the low half, so no min/max per axis is needed.  Returns the
    // (near, far) plane distances.  Two scalar FMAs: v_pk_fma_f32 measured 4 % slower here.
    static __device__ __forceinline__ f2 slab(uint32_t w, uint32_t rot, float a, float bn, float bf) {
        const uint32_t u = __builtin_amdgcn_alignbit(w, w, rot);
        return f2{__builtin_fmaf((float)(u & 0xffffu), a, bn), __builtin_fmaf((float)(u >> 16), a, bf)};
    }

    // One round: walk inner nodes until this lane holds a leaf (or has nothing left), then test
    // the triangles of the leaves it holds.
    __device__ __forceinline__ void step(const DevScene &S, TraceStats &st) {
        const uint32_t nodes0 = st.nodes, tris0 = st.tris;
        while (cur >= 0 && cur != kSentinel) {
            if (WIDE) {
                // Four children per visit: the hit ones are entered nearest first, the others pushed so that
                // they pop in order of distance.  A child that is missed sorts last with an infinite key.
                // Boxes: 8 bits per plane in the node's frame (origin on the scene grid, a shift per axis):
                // plane distance = q * (ga * 2^s) + (origin * ga + g), two roundings instead of one, both
                // inside the margin of gn / gf (four times the single-rounding bound).
                const uint4 *np = S.wide + 3 * (size_t)cur;
                const uint4 n0 = np[0], n1 = np[1], n3 = np[2];
                if (COUNT) {
                    ++st.nodes;
                    if (st.nodes - nodes0 > S.step_bound) {
                        abandon(st);
                        break;
                    }
                }
                const float fox = (float)(n1.z & 0xffffu), foy = (float)(n1.z >> 16), foz = (float)(n1.w & 0xffffu);
                const float ax = __builtin_ldexpf(ga.x, (int)((n1.w >> 16) & 15u)), ay = __builtin_ldexpf(ga.y, (int)((n1.w >> 20) & 15u)),
                            az = __builtin_ldexpf(ga.z, (int)((n1.w >> 24) & 15u));
                const float bnx = __builtin_fmaf(fox, ga.x, gn.x), bfx = __builtin_fmaf(fox, ga.x, gf.x);
                const float bny = __builtin_fmaf(foy, ga.y, gn.y), bfy = __builtin_fmaf(foy, ga.y, gf.y);
                const float bnz = __builtin_fmaf(foz, ga.z, gn.z), bfz = __builtin_fmaf(foz, ga.z, gf.z);
                // near planes are the hi ones where the direction is negative
                const uint32_t wnx = rx ? n0.y : n0.x, wfx = rx ? n0.x : n0.y, wny = ry ? n0.w : n0.z, wfy = ry ? n0.z : n0.w,
                               wnz = rz ? n1.y : n1.x, wfz = rz ? n1.x : n1.y;
                const uint32_t nch = n1.w >> 28;
#define RT_WIDE_SPAN(k)                                                                                                      \
    f2 {                                                                                                                     \
        __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)((wnx >> (8 * k)) & 0xffu), ax, bnx),          \
                                                        __builtin_fmaf((float)((wny >> (8 * k)) & 0xffu), ay, bny)),         \
                                        __builtin_fmaf((float)((wnz >> (8 * k)) & 0xffu), az, bnz)),                         \
                        0.f),                                                                                                \
            __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf((float)((wfx >> (8 * k)) & 0xffu), ax, bfx),      \
                                                            __builtin_fmaf((float)((wfy >> (8 * k)) & 0xffu), ay, bfy)),     \
                                            __builtin_fmaf((float)((wfz >> (8 * k)) & 0xffu), az, bfz)),                     \
                            tBest)                                                                                           \
    }
                const f2 s0 = RT_WIDE_SPAN(0), s1 = RT_WIDE_SPAN(1), s2 = RT_WIDE_SPAN(2), s3 = RT_WIDE_SPAN(3);
#undef RT_WIDE_SPAN
                const float none = 3.0e38f;
                float k0 = s0.x <= s0.y ? s0.x : none, k1 = s1.x <= s1.y ? s1.x : none,
                      k2 = (nch > 2u && s2.x <= s2.y) ? s2.x : none, k3 = (nch > 3u && s3.x <= s3.y) ? s3.x : none;
                int r0 = (int)n3.x, r1 = (int)n3.y, r2 = (int)n3.z, r3 = (int)n3.w;
                order(k0, r0, k1, r1);
                order(k2, r2, k3, r3);
                order(k0, r0, k2, r2);
                order(k1, r1, k3, r3);
                order(k1, r1, k2, r2);
                if (k3 < none) push(r3);
                if (k2 < none) push(r2);
                if (k1 < none) push(r1);
                cur = k0 < none ? r0 : pop();
                if (cur < 0 && leaf == 0) {
                    leaf = cur;
                    cur = pop();
                }
                if (__builtin_popcountll(__ballot(leaf == 0)) <= EXIT_LANES) break;
                continue;
            }
            const uint4 *np = S.nodes + 2 * (size_t)cur;
            const uint4 n0 = np[0], n1 = np[1];
            if (COUNT) {
                ++st.nodes;
                if (st.nodes - nodes0 > S.step_bound) {
                    abandon(st);
                    break;
                }
            }
            // Conservative slab test on the node's 16-bit grid boxes.  The host rounded every box
            // outward and added two guard cells per side, far more than the rounding of these FMAs
            // (a few ulps of the scene's extent), so no box is thinner than four cells and plain
            // near <= far needs no tolerance.  x = near plane, y = far plane after the rotation.
            const f2 lx = slab(n0.x, rx, ga.x, gn.x, gf.x), ly = slab(n0.y, ry, ga.y, gn.y, gf.y),
                     lz = slab(n0.z, rz, ga.z, gn.z, gf.z);
            const f2 qx = slab(n0.w, rx, ga.x, gn.x, gf.x), qy = slab(n1.x, ry, ga.y, gn.y, gf.y),
                     qz = slab(n1.y, rz, ga.z, gn.z, gf.z);
            const float tnL = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(lx.x, ly.x), lz.x), 0.f);
            const float tfL = __builtin_fminf(__builtin_fminf(__builtin_fminf(lx.y, ly.y), lz.y), tBest);
            const float tnR = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(qx.x, qy.x), qz.x), 0.f);
            const float tfR = __builtin_fminf(__builtin_fminf(__builtin_fminf(qx.y, qy.y), qz.y), tBest);
            const bool hL = tnL <= tfL, hR = tnR <= tfR;
            const int cl = (int)n1.z, cr = (int)n1.w;
            if (hL && hR) {
                const bool leftFirst = tnL <= tnR;
                push(leftFirst ? cr : cl);
                cur = leftFirst ? cl : cr;
            } else if (hL) {
                cur = cl;
            } else if (hR) {
                cur = cr;
            } else {
                cur = pop();
            }
            if (cur < 0 && leaf == 0) {  // postpone the first leaf and keep walking
                leaf = cur;
                cur = pop();
            }
            // Leave the node phase once at most EXIT_LANES of the lanes still walking hold no leaf yet:
            // for incoherent rays, waiting for the last few stragglers kept most of the wave idle (lane
            // occupancy of this phase was 44 %); they simply resume after the triangle phase.  Queue
            // tracing, million-triangle workload, with one leaf per round: 4: 3733, 12: 3849, 20: 3959,
            // 28: 3995, 32: 3989, 40: 3949, 48: 3753, 63: 2872 Mrays/s (Cornell and the glass scene are flat
            // from 4 to 32); with helper lanes in the triangle round 20: 4213, 28: 4308, 36: 4308, 44: 4201.  The megakernel and the primary-ray kernels lose 8-9 % with 12, so they use 0.
            if (__builtin_popcountll(__ballot(leaf == 0)) <= EXIT_LANES) break;
        }
        // One leaf per round: a lane whose `cur` is a leaf too keeps it for the next round instead of making
        // the wave loop here for the few lanes in that state (queue tracing +2 % on the million-triangle
        // scene, +5 % on the Cornell box; the megakernel loses 2 %).
        if (!HELP) {
            if (leaf != 0) {
                const uint32_t ref = ~(uint32_t)leaf;
                const uint32_t first = ref >> 3, cnt = (ref & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float4 *tp = S.tris + 4 * (size_t)(first + k);
                    float t;
                    if (COUNT) ++st.tris;
                    if (tri_test(tp[0], tp[1], tp[2], o, d, t)) consider(t, (int)(first + k));
                }
                leaf = 0;
                if (cur < 0) {
                    leaf = cur;
                    cur = pop();
                }
            }
        } else {
            // Lanes without a leaf lend a hand: a two-triangle leaf's second triangle is tested by an idle
            // lane with the owner's ray, in the same pass as everybody's first triangle, and the owner
            // merges the distance afterwards (after its own, so ties fall as in the sequential order).
            const uint32_t lane = threadIdx.x & 63u;
            const bool has = leaf != 0;
            const uint32_t ref = ~(uint32_t)leaf;
            const uint32_t first = ref >> 3, cnt = has ? (ref & 7u) + 1u : 0u;
            const unsigned long long donors = __ballot(cnt == 2u), helpers = __ballot(!has);
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t npair = (uint32_t)min(__builtin_popcountll(donors), __builtin_popcountll(helpers));
            const uint32_t drank = (uint32_t)__builtin_popcountll(donors & below);
            const uint32_t hrank = (uint32_t)__builtin_popcountll(helpers & below);
            const bool lend = cnt == 2u && drank < npair, help = !has && hrank < npair;
            if (lend) xchg[drank] = lane;
            __builtin_amdgcn_wave_barrier();  // LDS serves a wave's operations in order; keep the compiler from moving them
            int src = (int)lane;
            if (help) src = (int)xchg[hrank];
            const v3 ro = mk(lane_value(o.x, src), lane_value(o.y, src), lane_value(o.z, src));
            const v3 rd = mk(lane_value(d.x, src), lane_value(d.y, src), lane_value(d.z, src));
            float t = kInf;
            bool hit = false;
            const uint32_t firstOfSrc = (uint32_t)lane_value((int)first, src);  // every lane takes part in the shuffle
            const uint32_t myTri = has ? first : firstOfSrc + 1u;
            if (has || help) {
                const float4 *tp = S.tris + 4 * (size_t)myTri;
                hit = tri_test(tp[0], tp[1], tp[2], ro, rd, t);
            }
            // (a miss travels as a bit pattern no arithmetic produces, so that whatever the test returned -- a NaN
            // distance included -- reaches consider() exactly as it does when the owner runs the test itself)
            if (help) xchg[64 + hrank] = hit ? __float_as_uint(t) : kHelperMiss;
            __builtin_amdgcn_wave_barrier();
            if (COUNT) {  // passes of the wave through the triangle test: one, two if a second triangle found no helper
                const uint32_t passes = (has ? 1u : 0u) + ((cnt == 2u && !lend) ? 1u : 0u);
                st.triWave += wave_max(passes);
            }
            if (has) {
                if (COUNT) st.tris += cnt;
                if (hit) consider(t, (int)first);
                if (lend) {
                    const uint32_t b2 = xchg[64 + drank];
                    if (b2 != kHelperMiss) consider(__uint_as_float(b2), (int)first + 1);
                } else if (cnt == 2u) {  // more two-triangle leaves than idle lanes: test it here
                    const float4 *tp = S.tris + 4 * (size_t)(first + 1u);
                    float t2;
                    if (tri_test(tp[0], tp[1], tp[2], o, d, t2)) consider(t2, (int)first + 1);
                }
                leaf = 0;
                if (cur < 0) {
                    leaf = cur;
                    cur = pop();
                }
            }
        }
        if (COUNT) {  // all lanes of the wave are converged here
            st.nodeWave += wave_max(st.nodes - nodes0);
            if (!HELP) st.triWave += wave_max(st.tris - tris0);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Analytic primitives: Plane, Ellipsoid, Box (main.cpp:242-276, 278-365, 367-464), the intersection
// arithmetic of their intersectPrimitive restated operation for operation.  The reference's loader
// never constructs them (main.cpp:1089 makes triangles only); here they enter through the scene
// description and are tested one by one after the walk through the triangle tree, as the reference
// tests its planes at the root (main.cpp:552-558, 564-570).  A hit needs t < INF like a triangle's.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 cross(v3 x, v3 y) {  // glm::cross
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// glm: quat * vec3 = v + ((uv * q.w) + uuv) * 2 with uv = cross(q.xyz, v), uuv = cross(q.xyz, uv)
__device__ __forceinline__ v3 quat_rotate(float4 q, v3 v) {
    const v3 u = mk(q.x, q.y, q.z);
    const v3 uv = cross(u, v);
    const v3 uuv = cross(u, uv);
    return v + ((uv * q.w) + uuv) * 2.f;
}
__device__ __forceinline__ v3 vdiv(v3 a, v3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ float ref_min(float a, float b) { return (b < a) ? b : a; }  // std::min
__device__ __forceinline__ float ref_max(float a, float b) { return (a < b) ? b : a; }  // std::max

struct PrimHit {
    bool hit, inside;
    float t;
    v3 normal;  // as Intersection::normal leaves intersectPrimitive
};

// WANT_NORMAL = false: only hit / t (the closest-hit loop); true: the winner's full record for shading.
template <bool WANT_NORMAL>
__device__ __forceinline__ PrimHit prim_intersect(const float4 *rec, v3 o, v3 d) {
    const float4 q0 = rec[0], rot = rec[1], q2 = rec[2];
    const int type = __float_as_int(q0.w);
    const float4 conj = make_float4(-rot.x, -rot.y, -rot.z, rot.w);  // conjRotation
    const v3 no = quat_rotate(conj, o - mk(q0.x, q0.y, q0.z));
    const v3 nd = quat_rotate(conj, d);
    const v3 par = mk(q2.x, q2.y, q2.z);
    PrimHit h;
    h.hit = false;
    h.inside = false;
    h.t = kInf;
    h.normal = mk(0.f, 0.f, 0.f);
    if (type == 0) {  // Plane::intersectPrimitive, main.cpp:246-260
        const float t = -dot(no, par) / dot(nd, par);
        h.hit = t > 0.f;
        h.t = h.hit ? t : kInf;
        if (WANT_NORMAL) h.normal = quat_rotate(rot, par);  // rotatedNormal, main.cpp:274-276
    } else if (type == 1) {  // Ellipsoid::intersectPrimitive, main.cpp:282-317
        const v3 ndr = vdiv(nd, par), nor = vdiv(no, par);
        const float a = dot(ndr, ndr);
        const float b = dot(nor, ndr) / a;
        const float c = (dot(nor, nor) - 1.f) / a;
        const float discr = b * b - c;
        if (!(discr < 0.f)) {
            const float sq = __builtin_sqrtf(discr);
            const float mind = -b - sq, maxd = -b + sq;
            if (mind < 0.f && maxd > 0.f) {
                h.t = maxd;
                h.inside = true;
                h.hit = true;
            } else if (mind > 0.f) {
                h.t = mind;
                h.hit = true;
            }
            if (WANT_NORMAL && h.hit) {
                const v3 rsq = par * par;  // radiusSq
                const v3 g = normalize(quat_rotate(rot, vdiv(no + h.t * nd, rsq)));
                h.normal = h.inside ? -g : g;
            }
        }
    } else {  // Box::intersectPrimitive, main.cpp:370-419
        const v3 t1 = vdiv(par - no, nd), t2 = vdiv(-par - no, nd);
        const float d1 = ref_max(ref_max(ref_min(t1.x, t2.x), ref_min(t1.y, t2.y)), ref_min(t1.z, t2.z));
        const float d2 = ref_min(ref_min(ref_max(t1.x, t2.x), ref_max(t1.y, t2.y)), ref_max(t1.z, t2.z));
        if (!(d1 > d2 || d2 < 0.f)) {
            h.hit = true;
            if (d1 < 0.f) {
                h.t = d2;
                h.inside = true;
            } else {
                h.t = d1;
            }
            if (WANT_NORMAL) {
                const v3 ps = vdiv(no + h.t * nd, par);
                const v3 aps = mk(__builtin_fabsf(ps.x), __builtin_fabsf(ps.y), __builtin_fabsf(ps.z));
                v3 n;
                if (aps.x > aps.y && aps.x > aps.z) n = mk(ps.x < 0.f ? -1.f : 1.f, 0.f, 0.f);
                else if (aps.y > aps.x && aps.y > aps.z) n = mk(0.f, ps.y < 0.f ? -1.f : 1.f, 0.f);
                else n = mk(0.f, 0.f, ps.z < 0.f ? -1.f : 1.f);
                n = quat_rotate(rot, n);
                h.normal = h.inside ? -n : n;
            }
        }
    }
    return h;
}

// Folds the analytic primitives into a finished triangle walk: strict minimum, in list order.  Returns whether
// the reference's answer may depend on its own order here: a primitive came back at exactly the best distance,
// or with a distance that is not below INF (a plane met at a grazing angle: the reference tests planes first, at
// the root, with no distance cut, and keeps such a hit unless something nearer turns up, main.cpp:564-570).
__device__ __forceinline__ bool analytic_closest(const DevScene &S, v3 o, v3 d, float &tBest, int &triBest) {
    bool ask = false;
    for (uint32_t k = 0; k < S.nprims; ++k) {
        const PrimHit h = prim_intersect<false>(S.prims + 3 * (size_t)k, o, d);
        // (two hits within the tolerance of Traversal::consider: the reference's order decides)
        if (h.hit && h.t < tBest) {
            ask = triBest >= 0 && tBest - h.t <= __builtin_fmaf(kTieRelative, tBest, kTieAbsolute);
            tBest = h.t;
            triBest = (int)(S.ntri + k);
        } else if (h.hit && (h.t - tBest <= __builtin_fmaf(kTieRelative, h.t, kTieAbsolute) || (triBest < 0 && !(h.t < kInf)))) {
            ask = true;
        }
    }
    return ask;
}

}  // namespace rtd
#include "rt_refwalk.h"
namespace rtd {

// Reference order in the throughput kernels: a ray that starts far outside the scene -- it comes back from an
// analytic plane it met hundreds of scene sizes away -- is answered by the reference's walk, not by the fast one.
// Out there the reference's own intersection arithmetic is noise (its ellipsoid test "hits" a sphere of radius
// 0.15 from 1.4 units beside it at 4000 units distance, see ref_reach_missed), and WHICH of those phantom hits
// the reference reports is decided by where its doubled slabs let its walk go: only that walk knows.
__device__ __forceinline__ bool far_origin(const DevScene &S, v3 o) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x - S.far_c[0]), __builtin_fabsf(o.y - S.far_c[1])),
                           __builtin_fabsf(o.z - S.far_c[2])) > S.far_r;
}

// How a kernel answers where the reference's result depends on its traversal order:
//   kOrderLeaf       own tree, equal distances go to the triangle earlier in leaf order
//   kOrderRefTies    own tree; a ray that met two primitives at the best distance is traced again by ref_closest_hit
//   kOrderRefFull    ref_closest_hit for every ray (validation paths)
enum HitOrder { kOrderLeaf = 0, kOrderRefTies = 1, kOrderRefFull = 2 };

template <bool COUNT, int ORDER = kOrderLeaf>
__device__ __forceinline__ void closest_hit(const DevScene &S, v3 o, v3 d, int *stack, float &tBest, int &triBest,
                                            TraceStats &st) {
    if (ORDER == kOrderRefFull) {
        ref_closest_hit<COUNT>(S, o, d, tBest, triBest, st);
        return;
    }
    Traversal<COUNT, 0, false, false, ORDER == kOrderRefTies> T;
    T.bind(stack);
    T.start(S, o, d);
    uint32_t rounds = 0;
    while (!T.done()) {
        T.step(S, st);
        if (COUNT && ++rounds > S.step_bound) T.abandon(st);
    }
    tBest = T.tBest;
    triBest = T.best();
    bool tie = T.tied();
    if (S.nprims) tie = analytic_closest(S, o, d, tBest, triBest) || tie;
    if (ORDER == kOrderRefTies && (tie || far_origin(S, o))) ref_closest_hit<COUNT, true>(S, o, d, tBest, triBest, st);
}

// ---------------------------------------------------------------------------------------------
// Camera: generateSample's ray (main.cpp:1212-1215); consumes two uniforms.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 primary_direction(const DevScene &S, uint32_t px, uint32_t py, uint32_t W, uint32_t H,
                                                uint32_t &rng) {
    const float u1 = rng_uniform(rng);
    const float x = (2.f * ((float)px + u1) / (float)W - 1.f) * S.tan_x;
    const float u2 = rng_uniform(rng);
    const float y = -(2.f * ((float)py + u2) / (float)H - 1.f) * S.tan_y;
    const v3 right = mk(S.cam_right[0], S.cam_right[1], S.cam_right[2]);
    const v3 up = mk(S.cam_up[0], S.cam_up[1], S.cam_up[2]);
    const v3 fwd = mk(S.cam_fwd[0], S.cam_fwd[1], S.cam_fwd[2]);
    return normalize(x * right + y * up + fwd);
}

// ---------------------------------------------------------------------------------------------
// One path vertex of the reference estimator (applyLight*, main.cpp:1128-1210).  Every hit spawns
// exactly one continuation ray, so a sample is a chain of vertices.  sample_vertex() does what is
// common to every way of evaluating that chain: the random decisions, the next ray, and a record
// of HOW the vertex combines the radiance L_next coming back along that ray:
//   kOpNone      depth 0, or a miss under the reference's black background:      L = 0
//   kOpBackground  miss under a nonzero background colour: L = backgroundColor   (main.cpp:1190-1192)
//   kOpEmit      depth 1, or a diffuse vertex whose sampled direction is rejected:   L = e
//   kOpDiffuse   L = color * L_next * (1/pi) * wn / pdf + e      (main.cpp:1139-1140)
//   kOpMetal     L = color * L_next + e                          (main.cpp:1146)
//   kOpPass      L = L_next + e      (dielectric reflection, refraction on the way out)
//   kOpTint      L = L_next * color + e                          (refraction on the way in, 1181-1185)
// The throughput kernels fold the chain forward (L += T e; T *= k: same estimator and random
// numbers, different rounding of the same products); the replay kernel stores the records and
// evaluates the recursion backwards in the reference's own operation order.
// ---------------------------------------------------------------------------------------------
enum VertexOp { kOpNone = 0, kOpEmit, kOpDiffuse, kOpMetal, kOpPass, kOpTint, kOpBackground };

struct Vertex {
    int op;
    uint32_t mat;
    float wn, pdf;  // kOpDiffuse only
};

struct PathState {
    v3 o, d;      // current ray
    v3 T;         // throughput (forward evaluation)
    v3 L;         // radiance gathered so far by this sample (forward evaluation)
    uint32_t rng; // engine state
    uint32_t depth;  // rayDepth the next hit is shaded with (main.cpp:1188-1199)
};

// The first kLdsLights light triangles, copied into LDS by every workgroup of a shading kernel:
// every diffuse vertex tests its new direction against ALL lights (light_pdf below), the same few
// records for every lane, and fetching them through the vector-memory path was more than half of
// the shading kernel's load instructions.
// The material table's first kLdsMats entries ride along (two 16-byte gathers per vertex otherwise).
constexpr uint32_t kLdsLights = 64;
constexpr uint32_t kLdsMats = 64;
struct LightCache {
    const float4 *recs;  // LDS: 4 x float4 per cached light, light-list order
    const int *ids;      // LDS: leaf-order triangle position of each cached light
    const float4 *sph;   // LDS: a sphere around each cached light (centre, radius), generously padded
    const float4 *mats;  // LDS: 2 x float4 per cached material
    uint32_t n;          // lights cached: min(S.nlights, kLdsLights)
    uint32_t nm;         // materials cached: min(S.nmats, kLdsMats)
    int *stack;          // this lane's column of a [kStack][kBlock] LDS array, free while a vertex is shaded
};
struct LightCacheStorage {
    float4 recs[4 * kLdsLights];
    float4 mats[2 * kLdsMats];
    float4 sph[kLdsLights];
    int ids[kLdsLights];
};

// Call from all threads of the workgroup; ends with a barrier.
__device__ __forceinline__ LightCache load_light_cache(const DevScene &S, LightCacheStorage &st, int *lane_stack) {
    const uint32_t n = S.ntrilights < kLdsLights ? S.ntrilights : kLdsLights;
    const uint32_t nm = S.nmats < kLdsMats ? S.nmats : kLdsMats;
    for (uint32_t e = threadIdx.x; e < 4u * n; e += kBlock) st.recs[e] = S.tris[4 * (size_t)S.lights[e >> 2] + (e & 3u)];
    for (uint32_t e = threadIdx.x; e < 2u * nm; e += kBlock) st.mats[e] = reinterpret_cast<const float4 *>(S.mats)[e];
    for (uint32_t k = threadIdx.x; k < n; k += kBlock) {
        const uint32_t lt = S.lights[k];
        st.ids[k] = (int)lt;
        // sphere through the farthest vertex around the centroid, 1 % + 1e-4 larger: whatever point the exact
        // test can accept (it rounds at the 1e-6 level of the triangle's size and of the hit distance) is inside
        const float4 q0 = S.tris[4 * (size_t)lt], q1 = S.tris[4 * (size_t)lt + 1], q2 = S.tris[4 * (size_t)lt + 2];
        const v3 A = mk(q0.x, q0.y, q0.z), AB = mk(q0.w, q1.x, q1.y), AC = mk(q1.z, q1.w, q2.x);
        const v3 g = (AB + AC) * (1.f / 3.f);  // centroid - A
        const v3 gb = g - AB, gc = g - AC;
        const float r2 = __builtin_fmaxf(dot(g, g), __builtin_fmaxf(dot(gb, gb), dot(gc, gc)));
        const v3 c = A + g;
        st.sph[k] = make_float4(c.x, c.y, c.z, __builtin_sqrtf(r2) * 1.01f + 1.0e-4f);
    }
    __syncthreads();
    return LightCache{st.recs, st.ids, st.sph, st.mats, n, nm, lane_stack};
}

__device__ __forceinline__ DevMaterial material(const DevScene &S, const LightCache &LC, uint32_t m) {
    if (m < LC.nm) return DevMaterial{LC.mats[2 * m], LC.mats[2 * m + 1]};
    return S.mats[m];
}

// ---------------------------------------------------------------------------------------------
// Emissive analytic primitives (Ellipsoid, Box): samplePoint and pdf, with Primitive::intersectFull / pdfFull
// (main.cpp:218-239, 320-356, 425-455), operation for operation -- including what is odd about them: the
// ellipsoid's second density is computed from the FIRST intersection point (main.cpp:347-348), and a second
// intersection that is missed contributes through its default point (0, 0, 0) and zero normal (a division by
// zero: the sum is then inf or NaN and LightSurface::pdf returns 0, main.cpp:779-781).
// ---------------------------------------------------------------------------------------------
struct FullHit {  // the fields of Intersection that pdfFull and intersectFull read
    bool hit, inside;
    v3 p, normal, pPlus, pMinus;
};
__device__ __forceinline__ FullHit full_intersect(const float4 *rec, v3 o, v3 d) {
    const PrimHit h = prim_intersect<true>(rec, o, d);
    FullHit f;
    f.hit = h.hit;
    f.inside = h.inside;
    f.p = mk(0.f, 0.f, 0.f);
    f.normal = mk(0.f, 0.f, 0.f);
    f.pPlus = f.p;
    f.pMinus = f.p;
    if (h.hit) {  // Intersection::update, main.cpp:84-95
        f.normal = h.normal;
        f.p = o + h.t * d;
        const v3 nEps = h.normal * kEps;
        f.pPlus = f.p + nEps;
        f.pMinus = f.p - nEps;
    }
    return f;
}
__device__ __noinline__ float analytic_light_pdf(const float4 *rec, v3 o, v3 d) {
    const FullHit i1 = full_intersect(rec, o, d);
    if (!i1.hit) return 0.f;
    const FullHit i2 = i1.inside ? full_intersect(rec, i1.pPlus, -d) : full_intersect(rec, i1.pMinus, d);
    const float4 q0 = rec[0], rot = rec[1], q2 = rec[2];
    const v3 par = mk(q2.x, q2.y, q2.z);
    float p1, p2;
    if (__float_as_int(q0.w) == 1) {  // Ellipsoid::pdf
        float rs = par.x * par.y * par.z;
        rs = rs * rs;
        const float rxs = par.x * par.x, rys = par.y * par.y, rzs = par.z * par.z;
        const float4 conj = make_float4(-rot.x, -rot.y, -rot.z, rot.w);
        const v3 pos1 = quat_rotate(conj, i1.p - mk(q0.x, q0.y, q0.z));
        p1 = kOneOverPi / 4.f / __builtin_sqrtf(pos1.x * pos1.x * rs / rxs + pos1.y * pos1.y * rs / rys + pos1.z * pos1.z * rs / rzs);
        p2 = p1;  // pos2 = pos1, as the reference computes it
    } else {  // Box::pdf
        p1 = 1.f / 8.f / (par.y * par.z + par.x * par.z + par.x * par.y);
        p2 = p1;
    }
    const v3 r1 = i1.p - o, r2 = i2.p - o;  // Primitive::pdfFull
    const float fp1 = p1 * dot(r1, r1) / __builtin_fabsf(dot(normalize(r1), i1.normal));
    const float fp2 = p2 * dot(r2, r2) / __builtin_fabsf(dot(normalize(r2), i2.normal));
    return fp1 + fp2;
}
// Ellipsoid::samplePoint (3 normals) / Box::samplePoint (4 uniforms)
__device__ __noinline__ v3 analytic_sample_point(const float4 *rec, uint32_t &rng) {
    const float4 q0 = rec[0], rot = rec[1], q2 = rec[2];
    const v3 par = mk(q2.x, q2.y, q2.z);
    v3 point;
    if (__float_as_int(q0.w) == 1) {
        const float x = rng_normal(rng);
        const float y = rng_normal(rng);
        const float z = rng_normal(rng);
        point = normalize(mk(x, y, z));
        point = mk(point.x * par.x, point.y * par.y, point.z * par.z);
    } else {
        const float wx = par.y * par.z, wy = par.x * par.z, wz = par.x * par.y;
        const float u = rng_uniform(rng) * ((wx + wy + wz) - 0.f) + 0.f;  // uniform_real_distribution(a, b): u (b - a) + a
        const float side = 2.f * ((rng_uniform(rng) < 0.5f ? 1.f : 0.f) - 0.5f);
        const float c1 = (rng_uniform(rng) * (2.f - 0.f) + 0.f) - 1.f;
        const float c2 = (rng_uniform(rng) * (2.f - 0.f) + 0.f) - 1.f;
        if (u < wx) point = mk(side * par.x, c1 * par.y, c2 * par.z);
        else if (u < wx + wy) point = mk(c1 * par.x, side * par.y, c2 * par.z);
        else point = mk(c1 * par.x, c2 * par.y, side * par.z);
    }
    return quat_rotate(rot, point) + mk(q0.x, q0.y, q0.z);
}
// The analytic lights' share of LightSurface::pdf: they follow the triangles in the light list, so their terms
// are added after the triangles', in list order.  (Out of line, like the two functions above: a scene without
// such lights must not pay for their registers -- inlined into wf_shade they cost it 15 spilled registers and
// 16 % of its time.  The shading kernels take ALIGHT = false for such scenes and hold no call at all.)
__device__ __noinline__ float add_analytic_lights(const DevScene &S, float ps, v3 x, v3 w, int self, uint32_t &tests) {
    for (uint32_t k = S.ntrilights; k < S.nlights; ++k) {
        const uint32_t rec = S.lights[k];
        if ((int)rec == self) continue;
        ++tests;
        ps += analytic_light_pdf(S.prims + 3 * (size_t)(rec - S.ntri), x, w);
    }
    return ps;
}

// Triangle::pdf (main.cpp:510-517) of a light triangle the ray x + t w hits
__device__ __forceinline__ float light_term(float4 q2, float4 q3, v3 x, v3 w, float t) {
    v3 n = normalize(mk(q2.y, q2.z, q2.w));
    if (dot(n, w) > 0.f) n = n * -1.f;
    const v3 ip = x + t * w;
    const v3 r = ip - x;
    return q3.z * dot(r, r) / __builtin_fabsf(dot(normalize(r), n));
}

// LightSurface::pdf (main.cpp:771-783): sum over all lights except the surface's own triangle
// of Triangle::pdf, in light-list order, no occlusion test, divided by the light count.  A light
// the ray misses adds nothing (the reference skips it; here its term is never added either).
// The same sum through the tree over the light triangles.  Every triangle the ray hits counts, so the
// walk prunes nothing: it visits every node whose box the ray crosses, in "while-while" form (walk
// inner nodes until the lane holds a leaf, then test that leaf's triangles together with the other
// lanes).  The stack (this lane's LDS column) holds inner-node and leaf references alike and is
// bounded by the tree height, like the main traversal's.
// The reference adds the terms in light-list order and float addition does not associate, so one walk
// keeps the (up to four) hits with the smallest light indices above `kmin`, sorted; their terms are
// added in that order, and if the walk saw more hits it is repeated with kmin = the last index
// added.  Nearly every direction hits at most two lights (in and out of one emissive solid).
__device__ __forceinline__ float light_pdf_tree(const DevScene &S, int *stack, v3 x, v3 w, int self, uint32_t &tests) {
    typedef Traversal<false>::f2 f2;
    Traversal<false> T;
    T.setup(S.lgrid_origin, S.lgrid_cell, x, w);
    float ps = 0.f;
    int kmin = -1;
    bool more = T.cur != kSentinel;
    while (more) {
        more = false;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int k0 = kSentinel, k1 = kSentinel, k2 = kSentinel, k3 = kSentinel;
        int sp = 0, cur = 0;
        for (;;) {
            while (cur >= 0 && cur != kSentinel) {
                const uint4 n0 = S.lnodes[2 * (size_t)cur], n1 = S.lnodes[2 * (size_t)cur + 1];
                const f2 lx = T.slab(n0.x, T.rx, T.ga.x, T.gn.x, T.gf.x), ly = T.slab(n0.y, T.ry, T.ga.y, T.gn.y, T.gf.y),
                         lz = T.slab(n0.z, T.rz, T.ga.z, T.gn.z, T.gf.z);
                const f2 qx = T.slab(n0.w, T.rx, T.ga.x, T.gn.x, T.gf.x), qy = T.slab(n1.x, T.ry, T.ga.y, T.gn.y, T.gf.y),
                         qz = T.slab(n1.y, T.rz, T.ga.z, T.gn.z, T.gf.z);
                const bool hL = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(lx.x, ly.x), lz.x), 0.f) <=
                                __builtin_fminf(__builtin_fminf(lx.y, ly.y), lz.y);
                const bool hR = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(qx.x, qy.x), qz.x), 0.f) <=
                                __builtin_fminf(__builtin_fminf(qx.y, qy.y), qz.y);
                const int cl = (int)n1.z, cr = (int)n1.w;
                if (hL && hR) {
                    stack[sp++ * kBlock] = cr;
                    cur = cl;
                } else if (hL) {
                    cur = cl;
                } else if (hR) {
                    cur = cr;
                } else {
                    cur = sp ? stack[--sp * kBlock] : kSentinel;
                }
            }
            if (cur == kSentinel) break;
            const uint32_t ref = ~(uint32_t)cur;
            const uint32_t first = ref >> 3, n = (ref & 7u) + 1u;
            for (uint32_t j = 0; j < n; ++j) {
                const float4 *tp = S.lrecs + 4 * (size_t)(first + j);
                const float4 q3 = tp[3];
                const int k = __float_as_int(q3.x);
                if (__float_as_int(q3.y) == self || k <= kmin) continue;
                ++tests;
                float t;
                if (!tri_test(tp[0], tp[1], tp[2], x, w, t)) continue;
                if (k >= k3) {  // the four kept are all smaller: leave it to a later walk
                    more = true;
                    continue;
                }
                if (k3 != kSentinel) more = true;  // the largest kept one drops out
                k3 = k, t3 = light_term(tp[2], q3, x, w, t);
#define RT_ORDER(ka, ta, kb, tb)            \
    if (kb < ka) {                          \
        const int ki = ka;                  \
        const float ti = ta;                \
        ka = kb, ta = tb, kb = ki, tb = ti; \
    }
                RT_ORDER(k2, t2, k3, t3) RT_ORDER(k1, t1, k2, t2) RT_ORDER(k0, t0, k1, t1)
#undef RT_ORDER
            }
            cur = sp ? stack[--sp * kBlock] : kSentinel;
        }
        ps = (((ps + t0) + t1) + t2) + t3;  // empty slots hold 0 and sort last: adding them changes nothing
        kmin = k3;
    }
    return ps;
}

template <bool TREE, bool ALIGHT>
__device__ __forceinline__ float light_pdf(const DevScene &S, const LightCache &LC, v3 x, v3 w, int self,
                                           uint32_t &tests) {
    float ps = 0.f;
    if (TREE) {
        ps = light_pdf_tree(S, LC.stack, x, w, self, tests);
        if (ALIGHT && S.nlights > S.ntrilights) ps = add_analytic_lights(S, ps, x, w, self, tests);
        if (ps < kEps || __builtin_isnan(ps) || __builtin_isinf(ps)) return 0.f;
        return ps / (float)S.nlights;
    }
    // A ray that hits a light triangle crosses the sphere around it, and a light that is missed adds no term,
    // so only the lights whose sphere the ray crosses get the exact test -- each lane its own candidates, in
    // light-list order like the reference's loop: the sum is the same float.  Nearly every direction misses
    // nearly every light; all 64 lanes stepping through the exact test of every light was half of the
    // shading kernel's arithmetic.  `tests` counts what the reference tests: every light but the surface's own.
    unsigned long long cand = 0ull;
    for (uint32_t k = 0; k < LC.n; ++k) {
        const float4 sp = LC.sph[k];
        const v3 v = mk(sp.x - x.x, sp.y - x.y, sp.z - x.z);
        const float vv = dot(v, v), b = dot(v, w), d2 = vv - b * b;
        const bool other = LC.ids[k] != self;
        tests += other ? 1u : 0u;
        // vv - b^2 cancels: its rounding error grows with vv (a point 10^4 away on an analytic plane), so
        // the radius gives way by 4e-6 vv (ten times the bound).  Written so that a NaN keeps the light: the
        // exact test decides.
        const float slack = 4.0e-6f * vv;
        if (other && !(d2 > sp.w * sp.w + slack) && !(b < -(sp.w + slack))) cand |= 1ull << k;
    }
    while (cand) {
        const uint32_t k = (uint32_t)__builtin_ctzll(cand);
        cand &= cand - 1ull;
        const float4 *tp = LC.recs + 4 * k;
        float t;
        if (tri_test(tp[0], tp[1], tp[2], x, w, t)) ps += light_term(tp[2], tp[3], x, w, t);
    }
    for (uint32_t k = LC.n; k < S.ntrilights; ++k) {
        const int lt = (int)S.lights[k];
        if (lt == self) continue;
        ++tests;
        const float4 *tp = S.tris + 4 * (size_t)lt;
        float t;
        if (tri_test(tp[0], tp[1], tp[2], x, w, t)) ps += light_term(tp[2], tp[3], x, w, t);
    }
    if (ALIGHT && S.nlights > S.ntrilights) ps = add_analytic_lights(S, ps, x, w, self, tests);
    if (ps < kEps || __builtin_isnan(ps) || __builtin_isinf(ps)) return 0.f;
    return ps / (float)S.nlights;
}

// Processes the hit (t, tri) of ray (P.o, P.d): fills V, and when it returns true the path
// continues with the new ray in P.o / P.d at depth P.depth - 1.  Touches P.o, P.d, P.rng, P.depth only.
// ALIGHT: the light list may hold emissive analytic primitives (false: it holds triangles only).
template <bool TREE, bool ALIGHT = true>
__device__ __forceinline__ bool sample_vertex(const DevScene &S, const LightCache &LC, PathState &P, float t, int tri,
                                              Vertex &V, uint32_t &lightTests) {
    V.op = kOpNone;
    V.mat = 0u;
    V.wn = 0.f;
    V.pdf = 1.f;
    if (tri < 0) {                    // miss: backgroundColor, (0,0,0) unless the caller set one (main.cpp:694, 1190)
        if (S.has_background) V.op = kOpBackground;
        return false;
    }
    if (P.depth == 0) return false;   // main.cpp:1195
    v3 n;
    bool inside;
    if ((uint32_t)tri >= S.ntri) {    // an analytic primitive: its normal and side, recomputed for the winner
        const float4 *rec = S.prims + 3 * (size_t)((uint32_t)tri - S.ntri);
        V.mat = __float_as_uint(rec[2].w);
        V.op = kOpEmit;
        if (P.depth == 1) return false;
        const PrimHit h = prim_intersect<true>(rec, P.o, P.d);
        n = h.normal;
        inside = h.inside;
    } else {
        const float4 *tp = S.tris + 4 * (size_t)tri;
        const float4 q2 = tp[2], q3 = tp[3];
        V.mat = __float_as_uint(q3.y);
        V.op = kOpEmit;
        if (P.depth == 1) return false;   // main.cpp:1197
        // Triangle::intersectPrimitive's normal (main.cpp:489-494) for the winning triangle
        n = normalize(mk(q2.y, q2.z, q2.w));
        inside = dot(n, P.d) > 0.f;
        if (inside) n = n * -1.f;
    }
    const DevMaterial M = material(S, LC, V.mat);
    const int type = __float_as_int(M.a.w);

    // Intersection::update (main.cpp:84-95)
    const float nl = -dot(n, P.d);
    const v3 p = P.o + t * P.d;
    const v3 nEps = n * kEps;
    const v3 pPlus = p + nEps;
    const v3 pMinus = p - nEps;

    if (type == 0) {  // applyLightDiffuser, main.cpp:1128-1141, Mix 793-804
        v3 w;
        if (rng_uniform(P.rng) < 0.5f) {  // Cosine::sample, main.cpp:739-746
            const float gx = rng_normal(P.rng);
            const float gy = rng_normal(P.rng);
            const float gz = rng_normal(P.rng);
            w = normalize(normalize(mk(gx, gy, gz)) + n * (1.f + kEps));
        } else if (S.nlights == 0) {      // LightSurface::sample, main.cpp:764-769
            w = mk(0.f, 0.f, 0.f);
        } else {
            const uint32_t k = rng_uint(P.rng, S.nlights);
            v3 pt;
            if (ALIGHT && k >= S.ntrilights) {  // an emissive ellipsoid or box
                pt = analytic_sample_point(S.prims + 3 * (size_t)(S.lights[k] - S.ntri), P.rng);
            } else {
                float4 l0, l1, l2;
                if (k < LC.n) {
                    l0 = LC.recs[4 * k], l1 = LC.recs[4 * k + 1], l2 = LC.recs[4 * k + 2];
                } else {
                    const float4 *lp = S.tris + 4 * (size_t)S.lights[k];
                    l0 = lp[0], l1 = lp[1], l2 = lp[2];
                }
                float u = rng_uniform(P.rng);  // Triangle::samplePoint, main.cpp:500-508
                float v = rng_uniform(P.rng);
                if (u + v > 1.f) {
                    u = 1.f - u;
                    v = 1.f - v;
                }
                pt = mk(l0.x, l0.y, l0.z) + u * mk(l0.w, l1.x, l1.y) + v * mk(l1.z, l1.w, l2.x);
            }
            w = normalize(pt - pPlus);
        }
        const float dn = dot(w, n);  // Cosine::pdf, main.cpp:748-754
        const float pc = (dn < kEps) ? 0.f : dn * kOneOverPi;
        const float pdf = 0.5f * pc + 0.5f * light_pdf<TREE, ALIGHT>(S, LC, pPlus, w, tri, lightTests);
        const float wn = dot(w, n);
        if (wn < 0.f || pdf < kEps) return false;  // V.op stays kOpEmit, main.cpp:1134-1136
        V.op = kOpDiffuse;
        V.wn = wn;
        V.pdf = pdf;
        P.o = pPlus;
        P.d = w;
    } else if (type == 1) {  // applyLightMetallic + getReflectedLight, main.cpp:1120-1126, 1143-1147
        V.op = kOpMetal;
        P.o = pPlus;
        P.d = normalize(P.d + 2.f * n * nl);
    } else {  // applyLightDielectric, main.cpp:1149-1186
        float n1 = 1.f, n2 = M.b.w;
        if (inside) {
            n1 = n2;
            n2 = 1.f;
        }
        const float n12 = n1 / n2;
        const float sn = n12 * __builtin_sqrtf(1.f - nl * nl);
        bool reflect = sn > 1.f;
        if (!reflect) {
            float r0 = (n1 - n2) / (n1 + n2);
            r0 *= r0;
            const float mnl = 1.f - nl;
            const float mnlsq = mnl * mnl;
            const float r = r0 + (1.f - r0) * mnlsq * mnlsq * mnl;
            reflect = rng_uniform(P.rng) < r;
        }
        if (reflect) {
            V.op = kOpPass;
            P.o = pPlus;
            P.d = normalize(P.d + 2.f * n * nl);
        } else {
            const v3 rd = n12 * P.d + (n12 * nl - __builtin_sqrtf(1.f - sn * sn)) * n;
            P.d = normalize(rd);
            P.o = pMinus;
            V.op = inside ? kOpPass : kOpTint;
        }
    }
    --P.depth;
    return true;
}

// Forward evaluation of one vertex: L += T * e, T *= k.
__device__ __forceinline__ void fold_forward(const DevScene &S, const LightCache &LC, PathState &P, const Vertex &V) {
    if (V.op == kOpNone) return;
    if (V.op == kOpBackground) {
        P.L = P.L + P.T * mk(S.background[0], S.background[1], S.background[2]);
        return;
    }
    const DevMaterial M = material(S, LC, V.mat);
    const v3 color = mk(M.a.x, M.a.y, M.a.z);
    P.L = P.L + P.T * mk(M.b.x, M.b.y, M.b.z);
    if (V.op == kOpDiffuse)
        P.T = P.T * mk(color.x * kOneOverPi * V.wn / V.pdf, color.y * kOneOverPi * V.wn / V.pdf,
                       color.z * kOneOverPi * V.wn / V.pdf);
    else if (V.op == kOpMetal || V.op == kOpTint)
        P.T = P.T * color;
}

// Shades the hit (t, tri) of ray (P.o, P.d) in forward form.  Returns true when the path continues.
template <bool TREE, bool ALIGHT = true>
__device__ __forceinline__ bool shade_hit(const DevScene &S, const LightCache &LC, PathState &P, float t, int tri,
                                          uint32_t &lightTests) {
    Vertex V;
    const bool alive = sample_vertex<TREE, ALIGHT>(S, LC, P, t, tri, V, lightTests);
    fold_forward(S, LC, P, V);
    return alive;
}

// Backward evaluation of one vertex in the reference's operation order: returns this vertex's
// radiance given the radiance `l` that came back along its continuation ray.
__device__ __forceinline__ v3 fold_backward(const DevScene &S, const LightCache &LC, int op, uint32_t mat, float wn,
                                            float pdf, v3 l) {
    if (op == kOpNone) return mk(0.f, 0.f, 0.f);
    if (op == kOpBackground) return mk(S.background[0], S.background[1], S.background[2]);
    const DevMaterial M = material(S, LC, mat);
    const v3 color = mk(M.a.x, M.a.y, M.a.z);
    const v3 e = mk(M.b.x, M.b.y, M.b.z);
    if (op == kOpEmit) return e;
    if (op == kOpDiffuse) {  // color * l * one_over_pi * wn / p + emission, left to right
        v3 c = color * l;
        c = c * kOneOverPi;
        c = c * wn;
        c = mk(c.x / pdf, c.y / pdf, c.z / pdf);
        return c + e;
    }
    if (op == kOpMetal) return color * l + e;
    if (op == kOpTint) return l * color + e;
    return l + e;  // kOpPass
}

// ACES + gamma of generatePixel (main.cpp:1226-1231, 827-838) for one channel.
__device__ __forceinline__ float tonemap_channel(float sum, float spp) {
    const float x = sum / spp;
    float r = (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f);
    r = (r < 0.f) ? 0.f : r;  // glm::clamp = min(max(x, 0), 1) with glm's comparison forms: a NaN stays a NaN
    r = (1.f < r) ? 1.f : r;
    if (r != r) return r;     // powf(NaN, y) = NaN (an overflowed radiance sum: inf / inf in the tone curve)
    // glm::pow -> powf: glibc's kernel restated (rt_libm.h), bit-identical to libm on [0, 1]
    return rt_powf(r, 1.f / 2.2f);
}

}  // namespace rtd
