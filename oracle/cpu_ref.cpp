// oracle/cpu_ref.cpp -- TEST INFRASTRUCTURE ONLY (checker, never the product path).
//
// CPU restatement of the render hot path of sergeyrid/raytracing (main.cpp), written
// from the reference's behaviour: same algorithm, same fp32 operation order, same RNG
// (it links the very libstdc++ <random> engine/distributions the reference uses).
//
// PARITY STATUS: "parity unpinned".  The reference holds no tests, golden images or
// fixtures, and it cannot be built in this image (it includes <glm/...>, glm 0.9.9.8 is
// fetched from the network by its CMakeLists.txt:8-18 and is absent here; stand-in headers
// are not allowed).  The glm operation orders used below are restated from glm 0.9.9.8's
// published implementation (func_geometric.inl, func_matrix.inl compute_inverse<3,3>,
// type_mat3x3.inl operator*), anchored on the call sites cited next to each function.
// What IS pinned: the RNG layer (std::minstd_rand + libstdc++ distributions are used
// directly, and tests/golden/rng_kat.json holds known answers for them).
//
// Deliberate deviations from main.cpp (all widen limits, none changes results inside them):
//   * 32-bit node / primitive ids instead of uint16_t (main.cpp:541-544, 550, 609, 616, 641)
//   * SAH sweep buffers sized by the node's own count (main.cpp:639-640 sizes them by the
//     whole scene, which is O(N^2) and result-neutral)
//   * max path depth is a parameter (main.cpp:692 fixes it at 6)
//   * an un-splittable node with more than 4 primitives becomes a proper leaf over its own
//     range (main.cpp:675-679 stores an out-of-range first index there); the scene reports
//     how often that happened so parity scenes can assert "never"
//   * optional counter-keyed RNG mode (one freshly seeded minstd_rand per (pixel, sample))
//     so that sample ranges can be split across GPUs; mode 0 is the reference stream.
//   * analytic primitives (Plane / Ellipsoid / Box, dead code in the reference: no loader path constructs
//     them) can be placed as BVH::build would place them -- planes partitioned to the front and tested at
//     the root, the others in the tree under their calculateAABB boxes (main.cpp:552-558) -- or, for checking
//     kernels that keep them outside their own tree, folded in one by one after the triangle walk.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp, x86-64 baseline = SSE2,
// no FMA, matching the reference's Release flags: CMakeLists.txt:4, build.sh:3).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <utility>
#include <vector>

#include <omp.h>

namespace {

// main.cpp:22-25
const float EPS = 0.0001f;
const float INF = 10000.f;
const float BIG_INF = 20000.f;
const float NEG_INF = -10000.f;

// ---------------------------------------------------------------------------------------
// glm 0.9.9.8 subset, fp32, one rounding per operation, left-to-right.
// ---------------------------------------------------------------------------------------
struct vec3 {
    float x, y, z;
};
inline vec3 V(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 operator+(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, vec3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator*(vec3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return V(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
inline vec3 operator+(vec3 a, float s) { return V(a.x + s, a.y + s, a.z + s); }
inline vec3 operator-(vec3 a) { return V(-a.x, -a.y, -a.z); }
// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross
inline vec3 cross(vec3 a, vec3 b) {
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
// glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline vec3 normalize(vec3 v) { return v * (1.f / std::sqrt(dot(v, v))); }
inline float gmin(float x, float y) { return (y < x) ? y : x; }  // glm::min / std::min
inline float gmax(float x, float y) { return (x < y) ? y : x; }  // glm::max / std::max
inline vec3 vmin(vec3 a, vec3 b) { return V(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
inline vec3 vmax(vec3 a, vec3 b) { return V(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
const float ONE_OVER_PI = 0.318309886183790671537767526745028724f;  // glm::one_over_pi<float>()

// uvt = inverse(mat3{c0,c1,c2}) * w   (glm compute_inverse<3,3> then mat3*vec3; m[col][row])
inline vec3 inverse_times(vec3 c0, vec3 c1, vec3 c2, vec3 w) {
    const float m00 = c0.x, m01 = c0.y, m02 = c0.z;
    const float m10 = c1.x, m11 = c1.y, m12 = c1.z;
    const float m20 = c2.x, m21 = c2.y, m22 = c2.z;
    const float ood = 1.f / (+m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) +
                             m20 * (m01 * m12 - m11 * m02));
    const float i00 = +(m11 * m22 - m21 * m12) * ood;
    const float i10 = -(m10 * m22 - m20 * m12) * ood;
    const float i20 = +(m10 * m21 - m20 * m11) * ood;
    const float i01 = -(m01 * m22 - m21 * m02) * ood;
    const float i11 = +(m00 * m22 - m20 * m02) * ood;
    const float i21 = -(m00 * m21 - m20 * m01) * ood;
    const float i02 = +(m01 * m12 - m11 * m02) * ood;
    const float i12 = -(m00 * m12 - m10 * m02) * ood;
    const float i22 = +(m00 * m11 - m10 * m01) * ood;
    return V(i00 * w.x + i10 * w.y + i20 * w.z, i01 * w.x + i11 * w.y + i21 * w.z,
             i02 * w.x + i12 * w.y + i22 * w.z);
}

// ---------------------------------------------------------------------------------------
// Scene types
// ---------------------------------------------------------------------------------------
enum { MAT_DIFFUSER = 0, MAT_METALLIC = 1, MAT_DIELECTRIC = 2 };  // main.cpp:55-59

struct Material {  // main.cpp:61-66
    vec3 color;
    vec3 emission;
    int type;
    float ior;
};

struct AABB {  // main.cpp:113-197
    vec3 minPoint{INF, INF, INF};
    vec3 maxPoint{NEG_INF, NEG_INF, NEG_INF};
    vec3 center{0.f, 0.f, 0.f};
    vec3 sizePlusCenter{0.f, 0.f, 0.f};
    vec3 negSizePlusCenter{0.f, 0.f, 0.f};

    float area() const {  // main.cpp:120-130
        if (minPoint.x > maxPoint.x || minPoint.y > maxPoint.y || minPoint.z > maxPoint.z) return 0.f;
        float a = maxPoint.x - minPoint.x;
        float b = maxPoint.y - minPoint.y;
        float c = maxPoint.z - minPoint.z;
        return a * b + a * c + b * c;
    }
    bool isInside(vec3 p) const {  // main.cpp:132-135
        return maxPoint.x > p.x && maxPoint.y > p.y && maxPoint.z > p.z && minPoint.x < p.x &&
               minPoint.y < p.y && minPoint.z < p.z;
    }
    void extend(vec3 p) {  // main.cpp:137-141
        minPoint = vmin(minPoint, p);
        maxPoint = vmax(maxPoint, p);
    }
    void extend(const AABB &o) {  // main.cpp:143-147
        minPoint = vmin(minPoint, o.minPoint);
        maxPoint = vmax(maxPoint, o.maxPoint);
    }
    template <class Q>
    void rotateAndTranslate(const Q &rotation, vec3 translation) {  // main.cpp:149-166
        AABB zero = *this;
        minPoint = V(INF, INF, INF);
        maxPoint = V(-INF, -INF, -INF);
        extend(rotation * zero.maxPoint);
        extend(rotation * V(zero.maxPoint.x, zero.maxPoint.y, zero.minPoint.z));
        extend(rotation * V(zero.maxPoint.x, zero.minPoint.y, zero.maxPoint.z));
        extend(rotation * V(zero.minPoint.x, zero.maxPoint.y, zero.maxPoint.z));
        extend(rotation * V(zero.maxPoint.x, zero.minPoint.y, zero.minPoint.z));
        extend(rotation * V(zero.minPoint.x, zero.maxPoint.y, zero.minPoint.z));
        extend(rotation * V(zero.minPoint.x, zero.minPoint.y, zero.maxPoint.z));
        extend(rotation * zero.minPoint);
        minPoint = minPoint + translation;
        maxPoint = maxPoint + translation;
    }
    void calculateEverything() {  // main.cpp:168-178  (slabs are centre +- FULL size)
        center = V((maxPoint.x + minPoint.x) / 2.f, (maxPoint.y + minPoint.y) / 2.f,
                   (maxPoint.z + minPoint.z) / 2.f);
        vec3 size = maxPoint - minPoint;
        vec3 negSize = -size;
        sizePlusCenter = size + center;
        negSizePlusCenter = negSize + center;
    }
    float intersect(vec3 o, vec3 d) const {  // main.cpp:180-196
        if (isInside(o)) return 0.f;
        vec3 t1 = (sizePlusCenter - o) / d;
        vec3 t2 = (negSizePlusCenter - o) / d;
        float d1 = gmax(gmax(gmin(t1.x, t2.x), gmin(t1.y, t2.y)), gmin(t1.z, t2.z));
        float d2 = gmin(gmin(gmax(t1.x, t2.x), gmax(t1.y, t2.y)), gmax(t1.z, t2.z));
        if (d1 > d2 || d2 < 0.f) return BIG_INF;
        return d1;
    }
};

struct Intersection {  // main.cpp:70-96
    bool isIntersected = false;
    bool isInside = false;
    float t = INF;
    vec3 p{0.f, 0.f, 0.f};
    vec3 d{0.f, 0.f, 0.f};
    float nl = 0.f;
    vec3 normal{0.f, 0.f, 0.f};
    vec3 pPlusNormalEps{0.f, 0.f, 0.f};
    vec3 pMinusNormalEps{0.f, 0.f, 0.f};
    int32_t primitive = -1;  // loader-order triangle id, or -2 - k for analytic primitive k; reference keeps a pointer

    void update(vec3 o, vec3 newD, int32_t prim) {  // main.cpp:84-95
        if (!isIntersected) return;
        d = newD;
        nl = -dot(normal, d);
        p = o + t * d;
        primitive = prim;
        vec3 normalEps = normal * EPS;
        pPlusNormalEps = p + normalEps;
        pMinusNormalEps = p - normalEps;
    }
};

typedef std::minstd_rand Rng;

// main.cpp:98-111 -- distribution objects are created per call, exactly like the reference.
inline uint16_t sampleUInt(Rng &rng, uint16_t a, uint16_t b) {
    std::uniform_int_distribution<uint16_t> dis{a, b};
    return dis(rng);
}
inline float sampleUniform(Rng &rng, float a = 0.f, float b = 1.f) {
    std::uniform_real_distribution<float> dis{a, b};
    return dis(rng);
}
inline float sampleNormal(Rng &rng, float m = 0.f, float s = 1.f) {
    std::normal_distribution<float> dis{m, s};
    return dis(rng);
}

struct Triangle {  // main.cpp:466-537
    vec3 pointA, pointB, pointC;
    vec3 normal{0.f, 0.f, 1.f};
    vec3 sideAB{0.f, 0.f, 0.f}, sideAC{0.f, 0.f, 0.f};
    float pdfConst = 0.f;
    AABB aabb;
    uint32_t material = 0;
    uint32_t origId = 0;  // emission order of the loader (main.cpp:1095); reference has no id

    void update() {  // main.cpp:527-536, calculateAABB 519-525
        sideAB = pointB - pointA;
        sideAC = pointC - pointA;
        vec3 crossProduct = cross(sideAB, sideAC);
        normal = normalize(crossProduct);
        pdfConst = 2.f / length(crossProduct);
        aabb = AABB{};
        aabb.extend(pointA);
        aabb.extend(pointB);
        aabb.extend(pointC);
        aabb.calculateEverything();
    }

    Intersection intersectPrimitive(vec3 o, vec3 d, int32_t self) const {  // main.cpp:478-498
        Intersection it;
        vec3 uvt = inverse_times(sideAB, sideAC, -d, o - pointA);
        if (uvt.x < 0.f || uvt.y < 0.f || uvt.z < 0.f || uvt.x + uvt.y > 1.f) return it;
        it.isIntersected = true;
        it.t = uvt.z;
        it.normal = normal;
        if (dot(normal, d) > 0.f) {
            it.isInside = true;
            it.normal = it.normal * -1.f;
        }
        it.update(o, d, self);
        return it;
    }

    vec3 samplePoint(Rng &rng) const {  // main.cpp:500-508
        float u = sampleUniform(rng);
        float v = sampleUniform(rng);
        if (u + v > 1.f) {
            u = 1.f - u;
            v = 1.f - v;
        }
        return pointA + u * sideAB + v * sideAC;
    }

    float pdf(vec3 o, vec3 d, int32_t self) const {  // main.cpp:510-517
        Intersection i = intersectPrimitive(o, d, self);
        if (!i.isIntersected) return 0.f;
        vec3 r = i.p - o;
        return pdfConst * dot(r, r) / std::fabs(dot(normalize(r), i.normal));
    }
};

// Primitive + Plane / Ellipsoid / Box, main.cpp:199-464: the reference's loader never constructs them
// (main.cpp:1089 emits triangles only); restated for hosts that hand them in through the scene description.
// quat = glm::quat (members x, y, z, w); operator* is glm's quat * vec3.
struct quat {
    float x = 0.f, y = 0.f, z = 0.f, w = 1.f;
};
inline vec3 operator*(quat q, vec3 v) {
    vec3 qv = V(q.x, q.y, q.z);
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.f;
}

enum { PRIM_PLANE = 0, PRIM_ELLIPSOID = 1, PRIM_BOX = 2 };

struct Analytic {
    int type = PRIM_PLANE;
    vec3 position{0.f, 0.f, 0.f};
    quat rotation, conjRotation;
    vec3 param{0.f, 0.f, 1.f};          // Plane::normal / Ellipsoid::radius / Box::size
    vec3 rotatedNormal{0.f, 0.f, 1.f};  // Plane::update, main.cpp:274-276
    vec3 radiusSq{1.f, 1.f, 1.f};       // Ellipsoid: radius squared (the reference leaves setting it to its loader)
    uint32_t material = 0;
    AABB aabb;  // Ellipsoid / Box::calculateAABB; a plane has none (it is tested at the root, main.cpp:552-558)

    void update() {
        conjRotation = quat{-rotation.x, -rotation.y, -rotation.z, rotation.w};
        rotatedNormal = rotation * param;
        radiusSq = param * param;
        aabb = AABB{};
        if (type == PRIM_ELLIPSOID) {  // main.cpp:358-364
            aabb.extend(param);
            aabb.extend(-param);
        } else if (type == PRIM_BOX) {  // main.cpp:457-463: half of the extent intersectPrimitive uses
            aabb.extend(param / 2.f);
            aabb.extend(-param / 2.f);
        }
        if (type != PRIM_PLANE) {
            aabb.rotateAndTranslate(rotation, position);
            aabb.calculateEverything();
        }
    }

    // Primitive::intersectFull, main.cpp:218-230
    std::pair<Intersection, Intersection> intersectFull(vec3 o, vec3 d, int32_t self) const {
        Intersection i1 = intersectPrimitive(o, d, self);
        if (!i1.isIntersected) return {i1, i1};
        Intersection i2;
        if (i1.isInside) i2 = intersectPrimitive(i1.pPlusNormalEps, -d, self);
        else i2 = intersectPrimitive(i1.pMinusNormalEps, d, self);
        return {i1, i2};
    }
    // Primitive::pdfFull, main.cpp:232-239
    static float pdfFull(vec3 o, const Intersection &i1, const Intersection &i2, float p1, float p2) {
        vec3 r1 = i1.p - o;
        vec3 r2 = i2.p - o;
        float fp1 = p1 * dot(r1, r1) / std::fabs(dot(normalize(r1), i1.normal));
        float fp2 = p2 * dot(r2, r2) / std::fabs(dot(normalize(r2), i2.normal));
        return fp1 + fp2;
    }
    // Ellipsoid::samplePoint (main.cpp:320-330), Box::samplePoint (425-441); Plane::samplePoint throws (261-263)
    vec3 samplePoint(Rng &rng) const {
        if (type == PRIM_ELLIPSOID) {
            float x = sampleNormal(rng);
            float y = sampleNormal(rng);
            float z = sampleNormal(rng);
            vec3 point = normalize(V(x, y, z));
            point.x *= param.x;
            point.y *= param.y;
            point.z *= param.z;
            return rotation * point + position;
        }
        float wx = param.y * param.z;
        float wy = param.x * param.z;
        float wz = param.x * param.y;
        float u = sampleUniform(rng, 0.f, wx + wy + wz);
        float side = 2 * (float(sampleUniform(rng) < 0.5f) - 0.5f);
        float c1 = sampleUniform(rng, 0.f, 2.f) - 1.f, c2 = sampleUniform(rng, 0.f, 2.f) - 1.f;
        vec3 point;
        if (u < wx) point = V(side * param.x, c1 * param.y, c2 * param.z);
        else if (u < wx + wy) point = V(c1 * param.x, side * param.y, c2 * param.z);
        else point = V(c1 * param.x, c2 * param.y, side * param.z);
        return rotation * point + position;
    }
    // Ellipsoid::pdf (main.cpp:332-356: pos2 is computed from i1.p like pos1, as written there), Box::pdf (443-455)
    float pdf(vec3 o, vec3 d, int32_t self) const {
        std::pair<Intersection, Intersection> is = intersectFull(o, d, self);
        const Intersection &i1 = is.first, &i2 = is.second;
        if (!i1.isIntersected) return 0.f;
        if (type == PRIM_ELLIPSOID) {
            float rs = param.x * param.y * param.z;
            rs = rs * rs;
            float rxs = param.x * param.x;
            float rys = param.y * param.y;
            float rzs = param.z * param.z;
            vec3 pos1 = conjRotation * (i1.p - position);
            vec3 pos2 = conjRotation * (i1.p - position);
            float p1 = ONE_OVER_PI / 4.f / std::sqrt(pos1.x * pos1.x * rs / rxs + pos1.y * pos1.y * rs / rys + pos1.z * pos1.z * rs / rzs);
            float p2 = ONE_OVER_PI / 4.f / std::sqrt(pos2.x * pos2.x * rs / rxs + pos2.y * pos2.y * rs / rys + pos2.z * pos2.z * rs / rzs);
            return pdfFull(o, i1, i2, p1, p2);
        }
        float p = 1.f / 8.f / (param.y * param.z + param.x * param.z + param.x * param.y);
        return pdfFull(o, i1, i2, p, p);
    }

    Intersection intersectPrimitive(vec3 o, vec3 d, int32_t self) const {
        vec3 no = conjRotation * (o - position);
        vec3 nd = conjRotation * d;
        Intersection it;
        if (type == PRIM_PLANE) {  // main.cpp:246-260
            it.normal = rotatedNormal;
            it.t = -dot(no, param) / dot(nd, param);
            it.isIntersected = it.t > 0.f;
            if (!it.isIntersected) it.t = INF;
            it.update(o, d, self);
            return it;
        }
        if (type == PRIM_ELLIPSOID) {  // main.cpp:282-317
            vec3 ndr = nd / param;
            vec3 nor = no / param;
            float a = dot(ndr, ndr);
            float b = dot(nor, ndr) / a;
            float c = (dot(nor, nor) - 1) / a;
            float discr = b * b - c;
            if (discr < 0.f) return it;
            float sqrtDiscr = std::sqrt(discr);
            float mind = -b - sqrtDiscr;
            float maxd = -b + sqrtDiscr;
            if (mind < 0.f && maxd > 0.f) {
                it.t = maxd;
                it.normal = normalize(rotation * ((no + maxd * nd) / radiusSq)) * -1.f;
                it.isInside = true;
                it.isIntersected = true;
            } else if (mind > 0.f) {
                it.t = mind;
                it.normal = normalize(rotation * ((no + mind * nd) / radiusSq));
                it.isIntersected = true;
            }
            it.update(o, d, self);
            return it;
        }
        // Box, main.cpp:370-419
        vec3 t1 = (param - no) / nd;
        vec3 t2 = (param * -1.f - no) / nd;
        float d1 = std::max(std::max(std::min(t1.x, t2.x), std::min(t1.y, t2.y)), std::min(t1.z, t2.z));
        float d2 = std::min(std::min(std::max(t1.x, t2.x), std::max(t1.y, t2.y)), std::max(t1.z, t2.z));
        if (d1 > d2 || d2 < 0.f) return it;
        it.isIntersected = true;
        if (d1 < 0.f) {
            it.t = d2;
            it.isInside = true;
        } else {
            it.t = d1;
        }
        vec3 ps = (no + it.t * nd) / param;
        vec3 aps = V(std::fabs(ps.x), std::fabs(ps.y), std::fabs(ps.z));
        if (aps.x > aps.y && aps.x > aps.z) it.normal = V(ps.x < 0 ? -1.f : 1.f, 0.f, 0.f);
        else if (aps.y > aps.x && aps.y > aps.z) it.normal = V(0.f, ps.y < 0 ? -1.f : 1.f, 0.f);
        else it.normal = V(0.f, 0.f, ps.z < 0 ? -1.f : 1.f);
        it.normal = rotation * it.normal;
        if (it.isInside) it.normal = it.normal * -1.f;
        it.update(o, d, self);
        return it;
    }
};

struct BVHNode {  // main.cpp:539-546, ids widened to 32 bits
    AABB aabb;
    uint32_t left = 0, right = 0, firstPrimitiveId = 0, primitiveCount = 0;
    uint8_t division = 0;
};

struct Counters {
    uint64_t segments = 0;    // intersectScene calls (primary + continuation rays)
    uint64_t nodes = 0;       // BVH nodes entered
    uint64_t tri_tests = 0;   // intersectPrimitive calls made by traversal
    uint64_t box_tests = 0;   // AABB::intersect calls
    uint64_t light_tests = 0; // intersectPrimitive calls made by LightSurface::pdf
    void add(const Counters &o) {
        segments += o.segments;
        nodes += o.nodes;
        tri_tests += o.tri_tests;
        box_tests += o.box_tests;
        light_tests += o.light_tests;
    }
};

struct Camera {  // main.cpp:696-700
    vec3 position, right, up, forward;
    float fovTanX, fovTanY;
};

// One entry of InputData::primitives (main.cpp:702): the build sorts these, the triangles and analytic
// primitives themselves stay where the loader put them.
struct Item {
    AABB aabb;
    int32_t prim;  // loader-order triangle id, or -2 - k for analytic primitive k
    bool isPlane;
};

struct Scene {  // main.cpp:689-706 (InputData), minus width/height/samples which are call arguments
    std::vector<Triangle> tris;    // loader order
    std::vector<int32_t> lights;   // the emissive primitives (main.cpp:1096-1098): loader-order triangle ids in emission
                                   // order, then -2 - k for every emissive analytic primitive k in list order
    std::vector<Material> mats;
    std::vector<Analytic> prims;   // Intersection::primitive of analytic primitive k is -2 - k
    bool primsInTree = false;      // analytic primitives placed by BVH::build (true) or folded in after the walk
    std::vector<Item> items;       // InputData::primitives: re-ordered by the build
    std::vector<BVHNode> nodes;
    Camera cam;
    // material of whatever an Intersection names
    const Material &materialOf(int32_t primitive) const {
        return mats[primitive >= 0 ? tris[primitive].material : prims[-2 - primitive].material];
    }
    vec3 samplePointOf(int32_t primitive, Rng &rng) const {
        return primitive >= 0 ? tris[primitive].samplePoint(rng) : prims[-2 - primitive].samplePoint(rng);
    }
    float pdfOf(int32_t primitive, vec3 o, vec3 d) const {
        return primitive >= 0 ? tris[primitive].pdf(o, d, primitive) : prims[-2 - primitive].pdf(o, d, primitive);
    }
    Intersection intersectItem(const Item &it, vec3 o, vec3 d) const {
        return it.prim >= 0 ? tris[it.prim].intersectPrimitive(o, d, it.prim)
                            : prims[-2 - it.prim].intersectPrimitive(o, d, it.prim);
    }
    vec3 backgroundColor{0.f, 0.f, 0.f};  // never set by the loader (main.cpp:694)
    uint64_t unsplittable_nodes = 0;
    uint32_t max_depth_seen = 0;
};

// ---------------------------------------------------------------------------------------
// BVH build: main.cpp:552-558 and 608-686 (full-sweep SAH over 3 axes, leaf <= 4).
// ---------------------------------------------------------------------------------------
struct Builder {
    Scene &s;
    std::vector<Item> &prims;  // sorted in place, like the reference's shared_ptr vector
    explicit Builder(Scene &sc) : s(sc), prims(sc.items) {}

    void buildRecursive(size_t begin, size_t end, uint32_t depth) {
        uint32_t curNode = (uint32_t)s.nodes.size();
        s.nodes.push_back(BVHNode{});
        if (depth > s.max_depth_seen) s.max_depth_seen = depth;
        for (size_t i = begin; i != end; ++i) s.nodes[curNode].aabb.extend(prims[i].aabb);
        s.nodes[curNode].aabb.calculateEverything();
        uint32_t totalCount = (uint32_t)(end - begin);

        if (totalCount <= 4) {  // main.cpp:618-622
            s.nodes[curNode].firstPrimitiveId = (uint32_t)begin;
            s.nodes[curNode].primitiveCount = totalCount;
            return;
        }

        float bestCost = (float)totalCount * s.nodes[curNode].aabb.area();
        size_t middle = end;
        std::vector<AABB> leftAABBs(totalCount + 1), rightAABBs(totalCount + 1);

        for (uint8_t division = 0; division < 3; ++division) {  // main.cpp:627-663
            std::sort(prims.begin() + begin, prims.begin() + end,
                      [division](const Item &p1, const Item &p2) {
                          if (division == 0) return p1.aabb.center.x < p2.aabb.center.x;
                          if (division == 1) return p1.aabb.center.y < p2.aabb.center.y;
                          return p1.aabb.center.z < p2.aabb.center.z;
                      });
            AABB totalAABB;
            uint32_t j = 0;
            for (size_t i = begin; i != end; ++i) {
                ++j;
                totalAABB.extend(prims[i].aabb);
                leftAABBs[j] = totalAABB;
            }
            totalAABB = AABB{};
            for (size_t i = end - 1; i != begin; --i) {
                --j;
                totalAABB.extend(prims[i].aabb);
                rightAABBs[j] = totalAABB;
            }
            for (uint32_t i = 1; i < totalCount; ++i) {
                float newCost = leftAABBs[i].area() * (float)i +
                                rightAABBs[i].area() * (float)(totalCount - i);
                if (newCost < bestCost) {
                    bestCost = newCost;
                    middle = begin + i;
                    s.nodes[curNode].division = division;
                }
            }
        }

        const uint8_t div = s.nodes[curNode].division;  // main.cpp:665-673
        std::sort(prims.begin() + begin, prims.begin() + end,
                  [div](const Item &p1, const Item &p2) {
                      if (div == 0) return p1.aabb.center.x < p2.aabb.center.x;
                      if (div == 1) return p1.aabb.center.y < p2.aabb.center.y;
                      return p1.aabb.center.z < p2.aabb.center.z;
                  });

        if (middle == end) {  // main.cpp:675-679 (deviation: proper leaf, counted)
            s.nodes[curNode].firstPrimitiveId = (uint32_t)begin;
            s.nodes[curNode].primitiveCount = totalCount;
            ++s.unsplittable_nodes;
            return;
        }

        s.nodes[curNode].left = curNode + 1;  // main.cpp:681-685
        buildRecursive(begin, middle, depth + 1);
        s.nodes[curNode].right = (uint32_t)s.nodes.size();
        buildRecursive(middle, end, depth + 1);
    }
};

// BVH::build, main.cpp:552-558: InputData::primitives = the triangles in loader order, then (when they are
// placed in the tree) the analytic primitives in list order; planes are partitioned to the front and the
// root's range is widened over them.
void buildScene(Scene &s) {
    s.items.clear();
    s.nodes.clear();
    s.unsplittable_nodes = 0;
    s.max_depth_seen = 0;
    for (size_t i = 0; i < s.tris.size(); ++i) s.items.push_back(Item{s.tris[i].aabb, (int32_t)i, false});
    if (s.primsInTree)
        for (size_t k = 0; k < s.prims.size(); ++k)
            s.items.push_back(Item{s.prims[k].aabb, -2 - (int32_t)k, s.prims[k].type == PRIM_PLANE});
    auto begin = std::partition(s.items.begin(), s.items.end(), [](const Item &it) { return it.isPlane; });
    const size_t nplanes = (size_t)std::distance(s.items.begin(), begin);
    Builder b(s);
    b.buildRecursive(nplanes, s.items.size(), 0);
    s.nodes[0].firstPrimitiveId = 0;
    s.nodes[0].primitiveCount += (uint32_t)nplanes;
}

// main.cpp:560-605.  Root is node 0; leaves have left = right = 0 and are filtered by
// "!= root".  main.cpp:582-584 swaps the two child distances with
//   l += r; r = l - r; l -= r;
// which in floating point does not always reproduce the two values exactly.  Only the
// comparisons "leftDistance < minDistance" / "rightDistance < minDistance" consume them, so
// the restatement keeps that arithmetic to stay faithful in those compares.
inline void refSwap(float &l, float &r) {
    l += r;
    r = l - r;
    l -= r;
}

Intersection bvhIntersectRef(const Scene &s, vec3 o, vec3 d, uint32_t curNode, float minDistance,
                             Counters &c, bool &tie, float &tie_t) {
    Intersection closest;
    const BVHNode &node = s.nodes[curNode];
    ++c.nodes;
    for (uint32_t i = node.firstPrimitiveId; i < node.firstPrimitiveId + node.primitiveCount; ++i) {
        ++c.tri_tests;
        Intersection ni = s.intersectItem(s.items[i], o, d);
        if (ni.isIntersected && closest.isIntersected && ni.t == closest.t) {
            tie = true;
            tie_t = ni.t;
        }
        if (ni.isIntersected && (!closest.isIntersected || ni.t < closest.t)) closest = ni;
    }
    minDistance = closest.t;

    uint32_t left = node.left;
    uint32_t right = node.right;
    // Leaves: the reference evaluates the root's box twice here and then filters both
    // children by "!= root" (main.cpp:576-577, 587, 596); the result is the same as returning.
    if (left == 0 && right == 0) return closest;

    c.box_tests += 2;
    float leftDistance = s.nodes[left].aabb.intersect(o, d);
    float rightDistance = s.nodes[right].aabb.intersect(o, d);
    if (leftDistance > rightDistance) {
        left = right;
        right = node.left;
        refSwap(leftDistance, rightDistance);
    }
    if (left != 0 && leftDistance < minDistance) {
        Intersection ni = bvhIntersectRef(s, o, d, left, minDistance, c, tie, tie_t);
        if (ni.isIntersected && closest.isIntersected && ni.t == closest.t) {
            tie = true;
            tie_t = ni.t;
        }
        if (ni.isIntersected && (!closest.isIntersected || ni.t < closest.t)) {
            closest = ni;
            minDistance = closest.t;
        }
    }
    if (right != 0 && rightDistance < minDistance) {
        Intersection ni = bvhIntersectRef(s, o, d, right, minDistance, c, tie, tie_t);
        if (ni.isIntersected && closest.isIntersected && ni.t == closest.t) {
            tie = true;
            tie_t = ni.t;
        }
        if (ni.isIntersected && (!closest.isIntersected || ni.t < closest.t)) closest = ni;
    }
    return closest;
}

// Analytic primitives kept outside the tree (Scene::primsInTree false: not the reference's structure, but that
// of kernels that walk a triangle-only tree of their own): tested one by one after the walk.  Strict minimum,
// list order, and like a triangle's a hit needs t < INF.
inline void foldAnalytic(const Scene &s, vec3 o, vec3 d, Intersection &closest) {
    if (s.primsInTree) return;
    for (size_t k = 0; k < s.prims.size(); ++k) {
        Intersection ni = s.prims[k].intersectPrimitive(o, d, -2 - (int32_t)k);
        if (ni.isIntersected && ni.t < INF && (!closest.isIntersected || ni.t < closest.t)) closest = ni;
    }
}

// main.cpp:1113-1115
inline Intersection intersectScene(const Scene &s, vec3 o, vec3 d, Counters &c) {
    bool tie = false;
    float tie_t = 0.f;
    ++c.segments;
    Intersection closest = bvhIntersectRef(s, o, d, 0, INF, c, tie, tie_t);
    foldAnalytic(s, o, d, closest);
    return closest;
}

// ---------------------------------------------------------------------------------------
// Distributions: main.cpp:736-805 (Uniform 720-734 is never used)
// ---------------------------------------------------------------------------------------
inline vec3 cosineSample(vec3 n, Rng &rng) {  // main.cpp:739-746
    float x = sampleNormal(rng);
    float y = sampleNormal(rng);
    float z = sampleNormal(rng);
    vec3 d = V(x, y, z);
    d = normalize(d) + n * (1.f + EPS);
    return normalize(d);
}
inline float cosinePdf(vec3 n, vec3 d) {  // main.cpp:748-754
    float dn = dot(d, n);
    if (dn < EPS) return 0.f;
    return dn * ONE_OVER_PI;
}
inline vec3 lightSample(const Scene &s, vec3 x, Rng &rng) {  // main.cpp:764-769
    if (s.lights.empty()) return V(0.f, 0.f, 0.f);
    uint16_t k = sampleUInt(rng, 0, (uint16_t)(s.lights.size() - 1));
    return normalize(s.samplePointOf(s.lights[k], rng) - x);
}
inline float lightPdf(const Scene &s, vec3 x, vec3 d, int32_t primitive, Counters &c) {  // 771-783
    float ps = 0.f;
    for (int32_t light : s.lights) {
        if (light == primitive) continue;
        ++c.light_tests;
        ps += s.pdfOf(light, x, d);
    }
    if (ps < EPS || std::isnan(ps) || std::isinf(ps)) return 0.f;
    return ps / float(s.lights.size());
}

vec3 applyLight(const Scene &s, const Intersection &it, uint32_t rayDepth, Rng &rng, Counters &c);

// main.cpp:1120-1126
vec3 getReflectedLight(const Scene &s, const Intersection &it, uint32_t rayDepth, Rng &rng, Counters &c) {
    vec3 rd = it.d + 2.f * it.normal * it.nl;
    rd = normalize(rd);
    Intersection ri = intersectScene(s, it.pPlusNormalEps, rd, c);
    return applyLight(s, ri, rayDepth - 1, rng, c);
}

// main.cpp:1128-1141
vec3 applyLightDiffuser(const Scene &s, const Intersection &it, uint32_t rayDepth, Rng &rng, Counters &c) {
    const Material &m = s.materialOf(it.primitive);
    vec3 x = it.pPlusNormalEps, n = it.normal;
    // Mix::sample, main.cpp:793-800
    bool pickCos = sampleUniform(rng) < 0.5f;
    vec3 w = pickCos ? cosineSample(n, rng) : lightSample(s, x, rng);
    // Mix::pdf, main.cpp:802-804
    float p = 0.5f * cosinePdf(n, w) + 0.5f * lightPdf(s, x, w, it.primitive, c);
    float wn = dot(w, it.normal);
    if (wn < 0.f || p < EPS) return m.emission;
    Intersection ni = intersectScene(s, it.pPlusNormalEps, w, c);
    vec3 l = applyLight(s, ni, rayDepth - 1, rng, c);
    return m.color * l * ONE_OVER_PI * wn / p + m.emission;
}

// main.cpp:1143-1147
vec3 applyLightMetallic(const Scene &s, const Intersection &it, uint32_t rayDepth, Rng &rng, Counters &c) {
    const Material &m = s.materialOf(it.primitive);
    vec3 reflected = getReflectedLight(s, it, rayDepth, rng, c);
    return m.color * reflected + m.emission;
}

// main.cpp:1149-1186
vec3 applyLightDielectric(const Scene &s, const Intersection &it, uint32_t rayDepth, Rng &rng, Counters &c) {
    const Material &m = s.materialOf(it.primitive);
    float n1 = 1.f, n2 = m.ior;
    if (it.isInside) {
        n1 = n2;
        n2 = 1.f;
    }
    float n12 = n1 / n2;
    float nl = it.nl;
    float sn = n12 * std::sqrt(1.f - nl * nl);
    if (sn > 1.f) return getReflectedLight(s, it, rayDepth, rng, c) + m.emission;

    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float mnl = 1.f - nl;
    float mnlsq = mnl * mnl;
    float r = r0 + (1.f - r0) * mnlsq * mnlsq * mnl;
    if (sampleUniform(rng) < r) return getReflectedLight(s, it, rayDepth, rng, c) + m.emission;

    vec3 rd = n12 * it.d + (n12 * nl - std::sqrt(1 - sn * sn)) * it.normal;
    rd = normalize(rd);
    Intersection ri = intersectScene(s, it.pMinusNormalEps, rd, c);
    vec3 refracted = applyLight(s, ri, rayDepth - 1, rng, c);
    if (!it.isInside) refracted = refracted * m.color;
    return refracted + m.emission;
}

// main.cpp:1188-1210
vec3 applyLight(const Scene &s, const Intersection &it, uint32_t rayDepth, Rng &rng, Counters &c) {
    if (!it.isIntersected) return s.backgroundColor;
    vec3 color = V(0.f, 0.f, 0.f);
    if (rayDepth == 0) return color;
    const Material &m = s.materialOf(it.primitive);
    if (rayDepth == 1) return m.emission;
    if (m.type == MAT_DIFFUSER) color = applyLightDiffuser(s, it, rayDepth, rng, c);
    else if (m.type == MAT_METALLIC) color = applyLightMetallic(s, it, rayDepth, rng, c);
    else if (m.type == MAT_DIELECTRIC) color = applyLightDielectric(s, it, rayDepth, rng, c);
    return color;
}

// main.cpp:1212-1215: the jittered pinhole ray of one sample.
inline vec3 primaryDirection(const Scene &s, uint32_t px, uint32_t py, uint32_t W, uint32_t H, Rng &rng) {
    float x = (2 * (float(px) + sampleUniform(rng)) / float(W) - 1) * s.cam.fovTanX;
    float y = -(2 * (float(py) + sampleUniform(rng)) / float(H) - 1) * s.cam.fovTanY;
    return normalize(x * s.cam.right + y * s.cam.up + s.cam.forward);
}

// main.cpp:1212-1218
inline vec3 generateSample(const Scene &s, uint32_t px, uint32_t py, uint32_t W, uint32_t H,
                           uint32_t rayDepth, Rng &rng, Counters &c) {
    vec3 d = primaryDirection(s, px, py, W, H, rng);
    Intersection it = intersectScene(s, s.cam.position, d, c);
    return applyLight(s, it, rayDepth, rng, c);
}

// Seed of the per-(pixel, sample) stream in counter mode.  Not in the reference (its stream is
// one engine per pixel carried across samples, main.cpp:1221-1225); defined here so that the
// product and the oracle draw the same numbers when sample ranges are split over devices.
inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
inline uint32_t counterSeed(uint32_t pixel, uint32_t sample, uint32_t seed) {
    uint32_t h = fmix32(pixel ^ 0x9E3779B9u);
    h = fmix32(h ^ (sample * 0x85EBCA6Bu + 0x7F4A7C15u));
    h = fmix32(h ^ seed);
    return h;
}

// main.cpp:827-838
inline float saturate1(float v) { return gmin(gmax(v, 0.f), 1.f); }
inline vec3 acesTonemap(vec3 x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    vec3 r = (x * (a * x + b)) / (x * (c * x + d) + e);
    return V(saturate1(r.x), saturate1(r.y), saturate1(r.z));
}

Scene *asScene(void *h) { return reinterpret_cast<Scene *>(h); }

}  // namespace

// =========================================================================================
// C interface (ctypes).  Plain pointers and sizes only.
// =========================================================================================
extern "C" {

struct orc_camera {
    float position[3], right[3], up[3], forward[3];
    float tan_x, tan_y;
};
struct orc_material {
    float color[3];
    float emission[3];
    int32_t type;
    float ior;
};

// tri_verts: ntri * 9 floats (A, B, C in world space) in loader emission order;
// tri_material: ntri material indices.  Runs Triangle::update and BVH::build.
void *orc_scene_create(const float *tri_verts, const uint32_t *tri_material, uint32_t ntri,
                       const orc_material *mats, uint32_t nmat, const orc_camera *cam) {
    Scene *s = new Scene();
    s->mats.resize(nmat);
    for (uint32_t i = 0; i < nmat; ++i) {
        s->mats[i].color = V(mats[i].color[0], mats[i].color[1], mats[i].color[2]);
        s->mats[i].emission = V(mats[i].emission[0], mats[i].emission[1], mats[i].emission[2]);
        s->mats[i].type = mats[i].type;
        s->mats[i].ior = mats[i].ior;
    }
    s->cam.position = V(cam->position[0], cam->position[1], cam->position[2]);
    s->cam.right = V(cam->right[0], cam->right[1], cam->right[2]);
    s->cam.up = V(cam->up[0], cam->up[1], cam->up[2]);
    s->cam.forward = V(cam->forward[0], cam->forward[1], cam->forward[2]);
    s->cam.fovTanX = cam->tan_x;
    s->cam.fovTanY = cam->tan_y;

    s->tris.resize(ntri);
    std::vector<uint8_t> isLight(ntri, 0);
    for (uint32_t i = 0; i < ntri; ++i) {
        const float *v = tri_verts + 9 * (size_t)i;
        Triangle &t = s->tris[i];
        t.pointA = V(v[0], v[1], v[2]);
        t.pointB = V(v[3], v[4], v[5]);
        t.pointC = V(v[6], v[7], v[8]);
        t.material = tri_material[i];
        t.origId = i;
        t.update();
        // main.cpp:1096-1098
        if (length(s->mats[t.material].emission) > EPS) isLight[i] = 1;
    }
    // lights keep the loader's order; the reference stores raw pointers that survive the sort.
    for (uint32_t i = 0; i < ntri; ++i)
        if (isLight[i]) s->lights.push_back((int32_t)i);
    // main.cpp:552-558 (no planes exist so far: the loader only constructs triangles, main.cpp:1089)
    buildScene(*s);
    return s;
}

void orc_scene_destroy(void *h) { delete asScene(h); }

// The tree of BVH::build for checking another builder of the same tree against it.  nodes4: per node left,
// right, firstPrimitiveId, primitiveCount; boxes12: per node sizePlusCenter, negSizePlusCenter, minPoint,
// maxPoint; items: the re-ordered primitive list (loader-order triangle id, or ntri + k for analytic
// primitive k).  Any pointer may be null.  Returns the node count.
uint32_t orc_scene_tree(void *h, uint32_t *nodes4, float *boxes12, uint32_t *items) {
    Scene *s = asScene(h);
    for (size_t k = 0; k < s->nodes.size(); ++k) {
        const BVHNode &n = s->nodes[k];
        if (nodes4) {
            nodes4[4 * k] = n.left;
            nodes4[4 * k + 1] = n.right;
            nodes4[4 * k + 2] = n.firstPrimitiveId;
            nodes4[4 * k + 3] = n.primitiveCount;
        }
        if (boxes12) {
            const vec3 v[4] = {n.aabb.sizePlusCenter, n.aabb.negSizePlusCenter, n.aabb.minPoint, n.aabb.maxPoint};
            for (int j = 0; j < 4; ++j) {
                boxes12[12 * k + 3 * j] = v[j].x;
                boxes12[12 * k + 3 * j + 1] = v[j].y;
                boxes12[12 * k + 3 * j + 2] = v[j].z;
            }
        }
    }
    if (items)
        for (size_t i = 0; i < s->items.size(); ++i)
            items[i] = s->items[i].prim >= 0 ? (uint32_t)s->items[i].prim
                                             : (uint32_t)s->tris.size() + (uint32_t)(-2 - s->items[i].prim);
    return (uint32_t)s->nodes.size();
}

// info[0]=node count, [1]=light count, [2]=unsplittable nodes (deviation counter), [3]=max depth
void orc_scene_info(void *h, uint64_t *info) {
    Scene *s = asScene(h);
    info[0] = s->nodes.size();
    info[1] = s->lights.size();
    info[2] = s->unsplittable_nodes;
    info[3] = s->max_depth_seen;
}

// Closest hit for arbitrary rays.  prim_id = loader-order triangle id (0xFFFFFFFF on a miss),
// t = hit distance (INF=1e4 on a miss), tie[i]=1 when two triangles gave the identical
// winning t (the winner then depends on traversal order).  counters: 5 x uint64 (see Counters).
void orc_intersect_rays(void *h, uint64_t n, const float *o, const float *d, uint32_t *prim_id,
                        float *t, uint8_t *tie, uint64_t *counters, int use_bvh) {
    Scene *s = asScene(h);
    Counters total;
#pragma omp parallel
    {
        Counters c;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            vec3 ro = V(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
            vec3 rd = V(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
            bool tf = false;
            float tt = 0.f;
            Intersection it;
            ++c.segments;
            if (use_bvh) {
                it = bvhIntersectRef(*s, ro, rd, 0, INF, c, tf, tt);
            } else {  // brute force over all primitives in build order
                for (size_t k = 0; k < s->items.size(); ++k) {
                    Intersection ni = s->intersectItem(s->items[k], ro, rd);
                    if (ni.isIntersected && it.isIntersected && ni.t == it.t) {
                        tf = true;
                        tt = ni.t;
                    }
                    if (ni.isIntersected && (!it.isIntersected || ni.t < it.t)) it = ni;
                }
            }
            foldAnalytic(*s, ro, rd, it);
            prim_id[i] = !it.isIntersected ? 0xFFFFFFFFu
                         : it.primitive >= 0 ? (uint32_t)it.primitive
                                             : (uint32_t)s->tris.size() + (uint32_t)(-2 - it.primitive);
            t[i] = it.t;
            if (tie) tie[i] = (tf && it.isIntersected && tt == it.t) ? 1 : 0;
        }
#pragma omp critical
        total.add(c);
    }
    if (counters) {
        counters[0] = total.segments;
        counters[1] = total.nodes;
        counters[2] = total.tri_tests;
        counters[3] = total.box_tests;
        counters[4] = total.light_tests;
    }
}

// The primary ray of sample 0 of every pixel: engine seeded with the pixel index
// (main.cpp:1221), two uniforms (main.cpp:1213-1214).  Writes origin/direction per pixel.
void orc_primary_rays(void *h, uint32_t W, uint32_t H, float *o, float *d) {
    Scene *s = asScene(h);
#pragma omp parallel for schedule(static)
    for (int64_t ij = 0; ij < (int64_t)W * H; ++ij) {
        uint32_t px = (uint32_t)(ij % W), py = (uint32_t)(ij / W);
        Rng rng{uint32_t(py * W + px)};
        vec3 dir = primaryDirection(*s, px, py, W, H, rng);
        o[3 * ij] = s->cam.position.x;
        o[3 * ij + 1] = s->cam.position.y;
        o[3 * ij + 2] = s->cam.position.z;
        d[3 * ij] = dir.x;
        d[3 * ij + 1] = dir.y;
        d[3 * ij + 2] = dir.z;
    }
}

// generateScene (main.cpp:1235-1242) without the tonemap: radiance_sum[3*ij..] = sum over the
// sample range of generateSample, accumulated in sample order (main.cpp:1222-1225).
//   rng_mode 0: reference stream -- one engine per pixel seeded with py*W+px, carried across
//               samples; spp_begin must be 0.
//   rng_mode 1: counter mode -- sample k of pixel ij uses a fresh engine seeded with
//               counterSeed(ij, k, seed); any [spp_begin, spp_begin+spp_count) range is valid.
// max_depth is InputData::rayDepth (6 in the reference).  counters: 5 x uint64, may be null.
int orc_render(void *h, uint32_t W, uint32_t H, uint32_t spp_begin, uint32_t spp_count,
               uint32_t max_depth, int rng_mode, uint32_t seed, float *radiance_sum,
               uint64_t *counters, int nthreads) {
    Scene *s = asScene(h);
    if (rng_mode == 0 && spp_begin != 0) return -1;
    Counters total;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
    {
        Counters c;
#pragma omp for schedule(dynamic, 8)
        for (int64_t ij = 0; ij < (int64_t)W * H; ++ij) {
            uint32_t px = (uint32_t)(ij % W), py = (uint32_t)(ij / W);
            vec3 color = V(0.f, 0.f, 0.f);
            if (rng_mode == 0) {
                Rng rng{uint32_t(py * W + px)};
                for (uint32_t i = 0; i < spp_count; ++i) {
                    vec3 sm = generateSample(*s, px, py, W, H, max_depth, rng, c);
                    color = color + sm;
                }
            } else {
                for (uint32_t i = 0; i < spp_count; ++i) {
                    Rng rng{counterSeed((uint32_t)ij, spp_begin + i, seed)};
                    vec3 sm = generateSample(*s, px, py, W, H, max_depth, rng, c);
                    color = color + sm;
                }
            }
            radiance_sum[3 * ij] = color.x;
            radiance_sum[3 * ij + 1] = color.y;
            radiance_sum[3 * ij + 2] = color.z;
        }
#pragma omp critical
        total.add(c);
    }
    if (counters) {
        counters[0] = total.segments;
        counters[1] = total.nodes;
        counters[2] = total.tri_tests;
        counters[3] = total.box_tests;
        counters[4] = total.light_tests;
    }
    return 0;
}

// main.cpp:1226-1231 then 840-846: mean, ACES, gamma 1/2.2, truncating byte conversion.
void orc_tonemap(const float *radiance_sum, uint64_t npixels, uint32_t spp, float *out_float,
                 uint8_t *out_bytes) {
    const float p = 1.f / 2.2f;
    for (uint64_t i = 0; i < npixels; ++i) {
        vec3 color = V(radiance_sum[3 * i], radiance_sum[3 * i + 1], radiance_sum[3 * i + 2]);
        color = color / float(spp);  // glm vec3 /= uint16_t casts the scalar to float
        color = acesTonemap(color);
        color = V(std::pow(color.x, p), std::pow(color.y, p), std::pow(color.z, p));
        if (out_float) {
            out_float[3 * i] = color.x;
            out_float[3 * i + 1] = color.y;
            out_float[3 * i + 2] = color.z;
        }
        if (out_bytes) {
            out_bytes[3 * i] = uint8_t(color.x * 255);
            out_bytes[3 * i + 1] = uint8_t(color.y * 255);
            out_bytes[3 * i + 2] = uint8_t(color.z * 255);
        }
    }
}

// InputData::backgroundColor (main.cpp:694): the reference never sets it; a host that does (an opt-in
// of the build under test) is checked against applyLight's miss branch (main.cpp:1190-1192) with it.
void orc_scene_set_background(void *h, float r, float g, float b) { asScene(h)->backgroundColor = V(r, g, b); }

// Analytic primitives: n records of {int32 type; float position[3]; float rotation[4] (x, y, z, w);
// float param[3]; uint32 material}.  in_tree != 0: they join InputData::primitives behind the triangles and
// BVH::build places them (main.cpp:552-558); in_tree == 0: they are folded in after the walk (foldAnalytic).
struct orc_primitive {
    int32_t type;
    float position[3];
    float rotation[4];
    float param[3];
    uint32_t material;
};
void orc_scene_set_primitives(void *h, const orc_primitive *p, uint32_t n, int in_tree) {
    Scene *s = asScene(h);
    s->prims.resize(n);
    for (uint32_t k = 0; k < n; ++k) {
        Analytic &a = s->prims[k];
        a.type = p[k].type;
        a.position = V(p[k].position[0], p[k].position[1], p[k].position[2]);
        a.rotation = quat{p[k].rotation[0], p[k].rotation[1], p[k].rotation[2], p[k].rotation[3]};
        a.param = V(p[k].param[0], p[k].param[1], p[k].param[2]);
        a.material = p[k].material;
        a.update();
    }
    s->primsInTree = in_tree != 0;
    buildScene(*s);
    // the light list (main.cpp:1096-1098): the emissive triangles as before, then the emissive analytic primitives
    while (!s->lights.empty() && s->lights.back() < 0) s->lights.pop_back();
    for (uint32_t k = 0; k < n; ++k)
        if (length(s->mats[s->prims[k].material].emission) > EPS) s->lights.push_back(-2 - (int32_t)k);
}

// Known-answer probe of the RNG layer (main.cpp:98-111), each group from a freshly seeded engine.
// raw[4], uni[3], nrm[3], nrm_draws[1] (engine draws used by the 3 normals), ui2[4] =
// sampleUInt(0,1), ui12[4] = sampleUInt(0,11).
void orc_rng_kat(uint32_t seed, uint64_t *raw, float *uni, float *nrm, uint32_t *nrm_draws,
                 uint16_t *ui2, uint16_t *ui12) {
    {
        Rng r{seed};
        for (int i = 0; i < 4; ++i) raw[i] = r();
    }
    {
        Rng r{seed};
        for (int i = 0; i < 3; ++i) uni[i] = sampleUniform(r);
    }
    {
        Rng r{seed}, probe{seed};
        for (int i = 0; i < 3; ++i) nrm[i] = sampleNormal(r);
        uint32_t n = 0;
        while (!(probe == r)) {
            probe();
            ++n;
        }
        *nrm_draws = n;
    }
    {
        Rng r{seed};
        for (int i = 0; i < 4; ++i) ui2[i] = sampleUInt(r, 0, 1);
    }
    {
        Rng r{seed};
        for (int i = 0; i < 4; ++i) ui12[i] = sampleUInt(r, 0, 11);
    }
}

// A longer probe for comparing a device restatement of the RNG layer: from one engine seeded
// with `seed`, draw n triples (uniform, normal, uint in [0, L-1]) in that order.
void orc_rng_stream(uint32_t seed, uint32_t n, uint32_t L, float *uni, float *nrm, uint16_t *ui) {
    Rng r{seed};
    for (uint32_t i = 0; i < n; ++i) {
        uni[i] = sampleUniform(r);
        nrm[i] = sampleNormal(r);
        ui[i] = sampleUInt(r, 0, (uint16_t)(L - 1));
    }
}

uint32_t orc_counter_seed(uint32_t pixel, uint32_t sample, uint32_t seed) {
    return counterSeed(pixel, sample, seed);
}

// Per-triangle derived data in loader order, for checking a host loader / flattener:
// normal[3], pdfConst, per triangle.
void orc_triangle_derived(void *h, float *out4) {
    Scene *s = asScene(h);
    for (const Triangle &t : s->tris) {
        float *o = out4 + 4 * (size_t)t.origId;
        o[0] = t.normal.x;
        o[1] = t.normal.y;
        o[2] = t.normal.z;
        o[3] = t.pdfConst;
    }
}

// Light list in the reference's order (loader-order triangle ids; ntri + k for analytic primitive k).
uint32_t orc_lights(void *h, uint32_t *out, uint32_t cap) {
    Scene *s = asScene(h);
    uint32_t n = (uint32_t)s->lights.size();
    for (uint32_t i = 0; i < n && i < cap; ++i)
        out[i] = s->lights[i] >= 0 ? (uint32_t)s->lights[i] : (uint32_t)s->tris.size() + (uint32_t)(-2 - s->lights[i]);
    return n;
}

int orc_num_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
