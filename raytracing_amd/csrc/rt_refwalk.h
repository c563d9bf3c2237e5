// rt_refwalk.h -- the reference's own traversal, restated for the device (included by rt_device.h, which
// supplies DevScene, the vector helpers, tri_test and prim_intersect).
#pragma once

namespace rtd {

// ---------------------------------------------------------------------------------------------
// The reference's own walk (BVH::intersect, main.cpp:560-605) over the reference's own tree (S.rnodes /
// S.ritems, built by ref_tree.cpp): for the results that depend on the ORDER in which primitives are met.
//   * two primitives at exactly the same distance: the first met wins (strict `<`, main.cpp:566-567, 589-590);
//   * a ray in the plane of a triangle: the 3x3 inverse divides by zero, every comparison of main.cpp:483 fails
//     and the "hit" at distance NaN is kept if it is the first of its leaf (566), after which nothing
//     compares closer and no further subtree is entered (`distance < NaN`, 587, 596);
//   * a hit at t >= INF survives if it is the first of its leaf and nothing nearer is found later;
//   * every call starts from minDistance = its own closest hit (main.cpp:571): a subtree is cut only by its
//     sibling's hit, and the boxes are twice their size (center +- (max - min), 174-177), origin inside -> 0.
// The recursion is unrolled over a stack of frames, one per call that is waiting for a child: the call's
// closest hit so far and, while the first child runs, the second child with its box distance.  A frame's
// `right` is nonzero exactly while the first child is out (children are never the root, node 0).
// The frames live where the caller says (`frames`, one every `stride` elements): in private memory (1 KB per
// lane; the validation kernels and the megakernel) or in a global buffer the host sized for the tree's depth
// (wf_extend_retry_ref: lane-interleaved, so that neighbouring lanes' frames of one level share lines).  At most
// `depth` frames are used; the host guarantees tree depth <= depth <= kRefStack.
// ---------------------------------------------------------------------------------------------
constexpr int kRefStack = 64;
struct RefFrame {
    float rd, t;
    int rec;
    uint32_t right;
};
constexpr float kBigInf = 20000.f;  // BIG_INF, main.cpp:24

// AABB::intersect, main.cpp:180-196, on a node record (sizePlusCenter, negSizePlusCenter, minPoint, maxPoint)
__device__ __forceinline__ float ref_box_distance(const float4 *n, v3 o, v3 d) {
    const float4 q0 = n[0], q1 = n[1], q2 = n[2];
    const v3 spc = mk(q0.x, q0.y, q0.z), nspc = mk(q0.w, q1.x, q1.y), lo = mk(q1.z, q1.w, q2.x), hi = mk(q2.y, q2.z, q2.w);
    if (hi.x > o.x && hi.y > o.y && hi.z > o.z && lo.x < o.x && lo.y < o.y && lo.z < o.z) return 0.f;
    const v3 t1 = vdiv(spc - o, d), t2 = vdiv(nspc - o, d);
    const float d1 = ref_max(ref_max(ref_min(t1.x, t2.x), ref_min(t1.y, t2.y)), ref_min(t1.z, t2.z));
    const float d2 = ref_min(ref_min(ref_max(t1.x, t2.x), ref_max(t1.y, t2.y)), ref_max(t1.z, t2.z));
    if (d1 > d2 || d2 < 0.f) return kBigInf;
    return d1;
}

// tOut / recOut: distance and record index (triangle record, or S.ntri + k) of the reference's answer;
// recOut < 0 and tOut = INF when it hits nothing.  tOut may be NaN or >= INF for a hit (see above).
// PRUNE: a child is not entered when the ray misses the box of everything that can be HIT below it (S.rreach: the
// triangles' and primitives' real extents, padded), even where the reference enters it (its slabs are twice the
// node's box).  Nothing in there can be hit, so the reference's call returns "nothing" and changes neither its
// answer nor any later decision; skipping it keeps every answer that is a matter of real hits -- equal and nearly
// equal distances above all -- and takes the walk from hundreds of node visits, each a dependent trip to memory,
// to a few dozen (one ray of a 17 930-triangle scene: 1.1 ms -> under 0.1 ms; the kernel that runs this finds a
// handful of rays per bounce and lasts as long as the slowest of them).  What it gives up are the NaN "hits" of
// rays lying in a triangle's plane, which are not confined to any box: the validation kernels walk with
// PRUNE = false.  (The node's own corner points will not do: the reference files an analytic Box under a box of
// half its size.)
__device__ __forceinline__ bool ref_reach_missed(const float4 *n, v3 o, v3 d) {
    const float4 q0 = n[0], q1 = n[1];
    const v3 lo = mk(q0.x, q0.y, q0.z), hi = mk(q0.w, q1.x, q1.y);
    // Padded far beyond the rounding of a hit point (1e-6 of its coordinates and of the distance travelled), and
    // in proportion to the distance of the origin as well: what the reference's own arithmetic reports as hit
    // blurs with it.  Its ellipsoid test squares (o - centre) / radius and subtracts 1 (main.cpp:290-296); from
    // 4000 units away (a ray coming back from a plane it met out there) that difference is noise, and a ray
    // passing 1.4 units beside a sphere of radius 0.15 "hits" it: |o| * 2^-11.5.  The reference reports such hits
    // wherever its doubled slabs let the walk reach the primitive, so the pruned walk must not skip them either.
    const float far = 1.0e-3f * __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
    const v3 pad = mk(far + 1.0e-3f + 1.0e-3f * __builtin_fmaxf(__builtin_fabsf(lo.x), __builtin_fabsf(hi.x)),
                      far + 1.0e-3f + 1.0e-3f * __builtin_fmaxf(__builtin_fabsf(lo.y), __builtin_fabsf(hi.y)),
                      far + 1.0e-3f + 1.0e-3f * __builtin_fmaxf(__builtin_fabsf(lo.z), __builtin_fabsf(hi.z)));
    const v3 a = vdiv((lo - pad) - o, d), b = vdiv((hi + pad) - o, d);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)), __builtin_fminf(a.z, b.z));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)), __builtin_fmaxf(a.z, b.z));
    // fmin / fmax drop a NaN operand (0 / 0: origin on a padded plane of an axis the ray does not move along): that
    // axis then does not constrain.  Missed only when the comparison says so.
    return tn > tf * 1.0001f + 1.0e-3f || tf < -1.0e-3f;
}
template <bool COUNT, bool PRUNE = false>
__device__ __forceinline__ void ref_walk(const DevScene &S, v3 o, v3 d, RefFrame *frames, int stride, int depth, float &tOut,
                                         int &recOut, TraceStats &st) {
    typedef RefFrame Frame;
    int sp = 0;
    uint32_t cur = 0;
    float rt = kInf;  // what the call that just returned hands back
    int rrec = -1;
    for (;;) {
        const float4 q3 = S.rnodes[4 * (size_t)cur + 3];
        uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
        const uint32_t first = __float_as_uint(q3.z), count = __float_as_uint(q3.w);
        if (COUNT) ++st.nodes;
        float ct = kInf;  // closestIntersection of this call (main.cpp:562): t = INF until something is hit
        int crec = -1;
        for (uint32_t i = first; i < first + count; ++i) {
            const uint32_t rec = S.ritems[i];
            bool hit;
            float t;
            if (COUNT) ++st.tris;
            if (rec < S.ntri) {
                const float4 *tp = S.tris + 4 * (size_t)rec;
                hit = tri_test(tp[0], tp[1], tp[2], o, d, t);
            } else {
                const PrimHit h = prim_intersect<false>(S.prims + 3 * (size_t)(rec - S.ntri), o, d);
                hit = h.hit;
                t = h.t;
            }
            if (hit && (crec < 0 || t < ct)) {
                ct = t;
                crec = (int)rec;
            }
        }
        if (left | right) {  // a leaf's children are both node 0 = the root and filtered (main.cpp:587, 596)
            float ld = ref_box_distance(S.rnodes + 4 * (size_t)left, o, d);
            float rd = ref_box_distance(S.rnodes + 4 * (size_t)right, o, d);
            if (ld > rd) {  // main.cpp:579-585, the swap as written there
                const uint32_t l = left;
                left = right;
                right = l;
                ld += rd;
                rd = ld - rd;
                ld -= rd;
            }
            if (PRUNE) {  // as if the reference had walked the child and found nothing
                if (ref_reach_missed(S.rreach + 2 * (size_t)left, o, d)) ld = kBigInf;
                if (ref_reach_missed(S.rreach + 2 * (size_t)right, o, d)) rd = kBigInf;
            }
            if (sp < depth) {
                if (ld < ct) {  // minDistance = closestIntersection.t, main.cpp:571
                    frames[sp++ * stride] = Frame{rd, ct, crec, right};
                    cur = left;
                    continue;
                }
                if (rd < ct) {
                    frames[sp++ * stride] = Frame{0.f, ct, crec, 0u};
                    cur = right;
                    continue;
                }
            }
        }
        rt = ct;
        rrec = crec;
        bool descend = false;
        while (sp > 0) {  // return into the waiting calls
            Frame &f = frames[(sp - 1) * stride];
            if (rrec >= 0 && (f.rec < 0 || rt < f.t)) {  // main.cpp:589-593, 598-601
                f.t = rt;
                f.rec = rrec;
            }
            if (f.right != 0u) {  // back from the first child; minDistance is the closest hit's t either way
                const uint32_t r = f.right;
                f.right = 0u;
                if (f.rd < f.t) {
                    cur = r;
                    descend = true;
                    break;
                }
            }
            rt = f.t;
            rrec = f.rec;
            --sp;
        }
        if (!descend) break;
    }
    tOut = rt;
    recOut = rrec;
}
template <bool COUNT, bool PRUNE = false>
__device__ __noinline__ void ref_closest_hit(const DevScene &S, v3 o, v3 d, float &tOut, int &recOut, TraceStats &st) {
    RefFrame fr[kRefStack];
    ref_walk<COUNT, PRUNE>(S, o, d, fr, 1, kRefStack, tOut, recOut, st);
}

}  // namespace rtd
