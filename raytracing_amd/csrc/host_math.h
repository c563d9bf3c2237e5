// host_math.h -- fp32 vector / matrix helpers for the host side (scene loading, BVH build).
//
// Operation order follows glm 0.9.9.8 (the reference's math library) so that a scene file
// turns into the same float bits the reference computes: one rounding per operation, sums
// evaluated left to right.  Compile with -ffp-contract=off.
#pragma once
#include <cmath>

namespace hm {

struct vec3 {
    float x, y, z;
};
struct vec4 {
    float x, y, z, w;
};
struct quat {
    float w, x, y, z;
};
struct mat4 {
    vec4 c[4];  // columns; c[col].{x,y,z,w} = rows 0..3
};

inline vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 operator+(vec3 a, vec3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.f / std::sqrt(dot(v, v))); }

inline vec4 operator*(vec4 a, float s) { return vec4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator+(vec4 a, vec4 b) { return vec4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

inline float &elem(vec4 &v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }
inline float elem(const vec4 &v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

// glm's 16-scalar mat4 constructor: the arguments fill column 0 first.
inline mat4 mat4_from16(const float *v) {
    mat4 m;
    for (int c = 0; c < 4; ++c) m.c[c] = vec4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
    return m;
}
inline mat4 transpose(const mat4 &m) {
    mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) elem(r.c[c], k) = elem(m.c[k], c);
    return r;
}
// R[i] = ((A[0]*B[i][0] + A[1]*B[i][1]) + A[2]*B[i][2]) + A[3]*B[i][3]
inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int i = 0; i < 4; ++i)
        r.c[i] = ((a.c[0] * b.c[i].x + a.c[1] * b.c[i].y) + a.c[2] * b.c[i].z) + a.c[3] * b.c[i].w;
    return r;
}
// (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
inline vec4 operator*(const mat4 &m, vec4 v) {
    return (m.c[0] * v.x + m.c[1] * v.y) + (m.c[2] * v.z + m.c[3] * v.w);
}

// glm::mat3_cast widened to 4x4 (glm::toMat4)
inline mat4 to_mat4(quat q) {
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    mat4 m;
    m.c[0] = vec4{1.f - 2.f * (qyy + qzz), 2.f * (qxy + qwz), 2.f * (qxz - qwy), 0.f};
    m.c[1] = vec4{2.f * (qxy - qwz), 1.f - 2.f * (qxx + qzz), 2.f * (qyz + qwx), 0.f};
    m.c[2] = vec4{2.f * (qxz + qwy), 2.f * (qyz - qwx), 1.f - 2.f * (qxx + qyy), 0.f};
    m.c[3] = vec4{0.f, 0.f, 0.f, 1.f};
    return m;
}

// glm::quat_cast of the upper 3x3 (glm::toQuat)
inline quat to_quat(const mat4 &m) {
    float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z;
    float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z;
    float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z;
    float fx = m00 - m11 - m22;
    float fy = m11 - m00 - m22;
    float fz = m22 - m00 - m11;
    float fw = m00 + m11 + m22;
    int idx = 0;
    float big = fw;
    if (fx > big) {
        big = fx;
        idx = 1;
    }
    if (fy > big) {
        big = fy;
        idx = 2;
    }
    if (fz > big) {
        big = fz;
        idx = 3;
    }
    float val = std::sqrt(big + 1.f) * 0.5f;
    float mult = 0.25f / val;
    switch (idx) {
        case 0: return quat{val, (m12 - m21) * mult, (m20 - m02) * mult, (m01 - m10) * mult};
        case 1: return quat{(m12 - m21) * mult, val, (m01 + m10) * mult, (m20 + m02) * mult};
        case 2: return quat{(m20 - m02) * mult, (m01 + m10) * mult, val, (m12 + m21) * mult};
        default: return quat{(m01 - m10) * mult, (m20 + m02) * mult, (m12 + m21) * mult, val};
    }
}

// glm quat * vec3
inline vec3 rotate(quat q, vec3 v) {
    vec3 qv = V3(q.x, q.y, q.z);
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.f;
}

}  // namespace hm
