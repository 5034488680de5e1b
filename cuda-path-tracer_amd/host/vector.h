/*
 * vector.h -- math primitives of the host API: same types, members and operators as the
 * reference's src/vector.cuh:8-192 (ivec2, vec2, vec3, mat3, dot, cross, Hadamard * and /,
 * scalar ops, stream output), so scene scripts written against the reference compile unchanged.
 * Host-only: the device side has its own types.
 */
#ifndef PTAMD_HOST_VECTOR_H
#define PTAMD_HOST_VECTOR_H

#include <cmath>
#include <iostream>

struct ivec2 {
    int x, y;
    ivec2() {}
    ivec2(int x, int y) : x(x), y(y) {}
};

struct vec2 {
    float x, y;
    vec2() {}
    vec2(float x, float y) : x(x), y(y) {}
    vec2 operator+(vec2 v) const { return vec2(x + v.x, y + v.y); }
    vec2 operator-(vec2 v) const { return vec2(x - v.x, y - v.y); }
    vec2 normalized() const {
        float rnorm = 1.0f / std::sqrt(x * x + y * y);
        return vec2(x * rnorm, y * rnorm);
    }
    void normalize() { *this = normalized(); }
};

struct vec3 {
    float x, y, z;
    vec3() {}
    vec3(float v) : x(v), y(v), z(v) {}
    vec3(float x, float y, float z) : x(x), y(y), z(z) {}
    vec3 operator+(vec3 v) const { return vec3(x + v.x, y + v.y, z + v.z); }
    vec3 operator-(vec3 v) const { return vec3(x - v.x, y - v.y, z - v.z); }
    vec3 operator-() const { return vec3(-x, -y, -z); }
    /* 1/sqrt evaluated as the correctly rounded 1.0f / sqrtf(s), the definition the device side
     * and the CPU oracle share */
    vec3 normalized() const {
        float rnorm = 1.0f / std::sqrt(x * x + y * y + z * z);
        return vec3(x * rnorm, y * rnorm, z * rnorm);
    }
    void normalize() { *this = normalized(); }
};

inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return vec3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
inline float distance(vec3 a, vec3 b) {
    float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}
inline vec3 min(vec3 a, vec3 b) { return vec3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec2 operator*(float s, vec2 v) { return vec2(s * v.x, s * v.y); }
inline vec3 operator*(float s, vec3 v) { return vec3(s * v.x, s * v.y, s * v.z); }
inline vec3 operator*(vec3 v, float s) { return operator*(s, v); }
inline vec3 operator/(vec3 v, float s) { return vec3(v.x / s, v.y / s, v.z / s); }
inline vec3 operator-(float s, vec3 v) { return vec3(s - v.x, s - v.y, s - v.z); }

inline std::ostream &operator<<(std::ostream &os, const ivec2 &v) { return os << "(" << v.x << ", " << v.y << ")"; }
inline std::ostream &operator<<(std::ostream &os, const vec2 &v) { return os << "(" << v.x << ", " << v.y << ")"; }
inline std::ostream &operator<<(std::ostream &os, const vec3 &v) { return os << "(" << v.x << ", " << v.y << ", " << v.z << ")"; }

struct mat3 {
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;
    mat3() {}
    mat3(vec3 c0, vec3 c1, vec3 c2)
        : m00(c0.x), m01(c1.x), m02(c2.x), m10(c0.y), m11(c1.y), m12(c2.y), m20(c0.z), m21(c1.z), m22(c2.z) {}
    mat3(float m00, float m01, float m02, float m10, float m11, float m12, float m20, float m21, float m22)
        : m00(m00), m01(m01), m02(m02), m10(m10), m11(m11), m12(m12), m20(m20), m21(m21), m22(m22) {}
};

inline vec3 operator*(vec3 v, const mat3 &m) {
    return vec3(v.x * m.m00 + v.y * m.m01 + v.z * m.m02, v.x * m.m10 + v.y * m.m11 + v.z * m.m12,
                v.x * m.m20 + v.y * m.m21 + v.z * m.m22);
}

inline float luminosity(vec3 color) { return dot(color, vec3(0.2126f, 0.7152f, 0.0722f)); }   /* src/util.cuh:40-42 */

#endif
