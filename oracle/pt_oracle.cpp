/*
 * pt_oracle.cpp -- CPU oracle: a plain C++ / IEEE fp32 restatement of the reference path
 * tracer's hot path.  TEST INFRASTRUCTURE ONLY (see pt_oracle.h): it checks the HIP product and
 * serves as the timed CPU baseline; the product never links it.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference's
 * root).  Build with -ffp-contract=off and without -ffast-math so that every +,-,*,/ and sqrt is
 * a single correctly rounded fp32 operation in source order -- the HIP side is built the same
 * way, which is what makes hit ids, distances and barycentrics comparable bit for bit.
 *
 * Deliberate deviations from the reference (all are places where the reference has undefined
 * behaviour or a data race; each is repeated where it occurs):
 *   - RNG is Philox4x32-10 keyed by (seed, 0) with counter (pixel, sample, block, 0) instead of
 *     one shared curand XORWOW state per pixel (src/renderer.cu:26,43 races on it).
 *   - meshes without uvs keep the raw barycentrics in (u,v) instead of reading d_uvs[-1]
 *     (src/device_geometry_functions.cuh:216-218 reads out of bounds).
 *   - a sphere hit clears the hit's triangle and sets u = v = 0 (the reference leaves a stale
 *     triangle pointer / uninitialised u,v, src/device_geometry_functions.cuh:265-273).
 *   - texture and HDRI texel indices are clamped into the image (src/texture.cuh:22-43 and
 *     src/scene.cuh:52-54 can index one row/column past the end).
 */
#include "pt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace {

/* ------------------------------------------------------------------ math (src/vector.cuh) */
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };

inline vec3 V(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
inline vec3 V(float v) { return V(v, v, v); }
inline vec2 V2(float x, float y) { vec2 r = {x, y}; return r; }

inline vec3 operator+(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vector.cuh:60 */
inline vec3 operator-(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :63 */
inline vec3 operator-(vec3 a) { return V(-a.x, -a.y, -a.z); }                          /* :66 */
inline vec2 operator+(vec2 a, vec2 b) { return V2(a.x + b.x, a.y + b.y); }             /* :25 */
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         /* :89 */
inline vec3 cross(vec3 a, vec3 b) {                                                    /* :92 */
    return V((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
inline vec3 operator*(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }   /* :113 */
inline vec3 operator/(vec3 a, vec3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }   /* :118 */
inline vec2 operator*(float s, vec2 v) { return V2(s * v.x, s * v.y); }                /* :123 */
inline vec3 operator*(float s, vec3 v) { return V(s * v.x, s * v.y, s * v.z); }        /* :126 */
inline vec3 operator*(vec3 v, float s) { return s * v; }                               /* :129 */
inline vec3 operator/(vec3 v, float s) { return V(v.x / s, v.y / s, v.z / s); }        /* :132 */
inline vec3 operator-(float s, vec3 v) { return V(s - v.x, s - v.y, s - v.z); }        /* :135 */
inline vec3 vmin(vec3 a, vec3 b) { return V(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); } /* :108 */

/* The reference normalises with rsqrt (host) / __frsqrt_rn (device), vector.cuh:69-76.  Both
 * sides of this build define it as the correctly rounded 1/sqrt(x) in two IEEE operations. */
inline float rsqrt_ieee(float x) { return 1.0f / sqrtf(x); }
inline vec3 normalized(vec3 v) {
    float rnorm = rsqrt_ieee(v.x * v.x + v.y * v.y + v.z * v.z);
    return V(v.x * rnorm, v.y * rnorm, v.z * rnorm);
}

struct mat3 { vec3 c0, c1, c2; };                      /* vector.cuh:153-184: columns */
inline vec3 mul(vec3 v, const mat3 &m) {               /* vector.cuh:186-192 */
    return V(v.x * m.c0.x + v.y * m.c1.x + v.z * m.c2.x,
             v.x * m.c0.y + v.y * m.c1.y + v.z * m.c2.y,
             v.x * m.c0.z + v.y * m.c1.z + v.z * m.c2.z);
}

inline float luminosity(vec3 c) { return dot(c, V(0.2126f, 0.7152f, 0.0722f)); }       /* util.cuh:40 */

template <typename T> inline T bary_lerp(T a, T b, T c, float u, float v) {            /* math_util.cuh:9 */
    float w = 1.0f - (u + v);
    return u * b + v * c + w * a;
}
template <typename T> inline T lerp(T a, T b, float s) { return (1.0f - s) * a + s * b; } /* :15 */
template <typename T> inline T bilerp(T ul, T ur, T ll, T lr, float u, float v) {      /* :20 */
    T left = lerp(ul, ll, v);
    T right = lerp(ur, lr, v);
    return lerp(left, right, u);
}

/* ------------------------------------------------------------------ geometry types */
const float EPSILON_DET = 0.000000001f;   /* geometry.cuh:5 */
const float MAX_RDIR = 1000000.0f;        /* geometry.cuh:6 */
const float AABB_PADDING = 0.0001f;       /* geometry.cuh:7 */
const float PI_R = 3.14159265f;           /* renderer.cuh:4 */

struct Ray {                                /* geometry.cuh:17-32 */
    vec3 origin, direction, fracs;
    void recalc_fracs() {
        fracs.x = fmaxf(fminf(1.0f / direction.x, MAX_RDIR), -MAX_RDIR);
        fracs.y = fmaxf(fminf(1.0f / direction.y, MAX_RDIR), -MAX_RDIR);
        fracs.z = fmaxf(fminf(1.0f / direction.z, MAX_RDIR), -MAX_RDIR);
    }
};
inline Ray make_ray(vec3 o, vec3 d) { Ray r; r.origin = o; r.direction = d; r.recalc_fracs(); return r; }

struct Vertex { vec3 position, normal; };                     /* geometry.cuh:34-41 */
struct Triangle { int idx_a, idx_b, idx_c, vt_a, vt_b, vt_c; }; /* geometry.cuh:43-63 */

struct AABB {                                                 /* geometry.cuh:79-91 */
    vec3 min, max;
};
inline AABB padded_aabb(vec3 mn, vec3 mx) {                   /* geometry.cuh:81-84 */
    AABB b;
    b.min = mn - V(AABB_PADDING, AABB_PADDING, AABB_PADDING);
    b.max = mx + V(AABB_PADDING, AABB_PADDING, AABB_PADDING);
    return b;
}

struct Octree {                                               /* geometry.cuh:93-133 */
    Octree *children[8];
    std::vector<int> triangle_indices;
    int depth;
    AABB region;
    Octree(AABB r, int d) : depth(d), region(r) { for (int i = 0; i < 8; i++) children[i] = nullptr; }
    ~Octree() { for (int i = 0; i < 8; i++) delete children[i]; }
};

struct Texture { std::vector<uint8_t> data; int w = 0, h = 0; bool set = false; };

struct Material {                                             /* material.cuh:11-59 */
    vec3 albedo = V(0.8f), specular = V(0.0f);
    float emission = 0.0f, smoothness = 0.0f;
    bool smooth_shading = false;
    Texture albedo_map, specular_map, normal_map, smoothness_map;
};

enum Shape { SPHERE = 0, TRIANGLE_MESH = 1 };                 /* geometry.cuh:15 */

struct Entity {                                               /* geometry.cuh:135-206 */
    int shape = SPHERE;
    Octree *octree = nullptr;
    int max_depth = 6;
    std::vector<Vertex> vertices;
    std::vector<Triangle> triangles;
    std::vector<vec2> uvs;
    AABB aabb;
    float radius = 0.0f;
    vec3 center = V(0.0f);
    int material = 0;
    ~Entity() { delete octree; }
};

struct Intersection {                                         /* geometry.cuh:67-77 */
    vec3 position = V(0.0f), normal = V(0.0f);
    float distance = FLT_MAX;
    int entity = -1;
    int triangle = -1;
    float u = 0.0f, v = 0.0f;
};

struct Counters {
    uint64_t segments = 0, aabb_tests = 0, node_visits = 0, tri_tests = 0, env_lookups = 0,
             tex_fetches = 0, paths = 0;
    void add(const Counters &o) {
        segments += o.segments; aabb_tests += o.aabb_tests; node_visits += o.node_visits;
        tri_tests += o.tri_tests; env_lookups += o.env_lookups; tex_fetches += o.tex_fetches;
        paths += o.paths;
    }
};

struct TraceCtx {           /* per-thread bookkeeping that the reference does not have */
    Counters c;
    int32_t *log = nullptr; /* [0] = count, then triangle ids */
    int log_cap = 0;
};

} // namespace

struct orc_scene {
    std::vector<Entity *> entities;
    std::vector<Material> materials;
    std::vector<float> hdri;
    int hdri_w = 0, hdri_h = 0;
    float hdri_exposure = 1.0f, hdri_contrast = 1.0f, hdri_rot_offset = 0.0f;
    bool has_hdri = false, use_smooth_hdri = false;
    ~orc_scene() { for (Entity *e : entities) delete e; }
};

namespace {

/* ------------------------------------------------------------------ octree build (host) */

/* src/aabb.cu:64-120 -- SAT test, 9 edge axes + 3 box axes, plane test disabled. */
bool aabb_intersects_triangle(const AABB &box, vec3 u0, vec3 u1, vec3 u2) {
    vec3 aabb_center = 0.5f * (box.min + box.max);
    vec3 extents = (box.max - box.min) / 2.0f;

    vec3 v[3] = {u0 - aabb_center, u1 - aabb_center, u2 - aabb_center};
    vec3 f[3] = {v[1] - v[0], v[2] - v[1], v[0] - v[2]};
    vec3 e[3] = {V(1.0f, 0.0f, 0.0f), V(0.0f, 1.0f, 0.0f), V(0.0f, 0.0f, 1.0f)};

    vec3 a[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            a[i][j] = cross(e[i], f[j]);

    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            float p0 = dot(v[0], a[i][j]);
            float p1 = dot(v[1], a[i][j]);
            float p2 = dot(v[2], a[i][j]);
            float r = extents.x * fabsf(dot(e[0], a[i][j])) +
                      extents.y * fabsf(dot(e[1], a[i][j])) +
                      extents.z * fabsf(dot(e[2], a[i][j]));
            float p_min = fminf(p0, fminf(p1, p2));
            float p_max = fmaxf(p0, fmaxf(p1, p2));
            if (fmaxf(-p_max, p_min) > r)
                return false;
        }
    }

    float min_x = fminf(v[0].x, fminf(v[1].x, v[2].x));
    float min_y = fminf(v[0].y, fminf(v[1].y, v[2].y));
    float min_z = fminf(v[0].z, fminf(v[1].z, v[2].z));
    float max_x = fmaxf(v[0].x, fmaxf(v[1].x, v[2].x));
    float max_y = fmaxf(v[0].y, fmaxf(v[1].y, v[2].y));
    float max_z = fmaxf(v[0].z, fmaxf(v[1].z, v[2].z));
    if (min_x > extents.x || max_x < -extents.x) return false;
    if (min_y > extents.y || max_y < -extents.y) return false;
    if (min_z > extents.z || max_z < -extents.z) return false;
    return true;
}

/* src/octree.cu:83-128 -- top-down insertion; leaves only at max_depth. */
void insert_triangle(Octree *node, int max_depth, vec3 v0, vec3 v1, vec3 v2, int triangle_idx) {
    float x_min = node->region.min.x, y_min = node->region.min.y, z_min = node->region.min.z;
    float x_max = node->region.max.x, y_max = node->region.max.y, z_max = node->region.max.z;
    float x_mid = 0.5f * (x_min + x_max);
    float y_mid = 0.5f * (y_min + y_max);
    float z_mid = 0.5f * (z_min + z_max);

    if (node->depth == max_depth) {
        node->triangle_indices.push_back(triangle_idx);
        return;
    }

    AABB c[8];
    c[0] = padded_aabb(V(x_min, y_min, z_min), V(x_mid, y_mid, z_mid));
    c[4] = padded_aabb(V(x_mid, y_min, z_min), V(x_max, y_mid, z_mid));
    c[1] = padded_aabb(V(x_min, y_min, z_mid), V(x_mid, y_mid, z_max));
    c[5] = padded_aabb(V(x_mid, y_min, z_mid), V(x_max, y_mid, z_max));
    c[2] = padded_aabb(V(x_min, y_mid, z_min), V(x_mid, y_max, z_mid));
    c[6] = padded_aabb(V(x_mid, y_mid, z_min), V(x_max, y_max, z_mid));
    c[3] = padded_aabb(V(x_min, y_mid, z_mid), V(x_mid, y_max, z_max));
    c[7] = padded_aabb(V(x_mid, y_mid, z_mid), V(x_max, y_max, z_max));

    for (int i = 0; i < 8; i++) {
        if (!aabb_intersects_triangle(c[i], v0, v1, v2))
            continue;
        if (node->children[i] == nullptr)
            node->children[i] = new Octree(c[i], node->depth + 1);
        insert_triangle(node->children[i], max_depth, v0, v1, v2, triangle_idx);
    }
}

/* src/entity.cu:167-170 + src/octree.cu:133-142 */
void construct_octree(Entity *e, int max_depth) {
    e->max_depth = max_depth;
    e->octree = new Octree(e->aabb, 0);
    for (size_t i = 0; i < e->triangles.size(); i++) {
        const Triangle &t = e->triangles[i];
        insert_triangle(e->octree, max_depth, e->vertices[t.idx_a].position,
                        e->vertices[t.idx_b].position, e->vertices[t.idx_c].position, (int)i);
    }
}

/* src/aabb.cu:7-22 */
AABB recalc_aabb(const Vertex *vertices, int n) {
    vec3 mn = V(FLT_MAX), mx = V(-FLT_MAX);
    for (int i = 0; i < n; i++) {
        vec3 p = vertices[i].position;
        mn.x = (p.x < mn.x) ? p.x : mn.x;
        mn.y = (p.y < mn.y) ? p.y : mn.y;
        mn.z = (p.z < mn.z) ? p.z : mn.z;
        mx.x = (p.x > mx.x) ? p.x : mx.x;
        mx.y = (p.y > mx.y) ? p.y : mx.y;
        mx.z = (p.z > mx.z) ? p.z : mx.z;
    }
    return padded_aabb(mn, mx);
}

/* ------------------------------------------------------------------ intersection (device fns) */

/* src/device_geometry_functions.cuh:388-417 */
bool intersects_aabb(const AABB &b, const Ray &ray, const Intersection &bestHit) {
    float tx1 = (b.min.x - ray.origin.x) * ray.fracs.x;
    float tx2 = (b.max.x - ray.origin.x) * ray.fracs.x;
    float ty1 = (b.min.y - ray.origin.y) * ray.fracs.y;
    float ty2 = (b.max.y - ray.origin.y) * ray.fracs.y;
    float tz1 = (b.min.z - ray.origin.z) * ray.fracs.z;
    float tz2 = (b.max.z - ray.origin.z) * ray.fracs.z;

    float tmin = fminf(tx1, tx2);
    float tmax = fmaxf(tx1, tx2);
    tmin = fmaxf(tmin, fminf(ty1, ty2));
    tmax = fminf(tmax, fmaxf(ty1, ty2));
    tmin = fmaxf(tmin, fminf(tz1, tz2));
    tmax = fminf(tmax, fmaxf(tz1, tz2));

    if (tmax < 0.0f || tmin > tmax || bestHit.distance < tmin)
        return false;
    return true;
}

/* src/device_geometry_functions.cuh:319-365 -- Moller-Trumbore; normal is not flipped. */
bool intersect_triangle(vec3 v0, vec3 v1, vec3 v2, int triangle, int entity, Intersection &bestHit,
                        const Ray &ray, TraceCtx &ctx) {
    ctx.c.tri_tests++;
    if (ctx.log) {
        int n = ctx.log[0]++;
        if (n + 1 < ctx.log_cap) ctx.log[n + 1] = triangle;
    }
    vec3 e1 = v1 - v0;
    vec3 e2 = v2 - v0;

    vec3 pvec = cross(ray.direction, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < EPSILON_DET)
        return false;

    float inv_det = 1.0f / det;
    vec3 tvec = ray.origin - v0;
    float u = dot(tvec, pvec) * inv_det;
    if (u < -0.0001f || u > 1.0001f)
        return false;

    vec3 qvec = cross(tvec, e1);
    float v = dot(ray.direction, qvec) * inv_det;
    if (v < -0.0001f || u + v > 1.0001f)
        return false;

    float t = dot(e2, qvec) * inv_det;

    if (t > 0 && t < bestHit.distance) {
        bestHit.distance = t;
        bestHit.position = ray.origin + t * ray.direction;
        bestHit.normal = normalized(cross(e1, e2));
        bestHit.entity = entity;
        bestHit.triangle = triangle;
        bestHit.u = u;
        bestHit.v = v;
        return true;
    }
    return false;
}

bool entity_intersects_triangle(const Entity &e, int eid, int tri, Intersection &bestHit,
                                const Ray &ray, TraceCtx &ctx) {   /* dgf:306-311 */
    const Triangle &t = e.triangles[tri];
    return intersect_triangle(e.vertices[t.idx_a].position, e.vertices[t.idx_b].position,
                              e.vertices[t.idx_c].position, tri, eid, bestHit, ray, ctx);
}

/* src/device_geometry_functions.cuh:95-110 */
unsigned char find_first_node(vec3 t0, vec3 tM) {
    unsigned char answer = 0;
    if (t0.x > t0.y && t0.x > t0.z) {
        if (tM.y < t0.x) answer |= 2;
        if (tM.z < t0.x) answer |= 1;
        return answer;
    } else if (t0.y > t0.z) {
        if (tM.x < t0.y) answer |= 4;
        if (tM.z < t0.y) answer |= 1;
        return answer;
    }
    if (tM.x < t0.z) answer |= 4;
    if (tM.y < t0.z) answer |= 2;
    return answer;
}

/* src/device_geometry_functions.cuh:116-126 */
unsigned char next_node(float txm, float tym, float tzm, unsigned char x, unsigned char y,
                        unsigned char z) {
    if (txm < tym) {
        if (txm < tzm) return x;
    } else {
        if (tym < tzm) return y;
    }
    return z;
}

/* src/device_geometry_functions.cuh:132-191 -- recursive; note the (t1.y < 0) tested twice and
 * t1.z never (line 142), and the early-out as soon as a child reported a hit (line 188). */
bool proc_subtree(const Octree *node, const Entity &ent, int eid, unsigned char a, vec3 t0, vec3 t1,
                  const Ray &ray, Intersection &bestHit, TraceCtx &ctx) {
    ctx.c.node_visits++;
    if ((t1.x < 0.0f) || (t1.y < 0.0f) || (t1.y < 0.0f))
        return false;

    if (node->depth == ent.max_depth) {
        bool hit = false;
        for (size_t i = 0; i < node->triangle_indices.size(); i++) {
            int tri_idx = node->triangle_indices[i];
            hit = entity_intersects_triangle(ent, eid, tri_idx, bestHit, ray, ctx) || hit;
        }
        return hit;
    }

    vec3 tM = 0.5f * (t0 + t1);
    unsigned char curr_node = find_first_node(t0, tM);
    bool hit = false;
    do {
        unsigned char adj = curr_node ^ a;
        const Octree *ch = node->children[adj];
        if (ch != nullptr) {
            switch (curr_node) {
                case 0: hit = proc_subtree(ch, ent, eid, a, V(t0.x, t0.y, t0.z), V(tM.x, tM.y, tM.z), ray, bestHit, ctx) || hit; break;
                case 1: hit = proc_subtree(ch, ent, eid, a, V(t0.x, t0.y, tM.z), V(tM.x, tM.y, t1.z), ray, bestHit, ctx) || hit; break;
                case 2: hit = proc_subtree(ch, ent, eid, a, V(t0.x, tM.y, t0.z), V(tM.x, t1.y, tM.z), ray, bestHit, ctx) || hit; break;
                case 3: hit = proc_subtree(ch, ent, eid, a, V(t0.x, tM.y, tM.z), V(tM.x, t1.y, t1.z), ray, bestHit, ctx) || hit; break;
                case 4: hit = proc_subtree(ch, ent, eid, a, V(tM.x, t0.y, t0.z), V(t1.x, tM.y, tM.z), ray, bestHit, ctx) || hit; break;
                case 5: hit = proc_subtree(ch, ent, eid, a, V(tM.x, t0.y, tM.z), V(t1.x, tM.y, t1.z), ray, bestHit, ctx) || hit; break;
                case 6: hit = proc_subtree(ch, ent, eid, a, V(tM.x, tM.y, t0.z), V(t1.x, t1.y, tM.z), ray, bestHit, ctx) || hit; break;
                case 7: hit = proc_subtree(ch, ent, eid, a, V(tM.x, tM.y, tM.z), V(t1.x, t1.y, t1.z), ray, bestHit, ctx) || hit; break;
            }
        }
        switch (curr_node) {
            case 0: curr_node = next_node(tM.x, tM.y, tM.z, 4, 2, 1); break;
            case 1: curr_node = next_node(tM.x, tM.y, t1.z, 5, 3, 8); break;
            case 2: curr_node = next_node(tM.x, t1.y, tM.z, 6, 8, 3); break;
            case 3: curr_node = next_node(tM.x, t1.y, t1.z, 7, 8, 8); break;
            case 4: curr_node = next_node(t1.x, tM.y, tM.z, 8, 6, 5); break;
            case 5: curr_node = next_node(t1.x, tM.y, t1.z, 8, 7, 8); break;
            case 6: curr_node = next_node(t1.x, t1.y, tM.z, 8, 8, 7); break;
            case 7: curr_node = 8; break;
        }
    } while (curr_node < 8 && !hit);
    return hit;
}

/* src/device_geometry_functions.cuh:39-89 -- Revelles set-up; triangle tests use the original ray */
bool ray_step(const Entity &ent, int eid, const Ray &ray, Intersection &bestHit, TraceCtx &ctx) {
    const Octree *root = ent.octree;
    Ray r = ray;
    vec3 oct_mid = 0.5f * (root->region.max + root->region.min);

    unsigned char a = 0;
    if (r.direction.x < 0.0f) {
        float delta = r.origin.x - oct_mid.x;
        r.origin.x = oct_mid.x - delta;
        r.direction.x = -(r.direction.x);
        a |= 4;
    }
    if (r.direction.y < 0.0f) {
        float delta = r.origin.y - oct_mid.y;
        r.origin.y = oct_mid.y - delta;
        r.direction.y = -(r.direction.y);
        a |= 2;
    }
    if (r.direction.z < 0.0f) {
        float delta = r.origin.z - oct_mid.z;
        r.origin.z = oct_mid.z - delta;
        r.direction.z = -(r.direction.z);
        a |= 1;
    }
    if (a != 0)
        r.recalc_fracs();

    float tx0 = (root->region.min.x - r.origin.x) * r.fracs.x;
    float tx1 = (root->region.max.x - r.origin.x) * r.fracs.x;
    float ty0 = (root->region.min.y - r.origin.y) * r.fracs.y;
    float ty1 = (root->region.max.y - r.origin.y) * r.fracs.y;
    float tz0 = (root->region.min.z - r.origin.z) * r.fracs.z;
    float tz1 = (root->region.max.z - r.origin.z) * r.fracs.z;
    float tmin = fmaxf(fmaxf(tx0, ty0), tz0);
    float tmax = fminf(fminf(tx1, ty1), tz1);

    if ((tmin < tmax) && (tmax > 0.0f))
        return proc_subtree(root, ent, eid, a, V(tx0, ty0, tz0), V(tx1, ty1, tz1), ray, bestHit, ctx);
    return false;
}

/* src/device_geometry_functions.cuh:257-275 */
bool sphere_intersection(const Entity &e, int eid, const Ray &ray, Intersection &bestHit) {
    vec3 d = ray.origin - e.center;
    float p1 = -dot(ray.direction, d);
    float p2sqr = ((p1 * p1) - dot(d, d)) + (e.radius * e.radius);
    if (p2sqr < 0)
        return false;
    float p2 = 1.0f / rsqrt_ieee(p2sqr);
    float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
    if (t > 0 && t < bestHit.distance) {
        bestHit.distance = t;
        bestHit.position = ray.origin + t * ray.direction;
        bestHit.normal = normalized(bestHit.position - e.center);
        bestHit.entity = eid;
        bestHit.triangle = -1;   /* deviation: the reference leaves a stale triangle pointer */
        bestHit.u = 0.0f;
        bestHit.v = 0.0f;
        return true;
    }
    return false;
}

/* src/device_geometry_functions.cuh:282-299 */
bool mesh_intersection(const Entity &e, int eid, const Ray &ray, Intersection &bestHit, TraceCtx &ctx) {
    ctx.c.aabb_tests++;
    if (!intersects_aabb(e.aabb, ray, bestHit))
        return false;
    if (e.octree == nullptr) {
        bool hit = false;
        for (size_t i = 0; i < e.triangles.size(); i++)
            hit = entity_intersects_triangle(e, eid, (int)i, bestHit, ray, ctx) || hit;
        return hit;
    }
    return ray_step(e, eid, ray, bestHit, ctx);
}

/* src/device_geometry_functions.cuh:201-234; bu/bv return the raw barycentrics */
bool trace(const orc_scene &sc, const Ray &ray, Intersection &is, float *bu, float *bv, TraceCtx &ctx) {
    ctx.c.segments++;
    bool is_hit = false;
    for (size_t i = 0; i < sc.entities.size(); i++) {
        const Entity &e = *sc.entities[i];
        bool h = (e.shape == SPHERE) ? sphere_intersection(e, (int)i, ray, is)
                                     : mesh_intersection(e, (int)i, ray, is, ctx);
        is_hit = h || is_hit;
    }
    if (bu) { *bu = is.u; *bv = is.v; }
    if (is_hit && is.triangle >= 0) {
        const Entity &e = *sc.entities[is.entity];
        const Triangle &tr = e.triangles[is.triangle];
        float u = is.u, v = is.v;
        if (!e.uvs.empty()) {   /* deviation: the reference reads d_uvs unconditionally */
            vec2 v0_uv = e.uvs[tr.vt_a], v1_uv = e.uvs[tr.vt_b], v2_uv = e.uvs[tr.vt_c];
            vec2 intp = bary_lerp(v0_uv, v1_uv, v2_uv, is.u, is.v);
            is.u = intp.x;
            is.v = intp.y;
        }
        if (sc.materials[e.material].smooth_shading) {
            vec3 n0 = e.vertices[tr.idx_a].normal, n1 = e.vertices[tr.idx_b].normal,
                 n2 = e.vertices[tr.idx_c].normal;
            is.normal = normalized(bary_lerp(n0, n1, n2, u, v));
        }
    }
    return is_hit;
}

/* ------------------------------------------------------------------ materials / textures */
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int float_to_index(float f, int n) {   /* C truncation, NaN/neg -> 0, clamped */
    if (!(f >= 0.0f)) return 0;
    if (f >= (float)n) return n - 1;
    return clampi((int)f, 0, n - 1);
}

vec3 color_tex_sample(const Texture &t, float u, float v, TraceCtx &ctx) {   /* texture.cuh:22-31 */
    ctx.c.tex_fetches++;
    int x = float_to_index(u * t.w, t.w);
    int y = float_to_index((1.0f - v) * t.h, t.h);
    size_t idx = 3 * ((size_t)y * t.w + x);
    return V(float(t.data[idx]) / 255.0f, float(t.data[idx + 1]) / 255.0f, float(t.data[idx + 2]) / 255.0f);
}
float gray_tex_sample(const Texture &t, float u, float v, TraceCtx &ctx) {  /* texture.cuh:38-43 */
    ctx.c.tex_fetches++;
    int x = float_to_index(u * t.w, t.w);
    int y = float_to_index((1.0f - v) * t.h, t.h);
    return float(t.data[(size_t)y * t.w + x]) / 255.0f;
}
/* material.cuh:21-44 */
vec3 sample_albedo(const Material &m, float u, float v, TraceCtx &c) { return m.albedo_map.set ? color_tex_sample(m.albedo_map, u, v, c) : m.albedo; }
vec3 sample_specular(const Material &m, float u, float v, TraceCtx &c) { return m.specular_map.set ? color_tex_sample(m.specular_map, u, v, c) : m.specular; }
float sample_smoothness(const Material &m, float u, float v, TraceCtx &c) { return m.smoothness_map.set ? gray_tex_sample(m.smoothness_map, u, v, c) : m.smoothness; }
vec3 sample_normal(const Material &m, float u, float v, TraceCtx &c) {
    if (!m.normal_map.set) return V(0.0f, 0.0f, 1.0f);
    vec3 ret = 2.0f * color_tex_sample(m.normal_map, u, v, c) - V(1.0f);
    return normalized(ret);
}

/* src/scene.cuh:36-82 */
vec3 sample_hdri(const orc_scene &s, vec3 v, TraceCtx &ctx) {
    if (!s.has_hdri) return V(0, 0, 0);
    ctx.c.env_lookups++;
    const float *cols = s.hdri.data();
    int W = s.hdri_w, H = s.hdri_h;
    vec3 e_y = V(0, 1, 0), e_z = V(0, 0, 1);
    vec3 v_p = normalized(V(v.x, 0, v.z));
    float x_c = ((acosf(dot(v_p, e_z)) * (std::signbit(v_p.x) ? -1 : 1)) / (2.0f * 3.14159265f)) + 0.5f;
    float y_c = fminf(1.0f - ((dot(v, e_y) + 1.0f) / 2.0f), 0.99f);
    x_c += s.hdri_rot_offset;
    x_c = fmodf(x_c, 1.0f);

    if (!s.use_smooth_hdri) {
        int x_i = float_to_index(x_c * W, W);
        int y_i = float_to_index(y_c * H, H);
        int p = (y_i * W + x_i) * 3;
        return s.hdri_exposure * V(powf(cols[p], s.hdri_contrast), powf(cols[p + 1], s.hdri_contrast),
                                   powf(cols[p + 2], s.hdri_contrast));
    }
    int x_l = float_to_index(x_c * W, W);
    int x_r = (x_l + 1) % W;
    int y_l = float_to_index(y_c * H, H);
    int y_r = (y_l + 1) % H;
    float x_weight = fminf(fmaxf(x_c * W - x_l, 0.0f), 1.0f);
    float y_weight = fminf(fmaxf(y_c * H - y_l, 0.0f), 1.0f);
    int ul = (y_l * W + x_l) * 3, ur = (y_l * W + x_r) * 3, ll = (y_r * W + x_l) * 3, lr = (y_r * W + x_r) * 3;
    vec3 c_ul = V(cols[ul], cols[ul + 1], cols[ul + 2]);
    vec3 c_ur = V(cols[ur], cols[ur + 1], cols[ur + 2]);
    vec3 c_ll = V(cols[ll], cols[ll + 1], cols[ll + 2]);
    vec3 c_lr = V(cols[lr], cols[lr + 1], cols[lr + 2]);
    vec3 res = bilerp(c_ul, c_ur, c_ll, c_lr, x_weight, y_weight);
    res.x = powf(res.x, s.hdri_contrast);
    res.y = powf(res.y, s.hdri_contrast);
    res.z = powf(res.z, s.hdri_contrast);
    return s.hdri_exposure * res;
}

/* ------------------------------------------------------------------ RNG */
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* (0,1], the convention of curand_uniform that src/renderer.cu:63,121,158 relies on */
inline float u01(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

struct Rng {   /* one Philox block of 4 uniforms per (pixel, sample, block) */
    uint32_t pixel, sample, seed;
    void block(uint32_t b, float xi[4]) const {
        uint32_t ctr[4] = {pixel, sample, b, 0u}, key[2] = {seed, 0u}, o[4];
        philox4x32_10(ctr, key, o);
        for (int i = 0; i < 4; i++) xi[i] = u01(o[i]);
    }
};

/* ------------------------------------------------------------------ shading (src/renderer.cu) */
mat3 get_tangent_space(vec3 normal) {                  /* renderer.cu:168-175 */
    vec3 helper = V(1, 0, 0);
    if (fabsf(normal.x) > 0.99f)
        helper = V(0, 0, 1);
    vec3 tangent = normalized(cross(normal, helper));
    vec3 binormal = normalized(cross(normal, tangent));
    mat3 m = {tangent, binormal, normal};
    return m;
}
vec3 reflect(vec3 dir, vec3 normal) { return dir - 2.0f * dot(dir, normal) * normal; }  /* :147-149 */
vec3 sample_hemisphere(vec3 dir, float alpha, float xi_c, float xi_p) {                 /* :157-163 */
    float cos_theta = powf(xi_c, 1.0f / (alpha + 1.0f));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = 2 * PI_R * xi_p;
    vec3 ts = V(cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta);
    return mul(ts, get_tangent_space(dir));
}

/* src/renderer.cu:60-87; xi = Philox block 0 */
Ray create_camera_ray(const orc_camera &cam, int u, int v, const float xi[4]) {
    vec3 pos = V(cam.position[0], cam.position[1], cam.position[2]);
    vec3 dir = V(cam.direction[0], cam.direction[1], cam.direction[2]);
    vec3 up = V(cam.up[0], cam.up[1], cam.up[2]);
    vec3 ray_orig = pos;
    float n_u = (float(u + xi[0]) / float(cam.res_x)) - 0.5f;
    float n_v = (float(v + xi[1]) / float(cam.res_y)) - 0.5f;
    float aspect_ratio = float(cam.res_x) / float(cam.res_y);
    vec3 camera_right = -cross(dir, up);
    vec3 point = n_u * camera_right * aspect_ratio - n_v * up + pos + dir * cam.focal_length;
    vec3 ray_dir = normalized(point - pos);

    if (cam.aperture > 0.01f) {
        float r = sqrtf(xi[2]) * (cam.aperture / 2);
        float alpha = xi[3] * 2 * PI_R;
        float dx = cosf(alpha) * r;
        float dy = sinf(alpha) * r;
        vec3 orig_offset = dx * camera_right - dy * up;
        vec3 focal_point = ray_orig + cam.focus_distance * (1.0f / dot(dir, ray_dir)) * ray_dir;
        ray_orig = ray_orig + orig_offset;
        ray_dir = normalized(focal_point - ray_orig);
    }
    return make_ray(ray_orig, ray_dir);
}

/* src/renderer.cu:96-142; Philox block 1+i feeds bounce i: (roulette, cos-theta, phi, unused) */
vec3 color(Ray ray, const orc_scene &scene, const Rng &rng, TraceCtx &ctx) {
    ctx.c.paths++;
    vec3 attenuation = V(1.0f);
    vec3 result = V(0.0f);
    for (int i = 0; i < 6; i++) {
        Intersection hit;
        if (!trace(scene, ray, hit, nullptr, nullptr, ctx)) {
            result = result + attenuation * sample_hdri(scene, ray.direction, ctx);
            break;
        }
        const Material &m = scene.materials[scene.entities[hit.entity]->material];
        vec3 ts_normal = sample_normal(m, hit.u, hit.v, ctx);
        vec3 albedo = sample_albedo(m, hit.u, hit.v, ctx);
        float smoothness = sample_smoothness(m, hit.u, hit.v, ctx);
        vec3 specular = sample_specular(m, hit.u, hit.v, ctx);
        float emission = m.emission;
        vec3 normal = mul(ts_normal, get_tangent_space(hit.normal));

        albedo = vmin(1.0f - specular, albedo);
        float spec_chance = luminosity(specular);
        float diff_chance = luminosity(albedo);
        float sum = spec_chance + diff_chance;
        spec_chance /= sum;
        diff_chance /= sum;
        float xi[4];
        rng.block(1u + (uint32_t)i, xi);
        float roulette = xi[0];

        if (roulette < spec_chance) {
            float alpha = powf(1000.0f, smoothness * smoothness);
            ray.origin = hit.position + normal * 0.001f;
            ray.direction = sample_hemisphere(reflect(ray.direction, normal), alpha, xi[1], xi[2]);
            ray.recalc_fracs();
            float f = (alpha + 2) / (alpha + 1);
            attenuation = attenuation * (1.0f / spec_chance) * specular * f * dot(normal, ray.direction);
        } else {
            result = result + emission * albedo * attenuation;
            ray.origin = hit.position + normal * 0.001f;
            ray.direction = sample_hemisphere(normal, 1.0f, xi[1], xi[2]);
            ray.recalc_fracs();
            attenuation = attenuation * (1.0f / diff_chance) * albedo;
        }
    }
    return result;
}

void run_threads(int n_threads, int n_items, const std::function<void(int, int)> &fn) {
    if (n_threads <= 1 || n_items <= 1) { fn(0, n_items); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= n_items) break;
                fn(i, i + 1);
            }
        });
    for (auto &t : th) t.join();
}
} // namespace

/* ================================================================== C interface */
extern "C" {

orc_scene *orc_scene_new(void) { return new orc_scene(); }
void orc_scene_free(orc_scene *s) { delete s; }

int orc_add_material(orc_scene *s, const float albedo[3], const float specular[3], float emission,
                     float smoothness, int smooth_shading) {
    Material m;
    m.albedo = V(albedo[0], albedo[1], albedo[2]);
    m.specular = V(specular[0], specular[1], specular[2]);
    m.emission = emission;
    m.smoothness = smoothness;
    m.smooth_shading = smooth_shading != 0;
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}

int orc_material_set_map(orc_scene *s, int material, int which, const uint8_t *data, int w, int h) {
    if (material < 0 || material >= (int)s->materials.size() || which < 0 || which > 3) return -1;
    Material &m = s->materials[material];
    Texture *t = which == 0 ? &m.albedo_map : which == 1 ? &m.specular_map : which == 2 ? &m.normal_map : &m.smoothness_map;
    size_t n = (size_t)w * h * (which == 3 ? 1 : 3);
    t->data.assign(data, data + n);
    t->w = w; t->h = h; t->set = true;
    return 0;
}

/* vertex normals as src/entity.cu:110-133 computes them when the OBJ has no vn */
void orc_compute_vertex_normals(const float *positions, int n_vertices, const int32_t *tri,
                                int n_triangles, float *normals_out) {
    std::vector<vec3> n(n_vertices, V(0.0f));
    for (int f = 0; f < n_triangles; f++) {
        int a = tri[6 * f + 0], b = tri[6 * f + 1], c = tri[6 * f + 2];
        vec3 v0 = V(positions[3 * a], positions[3 * a + 1], positions[3 * a + 2]);
        vec3 v1 = V(positions[3 * b], positions[3 * b + 1], positions[3 * b + 2]);
        vec3 v2 = V(positions[3 * c], positions[3 * c + 1], positions[3 * c + 2]);
        vec3 tn = normalized(cross(v1 - v0, v2 - v0));
        n[a] = n[a] + tn;
        n[b] = n[b] + tn;
        n[c] = n[c] + tn;
    }
    for (int i = 0; i < n_vertices; i++) {
        vec3 r = normalized(n[i]);
        normals_out[3 * i] = r.x; normals_out[3 * i + 1] = r.y; normals_out[3 * i + 2] = r.z;
    }
}

/* src/entity.cu:185-258 */
void orc_transform(float *positions, float *normals, int n_vertices, float center[3], int op,
                   const float arg[3]) {
    vec3 c = V(center[0], center[1], center[2]);
    if (op == 0) {
        float factor = arg[0];
        for (int i = 0; i < n_vertices; i++) {
            vec3 pos = V(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
            pos = ((pos - c) * factor) + c;
            positions[3 * i] = pos.x; positions[3 * i + 1] = pos.y; positions[3 * i + 2] = pos.z;
        }
    } else if (op == 1) {
        vec3 delta = V(arg[0], arg[1], arg[2]);
        c = c + delta;
        center[0] = c.x; center[1] = c.y; center[2] = c.z;
        for (int i = 0; i < n_vertices; i++) {
            vec3 pos = V(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]) + delta;
            positions[3 * i] = pos.x; positions[3 * i + 1] = pos.y; positions[3 * i + 2] = pos.z;
        }
    } else if (op == 2) {
        /* double-precision libm cos/sin narrowed to float, as the host code does (entity.cu:232-237) */
        float cos_a = (float)cos((double)arg[2]), sin_a = (float)sin((double)arg[2]);
        float cos_b = (float)cos((double)arg[1]), sin_b = (float)sin((double)arg[1]);
        float cos_g = (float)cos((double)arg[0]), sin_g = (float)sin((double)arg[0]);
        vec3 R0 = V(cos_a * cos_b, cos_a * sin_b * sin_g - sin_a * cos_g, cos_a * sin_b * cos_g + sin_a * sin_g);
        vec3 R1 = V(sin_a * cos_b, sin_a * sin_b * sin_g + cos_a * cos_g, sin_a * sin_b * cos_g - cos_a * sin_g);
        vec3 R2 = V(-sin_b, cos_b * sin_g, cos_b * cos_g);
        for (int i = 0; i < n_vertices; i++) {
            vec3 v = V(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]) - c;
            v = V(dot(v, R0), dot(v, R1), dot(v, R2));
            v = v + c;
            positions[3 * i] = v.x; positions[3 * i + 1] = v.y; positions[3 * i + 2] = v.z;
            if (normals) {
                vec3 n = V(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
                n = V(dot(n, R0), dot(n, R1), dot(n, R2));
                normals[3 * i] = n.x; normals[3 * i + 1] = n.y; normals[3 * i + 2] = n.z;
            }
        }
    }
}

void orc_mesh_aabb(const float *positions, int n_vertices, float min_out[3], float max_out[3]) {
    std::vector<Vertex> v(n_vertices);
    for (int i = 0; i < n_vertices; i++) v[i].position = V(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    AABB b = recalc_aabb(v.data(), n_vertices);
    min_out[0] = b.min.x; min_out[1] = b.min.y; min_out[2] = b.min.z;
    max_out[0] = b.max.x; max_out[1] = b.max.y; max_out[2] = b.max.z;
}

int orc_add_mesh(orc_scene *s, const float *positions, const float *normals, int n_vertices,
                 const int32_t *tri, int n_triangles, const float *uvs, int n_uvs, int material,
                 int octree_depth) {
    Entity *e = new Entity();
    e->shape = TRIANGLE_MESH;
    e->material = material;
    e->vertices.resize(n_vertices);
    std::vector<float> nrm;
    if (!normals) {
        nrm.resize((size_t)3 * n_vertices);
        orc_compute_vertex_normals(positions, n_vertices, tri, n_triangles, nrm.data());
        normals = nrm.data();
    }
    for (int i = 0; i < n_vertices; i++) {
        e->vertices[i].position = V(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
        e->vertices[i].normal = V(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    }
    e->triangles.resize(n_triangles);
    for (int i = 0; i < n_triangles; i++) {
        Triangle t = {tri[6 * i], tri[6 * i + 1], tri[6 * i + 2], tri[6 * i + 3], tri[6 * i + 4], tri[6 * i + 5]};
        e->triangles[i] = t;
    }
    for (int i = 0; i < n_uvs; i++) e->uvs.push_back(V2(uvs[2 * i], uvs[2 * i + 1]));
    e->aabb = recalc_aabb(e->vertices.data(), n_vertices);   /* entity.cu:136 / aabb.cu:20-21 */
    if (octree_depth > 0) construct_octree(e, octree_depth);
    s->entities.push_back(e);
    return (int)s->entities.size() - 1;
}

int orc_add_sphere(orc_scene *s, const float center[3], float radius, int material) {
    Entity *e = new Entity();
    e->shape = SPHERE;
    e->center = V(center[0], center[1], center[2]);
    e->radius = radius;
    e->material = material;
    s->entities.push_back(e);
    return (int)s->entities.size() - 1;
}

void orc_set_hdri(orc_scene *s, const float *rgb, int w, int h, float exposure, float contrast,
                  float rot_offset, int smooth) {
    s->hdri.assign(rgb, rgb + (size_t)w * h * 3);
    s->hdri_w = w; s->hdri_h = h;
    s->hdri_exposure = exposure; s->hdri_contrast = contrast;
    s->hdri_rot_offset = fmodf(rot_offset, 1.0f);   /* scene.cu:52-55 */
    s->use_smooth_hdri = smooth != 0;
    s->has_hdri = true;
}

static void octree_walk_stats(const Octree *n, int64_t *per_depth, int64_t *refs, int64_t *max_leaf) {
    per_depth[n->depth]++;
    *refs += (int64_t)n->triangle_indices.size();
    *max_leaf = std::max<int64_t>(*max_leaf, (int64_t)n->triangle_indices.size());
    for (int i = 0; i < 8; i++) if (n->children[i]) octree_walk_stats(n->children[i], per_depth, refs, max_leaf);
}

int orc_octree_stats(orc_scene *s, int entity, int64_t *nodes_per_depth, int64_t *n_leaf_refs, int64_t *max_leaf) {
    if (entity < 0 || entity >= (int)s->entities.size() || !s->entities[entity]->octree) return -1;
    Entity *e = s->entities[entity];
    for (int i = 0; i <= e->max_depth; i++) nodes_per_depth[i] = 0;
    *n_leaf_refs = 0; *max_leaf = 0;
    octree_walk_stats(e->octree, nodes_per_depth, n_leaf_refs, max_leaf);
    return 0;
}

static void octree_walk_dump(const Octree *n, int slot, int32_t *out, int64_t cap, int64_t *pos) {
    int mask = 0;
    for (int i = 0; i < 8; i++) if (n->children[i]) mask |= 1 << i;
    int32_t hdr[4] = {n->depth, slot, mask, (int32_t)n->triangle_indices.size()};
    for (int k = 0; k < 4; k++) { if (out && *pos < cap) out[*pos] = hdr[k]; (*pos)++; }
    for (int t : n->triangle_indices) { if (out && *pos < cap) out[*pos] = t; (*pos)++; }
    for (int i = 0; i < 8; i++) if (n->children[i]) octree_walk_dump(n->children[i], i, out, cap, pos);
}

int64_t orc_octree_dump(orc_scene *s, int entity, int32_t *out, int64_t cap) {
    if (entity < 0 || entity >= (int)s->entities.size() || !s->entities[entity]->octree) return -1;
    int64_t pos = 0;
    octree_walk_dump(s->entities[entity]->octree, 0, out, cap, &pos);
    return pos;
}

int orc_aabb_intersects_triangle(const float bmin[3], const float bmax[3], const float v0[3],
                                 const float v1[3], const float v2[3], int pad_box) {
    AABB b;
    if (pad_box) b = padded_aabb(V(bmin[0], bmin[1], bmin[2]), V(bmax[0], bmax[1], bmax[2]));
    else { b.min = V(bmin[0], bmin[1], bmin[2]); b.max = V(bmax[0], bmax[1], bmax[2]); }
    return aabb_intersects_triangle(b, V(v0[0], v0[1], v0[2]), V(v1[0], v1[1], v1[2]), V(v2[0], v2[1], v2[2])) ? 1 : 0;
}

static void counters_out(const Counters &c, orc_counters *o) {
    if (!o) return;
    o->segments = c.segments; o->aabb_tests = c.aabb_tests; o->node_visits = c.node_visits;
    o->tri_tests = c.tri_tests; o->env_lookups = c.env_lookups; o->tex_fetches = c.tex_fetches;
    o->paths = c.paths;
}

void orc_trace(orc_scene *s, const float *rays, int64_t n, orc_hit *hits, int32_t *log, int log_stride,
               orc_counters *counters) {
    TraceCtx ctx;
    for (int64_t i = 0; i < n; i++) {
        Ray r = make_ray(V(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]),
                         V(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
        if (log) { ctx.log = log + i * log_stride; ctx.log_cap = log_stride; ctx.log[0] = 0; }
        Intersection is;
        float bu = 0, bv = 0;
        bool h = trace(*s, r, is, &bu, &bv, ctx);
        orc_hit &o = hits[i];
        o.entity = h ? is.entity : -1;
        o.triangle = h ? is.triangle : -1;
        o.t = is.distance;
        o.bu = h ? bu : 0.0f; o.bv = h ? bv : 0.0f;
        o.u = h ? is.u : 0.0f; o.v = h ? is.v : 0.0f;
        o.position[0] = h ? is.position.x : 0.0f; o.position[1] = h ? is.position.y : 0.0f; o.position[2] = h ? is.position.z : 0.0f;
        o.normal[0] = h ? is.normal.x : 0.0f; o.normal[1] = h ? is.normal.y : 0.0f; o.normal[2] = h ? is.normal.z : 0.0f;
    }
    counters_out(ctx.c, counters);
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
float orc_u01(uint32_t x) { return u01(x); }

void orc_camera_rays(const orc_camera *cam, uint32_t seed, int sample, float *rays_out) {
    for (int v = 0; v < cam->res_y; v++)
        for (int u = 0; u < cam->res_x; u++) {
            Rng rng = {(uint32_t)(v * cam->res_x + u), (uint32_t)sample, seed};
            float xi[4];
            rng.block(0, xi);
            Ray r = create_camera_ray(*cam, u, v, xi);
            float *o = rays_out + 6 * ((size_t)v * cam->res_x + u);
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
            o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
        }
}

/* src/renderer.cu:32-51 without the split-buffer race: samples are summed in index order */
void orc_render(orc_scene *s, const orc_camera *cam, uint32_t seed, int sample_begin, int sample_end,
                int row_begin, int row_end, float *accum, int n_threads, orc_counters *counters) {
    int n_rows = row_end - row_begin;
    std::vector<Counters> per_row(std::max(n_rows, 0));
    run_threads(n_threads, n_rows, [&](int r0, int r1) {
        for (int r = r0; r < r1; r++) {
            int v = row_begin + r;
            TraceCtx ctx;
            for (int u = 0; u < cam->res_x; u++) {
                vec3 result = V(0.0f);
                uint32_t pixel = (uint32_t)(v * cam->res_x + u);
                for (int smp = sample_begin; smp < sample_end; smp++) {
                    Rng rng = {pixel, (uint32_t)smp, seed};
                    float xi[4];
                    rng.block(0, xi);
                    Ray ray = create_camera_ray(*cam, u, v, xi);
                    result = result + color(ray, *s, rng, ctx);
                }
                float *o = accum + 3 * (size_t)pixel;
                o[0] += result.x; o[1] += result.y; o[2] += result.z;
            }
            per_row[r] = ctx.c;
        }
    });
    Counters tot;
    for (auto &c : per_row) tot.add(c);
    counters_out(tot, counters);
}

void orc_sample_hdri(orc_scene *s, const float *dirs, int64_t n, float *rgb_out) {
    TraceCtx ctx;
    for (int64_t i = 0; i < n; i++) {
        vec3 c = sample_hdri(*s, V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), ctx);
        rgb_out[3 * i] = c.x; rgb_out[3 * i + 1] = c.y; rgb_out[3 * i + 2] = c.z;
    }
}

void orc_sample_material(orc_scene *s, int material, const float *uv, int64_t n, float *albedo,
                         float *specular, float *smoothness, float *normal) {
    TraceCtx ctx;
    const Material &m = s->materials[material];
    for (int64_t i = 0; i < n; i++) {
        float u = uv[2 * i], v = uv[2 * i + 1];
        vec3 a = sample_albedo(m, u, v, ctx), sp = sample_specular(m, u, v, ctx), nn = sample_normal(m, u, v, ctx);
        albedo[3 * i] = a.x; albedo[3 * i + 1] = a.y; albedo[3 * i + 2] = a.z;
        specular[3 * i] = sp.x; specular[3 * i + 1] = sp.y; specular[3 * i + 2] = sp.z;
        normal[3 * i] = nn.x; normal[3 * i + 1] = nn.y; normal[3 * i + 2] = nn.z;
        smoothness[i] = sample_smoothness(m, u, v, ctx);
    }
}

/* ------------------------------------------------------------------ post-processing */

/* src/post_processing.cu:40-46,51-68: the factor is evaluated in double (M_PI), the power in float */
void orc_make_gaussian(unsigned size, float sigma, float *taps) {
    int ssize = (int)size;
    for (int y = 0; y < ssize * 2 + 1; y++)
        for (int x = 0; x < ssize * 2 + 1; x++) {
            float sx = float(x - ssize), sy = float(y - ssize);
            float fac = 1.0f / (sigma * sigma * 2 * M_PI);
            float ex = -((sx * sx + sy * sy) / (2 * sigma * sigma));
            taps[(size_t)y * (size * 2 + 1) + x] = fac * powf(M_E, ex);
        }
}
void orc_make_mean(unsigned size, float *taps) {     /* post_processing.cu:73-84 */
    size_t n = (size_t)(2 * size + 1) * (2 * size + 1);
    for (size_t i = 0; i < n; i++) taps[i] = 1.0f;
}

static inline float sample_gaussian_1d(float x, float sigma) {   /* post_processing.cu:31-35 */
    float fac = 1.0f / (sigma * sigma * 2 * 3.1415926535f);
    float ex = -((x * x) / (2 * sigma * sigma));
    return fac * powf(2.7182818284f, ex);
}

/* src/post_processing.cu:143-187 */
void orc_apply_filter(const float *in, float *out, int w, int h, const float *taps, int ksize,
                      int bilateral, int n_threads) {
    const vec3 *src = reinterpret_cast<const vec3 *>(in);
    vec3 *dst = reinterpret_cast<vec3 *>(out);
    int ks = ksize;
    run_threads(n_threads, h, [&](int r0, int r1) {
        for (int v = r0; v < r1; v++)
            for (int u = 0; u < w; u++) {
                size_t pixel_idx = (size_t)v * w + u;
                float kernel_weight = 0.0f;
                vec3 pixel = V(0.0f);
                for (int y = -ks; y <= ks; y++)
                    for (int x = -ks; x <= ks; x++) {
                        int ny = v + y, nx = u + x;
                        if (ny < 0 || ny >= h || nx < 0 || nx >= w) continue;
                        size_t nidx = (size_t)ny * w + nx;
                        float gs = taps[(size_t)(ks + y) * (ks * 2 + 1) + (ks + x)];
                        float gi = 1.0f;
                        if (bilateral) gi = sample_gaussian_1d(255 * luminosity(src[nidx] - src[pixel_idx]), 24.0f);
                        float wgt = gi * gs;
                        pixel = pixel + wgt * src[nidx];
                        kernel_weight += wgt;
                    }
                dst[pixel_idx] = pixel / kernel_weight;
            }
    });
}

void orc_compound_buffers(const float *in, float *out, int64_t n_pixels, int n_split) {   /* :243-256 */
    const vec3 *src = reinterpret_cast<const vec3 *>(in);
    vec3 *dst = reinterpret_cast<vec3 *>(out);
    for (int64_t i = 0; i < n_pixels; i++) {
        vec3 sum = V(0.0f);
        for (int j = 0; j < n_split; j++) sum = sum + src[i + (int64_t)j * n_pixels];
        dst[i] = sum;
    }
}
void orc_normalize_image(float *buf, int64_t n_pixels, unsigned n_samples) {              /* :211-218 */
    vec3 *b = reinterpret_cast<vec3 *>(buf);
    for (int64_t i = 0; i < n_pixels; i++) b[i] = b[i] / (float)n_samples;
}
void orc_apply_threshold(const float *in, float *out, int64_t n_pixels, float threshold) { /* :261-268 */
    const vec3 *src = reinterpret_cast<const vec3 *>(in);
    vec3 *dst = reinterpret_cast<vec3 *>(out);
    for (int64_t i = 0; i < n_pixels; i++) dst[i] = (luminosity(src[i]) < threshold) ? V(0.0f) : src[i];
}
void orc_image_add(float *buf, const float *layer, int64_t n_pixels, float opacity) {     /* :273-278 */
    vec3 *b = reinterpret_cast<vec3 *>(buf);
    const vec3 *l = reinterpret_cast<const vec3 *>(layer);
    for (int64_t i = 0; i < n_pixels; i++) b[i] = b[i] + opacity * l[i];
}
void orc_apply_aces(float *buf, int64_t n_pixels) {                                       /* :194-205,223-227 */
    vec3 *b = reinterpret_cast<vec3 *>(buf);
    const float a = 2.51f, bb = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    for (int64_t i = 0; i < n_pixels; i++) {
        vec3 x = b[i];
        vec3 res = (x * (a * x + V(bb))) / (x * (c * x + V(d)) + V(e));
        res.x = std::max(0.0f, std::min(1.0f, res.x));
        res.y = std::max(0.0f, std::min(1.0f, res.y));
        res.z = std::max(0.0f, std::min(1.0f, res.z));
        b[i] = res;
    }
}
void orc_gamma_correct(float *buf, int64_t n_pixels, float gamma) {                       /* :232-238 */
    for (int64_t i = 0; i < 3 * n_pixels; i++) buf[i] = powf(buf[i], 1.0f / gamma);
}
void orc_to_ppm_bytes(const float *buf, int64_t n_pixels, uint8_t *out) {                 /* io.cu:28-34 */
    for (int64_t i = 0; i < 3 * n_pixels; i++) {
        float f = (buf[i] < 0.999f) ? buf[i] : 0.999f;
        out[i] = (uint8_t)(f * 255);
    }
}

} // extern "C"
