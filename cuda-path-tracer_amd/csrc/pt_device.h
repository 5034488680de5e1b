/*
 * pt_device.h -- device functions of the hot path (gfx950, wave64).
 *
 * Geometry (everything that decides WHICH triangle is hit and at what t,u,v) is written in strict
 * IEEE fp32 in source order -- the translation unit is built with -ffp-contract=off and HIP's
 * default correctly rounded divide / sqrt -- so hit ids, distances, barycentrics and the order of
 * triangle tests are bit-identical to a scalar CPU evaluation of the same formulas.  Shading may
 * contract and uses the hardware transcendental units, like the reference's -use_fast_math build.
 *
 * Reference functions restated here (src/device_geometry_functions.cuh = dgf):
 *   intersects_aabb dgf:388-417, intersect_triangle dgf:319-365, sphere dgf:257-275,
 *   Octree::ray_step dgf:39-89, proc_subtree/find_first_node/next_node dgf:95-191 (made
 *   iterative), trace dgf:201-234, Ray::recalc_fracs src/geometry.cuh:23-28.
 */
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "pt_types.h"

namespace ptamd {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 v) { return mk(s * v.x, s * v.y, s * v.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
/* correctly rounded 1/sqrt in two IEEE operations: the definition both sides of the parity
 * check use for the reference's rsqrt / __frsqrt_rn (src/vector.cuh:69-76) */
__device__ __forceinline__ float rsqrt_ieee(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ V3 normalized(V3 v) {
    float r = rsqrt_ieee(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk(v.x * r, v.y * r, v.z * r);
}
/* src/geometry.cuh:23-28 */
__device__ __forceinline__ float clamp_frac(float d) { return fmaxf(fminf(1.0f / d, 1000000.0f), -1000000.0f); }

struct Stats {
    uint32_t segments, aabb_tests, node_visits, tri_tests, env_lookups, tex_fetches, paths;
};
__device__ __forceinline__ void stats_zero(Stats &s) {
    s.segments = s.aabb_tests = s.node_visits = s.tri_tests = s.env_lookups = s.tex_fetches = s.paths = 0;
}

struct Best {
    float t, bu, bv;
    int32_t entity;    /* -1 = none */
    int32_t tri;       /* -1 = sphere */
    uint32_t rec;      /* index of the accepted triangle record */
};

struct TriLog {        /* optional per-ray record of the triangle-test sequence */
    int32_t *p;
    int32_t cap;
};

/* dgf:319-365 on a pre-gathered record; e1,e2 were formed on the host by the same fp32
 * subtractions the reference performs per test. */
template <bool LOG>
__device__ __forceinline__ bool intersect_record(const PtTriRec *__restrict__ recs, uint32_t idx, V3 o, V3 d,
                                                 Best &best, int32_t entity, Stats &st, TriLog &lg) {
    const float4 *q = reinterpret_cast<const float4 *>(recs + idx);
    const float4 r0 = q[0], r1 = q[1], r2 = q[2];
    st.tri_tests++;
    const int32_t id = __float_as_int(r2.y);
    if (LOG) {
        int n = lg.p[0]++;
        if (n + 1 < lg.cap) lg.p[n + 1] = id;
    }
    const V3 v0 = mk(r0.x, r0.y, r0.z), e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
    const V3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    if (fabsf(det) < 0.000000001f) return false;
    const float inv_det = 1.0f / det;
    const V3 tvec = o - v0;
    const float u = dot3(tvec, pvec) * inv_det;
    if (u < -0.0001f || u > 1.0001f) return false;
    const V3 qvec = cross3(tvec, e1);
    const float v = dot3(d, qvec) * inv_det;
    if (v < -0.0001f || u + v > 1.0001f) return false;
    const float t = dot3(e2, qvec) * inv_det;
    if (t > 0 && t < best.t) {
        best.t = t; best.bu = u; best.bv = v;
        best.entity = entity; best.tri = id; best.rec = idx;
        return true;
    }
    return false;
}

/* dgf:388-417 */
__device__ __forceinline__ bool slab_test(const float mn[3], const float mx[3], V3 o, V3 fr, float best_t) {
    float tx1 = (mn[0] - o.x) * fr.x, tx2 = (mx[0] - o.x) * fr.x;
    float ty1 = (mn[1] - o.y) * fr.y, ty2 = (mx[1] - o.y) * fr.y;
    float tz1 = (mn[2] - o.z) * fr.z, tz2 = (mx[2] - o.z) * fr.z;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    tmin = fmaxf(tmin, fminf(ty1, ty2));
    tmax = fminf(tmax, fmaxf(ty1, ty2));
    tmin = fmaxf(tmin, fminf(tz1, tz2));
    tmax = fminf(tmax, fmaxf(tz1, tz2));
    return !(tmax < 0.0f || tmin > tmax || best_t < tmin);
}

/* dgf:116-126 with the per-child argument table of dgf:178-187 folded in: the exit parameters of
 * child `curr` are that child's t1, and the successor along an axis is curr with that axis bit
 * set, or 8 (leave the node) when the bit is already set. */
__device__ __forceinline__ int next_child(int curr, float ex, float ey, float ez) {
    const int cx = (curr & 4) ? 8 : (curr | 4);
    const int cy = (curr & 2) ? 8 : (curr | 2);
    const int cz = (curr & 1) ? 8 : (curr | 1);
    if (ex < ey) {
        if (ex < ez) return cx;
    } else {
        if (ey < ez) return cy;
    }
    return cz;
}

/* dgf:95-110 */
__device__ __forceinline__ int first_child(float t0x, float t0y, float t0z, float tMx, float tMy, float tMz) {
    int answer = 0;
    if (t0x > t0y && t0x > t0z) {
        if (tMy < t0x) answer |= 2;
        if (tMz < t0x) answer |= 1;
        return answer;
    } else if (t0y > t0z) {
        if (tMx < t0y) answer |= 4;
        if (tMz < t0y) answer |= 1;
        return answer;
    }
    if (tMx < t0z) answer |= 4;
    if (tMy < t0z) answer |= 2;
    return answer;
}

/*
 * Octree::ray_step + proc_subtree (dgf:39-191), iterative.  The recursion's frames live in a
 * per-lane stack of one 16-byte word per level in LDS: when descending, each axis of the child
 * interval inherits one bound from the parent ([t0,tM] or [tM,t1]); the word stores the three
 * bounds the child does NOT inherit plus {parent node, child slot}.  On return the parent's
 * t0/tM/t1 are rebuilt from the child's interval and the saved bounds by copies only, so every
 * value is the one the recursion would hold.  Reproduced quirks: the rejection tests t1.y twice
 * and never t1.z (dgf:142); the walk ends at the first leaf that reports an accepted test
 * (dgf:188 unwinds every level once `hit` is set).
 */
template <bool LOG>
__device__ __forceinline__ bool octree_traverse(const PtNode *__restrict__ nodes, const PtTriRec *__restrict__ tris,
                                                const PtEntity &e, int32_t entity, V3 ro, V3 rd, Best &best,
                                                Stats &st, TriLog &lg, float4 *stack, int stride) {
    V3 o = ro, d = rd;
    const float midx = 0.5f * (e.region_max[0] + e.region_min[0]);
    const float midy = 0.5f * (e.region_max[1] + e.region_min[1]);
    const float midz = 0.5f * (e.region_max[2] + e.region_min[2]);
    int a = 0;
    if (d.x < 0.0f) { float delta = o.x - midx; o.x = midx - delta; d.x = -d.x; a |= 4; }
    if (d.y < 0.0f) { float delta = o.y - midy; o.y = midy - delta; d.y = -d.y; a |= 2; }
    if (d.z < 0.0f) { float delta = o.z - midz; o.z = midz - delta; d.z = -d.z; a |= 1; }
    /* fracs of the mirrored ray; for un-mirrored components this is the value the original ray holds */
    const float frx = clamp_frac(d.x), fry = clamp_frac(d.y), frz = clamp_frac(d.z);

    float t0x = (e.region_min[0] - o.x) * frx, t1x = (e.region_max[0] - o.x) * frx;
    float t0y = (e.region_min[1] - o.y) * fry, t1y = (e.region_max[1] - o.y) * fry;
    float t0z = (e.region_min[2] - o.z) * frz, t1z = (e.region_max[2] - o.z) * frz;
    const float tmin = fmaxf(fmaxf(t0x, t0y), t0z);
    const float tmax = fminf(fminf(t1x, t1y), t1z);
    if (!((tmin < tmax) && (tmax > 0.0f))) return false;

    const int maxd = e.octree_depth;
    uint32_t node = e.root_node;
    int level = 0;
    bool entering = true;
    PtNode rec; rec.a = 0; rec.b = 0;
    float tMx = 0.f, tMy = 0.f, tMz = 0.f;
    int curr = 0;

    for (;;) {
        bool need_pop = false;
        if (entering) {
            st.node_visits++;
            bool ok = !((t1x < 0.0f) || (t1y < 0.0f) || (t1y < 0.0f));
            if (ok) rec = nodes[node];
            if (ok && level == maxd) {
                bool hit = false;
                for (uint32_t i = 0; i < rec.b; i++)
                    hit = intersect_record<LOG>(tris, rec.a + i, ro, rd, best, entity, st, lg) || hit;
                if (hit) return true;
                ok = false;
            }
            if (ok) {
                tMx = 0.5f * (t0x + t1x); tMy = 0.5f * (t0y + t1y); tMz = 0.5f * (t0z + t1z);
                curr = first_child(t0x, t0y, t0z, tMx, tMy, tMz);
                entering = false;
            } else {
                need_pop = true;
            }
        } else if (curr >= 8) {
            need_pop = true;
        }
        if (need_pop) {
            if (level == 0) return false;
            level--;
            const float4 s = stack[level * stride];
            const uint32_t w = __float_as_uint(s.w);
            const int c = (int)(w & 7u);
            node = w >> 3;
            /* (t0,t1) still hold the child's interval: its t1 is the exit parameter set */
            const float ex = t1x, ey = t1y, ez = t1z;
            if (c & 4) { tMx = t0x; t0x = s.x; } else { tMx = t1x; t1x = s.x; }
            if (c & 2) { tMy = t0y; t0y = s.y; } else { tMy = t1y; t1y = s.y; }
            if (c & 1) { tMz = t0z; t0z = s.z; } else { tMz = t1z; t1z = s.z; }
            rec = nodes[node];
            curr = next_child(c, ex, ey, ez);
            entering = false;
            continue;
        }
        const int adj = curr ^ a;
        if ((rec.b >> adj) & 1u) {
            float4 s;
            s.x = (curr & 4) ? t0x : t1x;
            s.y = (curr & 2) ? t0y : t1y;
            s.z = (curr & 1) ? t0z : t1z;
            s.w = __uint_as_float((node << 3) | (uint32_t)curr);
            stack[level * stride] = s;
            if (curr & 4) t0x = tMx; else t1x = tMx;
            if (curr & 2) t0y = tMy; else t1y = tMy;
            if (curr & 1) t0z = tMz; else t1z = tMz;
            node = rec.a + (uint32_t)__popc(rec.b & ((1u << adj) - 1u));
            level++;
            entering = true;
        } else {
            const float ex = (curr & 4) ? t1x : tMx;
            const float ey = (curr & 2) ? t1y : tMy;
            const float ez = (curr & 1) ? t1z : tMz;
            curr = next_child(curr, ex, ey, ez);
        }
    }
}

/* dgf:257-275; a sphere hit clears the triangle (the reference leaves a stale pointer) */
__device__ __forceinline__ bool sphere_hit(const PtEntity &e, int32_t entity, V3 o, V3 d, Best &best) {
    const V3 dd = o - mk(e.center[0], e.center[1], e.center[2]);
    const float p1 = -dot3(d, dd);
    const float p2sqr = ((p1 * p1) - dot3(dd, dd)) + (e.radius * e.radius);
    if (p2sqr < 0) return false;
    const float p2 = 1.0f / rsqrt_ieee(p2sqr);
    const float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
    if (t > 0 && t < best.t) {
        best.t = t; best.bu = 0.0f; best.bv = 0.0f;
        best.entity = entity; best.tri = -1; best.rec = 0;
        return true;
    }
    return false;
}

/* closest-hit part of trace() (dgf:201-206, 241-299): loops ALL entities with a shared best hit */
template <bool LOG>
__device__ __forceinline__ bool closest_hit(const PtSceneDev &sc, V3 o, V3 d, Best &best, Stats &st, TriLog &lg,
                                            float4 *stack, int stride) {
    st.segments++;
    best.t = FLT_MAX; best.bu = 0.f; best.bv = 0.f; best.entity = -1; best.tri = -1; best.rec = 0;
    const V3 fr = mk(clamp_frac(d.x), clamp_frac(d.y), clamp_frac(d.z));
    bool is_hit = false;
    for (int i = 0; i < sc.n_entities; i++) {
        const PtEntity &e = sc.entities[i];
        bool h = false;
        if (e.shape == PT_SHAPE_SPHERE) {
            h = sphere_hit(e, i, o, d, best);
        } else {
            st.aabb_tests++;
            if (slab_test(e.aabb_min, e.aabb_max, o, fr, best.t)) {
                if (e.octree_depth == 0) {
                    for (uint32_t k = 0; k < e.n_tris; k++)
                        h = intersect_record<LOG>(sc.tris, e.tri_first + k, o, d, best, i, st, lg) || h;
                } else {
                    h = octree_traverse<LOG>(sc.nodes, sc.tris, e, i, o, d, best, st, lg, stack, stride);
                }
            }
        }
        is_hit = h || is_hit;
    }
    return is_hit;
}

struct Surface {     /* what shading needs from a hit */
    V3 position, normal;
    float u, v;
};

/* position / normal / uv of the accepted hit: the tail of intersect_triangle (dgf:353-361) and of
 * trace() (dgf:208-231), evaluated once for the final hit instead of on every improvement. */
__device__ __forceinline__ void finish_hit(const PtSceneDev &sc, V3 o, V3 d, const Best &best, Surface &sf) {
    const PtEntity &e = sc.entities[best.entity];
    sf.position = o + best.t * d;
    sf.u = best.bu; sf.v = best.bv;
    if (best.tri < 0) {
        sf.normal = normalized(sf.position - mk(e.center[0], e.center[1], e.center[2]));
        return;
    }
    const float4 *q = reinterpret_cast<const float4 *>(sc.tris + best.rec);
    const float4 r0 = q[0], r1 = q[1], r2 = q[2];
    const V3 e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
    sf.normal = normalized(cross3(e1, e2));
    if (e.has_uvs || e.smooth_shading) {
        const PtShadeRec &sh = sc.shade[e.shade_base + (uint32_t)best.tri];
        const float u = best.bu, v = best.bv;
        const float w = 1.0f - (u + v);
        if (e.has_uvs) {   /* meshes without uvs keep the barycentrics (the reference reads d_uvs[-1]) */
            sf.u = (u * sh.uv1[0] + v * sh.uv2[0]) + w * sh.uv0[0];
            sf.v = (u * sh.uv1[1] + v * sh.uv2[1]) + w * sh.uv0[1];
        }
        if (e.smooth_shading) {
            V3 n = mk((u * sh.n1[0] + v * sh.n2[0]) + w * sh.n0[0], (u * sh.n1[1] + v * sh.n2[1]) + w * sh.n0[1],
                      (u * sh.n1[2] + v * sh.n2[2]) + w * sh.n0[2]);
            sf.normal = normalized(n);
        }
    }
}

/* ------------------------------------------------------------------ RNG: Philox4x32-10 */
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* (0,1], the convention of curand_uniform that the reference's sampling relies on */
__device__ __forceinline__ float u01(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }
__device__ __forceinline__ void rng_block(uint32_t pixel, uint32_t sample, uint32_t block, uint32_t seed, float xi[4]) {
    uint32_t o[4];
    philox4x32_10(pixel, sample, block, 0u, seed, 0u, o);
    xi[0] = u01(o[0]); xi[1] = u01(o[1]); xi[2] = u01(o[2]); xi[3] = u01(o[3]);
}

} // namespace ptamd
#endif
