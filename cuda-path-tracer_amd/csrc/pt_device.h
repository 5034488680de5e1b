/*
 * pt_device.h -- device functions of the hot path (gfx950, wave64).
 *
 * Geometry (everything that decides WHICH triangle is hit and at what t,u,v) is written in strict
 * IEEE fp32 in source order -- the translation unit is built with -ffp-contract=off and HIP's
 * default correctly rounded divide / sqrt -- so hit ids, distances, barycentrics and the order of
 * triangle tests are bit-identical to a scalar CPU evaluation of the same formulas.  Shading may
 * use the hardware transcendental units, like the reference's -use_fast_math build.
 *
 * Reference functions restated here (src/device_geometry_functions.cuh = dgf):
 *   intersects_aabb dgf:388-417, intersect_triangle dgf:319-365, sphere dgf:257-275,
 *   Octree::ray_step dgf:39-89, proc_subtree/find_first_node/next_node dgf:95-191,
 *   trace dgf:201-234, Ray::recalc_fracs src/geometry.cuh:23-28.
 *
 * Execution model.  The reference runs trace() as one recursive call chain per thread; on a 64-wide
 * wave that leaves most lanes idle (one lane in a 30-triangle leaf stalls 63 others).  Here every
 * lane carries its ray through an explicit state machine
 *     ENTITY -> (NODE <-> LEAF)* -> ENTITY ... -> DONE
 * and the wave repeatedly runs the block (entity set-up, octree node step, triangle test) that the
 * most lanes are waiting for.  One trip of a block is the same short instruction sequence for all
 * participating lanes, so divergence is bounded by the mix of states, not by the shape of each
 * lane's recursion.  The recursion's frames live in a per-lane stack of one 16-byte and one 4-byte
 * word per level in LDS plus three bits per level in a register.
 *
 * The kernel is bound by vector-instruction issue (a SIMD issues one wave64 VALU instruction per four
 * cycles), so the blocks are written for instruction count: the children a node's loop would step
 * through are listed once per node, branch-free; triangle tests run two at a time in packed fp32 --
 * and for lane occupancy: triangle runs, the one part of a ray's work that is parallel in itself, are
 * handed out to all lanes of the wave (leaf_coop), idle ones included.
 */
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "pt_child_table.h"
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
    /* scheduling diagnostics (per lane: trips of each block this lane took part in; lane 0 also
     * counts the trips its wave ran, so lanes / (64 * trips) is the block's lane occupancy) */
    uint32_t ent_lanes, node_lanes, leaf_lanes, front_lanes, ent_trips, node_trips, leaf_trips, front_trips;
    /* shader-clock cycles the wave spent inside each geometry block and inside trace_wave as a whole (wave-uniform; the
     * clock runs on while other waves of the SIMD issue, so these are shares of a wave's life time, not issue cycles) */
    uint64_t cyc_ent, cyc_node, cyc_leaf, cyc_trace;
    /* node trips this lane sat out because it had finished its ray (and waits for the shading quorum), because it waits at a
     * leaf for its triangle run, or for another reason (next entity due, no work left) */
    uint32_t idle_done, idle_leaf, idle_other;
};
__device__ __forceinline__ uint64_t wave_clock() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ void stats_zero(Stats &s) {
    s.segments = s.aabb_tests = s.node_visits = s.tri_tests = s.env_lookups = s.tex_fetches = s.paths = 0;
    s.ent_lanes = s.node_lanes = s.leaf_lanes = s.front_lanes = s.ent_trips = s.node_trips = s.leaf_trips = s.front_trips = 0;
    s.cyc_ent = s.cyc_node = s.cyc_leaf = s.cyc_trace = 0;
    s.idle_done = s.idle_leaf = s.idle_other = 0;
}

struct Best {
    float t, bu, bv;
    int32_t entity;    /* -1 = none */
    uint32_t rec;      /* accepted triangle record: (byte offset of its pair / 16) * 2 + slot (the triangle's index inside its
                        * entity is read from the record when the hit is final: best_triangle) */
};

struct TriLog {        /* optional per-ray record of the triangle-test sequence */
    int32_t *p;
    int32_t cap;
};

#ifndef PT_TRIPS
#define PT_TRIPS 8
#endif
/* a block keeps running while at least 1/PT_KEEP of the lanes it started with still wait for it */
#ifndef PT_KEEP_NODE
#define PT_KEEP_NODE 2
#endif
#ifndef PT_KEEP_LEAF
#define PT_KEEP_LEAF 2
#endif


enum : int { M_ENTITY = 0, M_NODE = 1, M_LEAF = 2, M_DONE = 3, M_IDLE = 4 };

/* per-lane state of the ray being traced */
struct Lane {
    V3 o, d;                 /* the ray (triangle and sphere tests always use the original ray) */
    Best best;
    int32_t mode;
    int32_t ent;             /* entity being processed (M_ENTITY: about to start it) */
    uint32_t emask;          /* builds with the entity pre-pass: bit e = entity e is still worth a visit */
    /* octree walk */
    float t0x, t0y, t0z, t1x, t1y, t1z;   /* parametric interval of the current node */
    uint32_t nb;             /* current node: index of its first child */
    uint32_t todo;           /* current node: its child list (see child_list) | child mask << 24 */
    uint32_t path;           /* 3 bits per level: the child slot taken at each ancestor, under a leading 1 (1 = at the root);
                              * 0 while the lane is not inside an octree (sphere, brute-force run) */
    int32_t rem;             /* levels between the current node's children and the leaves (0: its children are leaves); also
                              * the slot of the frame its parent pushed */
    uint32_t framed;         /* bit s: slot s holds a frame (a node with members left); see node_block */
    int32_t a;
    /* triangle run (octree leaf or brute-force mesh) */
    uint32_t tri_cur, tri_left;   /* byte offset of the run's next triangle pair, triangles still to test (0: no run pending) */
    V3 fr;                        /* clamped reciprocals of d, in the builds that keep them (lane_start_ray) */
    uint32_t spec_mark;           /* counting builds: the lane's node_visits when its pending leaf was entered (see node_block, SPEC) */
    /* a run that accepted a test shows in best.entity (== ent - 1: an entity is visited once per ray and an octree walk
     * ends with the first leaf that accepts); a run that is an octree leaf shows in path != 0 */
};

/*
 * The children of a node that proc_subtree's loop (dgf:159-188) would step through, in order.
 * The loop starts at find_first_node (dgf:95-110: the entry plane is the one with the largest t0;
 * every other axis whose mid plane lies before the entry point has its bit set) and moves on with next_node (dgf:116-126) applied to the exit
 * parameters of the current child (per axis: the node's t1 if the child's bit is set, else tM, the
 * argument table of dgf:178-187): the axis with the smallest exit parameter gets its bit set, or
 * the node is left when that bit is already set.  Bits are only ever set, so the sequence has at
 * most four members and does not depend on which children exist; each step adds a bit to the child number, so the
 * members come in ascending (mirrored) child number and the list is a set:
 *   bits  0..7   go:  child c (mirrored numbering) is a member, exists in the tree and passes the entry
 *                test of dgf:142
 *   bits  8..15  rej: child c is a member and exists but fails that test (entered and left at once: it
 *                only counts; kept in counting builds)
 *   bits 24..31  the node's child mask
 * Which children are members follows from ten comparisons between the interval's parameters (pt_child_table.h has
 * the derivation): they are shifted into a key through the carry flag and the set is read from a table in LDS, and
 * so is the node's child mask in the ray's mirrored numbering (ChildTables at LDS address 0 of the workgroup: a kernel that
 * walks octrees copies them there first).
 */
template <bool STATS>
__device__ __forceinline__ uint32_t child_list(float t0x, float t0y, float t0z, float t1x, float t1y, float t1z, uint32_t nm,
                                               int a) {
    const float tMx = 0.5f * (t0x + t1x), tMy = 0.5f * (t0y + t1y), tMz = 0.5f * (t0z + t1z);
    /* find_first_node (dgf:95-110): the entry plane is the one with the largest t0, and every OTHER axis whose mid plane
     * lies before the entry point te starts with its bit set.  The entry axis needs no special case: its own mid
     * parameter cannot lie before te = its t0, because t0 <= t1 on every axis (the ray was mirrored to run along
     * +x +y +z) and 0.5 * (t0 + t1) rounds monotonically, so tM >= t0; with ties for the largest t0 the tied axes all
     * have tM >= te.  So: te = max(t0), bit = tM < te on all three axes. */
    const float te = fmaxf(fmaxf(t0x, t0y), t0z);
    /* the key: per comparison key = 2 key + (a < b), the flag going from the compare into the addition through VCC.
     * (The two minima in assembly: the parameters are never NaN, and the compiler would quiet them first.) */
    uint32_t key = 0u;
    float myz, mxy;
    asm("v_min_f32_e32 %[myz], %[t1y], %[t1z]\n\t"
        "v_min_f32_e32 %[mxy], %[t1x], %[t1y]\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMx], %[te]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMy], %[te]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMz], %[te]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMx], %[myz]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[t1x], %[tMy]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMy], %[t1z]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[mxy], %[tMz]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMx], %[tMy]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMx], %[tMz]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[tMy], %[tMz]\n\t"
        "v_addc_co_u32_e32 %[k], vcc, %[k], %[k], vcc"
        : [k] "+v"(key), [myz] "=&v"(myz), [mxy] "=&v"(mxy)
        : [tMx] "v"(tMx), [tMy] "v"(tMy), [tMz] "v"(tMz), [te] "v"(te), [t1x] "v"(t1x), [t1y] "v"(t1y), [t1z] "v"(t1z)
        : "vcc");
    /* (the table reads in assembly, addressed from LDS address 0 where the kernels put the tables: through the pointer each
     * read carries an addition of the LDS segment's link-time base, which is zero.  The wait after the second read covers
     * both: LDS returns reads in order) */
    uint32_t set;
    asm volatile("ds_read_u8 %0, %1" : "=v"(set) : "v"(key));
    /* dgf:142, (ex < 0 || ey < 0 || ey < 0): y twice, never z.  The parameters are products of finite differences and
     * clamped reciprocals, never NaN, so "negative" is "bit pattern above -0.0".  A member's ex is the node's t1x when
     * its x bit is set, and the node would not have been entered with a negative t1x (the same test, one level up, or
     * the root's): only tMx can fail, for the four children without the x bit; likewise y */
    const uint32_t rej = (__float_as_uint(tMx) > 0x80000000u ? 0x0Fu : 0u) | (__float_as_uint(tMy) > 0x80000000u ? 0x33u : 0u);
    /* Everything above works on the interval alone.  The node's record, fetched at the start of the trip, is first
     * needed here; the fence and the empty statement after it (which pins the record to this point) keep the compiler
     * from pulling uses of the record, and with them the wait for the fetch, up into the comparisons */
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(nm), "+v"(key));
    uint32_t mir;
    asm volatile("ds_read_u8 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(mir) : "v"(((uint32_t)a << 8) | nm), "n"(kChildSetEntries));
    const uint32_t members = set & mir;
    uint32_t list = (nm << 24) | (members & ~rej);
    if (STATS) list |= (members & rej) << 8;
    return list;
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

/* The entity pre-pass (builds for scenes of many small entities: PREPASS) tests the boxes of all mesh entities against a
 * fresh ray in one uniform loop and leaves the set of entities worth a visit in L.emask: spheres, and meshes whose box the
 * ray does not miss.  The tests use an infinite closest distance, so they pass a superset of what the walk-time test
 * of the same box passes later (which still runs): skipping the others changes no result, only saves them a trip each.
 * Scenes with fewer than 6 entities (the pass costs more than the trips it saves: three-entity teapot scene 6110 against
 * 6925 Mpaths/s with it, twelve-entity box scene 8370 against 6915) or more than 31 (the mask is one word) go without. */
#ifndef PT_PREPASS_MIN
#define PT_PREPASS_MIN 6
#endif
static constexpr int kPrepassMinEntities = PT_PREPASS_MIN;
__device__ __forceinline__ bool uses_entity_mask(const PtSceneDev &sc) { return sc.n_entities >= kPrepassMinEntities && sc.n_entities <= 31; }

/* after an entity has been dealt with: on to the next one, or the ray is finished */
template <bool PREPASS>
__device__ __forceinline__ int next_entity_mode(const PtSceneDev &sc, const Lane &L) {
    if (PREPASS && uses_entity_mask(sc)) return (L.emask >> L.ent) == 0u ? M_DONE : M_ENTITY;      /* ent <= 31 here */
    return L.ent >= sc.n_entities ? M_DONE : M_ENTITY;
}

/* The clamped reciprocals of the ray direction (Ray::recalc_fracs, src/geometry.cuh: three IEEE divisions) are needed by every
 * entity visit.  The builds for scenes of several entities whose lanes have registers to spare (the ones without the cooperative
 * phase) keep them in the lane state from the start of the ray; the others make them in the visit, once per ray there anyway. */
#ifndef PT_KEEP_FRACS
#define PT_KEEP_FRACS 1
#endif
__host__ __device__ constexpr bool keeps_fracs(bool coop, bool prepass, bool multi) { return PT_KEEP_FRACS && !coop && (prepass || multi); }

template <bool KEEP_FR>
__device__ __forceinline__ void lane_start_ray(Lane &L, V3 o, V3 d) {
    L.o = o; L.d = d;
    if (KEEP_FR) L.fr = mk(clamp_frac(d.x), clamp_frac(d.y), clamp_frac(d.z));
    L.best.t = FLT_MAX; L.best.bu = 0.f; L.best.bv = 0.f; L.best.entity = -1; L.best.rec = 0;
    L.ent = 0;
    L.tri_left = 0u;
    L.mode = M_ENTITY;
}

/*
 * ENTITY block: starts processing entity L.ent (trace()'s loop body, dgf:204-206, 241-299).
 * Spheres are tested in place; a mesh passes the AABB gate and then either becomes a triangle run
 * over all its records (no octree) or an octree walk whose root has been entered (ray_step,
 * dgf:39-89: mirror the ray about the octree centre per negative direction component, root
 * interval, entry test).
 */
/* (the block proper, on a view E of the entity's record: the record itself, or a copy of its rows in registers) */
template <bool STATS, bool LOG, bool PREPASS, bool SPEC>
__device__ __forceinline__ void leaf_block(const PtSceneDev &sc, Lane &L, Stats &st, TriLog &lg);

/* An experiment, off (PT_TINY_INLINE = 0): a mesh without octree whose run is a single pair of triangles (a quad: a floor, the
 * walls of a box) tested right here in the builds whose lanes test their own runs (TINY), instead of sending the lane to wait for
 * a leaf trip -- on the teapot scene, where every ray meets the floor quad, 17 of 64 lanes sit out each node trip waiting for
 * that one pair.  Measured: teapot 7750 against 8375 Mpaths/s (the pair test then runs inside every entity trip, at that trip's
 * 58 % occupancy, instead of in a fifth as many leaf trips), Cornell box +2 % at 32 spp and equal at 256. */
#ifndef PT_TINY_INLINE
#define PT_TINY_INLINE 0
#endif
template <bool STATS, bool PREPASS, bool TINY, bool LOG, bool KEEP_FR, typename E>
__device__ __forceinline__ void entity_visit(const PtSceneDev &sc, Lane &L, Stats &st, const E &e, bool masked, TriLog &lg) {
    const int32_t ei = L.ent;
    L.ent = ei + 1;                       /* default: move on to the next entity, or finish */
    L.mode = next_entity_mode<PREPASS>(sc, L);
    if (e.shape == PT_SHAPE_SPHERE) {     /* dgf:257-275; a sphere hit clears the triangle */
        const V3 dd = L.o - mk(e.center[0], e.center[1], e.center[2]);
        const float p1 = -dot3(L.d, dd);
        const float p2sqr = ((p1 * p1) - dot3(dd, dd)) + (e.radius * e.radius);
        if (!(p2sqr < 0)) {
            const float p2 = 1.0f / rsqrt_ieee(p2sqr);
            const float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
            if (t > 0 && t < L.best.t) {
                L.best.t = t; L.best.bu = 0.0f; L.best.bv = 0.0f;
                L.best.entity = ei; L.best.rec = 0;
            }
        }
        return;
    }
    if (STATS && !masked) st.aabb_tests++;        /* (the pre-pass has counted the box tests of all meshes) */
    const V3 fr = KEEP_FR ? L.fr : mk(clamp_frac(L.d.x), clamp_frac(L.d.y), clamp_frac(L.d.z));
    if (!slab_test(e.aabb_min, e.aabb_max, L.o, fr, L.best.t)) return;
    if (e.octree_depth == 0) {            /* dgf:286-291 */
        if (e.n_tris == 0) return;
        L.tri_cur = e.tri_first << 4; L.tri_left = e.n_tris;
        L.path = 0;
        L.mode = M_LEAF;
        if (STATS) L.spec_mark = st.node_visits;
        if (TINY && PT_TINY_INLINE && e.n_tris <= 2u) leaf_block<STATS, LOG, PREPASS, false>(sc, L, st, lg);     /* (leaves the lane at its next entity) */
        return;
    }
    V3 o = L.o, d = L.d;
    const float midx = 0.5f * (e.region_max[0] + e.region_min[0]);
    const float midy = 0.5f * (e.region_max[1] + e.region_min[1]);
    const float midz = 0.5f * (e.region_max[2] + e.region_min[2]);
    int a = 0;
    if (d.x < 0.0f) { float delta = o.x - midx; o.x = midx - delta; d.x = -d.x; a |= 4; }
    if (d.y < 0.0f) { float delta = o.y - midy; o.y = midy - delta; d.y = -d.y; a |= 2; }
    if (d.z < 0.0f) { float delta = o.z - midz; o.z = midz - delta; d.z = -d.z; a |= 1; }
    /* fracs of the mirrored ray (dgf:67-68 recomputes them): 1 / -x == -(1 / x) exactly and the clamp bounds are
     * symmetric, so a mirrored component's frac is the negated frac of the original one -- no second division */
    const float frx = (a & 4) ? -fr.x : fr.x, fry = (a & 2) ? -fr.y : fr.y, frz = (a & 1) ? -fr.z : fr.z;
    const float t0x = (e.region_min[0] - o.x) * frx, t1x = (e.region_max[0] - o.x) * frx;
    const float t0y = (e.region_min[1] - o.y) * fry, t1y = (e.region_max[1] - o.y) * fry;
    const float t0z = (e.region_min[2] - o.z) * frz, t1z = (e.region_max[2] - o.z) * frz;
    const float tmin = fmaxf(fmaxf(t0x, t0y), t0z);
    const float tmax = fminf(fminf(t1x, t1y), t1z);
    if (!((tmin < tmax) && (tmax > 0.0f))) return;
    if (STATS) st.node_visits++;          /* the root is entered */
    if ((t1x < 0.0f) || (t1y < 0.0f) || (t1y < 0.0f)) return;   /* dgf:142 (tests y twice, never z) */
    L.t0x = t0x; L.t0y = t0y; L.t0z = t0z; L.t1x = t1x; L.t1y = t1y; L.t1z = t1z;
    L.nb = e.root_a;                      /* the root's node record, copied into the entity's */
    L.a = a; L.path = 1u; L.framed = 0u; L.rem = e.octree_depth - 1;
    L.todo = child_list<STATS>(t0x, t0y, t0z, t1x, t1y, t1z, e.root_b, a);
    L.mode = M_NODE;
}

/* Entity records in LDS (an experiment, off: PT_LDS_ENTITIES = 0).  A scene of at most PT_LDS_ENTITIES entities has its records
 * copied behind the node step's tables by every workgroup (pt_kernels.hip: load_lds_constants); the block then reads a record
 * with seven 16-byte LDS loads issued together instead of global loads through the vector cache.  Measured with 32: the
 * block's share of the waves' life time falls from a quarter to a seventh on the teapot and Cornell box scenes, and their
 * throughput stays where it was (+0.2 %, +0.5 %): the other waves of the SIMD were covering those latencies already -- the kernel
 * is bound by instruction issue.  The 3.5 KB of LDS cost the dragon scene (depth-8 stacks) its fourth workgroup per CU: -16 %. */
#ifndef PT_LDS_ENTITIES
#define PT_LDS_ENTITIES 0
#endif
static constexpr int kLdsEntities = PT_LDS_ENTITIES;
static constexpr uint32_t kEntLdsByte = 3072u;      /* = sizeof(ChildTables), asserted where the LDS is laid out */
struct EntityRows {
    int32_t shape, material, octree_depth; uint32_t root_node; float aabb_min[3]; uint32_t tri_first; float aabb_max[3]; uint32_t n_tris;
    float region_min[3]; uint32_t shade_base; float region_max[3]; int32_t has_uvs; float center[3]; float radius; uint32_t root_a, root_b;
};

template <bool STATS, bool PREPASS, bool TOGETHER, bool LOG, bool KEEP_FR>
__device__ __forceinline__ void entity_block(const PtSceneDev &sc, Lane &L, Stats &st, bool lds_ents, TriLog &lg) {
    const bool masked = PREPASS && uses_entity_mask(sc);
    if (masked) {
        const uint32_t left = L.emask >> L.ent;
        if (left == 0u) { L.mode = M_DONE; return; }
        L.ent += __builtin_ctz(left);     /* entities in between are not worth a visit */
    } else if (L.ent >= sc.n_entities) { L.mode = M_DONE; return; }
    typedef int iv4 __attribute__((ext_vector_type(4)));
    typedef float fv4 __attribute__((ext_vector_type(4)));
    const PtEntity *ep = &sc.entities[L.ent];
    iv4 r0, r6;
    fv4 r1, r2, r3, r4, r5;
    if (lds_ents) {                       /* (uniform) */
        const uint32_t at = kEntLdsByte + (uint32_t)L.ent * (uint32_t)sizeof(PtEntity);
        const __attribute__((address_space(3))) fv4 *rows = (const __attribute__((address_space(3))) fv4 *)(uintptr_t)at;
        const fv4 q0 = rows[0], q6 = rows[6];
        r1 = rows[1]; r2 = rows[2]; r3 = rows[3]; r4 = rows[4]; r5 = rows[5];
        r0 = __builtin_bit_cast(iv4, q0); r6 = __builtin_bit_cast(iv4, q6);
    } else {
        /* TOGETHER: the rows of the record that a mesh needs are fetched at once, before anything is decided on them.  Read
         * field by field where it is used, every decision (shape, box, octree depth, root region) waits for a fetch of its
         * own, four in a row.  Pays for scenes of one large mesh (+2 % with 871 k triangles); scenes of many small meshes lose by
         * it (box scene -6 %: most trips end at the box test). */
        if (!TOGETHER) { entity_visit<STATS, PREPASS, !TOGETHER, LOG, KEEP_FR>(sc, L, st, *ep, masked, lg); return; }
        r0 = reinterpret_cast<const iv4 *>(ep)[0];
        r1 = reinterpret_cast<const fv4 *>(ep)[1]; r2 = reinterpret_cast<const fv4 *>(ep)[2];
        r3 = reinterpret_cast<const fv4 *>(ep)[3]; r4 = reinterpret_cast<const fv4 *>(ep)[4];
        r5 = reinterpret_cast<const fv4 *>(ep)[5];
        r6 = reinterpret_cast<const iv4 *>(ep)[6];
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6));   /* all here, whole: not split up and moved to their uses */
    }
    EntityRows e;
    e.shape = r0.x; e.material = r0.y; e.octree_depth = r0.z; e.root_node = (uint32_t)r0.w;
    e.aabb_min[0] = r1.x; e.aabb_min[1] = r1.y; e.aabb_min[2] = r1.z; e.tri_first = __float_as_uint(r1.w);
    e.aabb_max[0] = r2.x; e.aabb_max[1] = r2.y; e.aabb_max[2] = r2.z; e.n_tris = __float_as_uint(r2.w);
    e.region_min[0] = r3.x; e.region_min[1] = r3.y; e.region_min[2] = r3.z; e.shade_base = __float_as_uint(r3.w);
    e.region_max[0] = r4.x; e.region_max[1] = r4.y; e.region_max[2] = r4.z; e.has_uvs = __float_as_int(r4.w);
    e.center[0] = r5.x; e.center[1] = r5.y; e.center[2] = r5.z; e.radius = r5.w;
    e.root_a = (uint32_t)r6.y; e.root_b = (uint32_t)r6.z;
    entity_visit<STATS, PREPASS, !TOGETHER, LOG, KEEP_FR>(sc, L, st, e, masked, lg);
}

template <bool STATS, bool KEEP_FR>
__device__ __forceinline__ void entity_prepass(const PtSceneDev &sc, Lane &L, Stats &st) {
    if (!uses_entity_mask(sc)) return;
    const V3 fr = KEEP_FR ? L.fr : mk(clamp_frac(L.d.x), clamp_frac(L.d.y), clamp_frac(L.d.z));
    uint32_t m = 0u;
    for (int e = 0; e < sc.n_entities; e++) {          /* the same trip count in every lane */
        const PtEntity &en = sc.entities[e];
        bool visit = true;
        if (en.shape != PT_SHAPE_SPHERE) {
            if (STATS) st.aabb_tests++;
            visit = slab_test(en.aabb_min, en.aabb_max, L.o, fr, FLT_MAX);
        }
        m |= visit ? (1u << e) : 0u;
    }
    L.emask = m;
}

/*
 * NODE block: proc_subtree's child loop (dgf:159-188) at the current inner node, one entered child
 * per trip.
 *   list exhausted  -> return to the nearest ancestor that still has members (pop), or finish the walk;
 *                      that ancestor's next child is taken in the same trip;
 *   next member     -> the child is entered: at leaf depth it becomes a triangle run, otherwise the
 *                      walk descends and the child's own list is built.
 * Members that do not exist cost nothing; members that fail the entry test are counted when the
 * walk passes over them.
 *
 * Frames.  Descending, each axis of the child interval inherits one bound from the parent ([t0,tM] or
 * [tM,t1]).  Every node with inner children owns the stack slot of its level (slot = rem - 1), and every
 * descent writes it:
 *   - the node has members left (a FRAME): the three bounds the child does not inherit, the node's first-child
 *     index and the rest of its list; the child taken is appended to `path`, the slot's bit is set in `framed`;
 *   - the member taken is the node's last one (nothing would be left to do on return: a third of all returns
 *     used to be of this kind, each a trip of its own): no frame; the slot gets the three bounds of the node's
 *     OWN interval on the sides it inherited from its parent, and its `framed` bit is cleared.
 * Returning from an exhausted node goes straight to the nearest ancestor A with a frame: A's interval is its
 * frame's three bounds plus, on the other sides, the bounds of A's child on the way down -- which is the
 * exhausted node itself, or else a node that descended without a frame and left exactly those bounds in its slot.
 * Copies only; tM is recomputed by the same expression, so every value is the one the recursion would hold.
 * (A counting build keeps a frame while rejected members are left to count, and may find a frame without members
 * on return: it then returns again in its next trip.)
 *
 * Walking ahead of a leaf (SPEC: the builds with the cooperative leaf phase).  A lane that enters a leaf does not stop
 * for it: the leaf's run becomes the lane's PENDING run (tri_left != 0) and the lane keeps walking as if the leaf accepted
 * nothing, which is what happens six times out of seven -- the walk itself never looks at the closest hit, only triangle
 * tests do.  The run is tested in the wave's next cooperative phase.  If it accepted a test, the reference's walk ended at
 * that leaf (dgf:188) and the steps taken since are dropped (the phase sends the lane on to its next entity; a counting
 * build takes its node count back to the mark set on entry); otherwise the lane simply is where the reference would be by
 * now.  One run at a time: a lane that reaches its next leaf, or the end of its walk, with a run still pending waits there
 * (M_LEAF) -- the leaf's member stays in the list and is taken again after the phase -- so triangle tests keep their order.
 */
template <bool STATS, bool PREPASS, bool SPEC>
__device__ __forceinline__ void node_block(const PtSceneDev &sc, Lane &L, Stats &st, float4 *stack, uint32_t *stack1, int stride) {
    if ((L.todo & 0xFFu) == 0u) {
        if (STATS) st.node_visits += __popc((L.todo >> 8) & 0xFFu);
        const uint32_t above = L.framed >> L.rem;              /* bit 0: the parent's slot */
        if (above == 0u) {                                     /* walk finished without an accepted leaf */
            if (SPEC && L.tri_left != 0u) { L.path = 0u; L.mode = M_LEAF; }       /* ... but for the pending one: like a run outside an octree */
            else L.mode = next_entity_mode<PREPASS>(sc, L);
            return;
        }
        const int up = __builtin_ctz(above);
        const int slot = L.rem + up;
        const float4 s = stack[slot * stride];
        const uint32_t w = stack1[slot * stride];
        const float4 h = stack[(slot > 0 ? slot - 1 : 0) * stride];        /* read by lanes with up > 0 only */
        const uint32_t c = L.path & 7u;
        L.path >>= 3;
        /* per axis: k = the bound the frame's owner shares with its child on the way down (the child's t1 where the
         * child is an upper half, else its t0) -- taken from this node if the frame is its parent's (bit 0 of `above`),
         * else from the slot below the frame; then owner.t0 = upper half ? frame word : k, owner.t1 = upper half ? k :
         * frame word.  Bitwise selects on 0 / ~0 masks, in place. */
        {
            uint32_t direct, cm, k;
            asm("v_bfe_i32 %[direct], %[above], 0, 1\n\t"
                "v_bfe_i32 %[cm], %[c], 2, 1\n\t"
                "v_bitop3_b32 %[k], %[cm], %[t1x], %[t0x] bitop3:0xca\n\t"
                "v_bitop3_b32 %[k], %[direct], %[k], %[hx] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t0x], %[cm], %[sx], %[k] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t1x], %[cm], %[k], %[sx] bitop3:0xca\n\t"
                "v_bfe_i32 %[cm], %[c], 1, 1\n\t"
                "v_bitop3_b32 %[k], %[cm], %[t1y], %[t0y] bitop3:0xca\n\t"
                "v_bitop3_b32 %[k], %[direct], %[k], %[hy] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t0y], %[cm], %[sy], %[k] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t1y], %[cm], %[k], %[sy] bitop3:0xca\n\t"
                "v_bfe_i32 %[cm], %[c], 0, 1\n\t"
                "v_bitop3_b32 %[k], %[cm], %[t1z], %[t0z] bitop3:0xca\n\t"
                "v_bitop3_b32 %[k], %[direct], %[k], %[hz] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t0z], %[cm], %[sz], %[k] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t1z], %[cm], %[k], %[sz] bitop3:0xca"
                : [direct] "=&v"(direct), [cm] "=&v"(cm), [k] "=&v"(k), [t0x] "+v"(L.t0x), [t1x] "+v"(L.t1x), [t0y] "+v"(L.t0y),
                  [t1y] "+v"(L.t1y), [t0z] "+v"(L.t0z), [t1z] "+v"(L.t1z)
                : [above] "v"(above), [c] "v"(c), [sx] "v"(s.x), [sy] "v"(s.y), [sz] "v"(s.z), [hx] "v"(h.x), [hy] "v"(h.y), [hz] "v"(h.z));
        }
        L.rem = slot + 1;
        L.nb = __float_as_uint(s.w);
        L.todo = w;
        if (STATS && (w & 0xFFu) == 0u) return;               /* only rejected members were left: counted in the next trip */
    }
    const uint32_t todo_before = L.todo;
    const uint32_t visits_before = STATS ? st.node_visits : 0u;
    const int curr = __builtin_ctz(L.todo);                   /* lowest go bit = next member (go != 0 here) */
    if (STATS) {
        const uint32_t below = (1u << curr) - 1u;
        st.node_visits += 1 + __popc((L.todo >> 8) & below);
        L.todo &= ~((1u << curr) | (below << 8));            /* consumed; passed-over rejects counted */
    } else {
        L.todo &= L.todo - 1u;                               /* consumed */
    }
    const uint32_t nm = L.todo >> 24;
    const float tMx = 0.5f * (L.t0x + L.t1x), tMy = 0.5f * (L.t0y + L.t1y), tMz = 0.5f * (L.t0z + L.t1z);
    const int adj = curr ^ L.a;
    const uint32_t child = L.nb + (uint32_t)__popc(nm & ((1u << adj) - 1u));
    const PtNode rec = sc.nodes[child];
    if (L.rem == 0) {                     /* leaf: dgf:146-157 */
        if (rec.b != 0u) {
            if (SPEC && L.tri_left != 0u) {               /* a run is pending already: wait here, the member goes back into the list */
                L.todo = todo_before;
                if (STATS) st.node_visits = visits_before;
                L.mode = M_LEAF;
            } else {
                L.tri_cur = rec.a << 4; L.tri_left = rec.b;
                if (STATS) L.spec_mark = st.node_visits;
                if (!SPEC) L.mode = M_LEAF;               /* (SPEC: the lane walks on, see above) */
            }
        }
        return;
    }
    /* nothing left at this node: no frame.  As a 0 / ~0 word, so that the three updates that depend on it are one
     * bitwise select each instead of a compare and selects on its mask */
    uint32_t last;                                           /* = (int32_t)((L.todo & 0xFFFF) - 1) >> 31, kept from being turned back into a compare */
    asm("v_add_u32_e32 %0, -1, %1\n\tv_ashrrev_i32_e32 %0, 31, %0" : "=v"(last) : "v"(L.todo & 0xFFFFu));
    /* which bound of each axis goes into the slot: frame -> the one the child does not inherit (t0 where it takes
     * the upper half); no frame -> the one this node inherited from its parent (t1 where IT is an upper half: the last
     * digit of `path`, which is the parent's if the parent has a frame; otherwise nobody reads the slot) */
    uint32_t pick;                                           /* last ? path : ~curr; bits 0..2 are read */
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xc5" : "=v"(pick) : "v"(last), "v"(L.path), "v"(curr));
    float4 s;
    uint32_t sel;                                            /* 0 / ~0: the child takes the upper half of the axis */
    asm("v_add_u32_e32 %0, -1, %0" : "+v"(L.rem));               /* in place, like the interval below */
    const uint32_t bit = 1u << L.rem;
    L.framed = (L.framed & ~bit) | (bit & ~last);
    /* the child's interval, selected in place (left to the compiler it lands in fresh registers and is copied back and
     * forth to join the lanes that do not descend); the slot words first: they read the bounds the others overwrite */
    asm("v_bfe_i32 %[sx], %[pick], 2, 1\n\t"
        "v_bfe_i32 %[sy], %[pick], 1, 1\n\t"
        "v_bfe_i32 %[sz], %[pick], 0, 1\n\t"
        "v_bitop3_b32 %[sx], %[sx], %[t1x], %[t0x] bitop3:0xca\n\t"
        "v_bitop3_b32 %[sy], %[sy], %[t1y], %[t0y] bitop3:0xca\n\t"
        "v_bitop3_b32 %[sz], %[sz], %[t1z], %[t0z] bitop3:0xca\n\t"
        "v_bfe_i32 %[m], %[curr], 2, 1\n\t"
        "v_bitop3_b32 %[t0x], %[m], %[tMx], %[t0x] bitop3:0xca\n\t"
        "v_bitop3_b32 %[t1x], %[m], %[t1x], %[tMx] bitop3:0xca\n\t"
        "v_bfe_i32 %[m], %[curr], 1, 1\n\t"
        "v_bitop3_b32 %[t0y], %[m], %[tMy], %[t0y] bitop3:0xca\n\t"
        "v_bitop3_b32 %[t1y], %[m], %[t1y], %[tMy] bitop3:0xca\n\t"
        "v_bfe_i32 %[m], %[curr], 0, 1\n\t"
        "v_bitop3_b32 %[t0z], %[m], %[tMz], %[t0z] bitop3:0xca\n\t"
        "v_bitop3_b32 %[t1z], %[m], %[t1z], %[tMz] bitop3:0xca"
        : [sx] "=&v"(s.x), [sy] "=&v"(s.y), [sz] "=&v"(s.z), [m] "=&v"(sel), [t0x] "+v"(L.t0x), [t1x] "+v"(L.t1x), [t0y] "+v"(L.t0y),
          [t1y] "+v"(L.t1y), [t0z] "+v"(L.t0z), [t1z] "+v"(L.t1z)
        : [pick] "v"(pick), [curr] "v"(curr), [tMx] "v"(tMx), [tMy] "v"(tMy), [tMz] "v"(tMz));
    s.w = __uint_as_float(L.nb);
    stack[L.rem * stride] = s;
    stack1[L.rem * stride] = L.todo;
    asm("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xca" : "+v"(L.path) : "v"(last), "v"((L.path << 3) | (uint32_t)curr));   /* last ? path : path.curr */
    L.todo = child_list<STATS>(L.t0x, L.t0y, L.t0z, L.t1x, L.t1y, L.t1z, rec.b, L.a);
    asm volatile("v_mov_b32_e32 %0, %1" : "+v"(L.nb) : "v"(rec.a));     /* after the list: the record is used as late as possible */
}

/*
 * LEAF block: Moller-Trumbore tests (dgf:319-365) on the pre-gathered records of the lane's run, two
 * per trip (the last trip of an odd run tests one); e1,e2 were formed on the host by the same fp32
 * subtractions the reference performs per test.
 *
 * The two tests are evaluated side by side in two-wide vectors, which the hardware executes as
 * packed fp32 instructions (separately rounded multiplies and adds, as in the scalar code).
 *
 * The reference divides by det before any comparison.  Here the three ratios are first formed
 * with the hardware reciprocal (relative error < 3e-7): if one of them fails its test by more
 * than a 1e-5 margin, the exactly rounded value fails it too and the test is rejected without the
 * IEEE division; otherwise the exact sequence runs, first for the first record, then for the
 * second, which sees the closest hit as the first one left it.  Accepted hits and every borderline
 * case go through the exact path, so the outcome is bit-identical to testing one by one.
 *
 * When the run ends: an octree leaf that accepted a test ends the whole walk of this entity
 * (dgf:188 unwinds every level once `hit` is set); otherwise the walk resumes; a brute-force run
 * moves on to the next entity.
 */
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool STATS, bool LOG, bool PREPASS, bool SPEC>
__device__ __forceinline__ void leaf_block(const PtSceneDev &sc, Lane &L, Stats &st, TriLog &lg) {
    const uint32_t idx = L.tri_cur;
    const bool two = L.tri_left != 1u;
    /* uniform base + 32-bit lane offset: the loads need no 64-bit address arithmetic */
    const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.tris) + idx);
    const float4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
    L.tri_cur = idx + (uint32_t)sizeof(PtTriPair);
    L.tri_left -= two ? 2u : 1u;
    if (STATS) st.tri_tests += two ? 2u : 1u;
    const int32_t id_a = __float_as_int(r4.z), id_b = __float_as_int(r4.w);
    if (LOG) {
        int n = lg.p[0]++;
        if (n + 1 < lg.cap) lg.p[n + 1] = id_a;
        if (two) {
            n = lg.p[0]++;
            if (n + 1 < lg.cap) lg.p[n + 1] = id_b;
        }
    }
    const f2 v0x = {r0.x, r0.y}, v0y = {r0.z, r0.w}, v0z = {r1.x, r1.y};
    const f2 e1x = {r1.z, r1.w}, e1y = {r2.x, r2.y}, e1z = {r2.z, r2.w};
    const f2 e2x = {r3.x, r3.y}, e2y = {r3.z, r3.w}, e2z = {r4.x, r4.y};
    float sdx = L.d.x, sdy = L.d.y, sdz = L.d.z, sox = L.o.x, soy = L.o.y, soz = L.o.z;
    /* keeps the six reads scalar: folded into one-element vector loads by the optimiser, they stop the
     * lane state from being promoted to registers */
    asm volatile("" : "+v"(sdx), "+v"(sdy), "+v"(sdz), "+v"(sox), "+v"(soy), "+v"(soz));
    const f2 dx = {sdx, sdx}, dy = {sdy, sdy}, dz = {sdz, sdz};
    /* pvec = d x e2 ; det = e1 . pvec */
    const f2 px = (dy * e2z) - (dz * e2y), py = (dz * e2x) - (dx * e2z), pz = (dx * e2y) - (dy * e2x);
    const f2 det = (e1x * px + e1y * py) + e1z * pz;
    /* tvec = o - v0 ; qvec = tvec x e1 */
    const f2 ox = {sox, sox}, oy = {soy, soy}, oz = {soz, soz};
    const f2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    const f2 qx = (ty * e1z) - (tz * e1y), qy = (tz * e1x) - (tx * e1z), qz = (tx * e1y) - (ty * e1x);
    const f2 un = (tx * px + ty * py) + tz * pz;
    const f2 vn = (dx * qx + dy * qy) + dz * qz;
    const f2 tn = (e2x * qx + e2y * qy) + e2z * qz;
    const f2 ri = {__builtin_amdgcn_rcpf(det.x), __builtin_amdgcn_rcpf(det.y)};
    const f2 ua = un * ri, va = vn * ri, ta = tn * ri;
    const f2 uv = ua + va;
    const float far = L.best.t * 1.00001f;
    uint32_t pend = 0;
    if (!(fabsf(det.x) < 0.000000001f) &&
        !((ua.x < -0.00011f) || (ua.x > 1.00011f) || (va.x < -0.00011f) || (uv.x > 1.00012f) || (ta.x < -1e-30f) || (ta.x > far)))
        pend = 1u;
    if (two && !(fabsf(det.y) < 0.000000001f) &&
        !((ua.y < -0.00011f) || (ua.y > 1.00011f) || (va.y < -0.00011f) || (uv.y > 1.00012f) || (ta.y < -1e-30f) || (ta.y > far)))
        pend |= 2u;
    /* the closest hit is carried in locals and written back with value selects: conditional stores
     * into the lane state inside this loop would push the state out of registers */
    float bt = L.best.t, bbu = L.best.bu, bbv = L.best.bv;
    int32_t bent = L.best.entity;
    uint32_t brec = L.best.rec;
    while (pend != 0u) {
        const bool second = !(pend & 1u);
        pend &= second ? ~2u : ~1u;
        const float det_s = second ? det.y : det.x, un_s = second ? un.y : un.x, vn_s = second ? vn.y : vn.x, tn_s = second ? tn.y : tn.x;
        const float inv_det = 1.0f / det_s;
        const float u = un_s * inv_det, v = vn_s * inv_det, t = tn_s * inv_det;
        const bool acc = !(u < -0.0001f || u > 1.0001f) && !(v < -0.0001f || u + v > 1.0001f) && (t > 0 && t < bt);
        bt = acc ? t : bt; bbu = acc ? u : bbu; bbv = acc ? v : bbv;
        bent = acc ? L.ent - 1 : bent;
        brec = acc ? ((idx >> 3) | (second ? 1u : 0u)) : brec;
    }
    L.best.t = bt; L.best.bu = bbu; L.best.bv = bbv; L.best.entity = bent; L.best.rec = brec;
    if (L.tri_left == 0u) {
        /* (SPEC: the lane may have walked on meanwhile, see node_block; an accepted test ends that walk, and the nodes it
         * visited do not count) */
        if (L.path != 0u && bent != L.ent - 1) L.mode = M_NODE;
        else {
            if (STATS && SPEC && bent == L.ent - 1) st.node_visits = L.spec_mark;
            L.mode = next_entity_mode<PREPASS>(sc, L);
        }
    }
}

/*
 * Cooperative LEAF phase.  In leaf_block a lane works through its own run, two records per trip, while the
 * lanes in other states wait: with the lanes of a wave split between the octree walk and triangle runs of
 * very different lengths, a trip has well under half of its lanes active.  Here the pending records of ALL
 * lanes in M_LEAF are laid end to end and dealt out, one pair per lane and trip, to all 64 lanes of the
 * wave whatever their own state: lane w of trip k takes pair 64k+w of the concatenation, finds the lane
 * that owns it, fetches that lane's ray through the cross-lane network and runs the same packed test
 * (pre-filter, then the exact IEEE sequence for candidates) against the owner's closest distance as it
 * was when the phase began.  Every run is finished when the phase ends (runs longer than kCoopMaxPairs
 * continue in the next phase), so the owners return to the walk together.
 *
 * Result.  Testing a run one by one accepts a test iff 0 < t < closest-so-far, so what the run leaves
 * behind is the smallest accepted t, the earliest record on ties.  The candidates here are the tests with
 * 0 < t < the closest distance at the start of the run, and the owner takes the minimum of (t, position in
 * the run) over them -- an LDS 64-bit unsigned minimum, positive floats ordering like their bit patterns:
 * the same hit, bit for bit, and "some test of the run was accepted" is "there is a candidate".
 *
 * Per-wave LDS: the best key and its barycentrics per owner, the table rank-of-run -> owner, and one bit
 * per pair position where a run starts.
 */
#ifndef PT_COOP_MAX_PAIRS
#define PT_COOP_MAX_PAIRS 16
#endif
static constexpr uint32_t kCoopMaxPairs = PT_COOP_MAX_PAIRS;     /* pairs per lane and phase (a power of two): at most 64 times as many per phase */
static constexpr int kCoopPlanes = 32 - __builtin_clz(PT_COOP_MAX_PAIRS);
struct CoopLds {
    unsigned long long res[64];
    float win[64][2];
    uint32_t own[64];
    unsigned long long starts[kCoopMaxPairs];
};

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ float lane_read(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool STATS>
__device__ __forceinline__ void leaf_coop(const PtSceneDev &sc, Lane &L, Stats &st, CoopLds *C, int lane) {
    /* every lane with a pending run takes part as an owner, whether it waits at the leaf or has walked on (a lane without a
     * run has tri_left == 0) */
    const uint32_t my_left = L.tri_left;
    const uint32_t my_pairs = (my_left + 1u) >> 1;
    const uint32_t np_all = min(my_pairs, kCoopMaxPairs);
    /* The pending pairs are laid end to end in two groups: first the runs of the lanes that WAIT for the phase (M_LEAF), in
     * lane order, then the runs of the lanes that have walked on.  Exclusive prefixes and totals of both groups come from one
     * scan of a word with a 16-bit count per group (at most 64 * kCoopMaxPairs pairs each): a scan on the cross-lane data
     * path, log steps inside each row of 16 lanes (a lane without a source adds zero), then the last lane of a row broadcast
     * into the rows after it.  All 64 lanes are active here. */
    const bool waits = L.mode == M_LEAF;
    const uint32_t both = waits ? np_all : np_all << 16;
    uint32_t incl = both;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);      /* row_shr:1 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);      /* row_shr:2 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);      /* row_shr:4 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);      /* row_shr:8 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);     /* row_bcast:15 into rows 1, 3 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);     /* row_bcast:31 into rows 2, 3 */
    const uint32_t before = incl - both;
    const uint32_t totals = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t T_wait = totals & 0xFFFFu, T_all = T_wait + (totals >> 16);
    const uint32_t excl = waits ? (before & 0xFFFFu) : T_wait + (before >> 16);      /* position of the lane's first pair */
    /* Whole trips only: of more than 64 pending pairs the phase takes a multiple of 64 and leaves the tail pending for the
     * next phase, where new runs fill the trip up -- the tail being runs of lanes that walk on meanwhile, as long as there
     * are such.  (A run that is cut continues where it stopped, like one longer than kCoopMaxPairs pairs.) */
    const uint32_t T = T_all >= 64u ? (T_all & ~63u) : T_all;
    const uint32_t np = min(np_all, T - min(excl, T));
    const unsigned long long runs_wait = __ballot(np != 0u && waits), runs_walk = __ballot(np != 0u && !waits);
    const uint32_t rank = waits ? lanes_below(runs_wait) : (uint32_t)__popcll(runs_wait) + lanes_below(runs_walk);      /* runs before this one */
    /* the two constants are made here: left to the compiler they are hoisted out of the kernel's work loop, where one
     * sat in a register pair for good and the other was spilled and came back from scratch memory in every phase */
    unsigned long long zeros, ones;
    asm volatile("v_mov_b64_e32 %0, 0\n\tv_mov_b64_e32 %1, -1" : "=v"(zeros), "=v"(ones));
    if (lane < (int)kCoopMaxPairs) C->starts[lane] = zeros;
    C->res[lane] = ones;
    wave_lds_fence();
    if (np != 0u) {
        atomicOr(&C->starts[excl >> 6], 1ull << (excl & 63u));
        C->own[rank] = (uint32_t)lane | (excl << 8);
    }
    wave_lds_fence();
    const uint32_t my_cur = L.tri_cur;
    const float my_bt = L.best.t;
    const int n_words = (int)((T + 63u) >> 6);
    uint32_t base = 0;
#pragma nounroll
    for (int k = 0; k < n_words; k++) {
        unsigned long long m = C->starts[k];
        m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
        const uint32_t g = (uint32_t)(64 * k + lane);
        const bool act = g < T;
        /* the run pair g belongs to: the last one that starts at or before g */
        const uint32_t r = base + lanes_below(m) + (uint32_t)((m >> lane) & 1ull) - 1u;
        base += (uint32_t)__popcll(m);
        const uint32_t ow = C->own[r & 63u];
        const int owner = (int)(ow & 63u);
        const uint32_t j = g - (ow >> 8);
        const float sox = lane_read(L.o.x, owner), soy = lane_read(L.o.y, owner), soz = lane_read(L.o.z, owner);
        const float sdx = lane_read(L.d.x, owner), sdy = lane_read(L.d.y, owner), sdz = lane_read(L.d.z, owner);
        const float bt0 = lane_read(my_bt, owner);
        const uint32_t o_cur = lane_read(my_cur, owner), o_left = lane_read(my_left, owner);
        if (STATS) st.leaf_trips++;
        if (act) {
            if (STATS) st.leaf_lanes++;
            const uint32_t idx = o_cur + j * (uint32_t)sizeof(PtTriPair);
            const bool two = !((o_left & 1u) && j == ((o_left + 1u) >> 1) - 1u);
            const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.tris) + idx);
            const float4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
            if (STATS) st.tri_tests += two ? 2u : 1u;
            const f2 v0x = {r0.x, r0.y}, v0y = {r0.z, r0.w}, v0z = {r1.x, r1.y};
            const f2 e1x = {r1.z, r1.w}, e1y = {r2.x, r2.y}, e1z = {r2.z, r2.w};
            const f2 e2x = {r3.x, r3.y}, e2y = {r3.z, r3.w}, e2z = {r4.x, r4.y};
            const f2 dx = {sdx, sdx}, dy = {sdy, sdy}, dz = {sdz, sdz};
            const f2 px = (dy * e2z) - (dz * e2y), py = (dz * e2x) - (dx * e2z), pz = (dx * e2y) - (dy * e2x);
            const f2 det = (e1x * px + e1y * py) + e1z * pz;
            const f2 ox = {sox, sox}, oy = {soy, soy}, oz = {soz, soz};
            const f2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
            const f2 qx = (ty * e1z) - (tz * e1y), qy = (tz * e1x) - (tx * e1z), qz = (tx * e1y) - (ty * e1x);
            const f2 un = (tx * px + ty * py) + tz * pz;
            const f2 vn = (dx * qx + dy * qy) + dz * qz;
            const f2 tn = (e2x * qx + e2y * qy) + e2z * qz;
            const f2 ri = {__builtin_amdgcn_rcpf(det.x), __builtin_amdgcn_rcpf(det.y)};
            const f2 ua = un * ri, va = vn * ri, ta = tn * ri;
            const f2 uv = ua + va;
            const float far = bt0 * 1.00001f;
            uint32_t pend = 0;
            {
                /* the six bounds as margins (the sign of a rounded sum or difference is the sign of the exact one), the
                 * smallest margin decides: packed adds serve both tests */
                const f2 m1 = ua + 0.00011f, m2 = 1.00011f - ua, m3 = va + 0.00011f, m4 = 1.00012f - uv, m5 = ta + 1e-30f;
                const f2 farv = {far, far};
                const f2 m6 = farv - ta;
                const float wx = fminf(fminf(fminf(m1.x, m2.x), m3.x), fminf(fminf(m4.x, m5.x), m6.x));
                const float wy = fminf(fminf(fminf(m1.y, m2.y), m3.y), fminf(fminf(m4.y, m5.y), m6.y));
                if (!(fabsf(det.x) < 0.000000001f) && !(wx < 0.0f)) pend = 1u;
                if (two && !(fabsf(det.y) < 0.000000001f) && !(wy < 0.0f)) pend |= 2u;
            }
            float bt = bt0, bu = 0.f, bv = 0.f;
            uint32_t slot = 0;
            bool has = false;
            while (pend != 0u) {
                const bool second = !(pend & 1u);
                pend &= second ? ~2u : ~1u;
                const float det_s = second ? det.y : det.x, un_s = second ? un.y : un.x, vn_s = second ? vn.y : vn.x, tn_s = second ? tn.y : tn.x;
                const float inv_det = 1.0f / det_s;
                const float u = un_s * inv_det, v = vn_s * inv_det, t = tn_s * inv_det;
                const bool acc = !(u < -0.0001f || u > 1.0001f) && !(v < -0.0001f || u + v > 1.0001f) && (t > 0 && t < bt);
                bt = acc ? t : bt; bu = acc ? u : bu; bv = acc ? v : bv;
                slot = acc ? (second ? 1u : 0u) : slot;
                has = has || acc;
            }
            if (has) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(bt) << 32) | (unsigned long long)(2u * j + slot);
                atomicMin(&C->res[owner], key);
                wave_lds_fence();
                /* (read back through an LDS pointer: as a volatile generic access it became a flat load with a full wait) */
                if (__hip_atomic_load((__attribute__((address_space(3))) unsigned long long *)&C->res[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == key) {
                    C->win[owner][0] = bu;
                    C->win[owner][1] = bv;
                }
            }
        }
        wave_lds_fence();
    }
    /* Results back into the lanes, as selects that every lane executes (no divergent region, so the lane state is
     * updated in place): only an owner's slot can hold a key, and a lane without pairs in the phase has np = 0 */
    {
        const unsigned long long key = C->res[lane];
        const bool got = key != ~0ull;
        const uint32_t oi = (uint32_t)key;
        const uint32_t off = my_cur + (oi >> 1) * (uint32_t)sizeof(PtTriPair);
        const float wu = C->win[lane][0], wv = C->win[lane][1];
        L.best.t = got ? __uint_as_float((uint32_t)(key >> 32)) : L.best.t;
        L.best.bu = got ? wu : L.best.bu;
        L.best.bv = got ? wv : L.best.bv;
        L.best.entity = got ? L.ent - 1 : L.best.entity;
        L.best.rec = got ? ((off >> 3) | (oi & 1u)) : L.best.rec;
        L.tri_cur += np * (uint32_t)sizeof(PtTriPair);
        L.tri_left -= min(my_left, 2u * np);
        const bool finished = np != 0u && L.tri_left == 0u;
        /* a finished run that accepted a test ends the walk of its entity, wherever the lane has walked to meanwhile; one that
         * did not leaves a lane that walked on where it is, and sends a lane that waited back to its node (or, after a run
         * outside an octree or at the end of a walk, on to the next entity) */
        const bool accepted = L.best.entity == L.ent - 1;
        const int after = (L.path != 0u && !accepted) ? M_NODE : next_entity_mode<false>(sc, L);
        if (STATS && finished && accepted) st.node_visits = L.spec_mark;
        L.mode = finished ? after : L.mode;
    }
}

/*
 * Runs the geometry blocks until no lane of the wave is in M_ENTITY / M_NODE / M_LEAF any more, or
 * until at least `done_quota` lanes wait in M_DONE (so the caller can shade / refill them together).
 * The block with the most waiting lanes runs next, for a few trips while it stays well filled.
 */
/* Lanes of the wave for which `p` holds.  The population count of a ballot is a 64-bit value to the compiler, which then
 * compares counts with 64-bit vector compares (there are no 64-bit scalar ones).  The empty statement hides the origin
 * of the count, so that the comparisons become 32-bit scalar ones: +2 % to +3 % in every build. */
__device__ __forceinline__ int count_lanes(bool p) {
    int n = __builtin_popcountll(__ballot(p));
    asm("" : "+s"(n));
    return n;
}

/* Node trips between two looks at the lane counts that end the node block: the counts cost two vector compares and
 * eight scalar instructions, 4 % of a trip.  Measured on the north-star scene: 2 -> +1.6 %, 3 and 4 -> +1.3 % and 0 %
 * (the block then runs on with too few lanes); no gain for the leaf trips of the builds without the cooperative phase */
#ifndef PT_NODE_GROUP
#define PT_NODE_GROUP 2
#endif
/* Walking ahead of a pending leaf in the builds without the cooperative phase too (not in the one that keeps a test log).
 * Off: measured -1.5 % on the teapot scene (8290 against 8400 Mpaths/s) and -1 % on the Cornell box -- the lanes at a
 * leaf get their trips soon there anyway, and a single object's leaves accept often, which throws the walk away. */
#ifndef PT_SPEC_LIGHT
#define PT_SPEC_LIGHT 0
#endif
template <bool STATS, bool LOG, bool COOP, bool PREPASS, bool MULTI>
__device__ __forceinline__ void trace_wave(const PtSceneDev &sc, Lane &L, Stats &st, TriLog &lg, float4 *stack, uint32_t *stack1,
                                           int stride, int done_quota, int leaf_quota, CoopLds *coop, int lane, bool lds_ents) {
    static_assert(!(LOG && COOP), "the cooperative leaf phase keeps no test log");
    static_assert(!(PREPASS && COOP), "the entity pre-pass belongs to the builds for small scenes");
    constexpr bool SPEC = COOP || (PT_SPEC_LIGHT && !LOG);
    bool phase_due = false;
    for (;;) {
        /* The lane counts that pick the next block cost a vector compare and a few scalar instructions each, and a wave in the
         * middle of its walk alternates between a cooperative phase and a couple of node trips: those two are decided on three
         * counts, in one pass of this loop (a phase when enough runs wait, then node trips while the walkers are a majority
         * and the finished lanes no quorum yet).  Everything else takes the full set of counts below. */
        int n_pend = 0;
        if (COOP) {
            n_pend = count_lanes(L.tri_left != 0u);
            if (n_pend >= leaf_quota || phase_due) {
                const uint64_t c0 = STATS ? wave_clock() : 0;
                leaf_coop<STATS>(sc, L, st, coop, lane);
                if (STATS) st.cyc_leaf += wave_clock() - c0;
                n_pend = 0; phase_due = false;
            }
        }
        const int n_node = count_lanes(L.mode == M_NODE);
        const int n_done = count_lanes(L.mode == M_DONE);
        int which = -1;
        int n_leaf = 0;
        if (COOP && n_node >= 32 && n_done < done_quota) {
            which = 1;
        } else {
            const int n_ent = count_lanes(L.mode == M_ENTITY);
            n_leaf = count_lanes(L.mode == M_LEAF);
            /* runs waiting for a cooperative phase: those of the lanes that wait at a leaf and those of lanes that have walked on */
            if (!SPEC) n_pend = n_leaf;
            else if (n_pend == 0) n_pend = count_lanes(L.tri_left != 0u);
            if (n_ent + n_node + n_leaf == 0) return;
            /* leave to shade / refill once the finished lanes are a quorum and no block has more lanes waiting */
            if (n_done >= done_quota && n_done >= n_node && n_done >= n_leaf) return;
            /* a cooperative leaf phase serves every waiting run with all 64 lanes: it runs as soon as `leaf_quota`
             * runs wait (about enough pairs for a full trip), and the walk stops for it */
            const bool leaf_first = COOP ? (n_pend >= leaf_quota || n_leaf > n_node) : (n_pend > n_node);
            /* the entity block needs twice the lanes of the other blocks to run first in a scene of several entities: a lane
             * that comes out of one entity's octree then waits for the next entity's trip -- on the teapot scene 16 of 64 lanes
             * sit out each node trip that way -- but the trip is dear (box test, octree entry, the root's child set), and
             * batching more lanes into it pays: teapot +4.5 %, 871 k-triangle mesh with floor +1.9 %; running it EARLIER, as
             * soon as 8 or 16 lanes wait, costs 10 % and 2.6 %.  Scenes of one entity meet the block only with the fresh rays
             * of a round, all at once, and the pre-pass builds skip most visits anyway: no weight there (-0.3 %, -0.7 %).
             * MULTI: the instantiations for scenes of two and more entities that have no pre-pass. */
            constexpr int ew = MULTI ? 2 : 1;           /* (an instantiation of its own: as a run-time weight it cost the single-entity scenes 0.6 %) */
            which = (n_ent >= ew * n_node && n_ent >= ew * n_leaf) ? 0 : (!leaf_first ? 1 : 2);
        }
        /* one of three blocks runs; written as three independent conditionals (not an if / else-if chain): every block
         * then is a plain divergent region that updates the lane state in place */
        const uint64_t c0 = STATS ? wave_clock() : 0;
        if (which == 0) {
            if (STATS) st.ent_trips++;
            if (L.mode == M_ENTITY) { if (STATS) st.ent_lanes++; entity_block<STATS, PREPASS, COOP, LOG, keeps_fracs(COOP, PREPASS, MULTI)>(sc, L, st, lds_ents, lg); }
            if (STATS) st.cyc_ent += wave_clock() - c0;
        }
        if (which == 1) {
            int trips = 0;
#pragma nounroll
            do {
#pragma unroll
                for (int u = 0; u < PT_NODE_GROUP; u++) {
                    if (STATS) {
                        st.node_trips++;
                        st.idle_done += L.mode == M_DONE ? 1u : 0u;
                        st.idle_leaf += L.mode == M_LEAF ? 1u : 0u;
                        st.idle_other += (L.mode == M_ENTITY || L.mode == M_IDLE) ? 1u : 0u;
                    }
                    if (L.mode == M_NODE) { if (STATS) st.node_lanes++; node_block<STATS, PREPASS, SPEC>(sc, L, st, stack, stack1, stride); }
                }
            } while ((trips += PT_NODE_GROUP) < PT_TRIPS && PT_KEEP_NODE * __popcll(__ballot(L.mode == M_NODE)) >= n_node &&
                     !(COOP && __popcll(__ballot(L.tri_left != 0u)) >= leaf_quota));
            if (STATS) st.cyc_node += wave_clock() - c0;
        }
        if (which == 2) {
            if (COOP) {
                phase_due = true;          /* (the one copy of the phase's code is at the head of the loop) */
            } else {
                int trips = 0;
#pragma nounroll
                do {
                    if (STATS) st.leaf_trips++;
                    if (L.tri_left != 0u) { if (STATS) st.leaf_lanes++; leaf_block<STATS, LOG, PREPASS, SPEC>(sc, L, st, lg); }
                } while (++trips < PT_TRIPS && PT_KEEP_LEAF * __popcll(__ballot(L.tri_left != 0u)) >= n_pend);
                if (STATS) st.cyc_leaf += wave_clock() - c0;
            }
        }
    }
}

struct Surface {     /* what shading needs from a hit */
    V3 position, normal;
    float u, v;
};

/* position / normal / uv of the accepted hit: the tail of intersect_triangle (dgf:353-361) and of
 * trace() (dgf:208-231), evaluated once for the final hit instead of on every improvement. */
/* index of the accepted triangle inside its entity, -1 for a sphere (and for no hit) */
__device__ __forceinline__ int32_t best_triangle(const PtSceneDev &sc, const Best &best) {
    if (best.entity < 0 || sc.entities[best.entity].shape == PT_SHAPE_SPHERE) return -1;
    return reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(sc.tris) + (size_t)(best.rec >> 1) * 16u)[18u + (best.rec & 1u)];
}

__device__ __forceinline__ void finish_hit(const PtSceneDev &sc, V3 o, V3 d, const Best &best, Surface &sf) {
    const PtEntity &e = sc.entities[best.entity];
    sf.position = o + best.t * d;
    sf.u = best.bu; sf.v = best.bv;
    if (e.shape == PT_SHAPE_SPHERE) {
        sf.normal = normalized(sf.position - mk(e.center[0], e.center[1], e.center[2]));
        return;
    }
    const float *pf = reinterpret_cast<const float *>(reinterpret_cast<const char *>(sc.tris) + (size_t)(best.rec >> 1) * 16u) + (best.rec & 1u);
    const V3 e1 = mk(pf[6], pf[8], pf[10]), e2 = mk(pf[12], pf[14], pf[16]);
    V3 nv = cross3(e1, e2);               /* the vector that becomes the normal: normalised once, below */
    if (e.has_uvs || e.smooth_shading) {
        const PtShadeRec &sh = sc.shade[e.shade_base + (uint32_t)reinterpret_cast<const int32_t *>(pf - (best.rec & 1u))[18u + (best.rec & 1u)]];
        const float u = best.bu, v = best.bv;
        const float w = 1.0f - (u + v);
        if (e.has_uvs) {   /* meshes without uvs keep the barycentrics (the reference reads d_uvs[-1]) */
            sf.u = (u * sh.uv1[0] + v * sh.uv2[0]) + w * sh.uv0[0];
            sf.v = (u * sh.uv1[1] + v * sh.uv2[1]) + w * sh.uv0[1];
        }
        if (e.smooth_shading) {
            nv = mk((u * sh.n1[0] + v * sh.n2[0]) + w * sh.n0[0], (u * sh.n1[1] + v * sh.n2[1]) + w * sh.n0[1],
                    (u * sh.n1[2] + v * sh.n2[2]) + w * sh.n0[2]);
        }
    }
    sf.normal = normalized(nv);
}

/* ------------------------------------------------------------------ RNG: Philox4x32-10 */
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        /* full 64-bit products: one multiply instruction each instead of separate high and low halves */
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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
