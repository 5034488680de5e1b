/*
 * octree_build.cpp -- see octree_build.h.  Host code; compiled with -ffp-contract=off so that
 * every fp32 operation rounds exactly as the reference's host build does.
 */
#include "octree_build.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>

namespace ptamd {

namespace {

const float kPad = 0.0001f;   /* AABB_PADDING, src/geometry.cuh:7 */

struct Box { float mn[3], mx[3]; };

inline Box padded(float x0, float y0, float z0, float x1, float y1, float z1) {   /* geometry.cuh:81-84 */
    Box b;
    b.mn[0] = x0 - kPad; b.mn[1] = y0 - kPad; b.mn[2] = z0 - kPad;
    b.mx[0] = x1 + kPad; b.mx[1] = y1 + kPad; b.mx[2] = z1 + kPad;
    return b;
}

/* The eight child boxes of src/octree.cu:84-103: slot bit 4 = upper x half, 2 = upper y, 1 = upper z. */
inline void child_boxes(const Box &r, Box c[8]) {
    float xm = 0.5f * (r.mn[0] + r.mx[0]);
    float ym = 0.5f * (r.mn[1] + r.mx[1]);
    float zm = 0.5f * (r.mn[2] + r.mx[2]);
    for (int i = 0; i < 8; i++) {
        float x0 = (i & 4) ? xm : r.mn[0], x1 = (i & 4) ? r.mx[0] : xm;
        float y0 = (i & 2) ? ym : r.mn[1], y1 = (i & 2) ? r.mx[1] : ym;
        float z0 = (i & 1) ? zm : r.mn[2], z1 = (i & 1) ? r.mx[2] : zm;
        c[i] = padded(x0, y0, z0, x1, y1, z1);
    }
}

/* One edge-axis test of src/aabb.cu:87-99 for axis cross(e_k, f): with e_k a unit axis the cross
 * product has a zero component and the two others are +-f components, so the three projections
 * reduce to one subtraction of two products each -- the same roundings as the general formula
 * (adding a signed zero and negating are exact). */
inline bool edge_axis_separates(float pa0, float pb0, float pa1, float pb1, float pa2, float pb2,
                                float fa, float fb, float ea, float eb) {
    /* p_i = pb_i * fa - pa_i * fb ; r = ea * |fb| + eb * |fa| */
    float p0 = pb0 * fa - pa0 * fb;
    float p1 = pb1 * fa - pa1 * fb;
    float p2 = pb2 * fa - pa2 * fb;
    float r = ea * std::fabs(fb) + eb * std::fabs(fa);
    float p_min = std::fmin(p0, std::fmin(p1, p2));
    float p_max = std::fmax(p0, std::fmax(p1, p2));
    return std::fmax(-p_max, p_min) > r;
}

bool sat_box_triangle(const Box &b, const float *u0, const float *u1, const float *u2) {
    float c[3], e[3], v0[3], v1[3], v2[3];
    for (int k = 0; k < 3; k++) {
        c[k] = 0.5f * (b.mn[k] + b.mx[k]);
        e[k] = (b.mx[k] - b.mn[k]) / 2.0f;
        v0[k] = u0[k] - c[k];
        v1[k] = u1[k] - c[k];
        v2[k] = u2[k] - c[k];
    }
    /* the three box axes (src/aabb.cu:103-111) first: cheapest rejection, order does not matter */
    for (int k = 0; k < 3; k++) {
        float mn = std::fmin(v0[k], std::fmin(v1[k], v2[k]));
        float mx = std::fmax(v0[k], std::fmax(v1[k], v2[k]));
        if (mn > e[k] || mx < -e[k]) return false;
    }
    float f[3][3];
    for (int k = 0; k < 3; k++) {
        f[0][k] = v1[k] - v0[k];
        f[1][k] = v2[k] - v1[k];
        f[2][k] = v0[k] - v2[k];
    }
    for (int j = 0; j < 3; j++) {
        const float fx = f[j][0], fy = f[j][1], fz = f[j][2];
        /* e_x x f = (0, -fz, fy):  p = v.z*fy - v.y*fz ; r = ey*|fz| + ez*|fy| */
        if (edge_axis_separates(v0[1], v0[2], v1[1], v1[2], v2[1], v2[2], fy, fz, e[1], e[2])) return false;
        /* e_y x f = (fz, 0, -fx):  p = v.x*fz - v.z*fx ; r = ex*|fz| + ez*|fx| */
        if (edge_axis_separates(v0[2], v0[0], v1[2], v1[0], v2[2], v2[0], fz, fx, e[2], e[0])) return false;
        /* e_z x f = (-fy, fx, 0):  p = v.y*fx - v.x*fy ; r = ex*|fy| + ey*|fx| */
        if (edge_axis_separates(v0[0], v0[1], v1[0], v1[1], v2[0], v2[1], fx, fy, e[0], e[1])) return false;
    }
    return true;
}

void parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int n_threads = (int)std::min<int64_t>(hw ? std::min(hw, 16u) : 1, (n + grain - 1) / std::max<int64_t>(grain, 1));
    if (n_threads <= 1) { fn(0, n); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                int64_t b = next.fetch_add(grain);
                if (b >= n) break;
                fn(b, std::min(n, b + grain));
            }
        });
    for (auto &t : th) t.join();
}

} // namespace

bool box_intersects_triangle(const float bmin[3], const float bmax[3], const float u0[3],
                             const float u1[3], const float u2[3]) {
    Box b;
    for (int k = 0; k < 3; k++) { b.mn[k] = bmin[k]; b.mx[k] = bmax[k]; }
    return sat_box_triangle(b, u0, u1, u2);
}

void padded_aabb(const float *positions, int64_t n_vertices, float mn[3], float mx[3]) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t i = 0; i < n_vertices; i++)
        for (int k = 0; k < 3; k++) {
            float p = positions[3 * i + k];
            lo[k] = (p < lo[k]) ? p : lo[k];
            hi[k] = (p > hi[k]) ? p : hi[k];
        }
    for (int k = 0; k < 3; k++) { mn[k] = lo[k] - kPad; mx[k] = hi[k] + kPad; }
}

void compute_vertex_normals(const float *positions, int64_t n_vertices, const int32_t *tri,
                            int64_t n_triangles, float *normals_out) {
    std::fill(normals_out, normals_out + 3 * n_vertices, 0.0f);
    for (int64_t f = 0; f < n_triangles; f++) {
        const int32_t ia = tri[6 * f], ib = tri[6 * f + 1], ic = tri[6 * f + 2];
        const float *v0 = positions + 3 * ia, *v1 = positions + 3 * ib, *v2 = positions + 3 * ic;
        float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
        float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
        float n[3] = {(e1[1] * e2[2]) - (e1[2] * e2[1]), (e1[2] * e2[0]) - (e1[0] * e2[2]),
                      (e1[0] * e2[1]) - (e1[1] * e2[0])};
        float rn = 1.0f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        float tn[3] = {n[0] * rn, n[1] * rn, n[2] * rn};
        const int32_t idx[3] = {ia, ib, ic};
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++)
                normals_out[3 * idx[c] + k] = normals_out[3 * idx[c] + k] + tn[k];
    }
    for (int64_t i = 0; i < n_vertices; i++) {
        float *n = normals_out + 3 * i;
        float rn = 1.0f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] = n[0] * rn; n[1] = n[1] * rn; n[2] = n[2] * rn;
    }
}

void build_octree(const float *positions, const int32_t *tri, int64_t n_triangles,
                  const float box_min[3], const float box_max[3], int max_depth, FlatOctree &out) {
    out = FlatOctree();
    out.max_depth = max_depth;
    out.nodes_per_depth.assign(max_depth + 1, 0);

    /* current level: nodes (regions) and their (node, triangle) pairs, pairs grouped by node in
     * ascending triangle order */
    std::vector<Box> regions(1);
    for (int k = 0; k < 3; k++) { regions[0].mn[k] = box_min[k]; regions[0].mx[k] = box_max[k]; }
    std::vector<int64_t> pair_begin = {0, n_triangles};   /* per node: [begin,end) into pair_tri */
    std::vector<int32_t> pair_tri(n_triangles);
    for (int64_t i = 0; i < n_triangles; i++) pair_tri[i] = (int32_t)i;

    out.nodes.clear();
    for (int depth = 0; depth < max_depth; depth++) {
        const int64_t n_nodes = (int64_t)regions.size();
        const int64_t n_pairs = (int64_t)pair_tri.size();
        out.nodes_per_depth[depth] = n_nodes;

        std::vector<Box> cboxes((size_t)n_nodes * 8);
        parallel_for(n_nodes, 4096, [&](int64_t b, int64_t e) {
            for (int64_t n = b; n < e; n++) child_boxes(regions[n], &cboxes[(size_t)n * 8]);
        });
        /* node of each pair */
        std::vector<int32_t> pair_node(n_pairs);
        parallel_for(n_nodes, 4096, [&](int64_t b, int64_t e) {
            for (int64_t n = b; n < e; n++)
                for (int64_t p = pair_begin[n]; p < pair_begin[n + 1]; p++) pair_node[p] = (int32_t)n;
        });
        /* 8-bit child overlap mask per pair */
        std::vector<uint8_t> mask(n_pairs);
        parallel_for(n_pairs, 8192, [&](int64_t b, int64_t e) {
            for (int64_t p = b; p < e; p++) {
                const int32_t *t = tri + 6 * (int64_t)pair_tri[p];
                const float *u0 = positions + 3 * (int64_t)t[0], *u1 = positions + 3 * (int64_t)t[1],
                            *u2 = positions + 3 * (int64_t)t[2];
                const Box *c = &cboxes[(size_t)pair_node[p] * 8];
                unsigned m = 0;
                for (int i = 0; i < 8; i++)
                    if (sat_box_triangle(c[i], u0, u1, u2)) m |= 1u << i;
                mask[p] = (uint8_t)m;
            }
        });
        /* counts per (node, slot) */
        std::vector<int64_t> cnt((size_t)n_nodes * 8 + 1, 0);
        parallel_for(n_nodes, 1024, [&](int64_t b, int64_t e) {
            for (int64_t n = b; n < e; n++)
                for (int64_t p = pair_begin[n]; p < pair_begin[n + 1]; p++) {
                    unsigned m = mask[p];
                    for (int i = 0; i < 8; i++) cnt[(size_t)n * 8 + i] += (m >> i) & 1;
                }
        });
        /* children in (node, slot) order -> breadth-first, contiguous per parent */
        std::vector<int64_t> child_of((size_t)n_nodes * 8, -1);
        std::vector<int64_t> next_begin = {0};
        std::vector<Box> next_regions;
        const int64_t level_base = (int64_t)out.nodes.size();
        const int64_t next_base = level_base + n_nodes;
        for (int64_t n = 0; n < n_nodes; n++) {
            unsigned m = 0;
            int64_t first_child = (int64_t)next_regions.size();
            for (int i = 0; i < 8; i++) {
                int64_t c = cnt[(size_t)n * 8 + i];
                if (c > 0) {
                    m |= 1u << i;
                    child_of[(size_t)n * 8 + i] = (int64_t)next_regions.size();
                    next_regions.push_back(cboxes[(size_t)n * 8 + i]);
                    next_begin.push_back(next_begin.back() + c);
                }
            }
            PtNode nd;
            nd.a = (uint32_t)(next_base + first_child);
            nd.b = m;
            out.nodes.push_back(nd);
        }
        /* scatter pairs to the children, preserving ascending triangle order */
        std::vector<int32_t> next_tri((size_t)next_begin.back());
        parallel_for(n_nodes, 1024, [&](int64_t b, int64_t e) {
            for (int64_t n = b; n < e; n++) {
                int64_t cursor[8];
                for (int i = 0; i < 8; i++) {
                    int64_t c = child_of[(size_t)n * 8 + i];
                    cursor[i] = c >= 0 ? next_begin[c] : -1;
                }
                for (int64_t p = pair_begin[n]; p < pair_begin[n + 1]; p++) {
                    unsigned m = mask[p];
                    for (int i = 0; i < 8; i++)
                        if ((m >> i) & 1) next_tri[(size_t)cursor[i]++] = pair_tri[p];
                }
            }
        });
        regions.swap(next_regions);
        pair_begin.swap(next_begin);
        pair_tri.swap(next_tri);
    }
    /* leaf level */
    const int64_t n_leaves = (int64_t)regions.size();
    out.nodes_per_depth[max_depth] = n_leaves;
    for (int64_t n = 0; n < n_leaves; n++) {
        PtNode nd;
        nd.a = (uint32_t)pair_begin[n];
        nd.b = (uint32_t)(pair_begin[n + 1] - pair_begin[n]);
        out.max_leaf = std::max<int64_t>(out.max_leaf, nd.b);
        out.nodes.push_back(nd);
    }
    out.leaf_tris.swap(pair_tri);
}

namespace {
void dump_rec(const FlatOctree &t, int64_t node, int depth, int slot, std::vector<int32_t> &out) {
    const PtNode &nd = t.nodes[node];
    if (depth == t.max_depth) {
        out.push_back(depth); out.push_back(slot); out.push_back(0); out.push_back((int32_t)nd.b);
        for (uint32_t i = 0; i < nd.b; i++) out.push_back(t.leaf_tris[nd.a + i]);
        return;
    }
    out.push_back(depth); out.push_back(slot); out.push_back((int32_t)nd.b); out.push_back(0);
    int64_t c = nd.a;
    for (int i = 0; i < 8; i++)
        if ((nd.b >> i) & 1) dump_rec(t, c++, depth + 1, i, out);
}
} // namespace

void dump_octree(const FlatOctree &t, std::vector<int32_t> &out) {
    out.clear();
    if (!t.nodes.empty()) dump_rec(t, 0, 0, 0, out);
}

} // namespace ptamd
