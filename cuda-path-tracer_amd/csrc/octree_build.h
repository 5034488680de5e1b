/*
 * octree_build.h -- host-side octree construction and flattening.
 *
 * Produces the tree that the reference's Entity::construct_octree / Octree::insert_triangle(s)
 * build (src/entity.cu:167-170, src/octree.cu:83-142, SAT test src/aabb.cu:64-120, box padding
 * src/geometry.cuh:81-84) -- same child boxes, same triangle-to-leaf assignment, same ascending
 * triangle order inside a leaf -- but level by level over (node, triangle) pairs with worker
 * threads, and emits it directly in the flat breadth-first layout of pt_types.h.
 */
#ifndef PT_OCTREE_BUILD_H
#define PT_OCTREE_BUILD_H

#include <cstdint>
#include <vector>

#include "pt_types.h"

namespace ptamd {

struct FlatOctree {
    int max_depth = 0;
    std::vector<PtNode> nodes;            /* local indices: children / leaf runs relative to this tree */
    std::vector<int32_t> leaf_tris;       /* concatenated leaf triangle lists (entity-local ids) */
    std::vector<int64_t> nodes_per_depth; /* max_depth + 1 entries */
    int64_t max_leaf = 0;
};

/* positions: n_vertices x 3; tri: n_triangles x 6 (a,b,c,vta,vtb,vtc); box = padded entity AABB */
void build_octree(const float *positions, const int32_t *tri, int64_t n_triangles,
                  const float box_min[3], const float box_max[3], int max_depth, FlatOctree &out);

/* true iff the (unpadded as given) box overlaps the triangle under the reference's SAT test */
bool box_intersects_triangle(const float bmin[3], const float bmax[3], const float u0[3],
                             const float u1[3], const float u2[3]);

/* padded AABB of a vertex array (src/aabb.cu:7-22) */
void padded_aabb(const float *positions, int64_t n_vertices, float mn[3], float mx[3]);

/* per-vertex normals as the reference computes them for OBJ files without vn (src/entity.cu:110-133) */
void compute_vertex_normals(const float *positions, int64_t n_vertices, const int32_t *tri,
                            int64_t n_triangles, float *normals_out);

/* pre-order dump {depth, slot, mask, n_tris, tris...} used by structure checks */
void dump_octree(const FlatOctree &t, std::vector<int32_t> &out);

} // namespace ptamd
#endif
