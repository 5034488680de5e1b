/*
 * pt_types.h -- device-visible data layout shared by the host scene builder and the HIP kernels.
 *
 * Layout in HBM (everything is a flat array; entities index into them by base offsets):
 *   nodes[]   8 B per octree node, breadth-first per entity, the (up to 8) children of a node
 *             stored contiguously in slot order.  inner: {first child index, child mask};
 *             leaf (depth == entity's octree depth): {first triangle record, record count}.
 *   tris[]    pre-gathered triangle records {v0, e1 = v1-v0, e2 = v2-v0, id}, 40 B per triangle,
 *             stored as 80-byte pairs (PtTriPair); one run per leaf (a triangle is repeated in
 *             every leaf it overlaps, ascending id inside a leaf), or one run in index order for a
 *             mesh without octree.  Five 16-byte loads per two tests instead of the reference's
 *             index -> Triangle -> 3 x Vertex pointer chase
 *             (src/device_geometry_functions.cuh:148-154).
 *   shade[]   64 B per unique triangle {3 vertex normals, 3 uvs}: read once per accepted hit.
 */
#ifndef PT_TYPES_H
#define PT_TYPES_H

#include <stdint.h>

#define PT_SHAPE_SPHERE 0
#define PT_SHAPE_MESH 1

struct PtNode {
    uint32_t a;   /* inner: index of first child;   leaf: first triangle pair, in 16-byte units */
    uint32_t b;   /* inner: 8-bit child mask;       leaf: number of triangles */
};

/* Two triangles side by side, component-interleaved: f[2*k + j] is component k of triangle j
 * (k = 0..8: v0.xyz, e1.xyz, e2.xyz), so that one 16-byte load yields operand pairs for the packed
 * two-wide arithmetic of the triangle test.  A run of n triangles (one octree leaf, or a whole mesh
 * without octree) occupies ceil(n/2) consecutive pairs; the second half of an odd run's last pair
 * repeats the first and is never tested. */
struct alignas(16) PtTriPair {
    float   f[18];
    int32_t id[2];             /* triangle index inside its entity */
};

struct alignas(16) PtShadeRec {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad;
};

struct alignas(16) PtEntity {
    int32_t  shape;
    int32_t  material;
    int32_t  octree_depth;     /* 0 = brute force */
    uint32_t root_node;        /* index into nodes[] */
    float    aabb_min[3];
    uint32_t tri_first;        /* brute force: first triangle pair, in 16-byte units */
    float    aabb_max[3];
    uint32_t n_tris;           /* brute force: number of triangles */
    float    region_min[3];    /* octree root region (the entity AABB when the octree was built) */
    uint32_t shade_base;
    float    region_max[3];
    int32_t  has_uvs;
    float    center[3];
    float    radius;
    int32_t  smooth_shading;
    uint32_t root_a, root_b;   /* copy of nodes[root_node]: entering the octree costs no fetch that depends on another */
    int32_t  pad;
};

/* Texel records of a material whose maps all have one resolution: the 8-bit texels of the maps side by side, so that
 * one load serves the three or four look-ups a hit makes at the same (u, v), in tiles of 4 x 4 texels (a tile of
 * 8-byte records is one 128-byte line: hits that are neighbours on the surface share lines in both directions).
 *   bytes 0..2 albedo rgb, 3..5 normal rgb, 6 smoothness, 7 unused;  16-byte records (a specular map is set): 8..10
 *   specular rgb, 11..15 unused.  Bytes of maps that are not set are zero and never read.
 * record index of texel (x, y) = ((y / 4) * tiles_x + x / 4) * 16 + (y % 4) * 4 + x % 4 */
#define PT_MAP_ALBEDO 0
#define PT_MAP_SPECULAR 1
#define PT_MAP_NORMAL 2
#define PT_MAP_SMOOTHNESS 3

struct PtMaterial {
    float albedo[3];
    float emission;
    float specular[3];
    float smoothness;
    int32_t smooth_shading;
    uint32_t packed_mask;      /* packed != null: bit k = map k is set */
    const uint8_t *map[4];     /* separate row-major maps (maps of different resolutions): albedo, specular, normal (RGB8), smoothness (G8);
                                * null = unset, or in `packed` */
    int32_t map_w[4];
    int32_t map_h[4];
    const uint8_t *packed;     /* texel records of all set maps (see above), or null */
    uint32_t packed_tiles_x;
    uint32_t packed_shift;     /* log2 of the record size: 3 or 4 */
    float packed_w_f, packed_h_f, packed_xmax_f, packed_ymax_f;   /* (float)w, (float)h, (float)(w - 1), (float)(h - 1) */
};

struct PtSceneDev {
    const PtEntity   *entities;
    const PtNode     *nodes;
    const PtTriPair  *tris;
    const PtShadeRec *shade;
    const PtMaterial *materials;
    const float      *hdri;    /* h x w x 3, row 0 first; null when hdri_quads is set */
    const float      *hdri_quads;   /* bilinear look-ups: per texel (x, y) one 64-byte record holding the four texels the look-up
                                     * at that texel blends -- (x, y), (x+1, y), (x, y+1), (x+1, y+1), indices wrapping -- as 12
                                     * floats + 4 unused: one aligned line fetch per look-up instead of two to four */
    int32_t n_entities;
    int32_t hdri_w, hdri_h;
    float   hdri_w_f, hdri_h_f, hdri_xmax_f, hdri_ymax_f;   /* (float)w, (float)h, (float)(w - 1), (float)(h - 1): made once on the
                                                             * host; converted in the kernel they are loop invariants that get
                                                             * hoisted out of its work loop and spilled */
    int32_t has_hdri, use_smooth_hdri;
    float   hdri_exposure, hdri_contrast, hdri_rot_offset;
    int32_t max_depth;         /* deepest octree in the scene: sizes the per-lane stack */
};

#endif
