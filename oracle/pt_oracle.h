/*
 * pt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library restates, on the CPU and in IEEE
 * fp32, the algorithm of the reference path tracer's hot path (geometry,
 * octree build and traversal, shading, environment lookup, post-processing).
 * It exists to CHECK the HIP product; only tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py may load it.  The product library
 * never links or calls it.
 *
 * Pinning: the reference cannot be built in this image (it needs the CUDA
 * toolkit: runtime API, curand, device intrinsics), so the restatement is
 * pinned by the known answers that the reference's own code produced for the
 * teapot (octree node counts per depth, leaf references, maximum leaf size,
 * hit count of a fixed pinhole sweep, octree == brute force on every ray);
 * everything those answers do not cover is "parity unpinned".
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct {
    float position[3];
    float direction[3];
    float up[3];
    int   res_x, res_y;
    float focal_length;
    float aperture;
    float focus_distance;
} orc_camera;

typedef struct {
    int32_t entity;     /* -1 = miss */
    int32_t triangle;   /* -1 = sphere or miss */
    float   t;          /* hit distance (FLT_MAX on miss) */
    float   bu, bv;     /* raw barycentrics of the accepted triangle test */
    float   u, v;       /* after uv interpolation (== bu,bv when mesh has no uvs) */
    float   position[3];
    float   normal[3];  /* geometric, or interpolated when smooth_shading */
} orc_hit;

typedef struct {
    uint64_t segments;      /* trace() calls */
    uint64_t aabb_tests;    /* entity AABB slab tests */
    uint64_t node_visits;   /* proc_subtree invocations */
    uint64_t tri_tests;     /* intersect_triangle invocations */
    uint64_t env_lookups;   /* sample_hdri calls with an HDRI set */
    uint64_t tex_fetches;   /* texel fetches */
    uint64_t paths;         /* color() calls */
} orc_counters;

orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *s);

int orc_add_material(orc_scene *s, const float albedo[3], const float specular[3],
                     float emission, float smoothness, int smooth_shading);
/* which: 0 albedo (RGB8), 1 specular (RGB8), 2 normal (RGB8), 3 smoothness (G8) */
int orc_material_set_map(orc_scene *s, int material, int which,
                         const uint8_t *data, int w, int h);

/* tri: 6 ints per triangle (idx a,b,c, vt a,b,c); uvs may be NULL (n_uvs = 0).
 * normals may be NULL -> recomputed as the reference does for OBJ files without vn.
 * octree_depth: 0 = no octree (brute force), else the leaf depth (reference: 6). */
int orc_add_mesh(orc_scene *s, const float *positions, const float *normals, int n_vertices,
                 const int32_t *tri, int n_triangles, const float *uvs, int n_uvs,
                 int material, int octree_depth);
int orc_add_sphere(orc_scene *s, const float center[3], float radius, int material);
void orc_set_hdri(orc_scene *s, const float *rgb, int w, int h, float exposure,
                  float contrast, float rot_offset, int smooth);

/* mesh helpers restating the reference's Entity ctor / transforms on plain arrays */
void orc_compute_vertex_normals(const float *positions, int n_vertices, const int32_t *tri,
                                int n_triangles, float *normals_out);
/* op: 0 scale(factor=arg[0]) about center, 1 translate(arg), 2 rotate(arg = euler xyz) about center.
 * center is updated in place (translate moves it). normals may be NULL. */
void orc_transform(float *positions, float *normals, int n_vertices, float center[3],
                   int op, const float arg[3]);
void orc_mesh_aabb(const float *positions, int n_vertices, float min_out[3], float max_out[3]);

/* octree structure queries (entity must be a mesh with an octree) */
int orc_octree_stats(orc_scene *s, int entity, int64_t *nodes_per_depth /*[max_depth+1]*/,
                     int64_t *n_leaf_refs, int64_t *max_leaf);
/* canonical pre-order dump: for each node: depth, slot, child mask, n_tris, tris...  Returns the
 * number of int32 words written (or needed when out == NULL). */
int64_t orc_octree_dump(orc_scene *s, int entity, int32_t *out, int64_t cap);
int orc_aabb_intersects_triangle(const float bmin[3], const float bmax[3], const float v0[3],
                                 const float v1[3], const float v2[3], int pad_box);

/* geometry: rays = n x 6 floats (origin, direction).  log (optional): per ray, log_stride ints:
 * [0] = number of triangle tests, then the tested triangle ids in order (truncated). */
void orc_trace(orc_scene *s, const float *rays, int64_t n, orc_hit *hits, int32_t *log,
               int log_stride, orc_counters *counters);
void orc_camera_rays(const orc_camera *cam, uint32_t seed, int sample, float *rays_out /* W*H*6 */);

/* Philox4x32-10 and the uniform mapping used by both sides */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_u01(uint32_t x);

/* accumulate sum of radiance samples sample_begin..sample_end-1 for pixels of rows
 * [row_begin,row_end) into accum (W*H*3 floats, += semantics on the given rows). */
void orc_render(orc_scene *s, const orc_camera *cam, uint32_t seed, int sample_begin,
                int sample_end, int row_begin, int row_end, float *accum, int n_threads,
                orc_counters *counters);
void orc_sample_hdri(orc_scene *s, const float *dirs, int64_t n, float *rgb_out);
void orc_sample_material(orc_scene *s, int material, const float *uv, int64_t n,
                         float *albedo, float *specular, float *smoothness, float *normal);

/* post-processing */
void orc_make_gaussian(unsigned size, float sigma, float *taps /* (2s+1)^2 */);
void orc_make_mean(unsigned size, float *taps);
void orc_apply_filter(const float *in, float *out, int w, int h, const float *taps, int ksize,
                      int bilateral, int n_threads);
void orc_compound_buffers(const float *in, float *out, int64_t n_pixels, int n_split);
void orc_normalize_image(float *buf, int64_t n_pixels, unsigned n_samples);
void orc_apply_threshold(const float *in, float *out, int64_t n_pixels, float threshold);
void orc_image_add(float *buf, const float *layer, int64_t n_pixels, float opacity);
void orc_apply_aces(float *buf, int64_t n_pixels);
void orc_gamma_correct(float *buf, int64_t n_pixels, float gamma);
void orc_to_ppm_bytes(const float *buf, int64_t n_pixels, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
