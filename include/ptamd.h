/*
 * ptamd.h -- C ABI of the MI355X path-tracing library (libptamd.so).
 *
 * The reference (henrikglass/cuda-path-tracer) has no FFI layer: its boundary is the host C++
 * class API that src/main.cu uses, and every class talks to the GPU itself (cudaMalloc /
 * cudaMemcpy / <<<>>> inside copy_to_device() and Renderer::render).  This header is the seam
 * that replaces those calls: the C++ facade in cuda-path-tracer_amd/host/ (same class names and
 * signatures as the reference) and any other binding (ctypes, cgo, JNI ...) call ONLY these
 * functions.  Plain pointers and sizes; no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; pt_last_error() gives the message
 *     (thread-local).  The C++ facade turns a non-zero status into the reference's
 *     print-and-exit behaviour (src/util.cuh:7-13).
 *   - "device pointer" arguments are HIP device addresses on the scene's (or the given) device;
 *     entry points that say "host or device" accept either and stage host memory themselves.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work
 *     and return; results are complete after the stream is synchronised, except where a function
 *     writes to a HOST pointer, in which case it synchronises the stream itself before returning.
 *   - vec3 images are float[H][W][3] (12 bytes per pixel), the layout of the reference's
 *     std::vector<vec3> image buffers.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_MAX_OCTREE_DEPTH 10   /* reference: fixed 6 (src/geometry.cuh:4); 3 path bits per level in 32 */

typedef struct pt_scene pt_scene;

/* src/camera.h:6-28 */
typedef struct {
    float position[3];
    float direction[3];
    float up[3];
    int   res_x, res_y;
    float focal_length;
    float aperture;
    float focus_distance;
} pt_camera;

/* src/material.cuh:46-58; maps are host pointers to 8-bit texels (RGB interleaved for the three
 * colour maps, one channel for smoothness), NULL = not set (src/texture.cuh:12). */
typedef struct {
    float albedo[3];
    float specular[3];
    float emission;
    float smoothness;
    int   smooth_shading;
    const uint8_t *albedo_map;     int albedo_w, albedo_h;
    const uint8_t *specular_map;   int specular_w, specular_h;
    const uint8_t *normal_map;     int normal_w, normal_h;
    const uint8_t *smoothness_map; int smoothness_w, smoothness_h;
} pt_material_desc;

/* result of one closest-hit query; replaces the reference's Intersection (src/geometry.cuh:67-77) */
typedef struct {
    int32_t entity;      /* index in insertion order, -1 = miss */
    int32_t triangle;    /* triangle index inside the entity, -1 = sphere or miss */
    float   t;           /* FLT_MAX on miss */
    float   bu, bv;      /* raw barycentrics of the accepted triangle test */
    float   u, v;        /* after uv interpolation (== bu,bv when the mesh has no uvs) */
    float   position[3];
    float   normal[3];   /* geometric, or interpolated vertex normal when smooth_shading */
} pt_hit;

/* work counters, summed over the launch */
typedef struct {
    uint64_t segments;     /* trace calls (ray segments) */
    uint64_t aabb_tests;   /* entity AABB slab tests */
    uint64_t node_visits;  /* octree nodes entered */
    uint64_t tri_tests;    /* ray/triangle tests */
    uint64_t env_lookups;  /* HDRI lookups */
    uint64_t tex_fetches;  /* texel fetches */
    uint64_t paths;        /* radiance samples */
} pt_counters;

/* flattened octree of one mesh entity, read back for structure checks */
typedef struct {
    int32_t  max_depth;
    int64_t  n_nodes;
    int64_t  n_leaf_refs;
    int64_t  max_leaf;
    int64_t  nodes_per_depth[PT_MAX_OCTREE_DEPTH + 1];
} pt_octree_info;

const char *pt_last_error(void);
/* replaces cudaGetDevice/cudaGetDeviceProperties in Renderer::render (src/renderer.cu:209-216) */
int pt_device_count(int *count);
int pt_device_name(int device, char *buf, size_t cap);

/* ---- scene: replaces Scene/Entity/Material/Texture/Octree::copy_to_device + free_from_device
 *      (src/scene.cu:67-110, src/entity.cu:263-337, src/octree.cu:24-61, src/material.cu:32-60,
 *      src/texture.cu:19-49, external/hdrloader.h:30-41). */
/* device < 0 creates a host-only scene: meshes and octrees can be built and inspected, commit fails */
int pt_scene_create(int device, pt_scene **out);
int pt_scene_destroy(pt_scene *scene);
int pt_scene_add_material(pt_scene *scene, const pt_material_desc *desc, int *material_out);
/* positions/normals: n_vertices x 3 floats (normals NULL -> area-unweighted face-normal average,
 * src/entity.cu:110-133); tri: n_triangles x 6 ints (vertex a,b,c, uv a,b,c; uv indices ignored
 * when n_uvs == 0); octree_depth 0 = no octree (brute force, src/device_geometry_functions.cuh:
 * 286-291), otherwise the leaf depth of the octree built as Entity::construct_octree does
 * (src/entity.cu:167-170, src/octree.cu:83-142; the reference uses 6). */
int pt_scene_add_mesh(pt_scene *scene, const float *positions, const float *normals, int n_vertices,
                      const int32_t *tri, int n_triangles, const float *uvs, int n_uvs,
                      int material, int octree_depth, int *entity_out);
int pt_scene_add_sphere(pt_scene *scene, const float center[3], float radius, int material,
                        int *entity_out);
/* rgb: h x w x 3 floats, row 0 first (src/scene.cu:19-55) */
int pt_scene_set_hdri(pt_scene *scene, const float *rgb, int w, int h, float exposure,
                      float contrast, float rot_offset, int smooth);
/* flattens every octree into the linear node / triangle-pair arrays and uploads the scene.  Also sets
 * the defaults of the two launch choices of the render kernel (see pt_scene_set_launch): the build --
 * heavy for scenes with a large octree, light otherwise -- and the quorum (62 without any octree; heavy
 * build: 48 with textured materials or an octree of depth 8 and more, else 40; light build: 40 / 12); the
 * environment variables PTAMD_KERNEL = heavy | light, PTAMD_QUORUM = 1..64 and PTAMD_LEAF_QUOTA = 1..65 (heavy
 * build: triangle runs waiting in a wave at which it tests them, default 12) fix them.  Limits: 2^29 octree nodes and
 * 4 GiB of triangle pairs per scene. */
int pt_scene_commit(pt_scene *scene);
/* ---- one scene, several GPUs (no counterpart in the reference, which builds and uploads one scene in one process for
 * device 0: src/main.cu:22-46, src/renderer.cu:201).  A scene's octrees are built when its meshes are added
 * (Entity::construct_octree, src/entity.cu:167-170); these entry points hand a built scene on so that a node builds it once:
 * pt_scene_clone: a new uncommitted scene for `device` (or a host-only one, device < 0) that SHARES the source's host-side
 *   description -- materials, maps, meshes, built octrees, environment map -- without copying or rebuilding anything, and
 *   takes over its launch choices; commit it to upload.  For one process driving several GPUs.  Either scene may be
 *   destroyed first.
 * pt_scene_save / pt_scene_load: the same description, octrees included, to and from a file (native byte order; put it
 *   on a RAM-backed file system such as /dev/shm): one process per GPU, the first builds and saves, the others load.  A
 *   loaded scene is uncommitted; it equals the saved one node for node (pt_scene_octree_dump).
 * pt_octree_builds: number of octrees this process has built since the library was loaded. */
int pt_scene_clone(pt_scene *src, int device, pt_scene **out);
int pt_scene_save(pt_scene *scene, const char *path);
int pt_scene_load(int device, const char *path, pt_scene **out);
int pt_octree_builds(uint64_t *count);
int pt_scene_octree_info(pt_scene *scene, int entity, pt_octree_info *info);
/* canonical pre-order dump of the flattened octree: per node {depth, slot, child mask, n_tris,
 * tris...}; returns the number of int32 words needed in *n_words (out may be NULL). */
int pt_scene_octree_dump(pt_scene *scene, int entity, int32_t *out, int64_t cap, int64_t *n_words);
/* bytes of device memory the committed scene occupies, by array */
int pt_scene_device_bytes(pt_scene *scene, uint64_t *nodes, uint64_t *tri_records,
                          uint64_t *shade_records, uint64_t *textures, uint64_t *hdri);

/* ---- geometry: the closest-hit query of trace() (src/device_geometry_functions.cuh:201-234).
 * rays: n x 6 floats (origin, direction), host or device.  hits: n records, host or device.
 * log (optional, host or device): n x log_stride ints, [0] = number of triangle tests of that ray,
 * then the tested triangle ids in order (truncated to log_stride-1).  counters: host, optional. */
int pt_trace_rays(pt_scene *scene, const float *rays, int64_t n, pt_hit *hits, int32_t *log,
                  int log_stride, pt_counters *counters, void *stream);
/* primary rays of sample `sample` for every pixel (src/renderer.cu:60-87), n = W*H rays x 6
 * floats to a host or device buffer; for parity tests of the ray generator. */
int pt_camera_rays(int device, const pt_camera *camera, uint32_t seed, int sample, float *rays,
                   void *stream);

/* ---- parity queries of single hot-path functions (for tooling and tests; host pointers, blocking).
 * pt_sample_hdri: Scene::sample_hdri (src/scene.cuh:36-82) of the committed scene for n directions
 * (n x 3 floats) -> n x 3 floats, exactly the lookup the render kernel performs for escaped paths.
 * pt_sample_material: Material::sample_albedo / sample_specular / sample_smoothness / sample_normal
 * (src/material.cuh:21-44, src/texture.cuh:22-43) of one material for n (u, v) pairs; texel indices
 * are clamped into the texture where the reference reads past it (u = 1 or v = 0).
 * pt_box_intersects_triangle: AABB::intersects_triangle (src/aabb.cu:64-120), the SAT test the octree
 * builder inserts triangles with; pad_box != 0 pads the box by AABB_PADDING first like the AABB(min, max)
 * constructor (src/geometry.cuh:81-84).  Host only. */
int pt_sample_hdri(pt_scene *scene, const float *dirs, int64_t n, float *rgb_out);
int pt_sample_material(pt_scene *scene, int material, const float *uv, int64_t n, float *albedo,
                       float *specular, float *smoothness, float *normal);
int pt_box_intersects_triangle(const float bmin[3], const float bmax[3], const float v0[3],
                               const float v1[3], const float v2[3], int pad_box, int *overlaps);
/* pt_child_tables: the two look-up tables of the octree walk's node step as the kernels use them, for tests.
 * set_out (1024 bytes): the children -- mirrored numbering, bit c = child c -- that the child loop of
 * Octree::proc_subtree (src/device_geometry_functions.cuh:159-188, with find_first_node :95-110 and next_node
 * :116-126) steps through, indexed by ten comparisons of the node's interval parameters, most significant bit
 * first: tMx < te, tMy < te, tMz < te (te = max(t0)), tMx < min(t1y, t1z), t1x < tMy, tMy < t1z,
 * min(t1x, t1y) < tMz, tMx < tMy, tMx < tMz, tMy < tMz.  mir_out (2048 bytes): [a << 8 | mask] = child mask in the
 * mirrored numbering of a ray with mirror bits a (:165).  Host only; either pointer may be NULL. */
int pt_child_tables(uint8_t *set_out, uint8_t *mir_out);

/* ---- render: replaces render_init + device_render (src/renderer.cu:19-51, 233-257).
 * Adds, for every pixel, the radiance samples sample_begin..sample_end-1 into n_split split
 * buffers: d_split is a DEVICE pointer to n_split x H x W x 3 floats (src/renderer.cu:194-197);
 * split buffer z receives the samples of the z-th of n_split equal sub-ranges.  += semantics: the
 * caller zeroes the buffers before the first call.  Sample s of pixel p draws from the Philox
 * stream (seed; p, s), so the image does not depend on how samples are sharded over calls/GPUs
 * beyond floating-point summation order.  counters: host pointer, optional (forces a stream sync).
 * A scene owns ONE render workspace (slice partials, queue cursors, counter block, profiling events)
 * shared by pt_render_accumulate and pt_trace_rays: at most one such call per scene may be in flight;
 * calls on the same stream serialise by themselves, calls on different streams must be ordered by the
 * caller (events), or use one scene object per stream. */
int pt_render_accumulate(pt_scene *scene, const pt_camera *camera, uint32_t seed,
                         int sample_begin, int sample_end, int n_split, float *d_split,
                         pt_counters *counters, void *stream);
/* Slices of samples a pt_render_accumulate call with these arguments cuts its sample range into (its work items are
 * 8x8-pixel tile x slice; its workspace and pt_render_accumulate_noalloc's scratch hold one W x H x 12-byte partial image
 * per slice): n_split x min(64 / n_split, ceil(ceil(samples / n_split) / L)), L = 8 samples per slice, or 4 for a launch
 * of fewer than 32768 work items.  No counterpart in the reference, whose slicing is fixed (8 x 4, src/renderer.cu:240-241). */
int pt_render_slices(const pt_camera *camera, int sample_begin, int sample_end, int n_split, int *slices);
/* pt_render_accumulate for one of tile_ranks interleaved shares of the image's 8x8-pixel tiles (share tile_rank takes every
 * tile_ranks-th tile of the tile sequence, so all shares see the same mix of scene content); pixels of other tiles are
 * left unchanged.  The reference's own grid has this tile axis (src/renderer.cu:219-223); it is the axis to shard over
 * GPUs when a render has too few samples per pixel to give every GPU a worthwhile slice of them.  Every sample of a pixel
 * is rendered by one share, with the summation order of the unsharded call: the shares' images add up to its bits. */
int pt_render_accumulate_tiles(pt_scene *scene, const pt_camera *camera, uint32_t seed,
                               int sample_begin, int sample_end, int n_split, float *d_split,
                               int tile_rank, int tile_ranks, pt_counters *counters, void *stream);
/* Variant of pt_render_accumulate that allocates nothing.  It has no counterpart in the reference; it is
 * for callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation.  Same arguments, same kernels and the same split-buffer bits as ONE
 * pt_render_accumulate call with these arguments (without counters), but the render kernel's work-queue
 * cursors and slice partials live in a fixed array of PT_NOALLOC_SCRATCH_BYTES in the library's device
 * image instead of a workspace grown with hipMalloc.  Loading the library costs that much memory on every
 * GPU it runs on, until the process ends, whether or not this entry point is called.  The plan of a call
 * has pt_render_slices() slices of samples, each with a partial image of W x H x 12 bytes.  When they fit pt_noalloc_max_partial_bytes() together (640x360: 24
 * slices) the call is one launch; otherwise it renders as many slices per launch as fit beside one image
 * that carries the running sum of the current split buffer (1920x1080: one slice per launch), adding them
 * up in pt_render_accumulate's order.  It fails, with a clear error and nothing written, only when two
 * images do not fit (frames of more than 2.79 M pixels).  No host or device allocation on the successful
 * path; all work is enqueued on `stream` as kernel launches (legal under stream capture; the cursors are
 * zeroed by a kernel: a captured memset in front of the first launch did not take effect on replay).  All calls on one GPU share the array, whatever their scene or stream: calls that overlap in
 * time on one GPU race and produce wrong images -- serialise them (one stream, or events). */
#define PT_NOALLOC_SCRATCH_BYTES (64u << 20)
int pt_render_accumulate_noalloc(pt_scene *scene, const pt_camera *camera, uint32_t seed,
                                 int sample_begin, int sample_end, int n_split, float *d_split,
                                 void *stream);
size_t pt_noalloc_max_partial_bytes(void);
/* Launch choices (no counterpart in the reference).  The render kernel has two launch choices that
 * change how a wave schedules its lanes and nothing a path computes (images are bit-identical under
 * all of them): the kernel build (0 = light: 4 waves per SIMD, every lane tests the triangle runs of
 * its own ray; 1 = heavy: 4 waves, or 3 where deep octree stacks fill the LDS, and the triangle runs waiting in a wave are
 * tested by all of its 64 lanes together) and the quorum (how many of a wave's 64 lanes must have finished their ray before the wave leaves
 * the geometry blocks to shade and refill them: high where shading dominates, low inside deep
 * trees).  pt_scene_commit picks defaults from the scene; pt_scene_set_launch fixes a choice for
 * this and later commits (kernel_build -1 / quorum 0 = leave as is); pt_scene_get_launch reports
 * the current ones (outputs may be NULL). */
int pt_scene_set_launch(pt_scene *scene, int kernel_build, int quorum);
int pt_scene_get_launch(pt_scene *scene, int *kernel_build, int *quorum);
/* 256-thread workgroups of the render kernel resident per CU for the committed scene = waves per SIMD =
 * the register budget the kernel was instantiated for: both builds run 4 (128 registers); the heavy build 3 (168)
 * when the LDS stacks of a deep octree (depth 8 and more) leave room for no more. */
int pt_scene_render_residency(pt_scene *scene, int *workgroups_per_cu);
/* Measurement aid (no counterpart in the reference): when enabled, every pt_render_accumulate
 * brackets its path-tracing kernel with two HIP events on the launch stream;
 * pt_scene_last_render_kernel_ms waits for the last launch and returns its duration. */
int pt_scene_set_profiling(pt_scene *scene, int enable);
/* Scheduling diagnostics of the last pt_render_accumulate that asked for counters: lanes that took
 * part in {entity, node, leaf, shading} block trips, then the number of wave-level trips of each. */
int pt_scene_schedule_counters(pt_scene *scene, uint64_t out[8]);
/* Where the waves of that launch spent their life time (no counterpart in the reference), in shader-clock cycles summed
 * over all waves: inside the {entity, node, leaf} blocks, inside the geometry scheduler as a whole (the three blocks plus the
 * lane counts that pick the next one), and from a wave's first instruction to its last.  The rest of a wave's life is the
 * shade / refill round.  The clock keeps running while other waves of the SIMD issue: these are shares of wall time,
 * not issue cycles, and the counting build of the kernel is slower than the one that is timed.  out[5..7]: lanes that sat out
 * a node trip, summed over all trips, because they had finished their ray and waited for the shading quorum, because they
 * waited at a leaf for its triangle run, or for another reason (next entity due, no work left).  out[8..10]: ticks of the chip-wide 100 MHz clock by which a wave
 * starts after the first wave on average, by which it ends before the last wave on average (the launch's tail), and from the first
 * wave's start to the last wave's end. */
int pt_scene_wave_profile(pt_scene *scene, uint64_t out[11]);
int pt_scene_last_render_kernel_ms(pt_scene *scene, float *ms);
/* d_out[p] = sum_z d_split[z][p]  (compound_buffers, src/post_processing.cu:243-256); device ptrs */
int pt_compound_buffers(int device, const float *d_split, float *d_out, int64_t n_pixels,
                        int n_split, void *stream);

/* ---- multi-GPU inside one process (no counterpart in the reference, which renders on device 0 only,
 * src/renderer.cu:209-216): d_dst (on dst_device) += d_src[0] + d_src[1] + ... in THAT order, each n_floats
 * floats on device src_devices[i].  Sources reach dst_device by peer copies into a staging buffer (peer access is
 * enabled where the two GPUs allow it, so that the copy uses the links between them; otherwise the runtime stages it
 * through host memory) and are added one after the other, so the result does not depend on timing.  Blocking: all
 * work on the source buffers must have completed before the call; the sum is complete when it returns.
 * One process per GPU setups use a collective instead (torch.distributed / RCCL reduce, see bench.py). */
int pt_reduce_sum(int dst_device, float *d_dst, int n_src, const int *src_devices,
                  const float *const *d_src, int64_t n_floats);

/* ---- post-processing: replaces apply_filter/device_apply_filter and the CPU loops
 *      (src/post_processing.cu:105-187, 194-278).  All image pointers are DEVICE pointers. */
/* taps: (2*ksize+1)^2 floats, host pointer (Kernel::make_gaussian / make_mean tables);
 * filter_type 0 = NORMAL, 1 = BILATERAL (src/post_processing.cuh:26-29) */
int pt_apply_filter(int device, const float *d_in, float *d_out, int w, int h, const float *taps,
                    int ksize, int filter_type, void *stream);
int pt_make_gaussian(unsigned size, float sigma, float *taps_out);   /* host, src/post_processing.cu:51-68 */
int pt_make_mean(unsigned size, float *taps_out);                    /* host, src/post_processing.cu:73-84 */
int pt_normalize_image(int device, float *d_buf, int64_t n_pixels, unsigned n_samples, void *stream);
int pt_apply_threshold(int device, const float *d_in, float *d_out, int64_t n_pixels, float threshold, void *stream);
int pt_image_add(int device, float *d_buf, const float *d_layer, int64_t n_pixels, float opacity, void *stream);
int pt_apply_aces(int device, float *d_buf, int64_t n_pixels, void *stream);
int pt_gamma_correct(int device, float *d_buf, int64_t n_pixels, float gamma, void *stream);
/* the fixed chain of Renderer::render (src/renderer.cu:277-297) run device-resident:
 * d_sum (un-normalised radiance sums, H x W x 3) -> d_out (tone-mapped LDR); d_sum is not modified */
int pt_post_chain(int device, const float *d_sum, float *d_out, int w, int h, unsigned n_samples,
                  void *stream);

/* plain device-memory helpers so that a host facade needs no HIP headers */
int pt_malloc(int device, size_t bytes, void **d_ptr);
int pt_free(int device, void *d_ptr);
int pt_memset(int device, void *d_ptr, int value, size_t bytes, void *stream);
int pt_upload(int device, void *d_dst, const void *h_src, size_t bytes, void *stream);
int pt_download(int device, void *h_dst, const void *d_src, size_t bytes, void *stream);
int pt_stream_synchronize(int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif
