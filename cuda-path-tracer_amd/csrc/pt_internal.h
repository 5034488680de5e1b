/* pt_internal.h -- declarations shared between the translation units of libptamd.so */
#ifndef PT_INTERNAL_H
#define PT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/ptamd.h"
#include "pt_types.h"

namespace ptamd {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PT_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) return ::ptamd::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define PT_FAIL(msg)                 \
    do {                             \
        ::ptamd::set_error(msg);     \
        return 1;                    \
    } while (0)

/* how a sample range is cut into work items; a pure function of the range, n_split and the image size so that
 * summation order never depends on launch geometry (grid, GPU, tile share) */
struct SlicePlan {
    int sample_begin;
    int n_samples;
    int n_split;
    int slices_per_split;
};
/* n_tiles: 8x8-pixel tile slots of the image (macro tiles x 64): small launches get shorter slices */
SlicePlan make_slice_plan(int sample_begin, int sample_end, int n_split, int64_t n_tiles);
/* samples [s0, s1) of slice `slice`: split buffer z = slice / slices_per_split takes the z-th of n_split equal parts
 * of the range, cut again into slices_per_split equal parts */
void slice_range(const SlicePlan &pl, int slice, int &s0, int &s1);

/* unsigned division by a launch-invariant divisor without a divide instruction (Granlund-Montgomery): exact for
 * every 32-bit numerator; the kernel evaluates it on wave-uniform values, i.e. on the scalar unit */
struct UDiv {
    uint32_t mul, sh1, sh2;
};
inline UDiv make_udiv(uint32_t d) {
    UDiv r;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;                                    /* ceil(log2 d) */
    r.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    r.sh1 = l < 1 ? l : 1;
    r.sh2 = l > 1 ? l - 1 : 0;
    return r;
}

/* what create_camera_ray (src/renderer.cu:60-87) derives from the camera alone, evaluated once on the host with the
 * same fp32 operations the per-ray code would use (this library's host code is built without contraction too) */
struct CamConsts {
    float right[3];          /* -cross(direction, up) */
    float fd[3];             /* focal_length * direction */
    float aspect;            /* (float)res_x / (float)res_y */
    float res_x_f, res_y_f;
    float half_aperture;     /* aperture / 2 */
};
inline CamConsts make_cam_consts(const pt_camera &c) {
    CamConsts k;
    const float *d = c.direction, *u = c.up;
    const float cx = (d[1] * u[2]) - (d[2] * u[1]), cy = (d[2] * u[0]) - (d[0] * u[2]), cz = (d[0] * u[1]) - (d[1] * u[0]);
    k.right[0] = -cx; k.right[1] = -cy; k.right[2] = -cz;
    for (int i = 0; i < 3; i++) k.fd[i] = c.focal_length * d[i];
    k.res_x_f = (float)c.res_x;
    k.res_y_f = (float)c.res_y;
    k.aspect = k.res_x_f / k.res_y_f;
    k.half_aperture = c.aperture / 2;
    return k;
}

static constexpr int kMaxSlices = 64;

constexpr int kMaxQueues = 2048;                               /* upper bound on the persistent grid (workgroups): one bit-map word per lane of a wave */
constexpr int kTicketMapWord = kMaxQueues + 2;
constexpr int kTicketWords = kTicketMapWord + kMaxQueues / 32;

struct RenderParams {
    PtSceneDev sc;
    pt_camera cam;
    CamConsts camk;
    uint32_t seed;
    SlicePlan plan;
    int n_slices;                /* slices this launch renders (the whole plan: n_split * slices_per_split; a group of a plan's slices:
                                  * fewer): slice k of the launch takes samples [slice_s0[k], slice_s1[k]) and writes partial image k */
    int tiles_x, tiles_y;        /* 8x8-pixel tiles covering the image */
    int mtiles_x, mtiles_y;      /* 8x8-tile macro tiles covering the tiles */
    uint32_t tile_first, tile_stride;   /* the launch renders tiles tile_first, tile_first + tile_stride, ... of the macro-tile sequence
                                         * (0, 1: every tile; r, n: the r-th of n interleaved shares) */
    uint32_t n_items;            /* (this launch's tiles of the mtiles_x * mtiles_y * 64) * slices */
    uint32_t n_queues;           /* == gridDim.x: every workgroup owns one contiguous range of the items */
    uint32_t items_per_queue;    /* ceil(n_items / n_queues): queue q owns items [q * per, (q + 1) * per) cut at n_items */
    uint32_t *tickets;           /* kTicketWords words, zeroed before the launch: n_queues cursors, then the number of exhausted queues;
                                  * from word kTicketMapWord on, one bit per queue that has been seen exhausted */
    float *partials;             /* [n_split * slices_per_split][H*W][3] */
    unsigned long long *stats;   /* 7 counters, or null */
    UDiv div_slices;             /* item / n_slices */
    UDiv div_mtiles_x;           /* macro tile / mtiles_x */
    int slice_s0[kMaxSlices];    /* sample range [s0, s1) of every slice of the plan (slice_range on the host) */
    int slice_s1[kMaxSlices];
    int done_quota;              /* finished lanes per wave at which it leaves the geometry blocks to shade / refill */
    int leaf_quota;              /* triangle runs waiting per wave at which it tests them before walking on */
};

/* kernel launchers (pt_kernels.hip); all asynchronous on `stream` */
int launch_trace(const PtSceneDev &sc, const float *d_rays, int64_t n, pt_hit *d_hits, int32_t *d_log,
                 int log_stride, unsigned long long *d_stats, hipStream_t stream);
int launch_sample_hdri(const PtSceneDev &sc, const float *d_dirs, int64_t n, float *d_out, hipStream_t stream);
int launch_sample_material(const PtSceneDev &sc, int material, const float *d_uv, int64_t n, float *d_out, hipStream_t stream);
/* scene arrays laid out for their access, made on the device at commit (pt_types.h: PtMaterial::packed, PtSceneDev::hdri_quads) */
int launch_pack_maps(const uint8_t *const d_maps[4], int w, int h, uint32_t tiles_x, uint32_t n_records, uint32_t shift, uint8_t *d_out, hipStream_t stream);
int launch_hdri_quads(const float *d_img, int w, int h, float *d_out, hipStream_t stream);
int launch_camera_rays(const pt_camera &cam, uint32_t seed, int sample, float *d_rays, hipStream_t stream);
/* heavy: the build for scenes dominated by octree traversal (more waves per SIMD, fewer registers) */
int render_workgroups_per_cu(int max_depth, bool heavy);
int launch_render(RenderParams p, int n_cu, int max_queues, bool heavy, hipStream_t stream);
int launch_compound_partials(const float *d_partials, float *d_split, int64_t n_pixels, const SlicePlan &plan,
                             hipStream_t stream);
/* d_sum = (first ? 0 : d_sum) + d_parts[0] + d_parts[1] + ... in that order (n_parts images of n_pixels); with d_split:
 * d_split += d_sum afterwards.  The running sum of one split buffer's slices when they are rendered a few at a time. */
int launch_fold_partials(float *d_sum, const float *d_parts, int n_parts, bool first, float *d_split, int64_t n_pixels, hipStream_t stream);
/* d_words[0..n) = 0, as a kernel (the allocation-free entry point enqueues kernel launches only) */
int launch_zero_words(uint32_t *d_words, int n, hipStream_t stream);
int launch_compound_split(const float *d_split, float *d_out, int64_t n_pixels, int n_split, hipStream_t stream);

} // namespace ptamd
#endif
