/*
 * pt_api.hip -- the C ABI of include/ptamd.h: scene construction / flattening / upload, the
 * closest-hit and render entry points, device-memory helpers.  Host code only (kernels live in
 * pt_kernels.hip and pt_post.hip).
 */
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "octree_build.h"
#include "pt_child_table.h"
#include "pt_internal.h"

namespace ptamd {

static thread_local std::string g_error;
static std::atomic<uint64_t> g_octree_builds{0};      /* octrees built by this process (pt_octree_builds) */

void set_error(const std::string &msg) { g_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error: %s (%s) at %s:%d", hipGetErrorString(e), what, file, line);
    g_error = buf;
    return 1;
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct MaterialHost {
    pt_material_desc d;
    std::vector<uint8_t> maps[4];
};

struct EntityHost {
    int shape = PT_SHAPE_MESH;
    int material = 0;
    int octree_depth = 0;
    std::vector<float> positions, normals, uvs;
    std::vector<int32_t> tri;
    float aabb_min[3] = {0, 0, 0}, aabb_max[3] = {0, 0, 0};
    float center[3] = {0, 0, 0};
    float radius = 0;
    FlatOctree tree;
};

} // namespace ptamd

using namespace ptamd;

struct pt_scene {
    int device = 0;
    int n_cu = 256;
    /* host-side description: shared, never modified once added, so that pt_scene_clone can hand it to a scene on another GPU */
    std::vector<std::shared_ptr<const MaterialHost>> materials;
    std::vector<std::shared_ptr<const EntityHost>> entities;
    std::shared_ptr<const std::vector<float>> hdri;
    int hdri_w = 0, hdri_h = 0;
    float hdri_exposure = 1.0f, hdri_contrast = 1.0f, hdri_rot = 0.0f;
    bool has_hdri = false, hdri_smooth = false;

    bool committed = false;
    /* launch choices of the render kernel (defaults picked at commit, pt_scene_set_launch overrides) */
    bool heavy = false;        /* traversal-dominated scene: render with the many-waves build of the kernel */
    int quorum = 12;           /* finished lanes per wave that make it leave the geometry blocks to shade / refill */
    int leaf_quota = 12;       /* triangle runs waiting per wave at which it tests them before walking on */
    bool build_fixed = false, quorum_fixed = false, leaf_quota_fixed = false;   /* set by the caller or the environment: commit keeps them */
    PtSceneDev dev{};
    std::vector<void *> allocs;
    uint64_t bytes_nodes = 0, bytes_tris = 0, bytes_shade = 0, bytes_tex = 0, bytes_hdri = 0;

    DevBuf partials;                    /* render workspace, grown on demand */
    uint32_t *d_tickets = nullptr;      /* kMaxQueues work-queue cursors */
    unsigned long long *d_stats = nullptr;
    bool stats_valid = false;           /* d_stats holds the scheduling counters of a counted render launch */
    unsigned char *scratch = nullptr;   /* this device's copy of the fixed scratch array (pt_render_accumulate_noalloc) */

    bool profiling = false;             /* bracket the render kernel with HIP events */
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool ev_valid = false;
};

namespace {

constexpr int kStatSlots = 28;       /* counter block of a counted launch: 7 work counters, 8 scheduling counters, 5 wave life times, 3 idle-lane counts */

/*
 * Scratch of pt_render_accumulate_noalloc: the work-queue cursors and the slice partials of the render
 * kernel, which pt_render_accumulate keeps in hipMalloc'ed workspace.  One copy of the array lives on
 * every GPU the library is loaded on, from module load to process end.  Host code never passes the
 * array's name to a kernel: the per-device address is resolved once with hipGetSymbolAddress
 * (scratch_address) and only that pointer is used for kernel arguments and memsets.
 */
constexpr size_t kScratchTicketBytes = 64 * 1024;                       /* kTicketWords cursor words, padded */
constexpr size_t kScratchBytes = PT_NOALLOC_SCRATCH_BYTES;
static_assert(kTicketWords * sizeof(uint32_t) <= kScratchTicketBytes, "cursor area too small");
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices];      /* per device: address of that GPU's copy of `scratch` */

/* address of device `dev`'s copy of the scratch array (resolved on first use, with that device current) */
int scratch_address(int dev, unsigned char **out) {
    if (dev < 0 || dev >= kMaxDevices) PT_FAIL("scratch_address: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        int cur = -1;
        PT_HIP(hipGetDevice(&cur));
        if (cur != dev) PT_FAIL("scratch_address: device not current");
        void *p = nullptr;
        PT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) PT_FAIL("scratch_address: hipGetSymbolAddress returned null");
        g_scratch_ptr[dev] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

/* true when p can be dereferenced by a kernel on the current device */
bool is_device_pointer(const void *p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

template <typename T>
int upload_vec(pt_scene *s, const std::vector<T> &v, const T **out, uint64_t *bytes_acc) {
    *out = nullptr;
    if (v.empty()) return 0;
    void *d = nullptr;
    PT_HIP(hipMalloc(&d, v.size() * sizeof(T)));
    s->allocs.push_back(d);
    PT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    if (bytes_acc) *bytes_acc += v.size() * sizeof(T);
    *out = static_cast<const T *>(d);
    return 0;
}

void free_device_side(pt_scene *s) {
    for (void *p : s->allocs) (void)hipFree(p);
    s->allocs.clear();
    if (s->partials.p) (void)hipFree(s->partials.p);
    s->partials = DevBuf();
    if (s->d_tickets) (void)hipFree(s->d_tickets);
    if (s->d_stats) (void)hipFree(s->d_stats);
    s->d_tickets = nullptr;
    s->d_stats = nullptr;
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
    s->ev_start = s->ev_stop = nullptr;
    s->ev_valid = false;
    s->stats_valid = false;
    s->committed = false;
    s->bytes_nodes = s->bytes_tris = s->bytes_shade = s->bytes_tex = s->bytes_hdri = 0;
}

/* appends the triangles ids[0..n) of one run as pairs; returns where the run starts, in 16-byte units
 * (the kernels address the pair array by 32-bit byte offsets) */
uint32_t append_run(std::vector<PtTriPair> &pairs, const float *pos, const int32_t *tri6, const int32_t *ids, int64_t first_id, int64_t n) {
    const uint32_t first = (uint32_t)pairs.size();
    for (int64_t k = 0; k < n; k += 2) {
        PtTriPair pr;
        for (int j = 0; j < 2; j++) {
            const int64_t kk = (k + j < n) ? k + j : k;       /* odd tail: repeat */
            const int32_t id = ids ? ids[kk] : (int32_t)(first_id + kk);
            const int32_t *t = tri6 + 6 * (size_t)id;
            const float *v0 = pos + 3 * (size_t)t[0], *v1 = pos + 3 * (size_t)t[1], *v2 = pos + 3 * (size_t)t[2];
            for (int c = 0; c < 3; c++) {
                pr.f[2 * c + j] = v0[c];
                /* e1 = v1 - v0, e2 = v2 - v0: the subtractions of src/device_geometry_functions.cuh:332-333 */
                pr.f[2 * (3 + c) + j] = v1[c] - v0[c];
                pr.f[2 * (6 + c) + j] = v2[c] - v0[c];
            }
            pr.id[j] = id;
        }
        pairs.push_back(pr);
    }
    return first * (uint32_t)(sizeof(PtTriPair) / 16);
}

/* runs launch(d_in, d_out) on device copies of a host input of in_bytes and downloads out_bytes of results */
template <typename F>
int staged_query(size_t in_bytes, const void *h_in, size_t out_bytes, void *h_out, F launch) {
    void *d_in = nullptr, *d_out = nullptr;
    int rc = 0;
    do {
        if (hipMalloc(&d_in, in_bytes) != hipSuccess || hipMalloc(&d_out, out_bytes) != hipSuccess) { rc = 1; set_error("query: out of device memory"); break; }
        if (hipMemcpy(d_in, h_in, in_bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = 1; set_error("query: upload failed"); break; }
        rc = launch(d_in, d_out);
        if (rc) break;
        hipError_t e = hipMemcpy(h_out, d_out, out_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpy", __FILE__, __LINE__); break; }
    } while (0);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

void counters_from(const unsigned long long h[7], pt_counters *c) {
    c->segments = h[0]; c->aabb_tests = h[1]; c->node_visits = h[2]; c->tri_tests = h[3];
    c->env_lookups = h[4]; c->tex_fetches = h[5]; c->paths = h[6];
}

} // namespace

extern "C" {

const char *pt_last_error(void) { return g_error.c_str(); }

int pt_device_count(int *count) {
    if (!count) PT_FAIL("pt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return 0;
}

int pt_device_name(int device, char *buf, size_t cap) {
    if (!buf || cap == 0) PT_FAIL("pt_device_name: null buffer");
    hipDeviceProp_t prop;
    PT_HIP(hipGetDeviceProperties(&prop, device));
    /* some driver stacks leave the marketing name empty: fall back to the architecture name */
    snprintf(buf, cap, "%s", prop.name[0] ? prop.name : prop.gcnArchName);
    return 0;
}

int pt_scene_create(int device, pt_scene **out) {
    if (!out) PT_FAIL("pt_scene_create: null argument");
    /* device < 0: host-only scene (build and inspect octrees; cannot be committed) */
    if (device >= 0) {
        int n = 0;
        PT_HIP(hipGetDeviceCount(&n));
        if (device >= n) PT_FAIL("pt_scene_create: no such device");
    }
    pt_scene *s = new pt_scene();
    s->device = device;
    int cu = 0;
    if (device >= 0 && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) s->n_cu = cu;
    *out = s;
    return 0;
}

int pt_scene_destroy(pt_scene *scene) {
    if (!scene) return 0;
    if (scene->device >= 0) {
        DeviceGuard g(scene->device);
        free_device_side(scene);
    }
    delete scene;
    return 0;
}

int pt_scene_add_material(pt_scene *scene, const pt_material_desc *desc, int *material_out) {
    if (!scene || !desc) PT_FAIL("pt_scene_add_material: null argument");
    MaterialHost m;
    m.d = *desc;
    const uint8_t *src[4] = {desc->albedo_map, desc->specular_map, desc->normal_map, desc->smoothness_map};
    const int w[4] = {desc->albedo_w, desc->specular_w, desc->normal_w, desc->smoothness_w};
    const int h[4] = {desc->albedo_h, desc->specular_h, desc->normal_h, desc->smoothness_h};
    for (int k = 0; k < 4; k++) {
        if (!src[k]) continue;
        if (w[k] <= 0 || h[k] <= 0) PT_FAIL("pt_scene_add_material: texture with non-positive size");
        const size_t n = (size_t)w[k] * h[k] * (k == 3 ? 1 : 3);
        m.maps[k].assign(src[k], src[k] + n);
    }
    scene->materials.push_back(std::make_shared<const MaterialHost>(std::move(m)));
    scene->committed = false;
    if (material_out) *material_out = (int)scene->materials.size() - 1;
    return 0;
}

int pt_scene_add_mesh(pt_scene *scene, const float *positions, const float *normals, int n_vertices, const int32_t *tri,
                      int n_triangles, const float *uvs, int n_uvs, int material, int octree_depth, int *entity_out) {
    if (!scene || !positions || !tri) PT_FAIL("pt_scene_add_mesh: null argument");
    if (n_vertices <= 0 || n_triangles <= 0) PT_FAIL("pt_scene_add_mesh: empty mesh");
    if (material < 0 || material >= (int)scene->materials.size()) PT_FAIL("pt_scene_add_mesh: unknown material");
    if (octree_depth < 0 || octree_depth > PT_MAX_OCTREE_DEPTH) PT_FAIL("pt_scene_add_mesh: octree depth out of range");
    for (int64_t i = 0; i < (int64_t)n_triangles; i++) {
        for (int k = 0; k < 3; k++) {
            const int32_t v = tri[6 * i + k];
            if (v < 0 || v >= n_vertices) PT_FAIL("pt_scene_add_mesh: vertex index out of range");
            if (n_uvs > 0) {
                const int32_t t = tri[6 * i + 3 + k];
                if (t < 0 || t >= n_uvs) PT_FAIL("pt_scene_add_mesh: uv index out of range");
            }
        }
    }
    std::unique_ptr<EntityHost> e(new EntityHost());
    e->shape = PT_SHAPE_MESH;
    e->material = material;
    e->octree_depth = octree_depth;
    e->positions.assign(positions, positions + 3 * (size_t)n_vertices);
    e->tri.assign(tri, tri + 6 * (size_t)n_triangles);
    if (uvs && n_uvs > 0) e->uvs.assign(uvs, uvs + 2 * (size_t)n_uvs);
    e->normals.resize(3 * (size_t)n_vertices);
    if (normals)
        memcpy(e->normals.data(), normals, sizeof(float) * 3 * (size_t)n_vertices);
    else
        compute_vertex_normals(positions, n_vertices, tri, n_triangles, e->normals.data());
    padded_aabb(positions, n_vertices, e->aabb_min, e->aabb_max);
    if (octree_depth > 0) {
        build_octree(e->positions.data(), e->tri.data(), n_triangles, e->aabb_min, e->aabb_max, octree_depth, e->tree);
        g_octree_builds.fetch_add(1, std::memory_order_relaxed);
        if (e->tree.nodes.size() >= (1ull << 29)) PT_FAIL("pt_scene_add_mesh: octree too large (node index needs 29 bits)");
    }
    scene->entities.push_back(std::shared_ptr<const EntityHost>(std::move(e)));
    scene->committed = false;
    if (entity_out) *entity_out = (int)scene->entities.size() - 1;
    return 0;
}

int pt_scene_add_sphere(pt_scene *scene, const float center[3], float radius, int material, int *entity_out) {
    if (!scene || !center) PT_FAIL("pt_scene_add_sphere: null argument");
    if (material < 0 || material >= (int)scene->materials.size()) PT_FAIL("pt_scene_add_sphere: unknown material");
    std::unique_ptr<EntityHost> e(new EntityHost());
    e->shape = PT_SHAPE_SPHERE;
    e->material = material;
    memcpy(e->center, center, sizeof(float) * 3);
    e->radius = radius;
    scene->entities.push_back(std::shared_ptr<const EntityHost>(std::move(e)));
    scene->committed = false;
    if (entity_out) *entity_out = (int)scene->entities.size() - 1;
    return 0;
}

int pt_scene_set_hdri(pt_scene *scene, const float *rgb, int w, int h, float exposure, float contrast, float rot_offset, int smooth) {
    if (!scene) PT_FAIL("pt_scene_set_hdri: null argument");
    if (!rgb || w <= 0 || h <= 0) PT_FAIL("pt_scene_set_hdri: empty image");
    if ((unsigned long long)w * (unsigned long long)h * 12ull >= (1ull << 32)) PT_FAIL("pt_scene_set_hdri: image of 4 GiB or more (texels are addressed by 32-bit byte offsets)");
    if (!(rot_offset >= 0.0f)) PT_FAIL("pt_scene_set_hdri: rotation must be >= 0");   /* src/scene.cu:53 */
    scene->hdri = std::make_shared<const std::vector<float>>(rgb, rgb + (size_t)w * h * 3);
    scene->hdri_w = w; scene->hdri_h = h;
    scene->hdri_exposure = exposure; scene->hdri_contrast = contrast;
    scene->hdri_rot = fmodf(rot_offset, 1.0f);                                        /* src/scene.cu:54 */
    scene->hdri_smooth = smooth != 0;
    scene->has_hdri = true;
    scene->committed = false;
    return 0;
}

int pt_octree_builds(uint64_t *count) {
    if (!count) PT_FAIL("pt_octree_builds: null argument");
    *count = g_octree_builds.load(std::memory_order_relaxed);
    return 0;
}

int pt_scene_clone(pt_scene *src, int device, pt_scene **out) {
    if (!src || !out) PT_FAIL("pt_scene_clone: null argument");
    pt_scene *s = nullptr;
    if (pt_scene_create(device, &s)) return 1;
    s->materials = src->materials;
    s->entities = src->entities;
    s->hdri = src->hdri;
    s->hdri_w = src->hdri_w; s->hdri_h = src->hdri_h;
    s->hdri_exposure = src->hdri_exposure; s->hdri_contrast = src->hdri_contrast; s->hdri_rot = src->hdri_rot;
    s->has_hdri = src->has_hdri; s->hdri_smooth = src->hdri_smooth;
    s->heavy = src->heavy; s->quorum = src->quorum; s->leaf_quota = src->leaf_quota;
    s->build_fixed = src->build_fixed; s->quorum_fixed = src->quorum_fixed; s->leaf_quota_fixed = src->leaf_quota_fixed;
    *out = s;
    return 0;
}

}  // extern "C"

namespace {

/* scene files (pt_scene_save / pt_scene_load): the host-side description with its built octrees, little endian, as
 * { "PTAMDSC1", counts, then per material / entity a fixed header and its arrays } */
struct FileOut {
    FILE *f;
    bool ok = true;
    void raw(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    template <typename T> void pod(const T &v) { raw(&v, sizeof v); }
    template <typename T> void vec(const std::vector<T> &v) { pod((uint64_t)v.size()); raw(v.data(), v.size() * sizeof(T)); }
};
struct FileIn {
    FILE *f;
    uint64_t left;      /* bytes of the file not read yet: bounds every array length before it is allocated */
    bool ok = true;
    void raw(void *p, size_t n) {
        if (!ok) return;
        if (n > left || (n && fread(p, 1, n, f) != n)) { ok = false; return; }
        left -= n;
    }
    template <typename T> void pod(T &v) { raw(&v, sizeof v); }
    template <typename T> void vec(std::vector<T> &v) {
        uint64_t n = 0;
        pod(n);
        if (!ok || n > left / sizeof(T)) { ok = false; return; }
        v.resize((size_t)n);
        raw(v.data(), (size_t)n * sizeof(T));
    }
};
const char kSceneMagic[8] = {'P', 'T', 'A', 'M', 'D', 'S', 'C', '1'};

struct MaterialFileHeader { float albedo[3], specular[3], emission, smoothness; int32_t smooth_shading, w[4], h[4]; };
struct EntityFileHeader { int32_t shape, material, octree_depth, tree_depth; float aabb_min[3], aabb_max[3], center[3], radius; int64_t max_leaf; };
struct HdriFileHeader { int32_t has, w, h, smooth; float exposure, contrast, rot; };

}  // namespace

extern "C" {

int pt_scene_save(pt_scene *scene, const char *path) {
    if (!scene || !path) PT_FAIL("pt_scene_save: null argument");
    FILE *f = fopen(path, "wb");
    if (!f) PT_FAIL(std::string("pt_scene_save: cannot open ") + path);
    FileOut o{f};
    o.raw(kSceneMagic, sizeof kSceneMagic);
    o.pod((uint32_t)scene->materials.size());
    for (const auto &m : scene->materials) {
        MaterialFileHeader h;
        memset(&h, 0, sizeof h);
        memcpy(h.albedo, m->d.albedo, 12); memcpy(h.specular, m->d.specular, 12);
        h.emission = m->d.emission; h.smoothness = m->d.smoothness; h.smooth_shading = m->d.smooth_shading;
        const int w[4] = {m->d.albedo_w, m->d.specular_w, m->d.normal_w, m->d.smoothness_w};
        const int hh[4] = {m->d.albedo_h, m->d.specular_h, m->d.normal_h, m->d.smoothness_h};
        for (int k = 0; k < 4; k++) { h.w[k] = m->maps[k].empty() ? 0 : w[k]; h.h[k] = m->maps[k].empty() ? 0 : hh[k]; }
        o.pod(h);
        for (int k = 0; k < 4; k++) o.vec(m->maps[k]);
    }
    o.pod((uint32_t)scene->entities.size());
    for (const auto &e : scene->entities) {
        EntityFileHeader h;
        memset(&h, 0, sizeof h);
        h.shape = e->shape; h.material = e->material; h.octree_depth = e->octree_depth; h.tree_depth = e->tree.max_depth;
        memcpy(h.aabb_min, e->aabb_min, 12); memcpy(h.aabb_max, e->aabb_max, 12); memcpy(h.center, e->center, 12);
        h.radius = e->radius; h.max_leaf = e->tree.max_leaf;
        o.pod(h);
        o.vec(e->positions); o.vec(e->normals); o.vec(e->uvs); o.vec(e->tri);
        o.vec(e->tree.nodes); o.vec(e->tree.leaf_tris); o.vec(e->tree.nodes_per_depth);
    }
    HdriFileHeader hh;
    memset(&hh, 0, sizeof hh);
    hh.has = scene->has_hdri ? 1 : 0; hh.w = scene->hdri_w; hh.h = scene->hdri_h; hh.smooth = scene->hdri_smooth ? 1 : 0;
    hh.exposure = scene->hdri_exposure; hh.contrast = scene->hdri_contrast; hh.rot = scene->hdri_rot;
    o.pod(hh);
    if (scene->has_hdri) o.vec(*scene->hdri);
    const bool closed = fclose(f) == 0;
    if (!o.ok || !closed) { remove(path); PT_FAIL(std::string("pt_scene_save: write failed: ") + path); }
    return 0;
}

int pt_scene_load(int device, const char *path, pt_scene **out) {
    if (!path || !out) PT_FAIL("pt_scene_load: null argument");
    FILE *f = fopen(path, "rb");
    if (!f) PT_FAIL(std::string("pt_scene_load: cannot open ") + path);
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    FileIn in{f, size > 0 ? (uint64_t)size : 0};
    pt_scene *s = nullptr;
    const auto fail = [&](const char *what) {
        fclose(f);
        if (s) pt_scene_destroy(s);
        set_error(std::string("pt_scene_load: ") + what + ": " + path);
        return 1;
    };
    char magic[8];
    in.raw(magic, sizeof magic);
    if (!in.ok || memcmp(magic, kSceneMagic, sizeof magic) != 0) return fail("not a scene file");
    if (pt_scene_create(device, &s)) { fclose(f); return 1; }
    uint32_t n_mat = 0, n_ent = 0;
    in.pod(n_mat);
    for (uint32_t i = 0; in.ok && i < n_mat; i++) {
        MaterialFileHeader h;
        in.pod(h);
        MaterialHost m;
        memset(&m.d, 0, sizeof m.d);
        memcpy(m.d.albedo, h.albedo, 12); memcpy(m.d.specular, h.specular, 12);
        m.d.emission = h.emission; m.d.smoothness = h.smoothness; m.d.smooth_shading = h.smooth_shading;
        m.d.albedo_w = h.w[0]; m.d.specular_w = h.w[1]; m.d.normal_w = h.w[2]; m.d.smoothness_w = h.w[3];
        m.d.albedo_h = h.h[0]; m.d.specular_h = h.h[1]; m.d.normal_h = h.h[2]; m.d.smoothness_h = h.h[3];
        for (int k = 0; k < 4 && in.ok; k++) {
            in.vec(m.maps[k]);
            if (in.ok && h.w[k] >= 0 && h.h[k] >= 0 && m.maps[k].size() != (size_t)h.w[k] * (size_t)h.h[k] * (k == 3 ? 1 : 3)) in.ok = false;
        }
        s->materials.push_back(std::make_shared<const MaterialHost>(std::move(m)));
    }
    in.pod(n_ent);
    for (uint32_t i = 0; in.ok && i < n_ent; i++) {
        EntityFileHeader h;
        in.pod(h);
        std::unique_ptr<EntityHost> e(new EntityHost());
        e->shape = h.shape; e->material = h.material; e->octree_depth = h.octree_depth; e->tree.max_depth = h.tree_depth;
        memcpy(e->aabb_min, h.aabb_min, 12); memcpy(e->aabb_max, h.aabb_max, 12); memcpy(e->center, h.center, 12);
        e->radius = h.radius; e->tree.max_leaf = h.max_leaf;
        in.vec(e->positions); in.vec(e->normals); in.vec(e->uvs); in.vec(e->tri);
        in.vec(e->tree.nodes); in.vec(e->tree.leaf_tris); in.vec(e->tree.nodes_per_depth);
        if (!in.ok) break;
        /* what pt_scene_add_mesh checks, and what the flattening in pt_scene_commit indexes */
        const size_t nv = e->positions.size() / 3, nt = e->tri.size() / 6, nuv = e->uvs.size() / 2;
        bool good = h.material >= 0 && h.material < (int)s->materials.size() && (h.shape == PT_SHAPE_SPHERE || h.shape == PT_SHAPE_MESH) &&
                    h.octree_depth >= 0 && h.octree_depth <= PT_MAX_OCTREE_DEPTH;
        if (good && h.shape == PT_SHAPE_MESH) {
            good = nv > 0 && nt > 0 && e->positions.size() == 3 * nv && e->normals.size() == 3 * nv && e->tri.size() == 6 * nt && e->uvs.size() == 2 * nuv;
            for (size_t t = 0; good && t < nt; t++)
                for (int k = 0; k < 3; k++) {
                    good = good && e->tri[6 * t + k] >= 0 && (size_t)e->tri[6 * t + k] < nv;
                    if (nuv) good = good && e->tri[6 * t + 3 + k] >= 0 && (size_t)e->tri[6 * t + 3 + k] < nuv;
                }
            if (good && h.octree_depth > 0) {
                const FlatOctree &T = e->tree;
                good = T.max_depth == h.octree_depth && T.nodes_per_depth.size() == (size_t)T.max_depth + 1 && !T.nodes.empty() && T.nodes.size() < (1ull << 29);
                int64_t total = 0;
                for (size_t d = 0; good && d < T.nodes_per_depth.size(); d++) { good = T.nodes_per_depth[d] >= 0; total += T.nodes_per_depth[d]; }
                good = good && (uint64_t)total == T.nodes.size();
                const size_t n_inner = good ? T.nodes.size() - (size_t)T.nodes_per_depth[T.max_depth] : 0;
                for (size_t n = 0; good && n < T.nodes.size(); n++) {
                    const PtNode &nd = T.nodes[n];
                    if (n < n_inner) good = nd.b < 256u && (uint64_t)nd.a + (uint32_t)__builtin_popcount(nd.b) <= T.nodes.size() && (nd.b == 0u || nd.a > n);
                    else good = (uint64_t)nd.a + nd.b <= T.leaf_tris.size();
                }
                for (size_t k = 0; good && k < T.leaf_tris.size(); k++) good = T.leaf_tris[k] >= 0 && (size_t)T.leaf_tris[k] < nt;
            }
        }
        if (!good) { in.ok = false; break; }
        s->entities.push_back(std::shared_ptr<const EntityHost>(std::move(e)));
    }
    HdriFileHeader hh;
    memset(&hh, 0, sizeof hh);
    in.pod(hh);
    if (in.ok && hh.has) {
        std::vector<float> img;
        in.vec(img);
        if (in.ok && (hh.w <= 0 || hh.h <= 0 || img.size() != (size_t)hh.w * (size_t)hh.h * 3 || (uint64_t)hh.w * (uint64_t)hh.h * 12ull >= (1ull << 32))) in.ok = false;
        s->hdri = std::make_shared<const std::vector<float>>(std::move(img));
        s->hdri_w = hh.w; s->hdri_h = hh.h; s->hdri_smooth = hh.smooth != 0; s->has_hdri = true;
        s->hdri_exposure = hh.exposure; s->hdri_contrast = hh.contrast; s->hdri_rot = hh.rot;
    }
    if (!in.ok) return fail("truncated or inconsistent scene file");
    fclose(f);
    *out = s;
    return 0;
}

int pt_scene_commit(pt_scene *scene) {
    if (!scene) PT_FAIL("pt_scene_commit: null argument");
    if (scene->device < 0) PT_FAIL("pt_scene_commit: host-only scene (created with device < 0)");
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_scene_commit: cannot select device");
    free_device_side(scene);

    std::vector<PtEntity> ents(scene->entities.size());
    std::vector<PtNode> nodes;
    std::vector<PtTriPair> recs;
    std::vector<PtShadeRec> shade;
    int max_depth = 0;
    for (size_t i = 0; i < scene->entities.size(); i++) {
        const EntityHost &h = *scene->entities[i];
        PtEntity e;
        memset(&e, 0, sizeof e);
        e.shape = h.shape;
        e.material = h.material;
        e.smooth_shading = scene->materials[h.material]->d.smooth_shading ? 1 : 0;
        memcpy(e.center, h.center, sizeof e.center);
        e.radius = h.radius;
        if (h.shape == PT_SHAPE_MESH) {
            const int64_t nt = (int64_t)h.tri.size() / 6;
            memcpy(e.aabb_min, h.aabb_min, sizeof e.aabb_min);
            memcpy(e.aabb_max, h.aabb_max, sizeof e.aabb_max);
            memcpy(e.region_min, h.aabb_min, sizeof e.region_min);
            memcpy(e.region_max, h.aabb_max, sizeof e.region_max);
            e.octree_depth = h.octree_depth;
            e.has_uvs = h.uvs.empty() ? 0 : 1;
            e.shade_base = (uint32_t)shade.size();
            for (int64_t t = 0; t < nt; t++) {
                const int32_t *tr = &h.tri[6 * t];
                PtShadeRec sr;
                memset(&sr, 0, sizeof sr);
                memcpy(sr.n0, &h.normals[3 * (size_t)tr[0]], 12);
                memcpy(sr.n1, &h.normals[3 * (size_t)tr[1]], 12);
                memcpy(sr.n2, &h.normals[3 * (size_t)tr[2]], 12);
                if (e.has_uvs) {
                    memcpy(sr.uv0, &h.uvs[2 * (size_t)tr[3]], 8);
                    memcpy(sr.uv1, &h.uvs[2 * (size_t)tr[4]], 8);
                    memcpy(sr.uv2, &h.uvs[2 * (size_t)tr[5]], 8);
                }
                shade.push_back(sr);
            }
            if (h.octree_depth == 0) {
                e.tri_first = append_run(recs, h.positions.data(), h.tri.data(), nullptr, 0, nt);
                e.n_tris = (uint32_t)nt;
            } else {
                const uint32_t node_base = (uint32_t)nodes.size();
                e.root_node = node_base;
                const FlatOctree &T = h.tree;
                const size_t n_inner = T.nodes.size() - (size_t)T.nodes_per_depth[T.max_depth];
                for (size_t n = 0; n < T.nodes.size(); n++) {
                    PtNode nd = T.nodes[n];
                    if (n < n_inner) nd.a += node_base;
                    else nd.a = append_run(recs, h.positions.data(), h.tri.data(), T.leaf_tris.data() + nd.a, 0, (int64_t)nd.b);
                    nodes.push_back(nd);
                }
                e.root_a = nodes[node_base].a;
                e.root_b = nodes[node_base].b;
                if (h.octree_depth > max_depth) max_depth = h.octree_depth;
            }
        }
        ents[i] = e;
    }
    if (nodes.size() >= (1ull << 29) || recs.size() * sizeof(PtTriPair) >= (1ull << 32)) PT_FAIL("pt_scene_commit: scene too large (limits: 4 GiB of triangle pairs, 2^29 octree nodes)");

    /* PTAMD_PLAIN_MAPS=1 keeps textures and the environment map in their row-major upload layouts (the code paths of maps
     * with different resolutions and of nearest-texel environment look-ups): for A/B measurements and tests */
    const char *plain_env = getenv("PTAMD_PLAIN_MAPS");
    const bool plain_layouts = plain_env && plain_env[0] == '1';
    std::vector<PtMaterial> mats(scene->materials.size());
    for (size_t i = 0; i < scene->materials.size(); i++) {
        const MaterialHost &h = *scene->materials[i];
        PtMaterial m;
        memset(&m, 0, sizeof m);
        memcpy(m.albedo, h.d.albedo, 12);
        memcpy(m.specular, h.d.specular, 12);
        m.emission = h.d.emission;
        m.smoothness = h.d.smoothness;
        m.smooth_shading = h.d.smooth_shading ? 1 : 0;
        const int w[4] = {h.d.albedo_w, h.d.specular_w, h.d.normal_w, h.d.smoothness_w};
        const int hh[4] = {h.d.albedo_h, h.d.specular_h, h.d.normal_h, h.d.smoothness_h};
        /* maps of one resolution go into texel records (PtMaterial::packed): one look-up per hit instead of one per map */
        int first = -1;
        bool same = true;
        for (int k = 0; k < 4; k++) {
            if (h.maps[k].empty()) continue;
            if (first < 0) first = k;
            same = same && w[k] == w[first] && hh[k] == hh[first];
        }
        const uint32_t shift = h.maps[PT_MAP_SPECULAR].empty() ? 3u : 4u;
        const uint64_t tiles_x = first >= 0 ? ((uint64_t)w[first] + 3) / 4 : 0, tiles_y = first >= 0 ? ((uint64_t)hh[first] + 3) / 4 : 0;
        const uint64_t n_records = tiles_x * tiles_y * 16;
        if (first >= 0 && same && !plain_layouts && (n_records << shift) < (1ull << 32)) {
            const uint8_t *d_maps[4] = {nullptr, nullptr, nullptr, nullptr};
            std::vector<void *> tmp;
            int rc = 0;
            void *d_packed = nullptr;
            do {
                for (int k = 0; k < 4 && !rc; k++) {
                    if (h.maps[k].empty()) continue;
                    void *d = nullptr;
                    if (hipMalloc(&d, h.maps[k].size()) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc map", __FILE__, __LINE__); break; }
                    tmp.push_back(d);
                    if (hipMemcpy(d, h.maps[k].data(), h.maps[k].size(), hipMemcpyHostToDevice) != hipSuccess) { rc = 1; set_error("pt_scene_commit: map upload failed"); break; }
                    d_maps[k] = static_cast<const uint8_t *>(d);
                    m.packed_mask |= 1u << k;
                }
                if (rc) break;
                if (hipMalloc(&d_packed, (size_t)(n_records << shift)) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc texel records", __FILE__, __LINE__); break; }
                scene->allocs.push_back(d_packed);
                rc = launch_pack_maps(d_maps, w[first], hh[first], (uint32_t)tiles_x, (uint32_t)n_records, shift, static_cast<uint8_t *>(d_packed), nullptr);
                if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) { rc = 1; set_error("pt_scene_commit: packing the maps failed"); }
            } while (0);
            for (void *d : tmp) (void)hipFree(d);
            if (rc) return 1;
            m.packed = static_cast<const uint8_t *>(d_packed);
            m.packed_tiles_x = (uint32_t)tiles_x;
            m.packed_shift = shift;
            m.packed_w_f = (float)w[first]; m.packed_h_f = (float)hh[first];
            m.packed_xmax_f = (float)(w[first] - 1); m.packed_ymax_f = (float)(hh[first] - 1);
            scene->bytes_tex += n_records << shift;
        } else {
            for (int k = 0; k < 4; k++) {
                if (h.maps[k].empty()) continue;
                const uint8_t *d = nullptr;
                if (upload_vec(scene, h.maps[k], &d, &scene->bytes_tex)) return 1;
                m.map[k] = d;
                m.map_w[k] = w[k];
                m.map_h[k] = hh[k];
            }
        }
        mats[i] = m;
    }

    PtSceneDev d;
    memset(&d, 0, sizeof d);
    if (upload_vec(scene, ents, &d.entities, nullptr)) return 1;
    if (upload_vec(scene, nodes, &d.nodes, &scene->bytes_nodes)) return 1;
    if (upload_vec(scene, recs, &d.tris, &scene->bytes_tris)) return 1;
    if (upload_vec(scene, shade, &d.shade, &scene->bytes_shade)) return 1;
    if (upload_vec(scene, mats, &d.materials, nullptr)) return 1;
    if (scene->has_hdri) {
        const uint64_t n_texels = (uint64_t)scene->hdri_w * (uint64_t)scene->hdri_h;
        if (scene->hdri_smooth && !plain_layouts && n_texels * 64 < (1ull << 32)) {
            /* bilinear look-ups read one 64-byte record of four texels (PtSceneDev::hdri_quads), expanded on the device */
            void *d_img = nullptr, *d_quads = nullptr;
            int rc = 0;
            do {
                if (hipMalloc(&d_img, scene->hdri->size() * sizeof(float)) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc hdri", __FILE__, __LINE__); break; }
                if (hipMemcpy(d_img, scene->hdri->data(), scene->hdri->size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { rc = 1; set_error("pt_scene_commit: environment map upload failed"); break; }
                if (hipMalloc(&d_quads, (size_t)(n_texels * 64)) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc hdri records", __FILE__, __LINE__); break; }
                scene->allocs.push_back(d_quads);
                rc = launch_hdri_quads(static_cast<const float *>(d_img), scene->hdri_w, scene->hdri_h, static_cast<float *>(d_quads), nullptr);
                if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) { rc = 1; set_error("pt_scene_commit: expanding the environment map failed"); }
            } while (0);
            if (d_img) (void)hipFree(d_img);
            if (rc) return 1;
            d.hdri_quads = static_cast<const float *>(d_quads);
            scene->bytes_hdri += n_texels * 64;
        } else if (upload_vec(scene, *scene->hdri, &d.hdri, &scene->bytes_hdri)) {
            return 1;
        }
    }
    d.n_entities = (int32_t)ents.size();
    d.hdri_w = scene->hdri_w; d.hdri_h = scene->hdri_h;
    d.hdri_w_f = (float)scene->hdri_w; d.hdri_h_f = (float)scene->hdri_h;
    d.hdri_xmax_f = (float)(scene->hdri_w - 1); d.hdri_ymax_f = (float)(scene->hdri_h - 1);
    d.has_hdri = scene->has_hdri ? 1 : 0;
    d.use_smooth_hdri = scene->hdri_smooth ? 1 : 0;
    d.hdri_exposure = scene->hdri_exposure;
    d.hdri_contrast = scene->hdri_contrast;
    d.hdri_rot_offset = scene->hdri_rot;
    d.max_depth = max_depth;
    scene->dev = d;
    /* which build of the render kernel: scenes with a large octree spend their time in the walk and
     * the triangle tests (measured cross-over between 24 k and 128 k triangle pairs); PTAMD_KERNEL =
     * heavy | light overrides */
    if (!scene->build_fixed) scene->heavy = max_depth > 0 && recs.size() >= 65536;
    if (const char *k = getenv("PTAMD_KERNEL")) {
        if (!strcmp(k, "heavy")) { scene->heavy = true; scene->build_fixed = true; }
        else if (!strcmp(k, "light")) { scene->heavy = false; scene->build_fixed = true; }
    }
    /* the quorum: a refill round (shading, new paths, their entity set-up) costs the same whatever the
     * number of lanes it serves, so it should serve many where it is expensive next to the walk that
     * separates two rounds -- textured materials, and most of all scenes without any octree, where there
     * is no walk at all (62: +34 %) -- and few inside deep trees, where lanes waiting for a refill idle
     * through many walk steps.  The heavy build tests triangle runs with the whole wave, waiting lanes
     * included, so waiting costs it less and its optimum lies higher (measured: 48 / 40 against 40 / 12) */
    if (!scene->quorum_fixed) {
        bool textured = false;
        for (const auto &m : scene->materials)
            for (int k = 0; k < 4; k++) textured = textured || !m->maps[k].empty();
        /* an octree so deep that its LDS stacks leave room for three waves per SIMD only: fewer waves cover the round's
         * fetches, so it should come less often (20 M triangles at depth 9, quorum x leaf quota: 48 x 12 is +2 % on 40 x 20) */
        const bool few_waves = scene->heavy && render_workgroups_per_cu(max_depth, true) < 4;
        if (max_depth == 0) scene->quorum = 62;
        else if (scene->heavy) scene->quorum = (textured || few_waves) ? 48 : 40;
        else scene->quorum = textured ? 40 : 12;
    }
    if (!scene->leaf_quota_fixed) scene->leaf_quota = 12;      /* (phases run whole trips of 64 pairs and serve waiting lanes first: 12 runs is the optimum of the four- and the three-wave instantiation alike) */
    if (const char *k = getenv("PTAMD_QUORUM")) {
        const int q = atoi(k);
        if (q >= 1 && q <= 64) { scene->quorum = q; scene->quorum_fixed = true; }
    }

    if (const char *k = getenv("PTAMD_LEAF_QUOTA")) {
        const int q = atoi(k);
        if (q >= 1 && q <= 65) { scene->leaf_quota = q; scene->leaf_quota_fixed = true; }
    }

    PT_HIP(hipMalloc((void **)&scene->d_tickets, kTicketWords * sizeof(uint32_t)));
    PT_HIP(hipMalloc((void **)&scene->d_stats, kStatSlots * sizeof(unsigned long long)));
    PT_HIP(hipMemset(scene->d_stats, 0, kStatSlots * sizeof(unsigned long long)));
    if (scratch_address(scene->device, &scene->scratch)) return 1;
    scene->committed = true;
    return 0;
}

int pt_scene_octree_info(pt_scene *scene, int entity, pt_octree_info *info) {
    if (!scene || !info) PT_FAIL("pt_scene_octree_info: null argument");
    if (entity < 0 || entity >= (int)scene->entities.size()) PT_FAIL("pt_scene_octree_info: no such entity");
    const EntityHost &h = *scene->entities[entity];
    if (h.shape != PT_SHAPE_MESH || h.octree_depth == 0) PT_FAIL("pt_scene_octree_info: entity has no octree");
    memset(info, 0, sizeof *info);
    info->max_depth = h.tree.max_depth;
    info->n_nodes = (int64_t)h.tree.nodes.size();
    info->n_leaf_refs = (int64_t)h.tree.leaf_tris.size();
    info->max_leaf = h.tree.max_leaf;
    for (int i = 0; i <= h.tree.max_depth; i++) info->nodes_per_depth[i] = h.tree.nodes_per_depth[i];
    return 0;
}

int pt_scene_octree_dump(pt_scene *scene, int entity, int32_t *out, int64_t cap, int64_t *n_words) {
    if (!scene || !n_words) PT_FAIL("pt_scene_octree_dump: null argument");
    if (entity < 0 || entity >= (int)scene->entities.size()) PT_FAIL("pt_scene_octree_dump: no such entity");
    const EntityHost &h = *scene->entities[entity];
    if (h.shape != PT_SHAPE_MESH || h.octree_depth == 0) PT_FAIL("pt_scene_octree_dump: entity has no octree");
    std::vector<int32_t> d;
    dump_octree(h.tree, d);
    *n_words = (int64_t)d.size();
    if (out && cap < 0) PT_FAIL("pt_scene_octree_dump: negative capacity");
    if (out) memcpy(out, d.data(), sizeof(int32_t) * (size_t)std::min<int64_t>(cap, (int64_t)d.size()));
    return 0;
}

int pt_scene_device_bytes(pt_scene *scene, uint64_t *nodes, uint64_t *tri_records, uint64_t *shade_records,
                          uint64_t *textures, uint64_t *hdri) {
    if (!scene) PT_FAIL("pt_scene_device_bytes: null argument");
    if (!scene->committed) PT_FAIL("pt_scene_device_bytes: scene not committed");
    if (nodes) *nodes = scene->bytes_nodes;
    if (tri_records) *tri_records = scene->bytes_tris;
    if (shade_records) *shade_records = scene->bytes_shade;
    if (textures) *textures = scene->bytes_tex;
    if (hdri) *hdri = scene->bytes_hdri;
    return 0;
}

int pt_trace_rays(pt_scene *scene, const float *rays, int64_t n, pt_hit *hits, int32_t *log, int log_stride,
                  pt_counters *counters, void *stream_) {
    if (!scene) PT_FAIL("pt_trace_rays: null scene");
    if (!scene->committed) PT_FAIL("pt_trace_rays: scene not committed");
    if (n < 0) PT_FAIL("pt_trace_rays: negative ray count");
    if (n == 0) {
        if (counters) memset(counters, 0, sizeof *counters);
        return 0;
    }
    if (!rays || !hits) PT_FAIL("pt_trace_rays: null buffer");
    if (log && log_stride < 1) PT_FAIL("pt_trace_rays: log_stride must be >= 1");
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_trace_rays: cannot select device");
    if (counters) scene->stats_valid = false;      /* the query kernel overwrites the counter block */
    hipStream_t stream = (hipStream_t)stream_;
    const bool rays_dev = is_device_pointer(rays), hits_dev = is_device_pointer(hits);
    const bool log_dev = log ? is_device_pointer(log) : true;
    const float *d_rays = rays;
    pt_hit *d_hits = hits;
    int32_t *d_log = log;
    void *tmp_rays = nullptr, *tmp_hits = nullptr, *tmp_log = nullptr;
    int rc = 0;
    do {
        if (!rays_dev) {
            if (hipMalloc(&tmp_rays, (size_t)n * 24) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc rays", __FILE__, __LINE__); break; }
            if (hipMemcpyAsync(tmp_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice, stream) != hipSuccess) { rc = 1; set_error("pt_trace_rays: ray upload failed"); break; }
            d_rays = (const float *)tmp_rays;
        }
        if (!hits_dev) {
            if (hipMalloc(&tmp_hits, (size_t)n * sizeof(pt_hit)) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc hits", __FILE__, __LINE__); break; }
            d_hits = (pt_hit *)tmp_hits;
        }
        if (log && !log_dev) {
            if (hipMalloc(&tmp_log, (size_t)n * log_stride * 4) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "hipMalloc log", __FILE__, __LINE__); break; }
            d_log = (int32_t *)tmp_log;
            /* entries past a ray's test count read as 0 */
            if (hipMemsetAsync(tmp_log, 0, (size_t)n * log_stride * 4, stream) != hipSuccess) { rc = 1; set_error("pt_trace_rays: memset failed"); break; }
        }
        if (counters && hipMemsetAsync(scene->d_stats, 0, kStatSlots * sizeof(unsigned long long), stream) != hipSuccess) { rc = 1; set_error("pt_trace_rays: memset failed"); break; }
        rc = launch_trace(scene->dev, d_rays, n, d_hits, d_log, log_stride, counters ? scene->d_stats : nullptr, stream);
        if (rc) break;
        if (!hits_dev && hipMemcpyAsync(hits, d_hits, (size_t)n * sizeof(pt_hit), hipMemcpyDeviceToHost, stream) != hipSuccess) { rc = 1; set_error("pt_trace_rays: hit download failed"); break; }
        if (log && !log_dev && hipMemcpyAsync(log, d_log, (size_t)n * log_stride * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) { rc = 1; set_error("pt_trace_rays: log download failed"); break; }
        if (counters) {
            unsigned long long h[7];
            if (hipMemcpyAsync(h, scene->d_stats, sizeof h, hipMemcpyDeviceToHost, stream) != hipSuccess) { rc = 1; set_error("pt_trace_rays: counter download failed"); break; }
            hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__); break; }
            counters_from(h, counters);
        }
        if (tmp_rays || tmp_hits || tmp_log) {
            hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__); break; }
        }
    } while (0);
    if (tmp_rays) (void)hipFree(tmp_rays);
    if (tmp_hits) (void)hipFree(tmp_hits);
    if (tmp_log) (void)hipFree(tmp_log);
    return rc;
}

int pt_camera_rays(int device, const pt_camera *camera, uint32_t seed, int sample, float *rays, void *stream_) {
    if (!camera || !rays) PT_FAIL("pt_camera_rays: null argument");
    if (camera->res_x <= 0 || camera->res_y <= 0) PT_FAIL("pt_camera_rays: empty image");
    DeviceGuard g(device);
    if (!g.ok) PT_FAIL("pt_camera_rays: cannot select device");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t bytes = (size_t)camera->res_x * camera->res_y * 24;
    if (is_device_pointer(rays)) return launch_camera_rays(*camera, seed, sample, rays, stream);
    void *tmp = nullptr;
    PT_HIP(hipMalloc(&tmp, bytes));
    int rc = launch_camera_rays(*camera, seed, sample, (float *)tmp, stream);
    if (!rc && hipMemcpyAsync(rays, tmp, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) { rc = 1; set_error("pt_camera_rays: download failed"); }
    if (!rc && hipStreamSynchronize(stream) != hipSuccess) { rc = 1; set_error("pt_camera_rays: sync failed"); }
    (void)hipFree(tmp);
    return rc;
}

namespace {

/* fills everything of the launch parameters that follows from the camera and the sample range; returns the bytes of
 * slice partials the launch needs in *need */
int plan_render(pt_scene *scene, const pt_camera *camera, uint32_t seed, int sample_begin, int sample_end, int n_split,
                RenderParams &p, size_t *need, int tile_rank = 0, int tile_ranks = 1) {
    memset(&p, 0, sizeof p);
    p.sc = scene->dev;
    p.cam = *camera;
    p.camk = make_cam_consts(*camera);
    p.seed = seed;
    p.tiles_x = (camera->res_x + 7) / 8;
    p.tiles_y = (camera->res_y + 7) / 8;
    p.mtiles_x = (p.tiles_x + 7) / 8;
    p.mtiles_y = (p.tiles_y + 7) / 8;
    p.plan = make_slice_plan(sample_begin, sample_end, n_split, (int64_t)p.mtiles_x * p.mtiles_y * 64);
    const int n_slices = p.plan.n_split * p.plan.slices_per_split;
    if (n_slices > kMaxSlices) PT_FAIL("pt_render_accumulate: slice plan exceeds the slice table");
    p.n_slices = n_slices;
    p.div_slices = make_udiv((uint32_t)n_slices);
    p.div_mtiles_x = make_udiv((uint32_t)p.mtiles_x);
    for (int k = 0; k < n_slices; k++) {
        slice_range(p.plan, k, p.slice_s0[k], p.slice_s1[k]);
        /* a lane counts the samples of its slice down in 16 bits */
        if (p.slice_s1[k] - p.slice_s0[k] > 0xFFFF) PT_FAIL("pt_render_accumulate: more than 65535 samples per slice (split the sample range over several calls)");
    }
    p.tile_first = (uint32_t)tile_rank;
    p.tile_stride = (uint32_t)tile_ranks;
    const uint64_t n_tiles = (uint64_t)p.mtiles_x * p.mtiles_y * 64;
    const uint64_t my_tiles = n_tiles > (uint64_t)tile_rank ? (n_tiles - (uint64_t)tile_rank + (uint64_t)tile_ranks - 1) / (uint64_t)tile_ranks : 0;
    const uint64_t n_items = my_tiles * n_slices;
    if (n_items >= 0xFFFF0000ull) PT_FAIL("pt_render_accumulate: too many work items");
    p.n_items = (uint32_t)n_items;
    const int64_t n_pixels = (int64_t)camera->res_x * camera->res_y;
    *need = (size_t)n_slices * (size_t)n_pixels * 12;
    p.stats = nullptr;
    p.done_quota = scene->quorum;
    p.leaf_quota = scene->leaf_quota;
    return 0;
}

/* (the message is only put together when a check fails: a successful call touches no heap) */
int fail_arg(const char *who, const char *what) {
    char buf[160];
    snprintf(buf, sizeof buf, "%s: %s", who, what);
    set_error(buf);
    return 1;
}

int check_render_args(const char *who, pt_scene *scene, const pt_camera *camera, int sample_begin, int sample_end, int n_split,
                      float *d_split) {
    if (!scene || !camera || !d_split) return fail_arg(who, "null argument");
    if (!scene->committed) return fail_arg(who, "scene not committed");
    if (camera->res_x <= 0 || camera->res_y <= 0) return fail_arg(who, "empty image");
    if (camera->res_x > 0xFFFF || camera->res_y > 0xFFFF) return fail_arg(who, "image larger than 65535 pixels on a side");
    if (sample_begin < 0 || sample_end < sample_begin) return fail_arg(who, "bad sample range");
    if (n_split < 1 || n_split > 64) return fail_arg(who, "n_split must be in [1,64]");
    if (!is_device_pointer(d_split)) return fail_arg(who, "d_split must be a device pointer");
    return 0;
}

int ensure_events(pt_scene *scene) {
    if (!scene->ev_start) {
        PT_HIP(hipEventCreate(&scene->ev_start));
        PT_HIP(hipEventCreate(&scene->ev_stop));
    }
    return 0;
}

}  // namespace

int pt_render_accumulate(pt_scene *scene, const pt_camera *camera, uint32_t seed, int sample_begin, int sample_end,
                         int n_split, float *d_split, pt_counters *counters, void *stream_) {
    return pt_render_accumulate_tiles(scene, camera, seed, sample_begin, sample_end, n_split, d_split, 0, 1, counters, stream_);
}

int pt_render_accumulate_tiles(pt_scene *scene, const pt_camera *camera, uint32_t seed, int sample_begin, int sample_end,
                               int n_split, float *d_split, int tile_rank, int tile_ranks, pt_counters *counters, void *stream_) {
    if (check_render_args("pt_render_accumulate", scene, camera, sample_begin, sample_end, n_split, d_split)) return 1;
    if (tile_ranks < 1 || tile_rank < 0 || tile_rank >= tile_ranks) PT_FAIL("pt_render_accumulate_tiles: need 0 <= tile_rank < tile_ranks");
    if (sample_end == sample_begin) {
        if (counters) memset(counters, 0, sizeof *counters);
        return 0;
    }
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_render_accumulate: cannot select device");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n_pixels = (int64_t)camera->res_x * camera->res_y;
    RenderParams p;
    size_t need = 0;
    if (plan_render(scene, camera, seed, sample_begin, sample_end, n_split, p, &need, tile_rank, tile_ranks)) return 1;
    if (p.n_items == 0) {
        if (counters) memset(counters, 0, sizeof *counters);
        return 0;
    }
    if (scene->partials.bytes < need) {
        /* the previous launch may still read the old workspace */
        PT_HIP(hipStreamSynchronize(stream));
        if (scene->partials.p) PT_HIP(hipFree(scene->partials.p));
        scene->partials = DevBuf();
        PT_HIP(hipMalloc(&scene->partials.p, need));
        scene->partials.bytes = need;
    }
    p.partials = (float *)scene->partials.p;
    p.tickets = scene->d_tickets;
    p.stats = counters ? scene->d_stats : nullptr;
    PT_HIP(hipMemsetAsync(scene->d_tickets, 0, kTicketWords * sizeof(uint32_t), stream));
    /* a share of the tiles leaves the other tiles' partials unwritten: they must read as zero in the compounding kernel */
    if (tile_ranks > 1) PT_HIP(hipMemsetAsync(p.partials, 0, need, stream));
    if (counters) PT_HIP(hipMemsetAsync(scene->d_stats, 0, kStatSlots * sizeof(unsigned long long), stream));
    if (scene->profiling) {
        if (ensure_events(scene)) return 1;
        PT_HIP(hipEventRecord(scene->ev_start, stream));
    }
    if (launch_render(p, scene->n_cu, kMaxQueues, scene->heavy, stream)) return 1;
    if (scene->profiling) {
        PT_HIP(hipEventRecord(scene->ev_stop, stream));
        scene->ev_valid = true;
    }
    if (launch_compound_partials(p.partials, d_split, n_pixels, p.plan, stream)) return 1;
    if (counters) {
        unsigned long long h[7];
        PT_HIP(hipMemcpyAsync(h, scene->d_stats, sizeof h, hipMemcpyDeviceToHost, stream));
        PT_HIP(hipStreamSynchronize(stream));
        counters_from(h, counters);
        scene->stats_valid = true;
    }
    return 0;
}

int pt_render_slices(const pt_camera *camera, int sample_begin, int sample_end, int n_split, int *slices) {
    if (!camera || !slices) PT_FAIL("pt_render_slices: null argument");
    if (camera->res_x <= 0 || camera->res_y <= 0 || sample_begin < 0 || sample_end < sample_begin || n_split < 1 || n_split > 64)
        PT_FAIL("pt_render_slices: bad argument");
    const int64_t mtx = ((camera->res_x + 7) / 8 + 7) / 8, mty = ((camera->res_y + 7) / 8 + 7) / 8;
    const SlicePlan pl = make_slice_plan(sample_begin, sample_end, n_split, mtx * mty * 64);
    *slices = sample_end == sample_begin ? 0 : pl.n_split * pl.slices_per_split;
    return 0;
}

size_t pt_noalloc_max_partial_bytes(void) { return kScratchBytes - kScratchTicketBytes; }

int pt_render_accumulate_noalloc(pt_scene *scene, const pt_camera *camera, uint32_t seed, int sample_begin, int sample_end,
                                 int n_split, float *d_split, void *stream_) {
    if (check_render_args("pt_render_accumulate_noalloc", scene, camera, sample_begin, sample_end, n_split, d_split)) return 1;
    if (sample_end == sample_begin) return 0;
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_render_accumulate_noalloc: cannot select device");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n_pixels = (int64_t)camera->res_x * camera->res_y;
    RenderParams p;
    size_t need = 0;
    if (plan_render(scene, camera, seed, sample_begin, sample_end, n_split, p, &need)) return 1;
    unsigned char *base = scene->scratch;        /* this device's copy of the scratch array, resolved at commit */
    if (!base) PT_FAIL("pt_render_accumulate_noalloc: scratch address not resolved");
    p.tickets = reinterpret_cast<uint32_t *>(base);
    float *images = reinterpret_cast<float *>(base + kScratchTicketBytes);
    const size_t image_bytes = (size_t)n_pixels * 12;
    const size_t max_images = (kScratchBytes - kScratchTicketBytes) / image_bytes;
    const int n_slices = p.n_slices;
    if ((size_t)n_slices <= max_images) {
        /* the whole plan at once, as pt_render_accumulate does.  The array keeps the previous call's contents: the cursors
         * must start at zero; every partial the compounding kernel reads is written by the render kernel first */
        p.partials = images;
        if (launch_zero_words(p.tickets, kTicketWords, stream)) return 1;
        if (launch_render(p, scene->n_cu, kMaxQueues, scene->heavy, stream)) return 1;
        if (launch_compound_partials(p.partials, d_split, n_pixels, p.plan, stream)) return 1;
        return 0;
    }
    /* The plan's slices do not fit the scratch together: render them a few at a time.  Image 0 of the scratch carries the
     * running sum of the current split buffer's slices, the other images take the slices of one launch; the sums are the
     * additions pt_render_accumulate's compounding kernel makes, in its order -- (((0 + p0) + p1) + ...), then split += sum --
     * so the split buffers end up with the same bits as after ONE pt_render_accumulate call with the same arguments. */
    if (max_images < 2) {
        char buf[256];
        snprintf(buf, sizeof buf,
                 "pt_render_accumulate_noalloc: a frame of %lld pixels needs two images of %zu bytes in the fixed scratch of %zu bytes "
                 "(one slice of samples and the running sum): render a smaller frame per call",
                 (long long)n_pixels, image_bytes, kScratchBytes - kScratchTicketBytes);
        PT_FAIL(buf);
    }
    const int per_launch = (int)std::min<size_t>(max_images - 1, (size_t)kMaxSlices);
    const int sps = p.plan.slices_per_split;
    float *sum = images, *parts = images + (size_t)n_pixels * 3;
    for (int z = 0; z < p.plan.n_split; z++) {
        for (int j = 0; j < sps; j += per_launch) {
            const int g = std::min(per_launch, sps - j);
            RenderParams q = p;
            q.n_slices = g;
            q.div_slices = make_udiv((uint32_t)g);
            for (int k = 0; k < g; k++) slice_range(p.plan, z * sps + j + k, q.slice_s0[k], q.slice_s1[k]);
            q.n_items = (uint32_t)((uint64_t)p.mtiles_x * p.mtiles_y * 64 * g);      /* (every tile: tile_first 0, tile_stride 1) */
            q.partials = parts;
            if (launch_zero_words(q.tickets, kTicketWords, stream)) return 1;
            if (launch_render(q, scene->n_cu, kMaxQueues, scene->heavy, stream)) return 1;
            const bool last = j + g >= sps;
            if (launch_fold_partials(sum, parts, g, j == 0, last ? d_split + (size_t)z * n_pixels * 3 : nullptr, n_pixels, stream)) return 1;
        }
    }
    return 0;
}

int pt_scene_set_launch(pt_scene *scene, int kernel_build, int quorum) {
    if (!scene) PT_FAIL("pt_scene_set_launch: null argument");
    if (kernel_build < -1 || kernel_build > 1) PT_FAIL("pt_scene_set_launch: kernel_build must be -1 (keep), 0 (light) or 1 (heavy)");
    if (quorum < 0 || quorum > 64) PT_FAIL("pt_scene_set_launch: quorum must be 0 (keep) or in [1,64]");
    if (kernel_build >= 0) { scene->heavy = kernel_build == 1; scene->build_fixed = true; }
    if (quorum > 0) { scene->quorum = quorum; scene->quorum_fixed = true; }
    return 0;
}

int pt_scene_get_launch(pt_scene *scene, int *kernel_build, int *quorum) {
    if (!scene) PT_FAIL("pt_scene_get_launch: null argument");
    if (kernel_build) *kernel_build = scene->heavy ? 1 : 0;
    if (quorum) *quorum = scene->quorum;
    return 0;
}

int pt_sample_hdri(pt_scene *scene, const float *dirs, int64_t n, float *rgb_out) {
    if (!scene) PT_FAIL("pt_sample_hdri: null scene");
    if (!scene->committed) PT_FAIL("pt_sample_hdri: scene not committed");
    if (n < 0) PT_FAIL("pt_sample_hdri: negative count");
    if (n == 0) return 0;
    if (!dirs || !rgb_out) PT_FAIL("pt_sample_hdri: null buffer");
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_sample_hdri: cannot select device");
    return staged_query((size_t)n * 12, dirs, (size_t)n * 12, rgb_out, [&](void *d_in, void *d_out) {
        return launch_sample_hdri(scene->dev, (const float *)d_in, n, (float *)d_out, nullptr);
    });
}

int pt_sample_material(pt_scene *scene, int material, const float *uv, int64_t n, float *albedo, float *specular, float *smoothness,
                       float *normal) {
    if (!scene) PT_FAIL("pt_sample_material: null scene");
    if (!scene->committed) PT_FAIL("pt_sample_material: scene not committed");
    if (material < 0 || material >= (int)scene->materials.size()) PT_FAIL("pt_sample_material: unknown material");
    if (n < 0) PT_FAIL("pt_sample_material: negative count");
    if (n == 0) return 0;
    if (!uv || !albedo || !specular || !smoothness || !normal) PT_FAIL("pt_sample_material: null buffer");
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_sample_material: cannot select device");
    std::vector<float> packed((size_t)n * 10);
    if (staged_query((size_t)n * 8, uv, packed.size() * sizeof(float), packed.data(), [&](void *d_in, void *d_out) {
            return launch_sample_material(scene->dev, material, (const float *)d_in, n, (float *)d_out, nullptr);
        }))
        return 1;
    for (int64_t i = 0; i < n; i++) {
        const float *o = &packed[(size_t)i * 10];
        memcpy(albedo + 3 * i, o, 12);
        memcpy(specular + 3 * i, o + 3, 12);
        smoothness[i] = o[6];
        memcpy(normal + 3 * i, o + 7, 12);
    }
    return 0;
}

int pt_box_intersects_triangle(const float bmin[3], const float bmax[3], const float v0[3], const float v1[3], const float v2[3],
                               int pad_box, int *overlaps) {
    if (!bmin || !bmax || !v0 || !v1 || !v2 || !overlaps) PT_FAIL("pt_box_intersects_triangle: null argument");
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) {
        mn[k] = pad_box ? bmin[k] - 0.0001f : bmin[k];      /* AABB_PADDING, src/geometry.cuh:7,81-84 */
        mx[k] = pad_box ? bmax[k] + 0.0001f : bmax[k];
    }
    *overlaps = box_intersects_triangle(mn, mx, v0, v1, v2) ? 1 : 0;
    return 0;
}

int pt_child_tables(uint8_t *set_out, uint8_t *mir_out) {
    static constexpr ChildTables t = make_child_tables();
    if (set_out) memcpy(set_out, t.set, sizeof t.set);
    if (mir_out) memcpy(mir_out, t.mir, sizeof t.mir);
    return 0;
}

int pt_scene_render_residency(pt_scene *scene, int *workgroups_per_cu) {
    if (!scene || !workgroups_per_cu) PT_FAIL("pt_scene_render_residency: null argument");
    if (!scene->committed) PT_FAIL("pt_scene_render_residency: scene not committed");
    *workgroups_per_cu = render_workgroups_per_cu(scene->dev.max_depth, scene->heavy);
    return 0;
}

int pt_scene_schedule_counters(pt_scene *scene, uint64_t out[8]) {
    if (!scene || !out) PT_FAIL("pt_scene_schedule_counters: null argument");
    if (!scene->committed) PT_FAIL("pt_scene_schedule_counters: scene not committed");
    if (!scene->stats_valid) PT_FAIL("pt_scene_schedule_counters: no render launch with counters since the scene was committed");
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_scene_schedule_counters: cannot select device");
    unsigned long long h[16];
    PT_HIP(hipMemcpy(h, scene->d_stats, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = h[7 + i];
    return 0;
}

int pt_scene_wave_profile(pt_scene *scene, uint64_t out[11]) {
    if (!scene || !out) PT_FAIL("pt_scene_wave_profile: null argument");
    if (!scene->committed) PT_FAIL("pt_scene_wave_profile: scene not committed");
    if (!scene->stats_valid) PT_FAIL("pt_scene_wave_profile: no render launch with counters since the scene was committed");
    DeviceGuard g(scene->device);
    if (!g.ok) PT_FAIL("pt_scene_wave_profile: cannot select device");
    unsigned long long h[kStatSlots];
    PT_HIP(hipMemcpy(h, scene->d_stats, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = h[15 + i];
    const unsigned long long waves = h[26] ? h[26] : 1, first_start = ~h[24], last_end = h[25];
    const unsigned long long mean_start = h[23] / waves, mean_end = (h[23] + h[27]) / waves;
    out[8] = mean_start - first_start;          /* how long after the first wave a wave starts, on average */
    out[9] = last_end - mean_end;               /* how long before the last wave a wave ends, on average */
    out[10] = last_end - first_start;           /* from the first wave's start to the last wave's end */
    return 0;
}

int pt_scene_set_profiling(pt_scene *scene, int enable) {
    if (!scene) PT_FAIL("pt_scene_set_profiling: null argument");
    scene->profiling = enable != 0;
    scene->ev_valid = false;
    return 0;
}

int pt_scene_last_render_kernel_ms(pt_scene *scene, float *ms) {
    if (!scene || !ms) PT_FAIL("pt_scene_last_render_kernel_ms: null argument");
    if (!scene->profiling || !scene->ev_valid) PT_FAIL("pt_scene_last_render_kernel_ms: no profiled render launch");
    DeviceGuard g(scene->device);
    PT_HIP(hipEventSynchronize(scene->ev_stop));
    PT_HIP(hipEventElapsedTime(ms, scene->ev_start, scene->ev_stop));
    return 0;
}

int pt_compound_buffers(int device, const float *d_split, float *d_out, int64_t n_pixels, int n_split, void *stream) {
    if (!d_split || !d_out) PT_FAIL("pt_compound_buffers: null argument");
    if (n_pixels < 0 || n_split < 1) PT_FAIL("pt_compound_buffers: bad size");
    if (n_pixels == 0) return 0;
    DeviceGuard g(device);
    return launch_compound_split(d_split, d_out, n_pixels, n_split, (hipStream_t)stream);
}

int pt_malloc(int device, size_t bytes, void **d_ptr) {
    if (!d_ptr) PT_FAIL("pt_malloc: null argument");
    DeviceGuard g(device);
    if (!g.ok) PT_FAIL("pt_malloc: cannot select device");
    PT_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return 0;
}
int pt_free(int device, void *d_ptr) {
    if (!d_ptr) return 0;
    DeviceGuard g(device);
    PT_HIP(hipFree(d_ptr));
    return 0;
}
int pt_memset(int device, void *d_ptr, int value, size_t bytes, void *stream) {
    if (!d_ptr) PT_FAIL("pt_memset: null argument");
    DeviceGuard g(device);
    PT_HIP(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
    return 0;
}
int pt_upload(int device, void *d_dst, const void *h_src, size_t bytes, void *stream) {
    if (!d_dst || !h_src) PT_FAIL("pt_upload: null argument");
    DeviceGuard g(device);
    PT_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    PT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int pt_download(int device, void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (!h_dst || !d_src) PT_FAIL("pt_download: null argument");
    DeviceGuard g(device);
    PT_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    PT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int pt_stream_synchronize(int device, void *stream) {
    DeviceGuard g(device);
    PT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

} // extern "C"
