/*
 * path_tracer.h -- the host C++ API of the reference (henrikglass/cuda-path-tracer), kept source
 * compatible: the same class names, constructors, public members and free functions that its
 * scene script (src/main.cu:7-69) uses.  Every class that talked to the GPU itself in the reference
 * (cudaMalloc / cudaMemcpy / kernel launches inside copy_to_device() and Renderer::render) now goes
 * through the C ABI of include/ptamd.h; no HIP header is needed to compile against this file.
 *
 *   reference header            here
 *   src/camera.h:6-28           Camera
 *   src/image.h:4-11            Image
 *   src/texture.cuh:7-44        Texture, ColorTexture, GrayscaleTexture
 *   src/material.cuh:11-59      Material
 *   src/geometry.cuh:15,79-206  Shape, AABB, Entity          (Octree: internal to the library)
 *   src/scene.cuh:9-31          Scene
 *   src/renderer.cuh:24-32      Renderer
 *   src/post_processing.cuh     Kernel, FilterType, apply_filter, compound_buffers, ...
 *   src/io.cuh                  save_ppm
 *
 * Error behaviour follows the reference: no exceptions, no status codes -- a failed library call
 * prints the message and exits (src/util.cuh:7-13, src/entity.cu:30-32, src/scene.cu:21-24).
 *
 * Deviations (each is a place where the reference misbehaves): Entity::construct_octree() only
 * records that an octree is wanted -- it is built from the vertices as they are when the scene is
 * copied to the device, so a transform issued after construct_octree() is honoured instead of
 * leaving a stale tree (Entity::octree_statistics() builds it on demand for inspection); textures are read from binary PPM / PGM files (the reference's stb_image
 * decoder is third-party code that is not re-created) or set from memory; .hdr files are read by an
 * own Radiance RGBE reader.
 */
#ifndef PTAMD_HOST_PATH_TRACER_H
#define PTAMD_HOST_PATH_TRACER_H

#include <string>
#include <vector>

#include "vector.h"

#define MAX_OCTREE_DEPTH 6          /* src/geometry.cuh:4: the reference's fixed leaf depth, the default here */
#define OCTREE_DEPTH_LIMIT 10       /* deepest octree the library builds (PT_MAX_OCTREE_DEPTH) */
#define AABB_PADDING 0.0001f        /* src/geometry.cuh:7 */

struct pt_scene;

/* ------------------------------------------------------------------ camera / image */
struct Camera {
    Camera() = default;
    Camera(vec3 position, vec3 direction, ivec2 resolution) {
        this->position = position;
        this->direction = direction;
        this->resolution = resolution;
        this->focal_length = 1.0f;
        this->up = vec3(0, 1, 0);
        this->aperture = 0.0f;
        this->focus_distance = 100.0f;
    }
    vec3 position;
    vec3 direction;
    vec3 up;
    ivec2 resolution;
    float focal_length;
    float aperture;
    float focus_distance;
};

struct Image {
    Image(std::vector<vec3> buf, ivec2 resolution) : buf(std::move(buf)), resolution(resolution) {}
    std::vector<vec3> buf;
    ivec2 resolution;
};

/* ------------------------------------------------------------------ textures / material */
struct Texture {
    Texture() {}
    Texture(const Texture &other);                   /* deep copies: a Material can be copied safely */
    Texture &operator=(const Texture &other);
    ~Texture();
    void free_from_device() {}                       /* device copies are owned by the scene */
    bool is_set() const { return this->data != nullptr; }
    unsigned char *data = nullptr;                   /* 8-bit texels, row 0 first */
    ivec2 resolution = ivec2(0, 0);
    int channels = 0;                                /* bytes per texel of `data` */
};

struct ColorTexture : Texture {
    void copy_to_device() {}
    void set(const std::string &path);               /* binary PPM (P6), 3 channels */
    void set(const unsigned char *rgb, ivec2 resolution);
};

struct GrayscaleTexture : Texture {
    void copy_to_device() {}
    void set(const std::string &path);               /* binary PGM (P5), or PPM reduced to its red channel */
    void set(const unsigned char *gray, ivec2 resolution);
};

struct Material {
    Material() {}
    Material(vec3 albedo, vec3 specular, float emission, float smoothness, bool smooth_shading) {
        this->albedo = albedo;
        this->specular = specular;
        this->emission = emission;
        this->smoothness = smoothness;
        this->smooth_shading = smooth_shading;
    }
    void copy_to_device() {}                         /* uploaded with the scene */
    void free_from_device() {}

    vec3 albedo = vec3(0.8f);
    vec3 specular = vec3(0.0f);
    float emission = 0.0f;
    float smoothness = 0.0f;
    bool smooth_shading = false;
    bool on_device = false;

    ColorTexture albedo_map;
    ColorTexture specular_map;
    ColorTexture normal_map;
    GrayscaleTexture smoothness_map;
};

/* ------------------------------------------------------------------ geometry */
enum Shape { SPHERE, TRIANGLE_MESH };

struct Vertex {
    Vertex() {}
    Vertex(vec3 position, vec3 normal) : position(position), normal(normal) {}
    vec3 position, normal;
};

struct Triangle {
    Triangle() {}
    Triangle(int a, int b, int c) : idx_a(a), idx_b(b), idx_c(c), vt_idx_a(-1), vt_idx_b(-1), vt_idx_c(-1) {}
    Triangle(int a, int b, int c, int ta, int tb, int tc) : idx_a(a), idx_b(b), idx_c(c), vt_idx_a(ta), vt_idx_b(tb), vt_idx_c(tc) {}
    int idx_a, idx_b, idx_c;
    int vt_idx_a, vt_idx_b, vt_idx_c;
};

struct AABB {
    AABB() {}
    AABB(vec3 min, vec3 max) {
        this->min = min - vec3(AABB_PADDING, AABB_PADDING, AABB_PADDING);
        this->max = max + vec3(AABB_PADDING, AABB_PADDING, AABB_PADDING);
    }
    void recalculate(Vertex *vertices, int n_vertices);
    /* separating-axis test of the box against a triangle (src/aabb.cu:64-120), as the octree builder uses it */
    bool intersects_triangle(vec3 u0, vec3 u1, vec3 u2);
    vec3 min;
    vec3 max;
};

/* The reference's pointer octree (src/geometry.cuh:93-133, src/octree.cu:9-142) for code that builds or inspects a
 * tree by hand: same constructor, members and insertion semantics (a triangle goes into every child whose padded
 * box it overlaps; leaves, at depth `leaf_depth`, list triangle indices in insertion order).  Rendering does not use
 * it: Entity::construct_octree builds the same tree level by level inside the library and flattens it for the GPU,
 * and the tests compare the two node for node.  The leaf depth is the reference's MAX_OCTREE_DEPTH unless given. */
struct Octree {
    Octree(AABB aabb, int depth, int leaf_depth = MAX_OCTREE_DEPTH) : children(), depth(depth), region(aabb), leaf_depth(leaf_depth) {}
    ~Octree();
    Octree(const Octree &) = delete;
    Octree &operator=(const Octree &) = delete;
    void pretty_print(int child_nr);
    void copy_to_device() {}                         /* the flattened tree travels with the scene */
    void free_from_device() {}
    void insert_triangle(vec3 v0, vec3 v1, vec3 v2, size_t triangle_idx);
    void insert_triangles(Vertex *vertices, Triangle *triangles, size_t n_triangles);
    Octree *children[8];
    std::vector<int> triangle_indices;
    int n_triangle_indices = 0;
    int depth;
    AABB region;
    bool on_device = false;
    int leaf_depth;
};

class Entity {
public:
    Shape shape;
    /* triangle mesh from an .obj file (one shape, triangles only; src/entity.cu:13-148) */
    Entity(const std::string &path, Material *material);
    /* sphere (src/entity.cu:157-162) */
    Entity(vec3 center, float radius, Material *material);
    ~Entity();

    void copy_to_device() {}                         /* uploaded with the scene */
    void free_from_device() {}
    /* The reference fixes the leaf depth at compile time (MAX_OCTREE_DEPTH, src/geometry.cuh:4,
     * src/entity.cu:167-170); here it is a run-time argument with that value as its default, so
     * existing scene scripts compile unchanged and large meshes can ask for a deeper tree (1..10). */
    void construct_octree(int depth = MAX_OCTREE_DEPTH);
    void scale(float factor);
    void translate(vec3 delta);
    void rotate(vec3 rot);
    void print();

    Material *material;
    bool on_device = false;

    /* read access for tools and tests */
    size_t vertex_count() const { return n_vertices; }
    size_t triangle_count() const { return n_triangles; }
    size_t uv_count() const { return n_uvs; }
    const Vertex *vertex_data() const { return vertices; }
    const Triangle *triangle_data() const { return triangles; }
    const vec2 *uv_data() const { return uvs; }
    const AABB &bounds() const { return aabb; }
    vec3 get_center() const { return center; }
    float get_radius() const { return radius; }
    bool wants_octree() const { return requested_octree_depth > 0; }
    int octree_depth() const { return requested_octree_depth; }     /* 0 = no octree requested */
    /* builds the octree of the mesh as it is NOW on the host (nothing is uploaded) and reports its shape:
     * nodes per depth (depth + 1 entries), leaf references and the largest leaf */
    struct OctreeStatistics { std::vector<long long> nodes_per_depth; long long n_nodes = 0, n_leaf_refs = 0, max_leaf = 0; };
    OctreeStatistics octree_statistics() const;

private:
    template <typename F> void transform_vertices(F f);
    Vertex *vertices = nullptr;
    Triangle *triangles = nullptr;
    vec2 *uvs = nullptr;
    size_t n_triangles = 0;
    size_t n_vertices = 0;
    size_t n_uvs = 0;
    AABB aabb;
    float radius = 0.0f;
    vec3 center = vec3(0.0f);
    int requested_octree_depth = 0;
};

/* ------------------------------------------------------------------ scene */
class Scene {
public:
    Scene();
    ~Scene();
    std::vector<Entity *> entities;
    size_t n_entities = 0;
    void set_hdri(const std::string &path);          /* Radiance .hdr (RGBE) */
    void set_hdri(const float *rgb, int width, int height);
    void set_hdri_exposure(float exposure);
    void set_hdri_contrast(float contrast);
    void rotate_hdri(float amount);
    void add_entity(Entity *entity);
    void copy_to_device();
    void free_from_device();
    void use_hdri_smoothing(bool b);

    pt_scene *device_scene() const { return handle; }   /* valid between copy_to_device and free_from_device */
    int device = 0;                                      /* GPU the scene is copied to */
    /* builds this scene on `device` and returns the library handle; the caller owns it (pt_scene_destroy).
     * copy_to_device() is build_on(this->device) kept in the scene object. */
    pt_scene *build_on(int device) const;

private:
    std::vector<float> hdri;
    int hdri_width = 0, hdri_height = 0;
    float hdri_exposure = 1.0f;
    float hdri_contrast = 1.0f;
    float hdri_rot_offset = 0.0f;
    bool on_device = false;
    bool has_hdri = false;
    bool use_smooth_hdri = false;
    pt_scene *handle = nullptr;
};

/* ------------------------------------------------------------------ renderer */
class Renderer {
public:
    unsigned int n_split_buffers = 4;
    unsigned int n_blocks_per_tile = 8;
    unsigned int n_samples_per_pass = 32;
    unsigned int tile_size = 4;
    void set_samples_per_pixel(unsigned int spp);
    Image render(const Camera &camera, Scene &scene);

    /* extensions: the RNG seed (the reference hard-codes 1337, src/renderer.cu:26), the GPUs to render
     * on, and the linear HDR image (un-normalised sample sums) of the last render for tooling.
     * devices empty (default): the scene's own device, like the reference's single-GPU render
     * (src/renderer.cu:209-216).  With several entries the scene is copied to every listed GPU, one host
     * thread per GPU renders its share of the sample range (samples are drawn from the Philox stream
     * (seed; pixel, sample), so the shares add up to the single-GPU image), the per-GPU sums are added on
     * devices[0] in list order, and the post chain runs there. */
    std::vector<int> devices;
    unsigned int seed = 1337;
    std::vector<vec3> last_hdr_sum;
    unsigned int last_samples_per_pixel = 0;
};

/* ------------------------------------------------------------------ post-processing */
struct Kernel {
    std::vector<float> taps;          /* (2*size+1)^2 weights, row major (the reference keeps them in d_buf) */
    unsigned int size = 0;            /* 3x3 kernel has size = 1, 5x5 has size = 2 */
    float weight = 0.0f;
    void print();
    void make_gaussian(unsigned int _size, float sigma);
    void make_mean(unsigned int _size);
};

enum FilterType { NORMAL, BILATERAL };

void compound_buffers(std::vector<vec3> &out_image, const std::vector<vec3> &in_buf, unsigned int n_split_buffers);
void normalize_image(std::vector<vec3> &buf, unsigned int n_samples_per_pixel);
void gamma_correct(std::vector<vec3> &buf, float gamma);
void apply_aces(std::vector<vec3> &buf);
void image_add(std::vector<vec3> &buf, const std::vector<vec3> &layer, float opacity = 1.0f);
std::vector<vec3> apply_threshold(const std::vector<vec3> &in_buf, float threshold);
void apply_filter(std::vector<vec3> &buf, ivec2 resolution, const Kernel &kernel, FilterType filter_type = NORMAL);
/* GPU on which the free post-processing functions above run (default 0; the reference always used device 0) */
void set_post_processing_device(int device);
int post_processing_device();

/* ------------------------------------------------------------------ io */
void save_ppm(const std::string &filepath, const Image &img);
/* Radiance .hdr (RGBE) -> h x w x 3 floats, first scanline first; what Scene::set_hdri(path) reads with */
bool load_hdr_image(const std::string &path, std::vector<float> &rgb, int &width, int &height);

#endif
