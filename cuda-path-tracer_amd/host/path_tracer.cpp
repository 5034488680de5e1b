/*
 * path_tracer.cpp -- implementation of the host C++ API (path_tracer.h) on top of the C ABI of
 * include/ptamd.h.  Host code only; links libptamd.so.  Compiled with -ffp-contract=off so that
 * mesh preparation (normals, transforms, bounding boxes) rounds like the reference's host code.
 */
#include "path_tracer.h"

#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "../../include/ptamd.h"

namespace {

/* print-and-exit, like gpuErrchk (src/util.cuh:7-13) */
void check(int rc, const char *what) {
    if (rc != 0) {
        fprintf(stderr, "ptamd: %s failed: %s\n", what, pt_last_error());
        exit(rc);
    }
}
#define PT_CHECK(call) check((call), #call)

/* reads one whitespace-delimited PNM header token, skipping comments */
bool pnm_token(FILE *f, std::string &tok) {
    tok.clear();
    int c = fgetc(f);
    while (c != EOF) {
        if (c == '#') {
            while (c != EOF && c != '\n') c = fgetc(f);
        } else if (!isspace(c)) {
            break;
        }
        c = fgetc(f);
    }
    while (c != EOF && !isspace(c)) {
        tok.push_back((char)c);
        c = fgetc(f);
    }
    return !tok.empty();
}

/* binary PPM (P6) / PGM (P5) with maxval 255 -> malloc'd texels with `want` channels */
unsigned char *load_pnm(const std::string &path, int want, ivec2 &res) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    std::string magic, sw, sh, smax;
    if (!pnm_token(f, magic) || !pnm_token(f, sw) || !pnm_token(f, sh) || !pnm_token(f, smax)) { fclose(f); return nullptr; }
    const int have = magic == "P6" ? 3 : magic == "P5" ? 1 : 0;
    const int w = atoi(sw.c_str()), h = atoi(sh.c_str());
    if (!have || w <= 0 || h <= 0 || atoi(smax.c_str()) != 255) { fclose(f); return nullptr; }
    std::vector<unsigned char> raw((size_t)w * h * have);
    if (fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); return nullptr; }
    fclose(f);
    unsigned char *out = (unsigned char *)malloc((size_t)w * h * want);
    for (size_t i = 0; i < (size_t)w * h; i++)
        for (int c = 0; c < want; c++) out[i * want + c] = raw[i * have + (have == 3 ? (want == 3 ? c : 0) : 0)];
    res = ivec2(w, h);
    return out;
}

} // namespace

/* Radiance RGBE (.hdr) reader: header, "-Y h +X w", new-style RLE or flat scanlines; row 0 of the
 * result is the first scanline of the file, as in the reference's loader (external/hdrloader.h:207-212);
 * component = mantissa / 256 * 2^(e - 128) (external/hdrloader.h:54-59). */
bool load_hdr_image(const std::string &path, std::vector<float> &rgb, int &w, int &h) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char line[256];
    if (!fgets(line, sizeof line, f) || strncmp(line, "#?", 2) != 0) { fclose(f); return false; }
    for (;;) {       /* header lines up to the empty line */
        if (!fgets(line, sizeof line, f)) { fclose(f); return false; }
        if (line[0] == '\n' || (line[0] == '\r' && line[1] == '\n')) break;
    }
    if (!fgets(line, sizeof line, f) || sscanf(line, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { fclose(f); return false; }
    rgb.assign((size_t)w * h * 3, 0.0f);
    std::vector<unsigned char> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        unsigned char hd[4];
        if (fread(hd, 1, 4, f) != 4) { fclose(f); return false; }
        if (w >= 8 && w < 32768 && hd[0] == 2 && hd[1] == 2 && !(hd[2] & 128)) {
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    int code = fgetc(f);
                    if (code == EOF) { fclose(f); return false; }
                    if (code > 128) {
                        int n = code & 127, val = fgetc(f);
                        while (n-- > 0 && x < w) scan[(size_t)x++ * 4 + c] = (unsigned char)val;
                    } else {
                        int n = code;
                        while (n-- > 0 && x < w) scan[(size_t)x++ * 4 + c] = (unsigned char)fgetc(f);
                    }
                }
            }
        } else {
            memcpy(scan.data(), hd, 4);
            if (w > 1 && fread(scan.data() + 4, 1, (size_t)(w - 1) * 4, f) != (size_t)(w - 1) * 4) { fclose(f); return false; }
        }
        float *row = &rgb[(size_t)y * w * 3];
        for (int x = 0; x < w; x++) {
            const float d = (float)pow(2, (int)scan[(size_t)x * 4 + 3] - 128);
            for (int c = 0; c < 3; c++) row[x * 3 + c] = (scan[(size_t)x * 4 + c] / 256.0f) * d;
        }
    }
    fclose(f);
    return true;
}

namespace {

struct ObjData {
    std::vector<float> v, vt, vn;
    std::vector<int> idx;      /* per face corner: vertex, texcoord (0-based, -1 when absent) */
    std::vector<int> face_sizes;
    int n_shapes = 0;
};

/* the subset of OBJ the reference accepts through tinyobjloader: v / vt / vn / f, index
 * conventions 0-based with -1 for "absent", negative (relative) indices resolved */
bool parse_obj(const std::string &path, ObjData &o) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    bool saw_face = false, named = false;
    while (std::getline(in, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            o.v.push_back(x); o.v.push_back(y); o.v.push_back(z);
        } else if (tag == "vt") {
            float u = 0, v = 0;
            ss >> u >> v;
            o.vt.push_back(u); o.vt.push_back(v);
        } else if (tag == "vn") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            o.vn.push_back(x); o.vn.push_back(y); o.vn.push_back(z);
        } else if (tag == "f") {
            std::string corner;
            int n = 0;
            while (ss >> corner) {
                int vi = 0, ti = 0;
                bool has_t = false;
                const size_t s1 = corner.find('/');
                vi = atoi(corner.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    const size_t s2 = corner.find('/', s1 + 1);
                    const std::string t = corner.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) { ti = atoi(t.c_str()); has_t = true; }
                }
                o.idx.push_back(vi > 0 ? vi - 1 : (int)(o.v.size() / 3) + vi);
                o.idx.push_back(has_t ? (ti > 0 ? ti - 1 : (int)(o.vt.size() / 2) + ti) : -1);
                n++;
            }
            o.face_sizes.push_back(n);
            if (!saw_face) { saw_face = true; o.n_shapes++; }
        } else if (tag == "o" || tag == "g") {
            /* a new named group after faces starts another shape, as tinyobjloader counts them */
            if (saw_face && named) saw_face = false;
            named = true;
        }
    }
    return true;
}

} // namespace

/* ================================================================== textures */
Texture::~Texture() {
    if (this->data != nullptr) free(this->data);
}

Texture::Texture(const Texture &other) { *this = other; }

Texture &Texture::operator=(const Texture &other) {
    if (this == &other) return *this;
    if (this->data) { free(this->data); this->data = nullptr; }
    this->resolution = other.resolution;
    if (other.data) {
        /* the channel count is not stored: copy what the allocation holds (3 bytes per texel at most) */
        const size_t texels = (size_t)other.resolution.x * other.resolution.y;
        const size_t bytes = texels * (size_t)other.channels;
        this->data = (unsigned char *)malloc(bytes);
        memcpy(this->data, other.data, bytes);
    }
    this->channels = other.channels;
    return *this;
}

void ColorTexture::set(const std::string &path) {
    if (this->data) { free(this->data); this->data = nullptr; }
    this->data = load_pnm(path, 3, this->resolution);     /* nullptr when unreadable: is_set() stays false */
    this->channels = 3;
}
void ColorTexture::set(const unsigned char *rgb, ivec2 res) {
    if (this->data) free(this->data);
    const size_t n = (size_t)res.x * res.y * 3;
    this->data = (unsigned char *)malloc(n);
    memcpy(this->data, rgb, n);
    this->resolution = res;
    this->channels = 3;
}
void GrayscaleTexture::set(const std::string &path) {
    if (this->data) { free(this->data); this->data = nullptr; }
    this->data = load_pnm(path, 1, this->resolution);
    this->channels = 1;
}
void GrayscaleTexture::set(const unsigned char *gray, ivec2 res) {
    if (this->data) free(this->data);
    const size_t n = (size_t)res.x * res.y;
    this->data = (unsigned char *)malloc(n);
    memcpy(this->data, gray, n);
    this->resolution = res;
    this->channels = 1;
}

/* ================================================================== geometry */
void AABB::recalculate(Vertex *vertices, int n_vertices) {      /* src/aabb.cu:7-22 */
    vec3 _min(FLT_MAX, FLT_MAX, FLT_MAX);
    vec3 _max(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < n_vertices; i++) {
        vec3 p = vertices[i].position;
        _min.x = (p.x < _min.x) ? p.x : _min.x;
        _min.y = (p.y < _min.y) ? p.y : _min.y;
        _min.z = (p.z < _min.z) ? p.z : _min.z;
        _max.x = (p.x > _max.x) ? p.x : _max.x;
        _max.y = (p.y > _max.y) ? p.y : _max.y;
        _max.z = (p.z > _max.z) ? p.z : _max.z;
    }
    this->min = _min - vec3(AABB_PADDING, AABB_PADDING, AABB_PADDING);
    this->max = _max + vec3(AABB_PADDING, AABB_PADDING, AABB_PADDING);
}

Entity::Entity(const std::string &path, Material *material) {   /* src/entity.cu:13-148 */
    ObjData obj;
    if (!parse_obj(path, obj) || obj.n_shapes != 1) {
        std::cout << "cannot load " << path << " (missing, or not exactly one shape)" << std::endl;
        exit(1);
    }
    const bool no_normals = (obj.vn.size() / 3) == 0;
    const bool has_uv = obj.vt.size() != 0;
    this->n_vertices = obj.v.size() / 3;
    this->n_triangles = obj.face_sizes.size();
    this->n_uvs = obj.vt.size() / 2;
    this->vertices = new Vertex[this->n_vertices];
    this->triangles = new Triangle[this->n_triangles];
    if (has_uv) this->uvs = new vec2[this->n_uvs];

    vec3 mn(FLT_MAX, FLT_MAX, FLT_MAX), mx(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (size_t v = 0; v < this->n_vertices; v++) {
        vec3 position(obj.v[3 * v], obj.v[3 * v + 1], obj.v[3 * v + 2]);
        vec3 normal(0.0f, 0.0f, 0.0f);
        /* the reference indexes vn by VERTEX id (src/entity.cu:58-65, a known flaw of it); kept,
         * bounded by the number of normals present */
        if (!no_normals && 3 * v + 2 < obj.vn.size()) {
            normal = vec3(obj.vn[3 * v], obj.vn[3 * v + 1], obj.vn[3 * v + 2]);
            normal.normalize();
        }
        mn.x = (position.x < mn.x) ? position.x : mn.x;
        mn.y = (position.y < mn.y) ? position.y : mn.y;
        mn.z = (position.z < mn.z) ? position.z : mn.z;
        mx.x = (position.x > mx.x) ? position.x : mx.x;
        mx.y = (position.y > mx.y) ? position.y : mx.y;
        mx.z = (position.z > mx.z) ? position.z : mx.z;
        this->vertices[v] = Vertex(position, normal);
    }
    for (size_t t = 0; t < this->n_uvs; t++) this->uvs[t] = vec2(obj.vt[2 * t], obj.vt[2 * t + 1]);

    size_t off = 0;
    for (size_t f = 0; f < this->n_triangles; f++) {
        if (obj.face_sizes[f] != 3) {
            std::cerr << "OBJ file faces must be triangles" << std::endl;
            exit(1);
        }
        this->triangles[f] = Triangle(obj.idx[off + 0], obj.idx[off + 2], obj.idx[off + 4], obj.idx[off + 1], obj.idx[off + 3], obj.idx[off + 5]);
        const Triangle &tr = this->triangles[f];
        if (tr.idx_a < 0 || tr.idx_b < 0 || tr.idx_c < 0 || (size_t)tr.idx_a >= n_vertices || (size_t)tr.idx_b >= n_vertices ||
            (size_t)tr.idx_c >= n_vertices) {
            std::cerr << "OBJ face references a missing vertex" << std::endl;
            exit(1);
        }
        if (no_normals) {     /* src/entity.cu:110-123 */
            vec3 v0 = this->vertices[tr.idx_a].position, v1 = this->vertices[tr.idx_b].position, v2 = this->vertices[tr.idx_c].position;
            vec3 t_normal = cross(v1 - v0, v2 - v0).normalized();
            this->vertices[tr.idx_a].normal = this->vertices[tr.idx_a].normal + t_normal;
            this->vertices[tr.idx_b].normal = this->vertices[tr.idx_b].normal + t_normal;
            this->vertices[tr.idx_c].normal = this->vertices[tr.idx_c].normal + t_normal;
        }
        off += 6;
    }
    if (no_normals)
        for (size_t i = 0; i < this->n_vertices; i++) this->vertices[i].normal.normalize();

    this->aabb = AABB(mn, mx);
    this->center = vec3((mn.x + mx.x) / 2, (mn.y + mx.y) / 2, (mn.z + mx.z) / 2);
    this->material = material;
    this->shape = TRIANGLE_MESH;
}

Entity::Entity(vec3 center, float radius, Material *material) {
    this->shape = SPHERE;
    this->center = center;
    this->radius = radius;
    this->material = material;
}

Entity::~Entity() {
    delete[] this->vertices;
    delete[] this->triangles;
    delete[] this->uvs;
}

void Entity::construct_octree() { this->octree_requested = true; }

void Entity::print() {      /* src/entity.cu:175-180 */
    std::cout << "center: " << this->center << std::endl;
    std::cout << "x: " << this->aabb.min.x << " to " << this->aabb.max.x << std::endl;
    std::cout << "y: " << this->aabb.min.y << " to " << this->aabb.max.y << std::endl;
    std::cout << "z: " << this->aabb.min.z << " to " << this->aabb.max.z << std::endl;
}

void Entity::scale(float factor) {      /* src/entity.cu:185-202 */
    if (this->shape == SPHERE) { radius *= factor; return; }
    for (size_t i = 0; i < n_vertices; i++) {
        vec3 pos = vertices[i].position;
        pos = ((pos - this->center) * factor) + this->center;
        vertices[i].position = pos;
    }
    this->aabb.recalculate(this->vertices, (int)this->n_vertices);
}

void Entity::translate(vec3 delta) {    /* src/entity.cu:207-222 */
    this->center = this->center + delta;
    if (this->shape == TRIANGLE_MESH) {
        for (size_t i = 0; i < n_vertices; i++) vertices[i].position = vertices[i].position + delta;
        this->aabb.recalculate(this->vertices, (int)this->n_vertices);
    }
}

void Entity::rotate(vec3 rot) {         /* src/entity.cu:227-258 */
    if (this->shape == SPHERE) return;
    /* the global ::cos / ::sin of the reference's translation unit are the double versions; narrowed */
    float cos_a = (float)std::cos((double)rot.z), sin_a = (float)std::sin((double)rot.z);
    float cos_b = (float)std::cos((double)rot.y), sin_b = (float)std::sin((double)rot.y);
    float cos_g = (float)std::cos((double)rot.x), sin_g = (float)std::sin((double)rot.x);
    vec3 R0(cos_a * cos_b, cos_a * sin_b * sin_g - sin_a * cos_g, cos_a * sin_b * cos_g + sin_a * sin_g);
    vec3 R1(sin_a * cos_b, sin_a * sin_b * sin_g + cos_a * cos_g, sin_a * sin_b * cos_g - cos_a * sin_g);
    vec3 R2(-sin_b, cos_b * sin_g, cos_b * cos_g);
    for (size_t i = 0; i < this->n_vertices; i++) {
        vec3 v = this->vertices[i].position - this->center;
        v = vec3(dot(v, R0), dot(v, R1), dot(v, R2));
        this->vertices[i].position = v + this->center;
        vec3 n = this->vertices[i].normal;
        this->vertices[i].normal = vec3(dot(n, R0), dot(n, R1), dot(n, R2));
    }
    this->aabb.recalculate(this->vertices, (int)this->n_vertices);
}

/* ================================================================== scene */
Scene::Scene() {}
Scene::~Scene() { this->free_from_device(); }

void Scene::set_hdri(const std::string &path) {       /* src/scene.cu:19-26 */
    if (!load_hdr_image(path, this->hdri, this->hdri_width, this->hdri_height)) {
        std::cerr << "Error parsing .hdr file, or file doesn't exist." << std::endl;
        exit(1);
    }
    has_hdri = true;
}
void Scene::set_hdri(const float *rgb, int width, int height) {
    this->hdri.assign(rgb, rgb + (size_t)width * height * 3);
    this->hdri_width = width;
    this->hdri_height = height;
    has_hdri = true;
}
void Scene::set_hdri_exposure(float exposure) { this->hdri_exposure = exposure; }
void Scene::set_hdri_contrast(float contrast) { this->hdri_contrast = contrast; }
void Scene::use_hdri_smoothing(bool b) { this->use_smooth_hdri = b; }
void Scene::rotate_hdri(float amount) {               /* src/scene.cu:52-55 */
    assert(amount >= 0.0f);
    this->hdri_rot_offset = fmod(amount, 1.0f);
}
void Scene::add_entity(Entity *entity) { this->entities.push_back(entity); }

/* replaces the per-object cudaMalloc / cudaMemcpy cascade of src/scene.cu:67-90, src/entity.cu:263-303,
 * src/octree.cu:24-41, src/material.cu:32-43, src/texture.cu:29-49 with one flattened upload */
void Scene::copy_to_device() {
    if (this->on_device) this->free_from_device();
    PT_CHECK(pt_scene_create(this->device, &this->handle));
    std::map<Material *, int> mat_ids;      /* materials may be shared between entities (src/entity.cu:363) */
    for (Entity *e : this->entities) {
        if (mat_ids.find(e->material) == mat_ids.end()) {
            Material *m = e->material;
            pt_material_desc d;
            memset(&d, 0, sizeof d);
            d.albedo[0] = m->albedo.x; d.albedo[1] = m->albedo.y; d.albedo[2] = m->albedo.z;
            d.specular[0] = m->specular.x; d.specular[1] = m->specular.y; d.specular[2] = m->specular.z;
            d.emission = m->emission;
            d.smoothness = m->smoothness;
            d.smooth_shading = m->smooth_shading ? 1 : 0;
            d.albedo_map = m->albedo_map.data; d.albedo_w = m->albedo_map.resolution.x; d.albedo_h = m->albedo_map.resolution.y;
            d.specular_map = m->specular_map.data; d.specular_w = m->specular_map.resolution.x; d.specular_h = m->specular_map.resolution.y;
            d.normal_map = m->normal_map.data; d.normal_w = m->normal_map.resolution.x; d.normal_h = m->normal_map.resolution.y;
            d.smoothness_map = m->smoothness_map.data; d.smoothness_w = m->smoothness_map.resolution.x; d.smoothness_h = m->smoothness_map.resolution.y;
            int id = -1;
            PT_CHECK(pt_scene_add_material(this->handle, &d, &id));
            mat_ids[m] = id;
            m->on_device = true;
        }
    }
    for (Entity *e : this->entities) {
        const int mat = mat_ids[e->material];
        if (e->shape == SPHERE) {
            const vec3 c = e->get_center();
            const float cc[3] = {c.x, c.y, c.z};
            PT_CHECK(pt_scene_add_sphere(this->handle, cc, e->get_radius(), mat, nullptr));
        } else {
            const size_t nv = e->vertex_count(), nt = e->triangle_count();
            std::vector<float> pos(3 * nv), nrm(3 * nv);
            for (size_t i = 0; i < nv; i++) {
                const Vertex &v = e->vertex_data()[i];
                pos[3 * i] = v.position.x; pos[3 * i + 1] = v.position.y; pos[3 * i + 2] = v.position.z;
                nrm[3 * i] = v.normal.x; nrm[3 * i + 1] = v.normal.y; nrm[3 * i + 2] = v.normal.z;
            }
            static_assert(sizeof(Triangle) == 6 * sizeof(int), "Triangle is six ints");
            static_assert(sizeof(vec2) == 2 * sizeof(float), "vec2 is two floats");
            PT_CHECK(pt_scene_add_mesh(this->handle, pos.data(), nrm.data(), (int)nv, (const int32_t *)e->triangle_data(), (int)nt,
                                       (const float *)e->uv_data(), (int)e->uv_count(), mat, e->wants_octree() ? MAX_OCTREE_DEPTH : 0,
                                       nullptr));
        }
        e->on_device = true;
    }
    if (has_hdri)
        PT_CHECK(pt_scene_set_hdri(this->handle, this->hdri.data(), this->hdri_width, this->hdri_height, this->hdri_exposure,
                                   this->hdri_contrast, this->hdri_rot_offset, this->use_smooth_hdri ? 1 : 0));
    PT_CHECK(pt_scene_commit(this->handle));
    this->on_device = true;
    this->n_entities = this->entities.size();
}

void Scene::free_from_device() {       /* src/scene.cu:95-110 */
    if (!this->on_device) return;
    PT_CHECK(pt_scene_destroy(this->handle));
    this->handle = nullptr;
    for (Entity *e : this->entities) {
        e->on_device = false;
        e->material->on_device = false;
    }
    this->on_device = false;
}

/* ================================================================== renderer */
void Renderer::set_samples_per_pixel(unsigned int spp) {      /* src/renderer.cu:181-183 */
    unsigned int per = spp / (this->n_blocks_per_tile * this->n_split_buffers);
    this->n_samples_per_pass = per > 1 ? per : 1;
}

/* src/renderer.cu:192-312: upload, trace n_split * n_blocks * n_samples_per_pass samples per pixel
 * into n_split split buffers, compound, then the fixed post chain -- all device resident here. */
Image Renderer::render(const Camera &camera, Scene &scene) {
    const int64_t n_pixels = (int64_t)camera.resolution.x * camera.resolution.y;
    const unsigned int spp = this->n_split_buffers * this->n_blocks_per_tile * this->n_samples_per_pass;
    std::cout << "copying scene to device..." << std::endl;
    scene.copy_to_device();
    std::cout << "done!" << std::endl;
    const int dev = scene.device;
    char name[256];
    if (pt_device_name(dev, name, sizeof name) == 0) printf("Device %d: \"%s\"\n", dev, name);

    void *d_split = nullptr, *d_sum = nullptr, *d_out = nullptr;
    const size_t img_bytes = (size_t)n_pixels * sizeof(vec3);
    PT_CHECK(pt_malloc(dev, img_bytes * this->n_split_buffers, &d_split));
    PT_CHECK(pt_malloc(dev, img_bytes, &d_sum));
    PT_CHECK(pt_malloc(dev, img_bytes, &d_out));
    PT_CHECK(pt_memset(dev, d_split, 0, img_bytes * this->n_split_buffers, nullptr));    /* the reference never clears it */

    pt_camera cam;
    cam.position[0] = camera.position.x; cam.position[1] = camera.position.y; cam.position[2] = camera.position.z;
    cam.direction[0] = camera.direction.x; cam.direction[1] = camera.direction.y; cam.direction[2] = camera.direction.z;
    cam.up[0] = camera.up.x; cam.up[1] = camera.up.y; cam.up[2] = camera.up.z;
    cam.res_x = camera.resolution.x; cam.res_y = camera.resolution.y;
    cam.focal_length = camera.focal_length; cam.aperture = camera.aperture; cam.focus_distance = camera.focus_distance;

    std::cout << "start render" << std::endl;
    PT_CHECK(pt_render_accumulate(scene.device_scene(), &cam, this->seed, 0, (int)spp, (int)this->n_split_buffers, (float *)d_split,
                                  nullptr, nullptr));
    PT_CHECK(pt_compound_buffers(dev, (const float *)d_split, (float *)d_sum, n_pixels, (int)this->n_split_buffers, nullptr));
    PT_CHECK(pt_post_chain(dev, (const float *)d_sum, (float *)d_out, camera.resolution.x, camera.resolution.y, spp, nullptr));

    std::vector<vec3> result_pixels((size_t)n_pixels);
    PT_CHECK(pt_download(dev, result_pixels.data(), d_out, img_bytes, nullptr));
    this->last_hdr_sum.resize((size_t)n_pixels);
    PT_CHECK(pt_download(dev, this->last_hdr_sum.data(), d_sum, img_bytes, nullptr));
    this->last_samples_per_pixel = spp;
    PT_CHECK(pt_free(dev, d_split));
    PT_CHECK(pt_free(dev, d_sum));
    PT_CHECK(pt_free(dev, d_out));

    std::cout << "freeing scene from device..." << std::endl;
    scene.free_from_device();
    std::cout << "done!" << std::endl;
    return Image(result_pixels, camera.resolution);
}

/* ================================================================== post-processing */
void Kernel::make_gaussian(unsigned int _size, float sigma) {
    this->size = _size;
    this->taps.assign((size_t)(2 * _size + 1) * (2 * _size + 1), 0.0f);
    PT_CHECK(pt_make_gaussian(_size, sigma, this->taps.data()));
}
void Kernel::make_mean(unsigned int _size) {
    this->size = _size;
    this->taps.assign((size_t)(2 * _size + 1) * (2 * _size + 1), 0.0f);
    PT_CHECK(pt_make_mean(_size, this->taps.data()));
}
void Kernel::print() {       /* src/post_processing.cu:11-26 */
    float sum = 0.0f;
    for (size_t y = 0; y < size * 2 + 1; y++) {
        for (size_t x = 0; x < size * 2 + 1; x++) {
            printf("%d ", int(10 * taps[y * (2 * size + 1) + x]));
            sum += taps[y * (2 * size + 1) + x];
        }
        printf("\n");
    }
    printf("sum: %g\n", sum);
}

namespace {
/* runs one device image operation on a host vector: upload, op, download (the reference's
 * apply_filter does the same round trip, src/post_processing.cu:105-131) */
template <typename F>
void on_device(std::vector<vec3> &buf, F op) {
    const size_t bytes = buf.size() * sizeof(vec3);
    void *d = nullptr;
    PT_CHECK(pt_malloc(0, bytes, &d));
    PT_CHECK(pt_upload(0, d, buf.data(), bytes, nullptr));
    op((float *)d);
    PT_CHECK(pt_download(0, buf.data(), d, bytes, nullptr));
    PT_CHECK(pt_free(0, d));
}
} // namespace

void apply_filter(std::vector<vec3> &buf, ivec2 resolution, const Kernel &kernel, FilterType filter_type) {
    const size_t bytes = buf.size() * sizeof(vec3);
    void *d_in = nullptr, *d_out = nullptr;
    PT_CHECK(pt_malloc(0, bytes, &d_in));
    PT_CHECK(pt_malloc(0, bytes, &d_out));
    PT_CHECK(pt_upload(0, d_in, buf.data(), bytes, nullptr));
    PT_CHECK(pt_apply_filter(0, (const float *)d_in, (float *)d_out, resolution.x, resolution.y, kernel.taps.data(), (int)kernel.size,
                             filter_type == BILATERAL ? 1 : 0, nullptr));
    PT_CHECK(pt_download(0, buf.data(), d_out, bytes, nullptr));
    PT_CHECK(pt_free(0, d_in));
    PT_CHECK(pt_free(0, d_out));
}

void compound_buffers(std::vector<vec3> &out_image, const std::vector<vec3> &in_buf, unsigned int n_split_buffers) {
    const size_t n = out_image.size();      /* the reference uses capacity(); size() is what it means */
    assert(in_buf.size() >= n * n_split_buffers);
    void *d_in = nullptr, *d_out = nullptr;
    PT_CHECK(pt_malloc(0, n * n_split_buffers * sizeof(vec3), &d_in));
    PT_CHECK(pt_malloc(0, n * sizeof(vec3), &d_out));
    PT_CHECK(pt_upload(0, d_in, in_buf.data(), n * n_split_buffers * sizeof(vec3), nullptr));
    PT_CHECK(pt_compound_buffers(0, (const float *)d_in, (float *)d_out, (int64_t)n, (int)n_split_buffers, nullptr));
    PT_CHECK(pt_download(0, out_image.data(), d_out, n * sizeof(vec3), nullptr));
    PT_CHECK(pt_free(0, d_in));
    PT_CHECK(pt_free(0, d_out));
}

void normalize_image(std::vector<vec3> &buf, unsigned int n_samples_per_pixel) {
    on_device(buf, [&](float *d) { PT_CHECK(pt_normalize_image(0, d, (int64_t)buf.size(), n_samples_per_pixel, nullptr)); });
}
void gamma_correct(std::vector<vec3> &buf, float gamma) {
    on_device(buf, [&](float *d) { PT_CHECK(pt_gamma_correct(0, d, (int64_t)buf.size(), gamma, nullptr)); });
}
void apply_aces(std::vector<vec3> &buf) {
    on_device(buf, [&](float *d) { PT_CHECK(pt_apply_aces(0, d, (int64_t)buf.size(), nullptr)); });
}
void image_add(std::vector<vec3> &buf, const std::vector<vec3> &layer, float opacity) {
    assert(buf.size() == layer.size());
    const size_t bytes = buf.size() * sizeof(vec3);
    void *d_l = nullptr;
    PT_CHECK(pt_malloc(0, bytes, &d_l));
    PT_CHECK(pt_upload(0, d_l, layer.data(), bytes, nullptr));
    on_device(buf, [&](float *d) { PT_CHECK(pt_image_add(0, d, (const float *)d_l, (int64_t)buf.size(), opacity, nullptr)); });
    PT_CHECK(pt_free(0, d_l));
}
std::vector<vec3> apply_threshold(const std::vector<vec3> &in_buf, float threshold) {
    std::vector<vec3> result(in_buf.size());
    const size_t bytes = in_buf.size() * sizeof(vec3);
    void *d_in = nullptr, *d_out = nullptr;
    PT_CHECK(pt_malloc(0, bytes, &d_in));
    PT_CHECK(pt_malloc(0, bytes, &d_out));
    PT_CHECK(pt_upload(0, d_in, in_buf.data(), bytes, nullptr));
    PT_CHECK(pt_apply_threshold(0, (const float *)d_in, (float *)d_out, (int64_t)in_buf.size(), threshold, nullptr));
    PT_CHECK(pt_download(0, result.data(), d_out, bytes, nullptr));
    PT_CHECK(pt_free(0, d_in));
    PT_CHECK(pt_free(0, d_out));
    return result;
}

/* ================================================================== io */
void save_ppm(const std::string &filepath, const Image &img) {      /* src/io.cu:15-39 */
    FILE *fp = fopen(filepath.c_str(), "w");
    if (!fp) {
        fprintf(stderr, "cannot write %s\n", filepath.c_str());
        exit(1);
    }
    fputs("P6\n", fp);
    fputs("# Generated by ptamd\n", fp);
    fprintf(fp, "%d %d\n", img.resolution.x, img.resolution.y);
    fprintf(fp, "%d\n", 255);
    for (int i = 0; i < img.resolution.x * img.resolution.y; i++) {
        unsigned char color[3];
        color[0] = (unsigned char)(((img.buf[i].x < 0.999f) ? img.buf[i].x : 0.999f) * 255);
        color[1] = (unsigned char)(((img.buf[i].y < 0.999f) ? img.buf[i].y : 0.999f) * 255);
        color[2] = (unsigned char)(((img.buf[i].z < 0.999f) ? img.buf[i].z : 0.999f) * 255);
        fwrite(color, sizeof(char), 3, fp);
    }
    fclose(fp);
}
