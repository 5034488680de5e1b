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
#include <thread>

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
namespace {

/* componentwise bounds of a vertex array (min / max select one of their arguments: no rounding) */
struct Extent {
    vec3 lo = vec3(FLT_MAX), hi = vec3(-FLT_MAX);
    void include(const vec3 &p) {
        lo = vec3(std::fmin(lo.x, p.x), std::fmin(lo.y, p.y), std::fmin(lo.z, p.z));
        hi = vec3(std::fmax(hi.x, p.x), std::fmax(hi.y, p.y), std::fmax(hi.z, p.z));
    }
};

/* a 3x3 matrix as three rows; apply() forms each output component as dot(v, row): x*rx + y*ry + z*rz in that
 * order, the rounding sequence of the reference's per-vertex rotation (src/entity.cu:244-256) */
struct Rows3 {
    vec3 row[3];
    vec3 apply(const vec3 &v) const { return vec3(dot(v, row[0]), dot(v, row[1]), dot(v, row[2])); }
};

/* R = Rz(rot.z) * Ry(rot.y) * Rx(rot.x) with the sines and cosines evaluated in double and narrowed
 * (src/entity.cu:232-242: the unqualified cos / sin of that translation unit are the double overloads) */
Rows3 euler_zyx(const vec3 &rot) {
    const float cz = (float)std::cos((double)rot.z), sz = (float)std::sin((double)rot.z);
    const float cy = (float)std::cos((double)rot.y), sy = (float)std::sin((double)rot.y);
    const float cx = (float)std::cos((double)rot.x), sx = (float)std::sin((double)rot.x);
    Rows3 m;
    m.row[0] = vec3(cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx);
    m.row[1] = vec3(sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx);
    m.row[2] = vec3(-sy, cy * sx, cy * cx);
    return m;
}

}  // namespace

bool AABB::intersects_triangle(vec3 u0, vec3 u1, vec3 u2) {
    const float lo[3] = {min.x, min.y, min.z}, hi[3] = {max.x, max.y, max.z};
    const float a[3] = {u0.x, u0.y, u0.z}, b[3] = {u1.x, u1.y, u1.z}, c[3] = {u2.x, u2.y, u2.z};
    int overlaps = 0;
    PT_CHECK(pt_box_intersects_triangle(lo, hi, a, b, c, 0, &overlaps));
    return overlaps != 0;
}

/* ================================================================== pointer octree */
Octree::~Octree() {
    for (Octree *child : children) delete child;
}

void Octree::pretty_print(int child_nr) {       /* src/octree.cu:9-19: one line per node, indented by depth */
    printf("%*s[%d] : ", 2 * depth, "", child_nr);
    for (int t : triangle_indices) printf("%d, ", t);
    printf("\n");
    for (int i = 0; i < 8; i++)
        if (children[i]) children[i]->pretty_print(i);
}

void Octree::insert_triangle(vec3 v0, vec3 v1, vec3 v2, size_t triangle_idx) {     /* src/octree.cu:83-127 */
    if (depth == leaf_depth) {
        triangle_indices.push_back((int)triangle_idx);
        n_triangle_indices++;
        return;
    }
    /* child i takes the upper half in x / y / z where bit 2 / 1 / 0 of i is set; its box is padded again by the
     * AABB constructor, like every box of the tree */
    const vec3 lo = region.min, hi = region.max;
    const vec3 mid(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
    bool placed = false;
    for (int i = 0; i < 8; i++) {
        const bool ux = i & 4, uy = i & 2, uz = i & 1;
        AABB box(vec3(ux ? mid.x : lo.x, uy ? mid.y : lo.y, uz ? mid.z : lo.z),
                 vec3(ux ? hi.x : mid.x, uy ? hi.y : mid.y, uz ? hi.z : mid.z));
        if (!box.intersects_triangle(v0, v1, v2)) continue;
        placed = true;
        if (!children[i]) children[i] = new Octree(box, depth + 1, leaf_depth);
        children[i]->insert_triangle(v0, v1, v2, triangle_idx);
    }
    if (!placed) std::cout << "Miss triangle!" << std::endl;
}

void Octree::insert_triangles(Vertex *vertices, Triangle *triangles, size_t n_triangles) {     /* src/octree.cu:132-142 */
    for (size_t t = 0; t < n_triangles; t++) {
        const Triangle &tri = triangles[t];
        insert_triangle(vertices[tri.idx_a].position, vertices[tri.idx_b].position, vertices[tri.idx_c].position, t);
    }
}

void AABB::recalculate(Vertex *vertices, int n_vertices) {      /* src/aabb.cu:7-22: padded bounds of the vertices */
    Extent ext;
    for (const Vertex *v = vertices; v != vertices + n_vertices; ++v) ext.include(v->position);
    *this = AABB(ext.lo, ext.hi);
}

Entity::Entity(const std::string &path, Material *material) {   /* src/entity.cu:13-148 */
    ObjData obj;
    if (!parse_obj(path, obj) || obj.n_shapes != 1) {
        std::cout << "cannot load " << path << " (missing, or not exactly one shape)" << std::endl;
        exit(1);
    }
    this->n_vertices = obj.v.size() / 3;
    this->n_triangles = obj.face_sizes.size();
    this->n_uvs = obj.vt.size() / 2;
    const size_t n_file_normals = obj.vn.size() / 3;
    this->vertices = new Vertex[this->n_vertices];
    this->triangles = new Triangle[this->n_triangles];
    if (this->n_uvs) {
        this->uvs = new vec2[this->n_uvs];
        for (size_t t = 0; t < this->n_uvs; t++) this->uvs[t] = vec2(obj.vt[2 * t], obj.vt[2 * t + 1]);
    }

    Extent ext;
    for (size_t v = 0; v < this->n_vertices; v++) {
        Vertex &out = this->vertices[v];
        out.position = vec3(obj.v[3 * v], obj.v[3 * v + 1], obj.v[3 * v + 2]);
        out.normal = vec3(0.0f);
        /* the reference pairs vn entries with vertices by VERTEX index (src/entity.cu:58-65, a known flaw of
         * it); kept, bounded by the number of normals present */
        if (v < n_file_normals) out.normal = vec3(obj.vn[3 * v], obj.vn[3 * v + 1], obj.vn[3 * v + 2]).normalized();
        ext.include(out.position);
    }

    const int *corner = obj.idx.data();       /* per face corner: vertex index, texcoord index */
    for (size_t f = 0; f < this->n_triangles; f++, corner += 6) {
        if (obj.face_sizes[f] != 3) {
            std::cerr << "OBJ file faces must be triangles" << std::endl;
            exit(1);
        }
        const Triangle tr(corner[0], corner[2], corner[4], corner[1], corner[3], corner[5]);
        for (int vi : {tr.idx_a, tr.idx_b, tr.idx_c})
            if (vi < 0 || (size_t)vi >= this->n_vertices) {
                std::cerr << "OBJ face references a missing vertex" << std::endl;
                exit(1);
            }
        this->triangles[f] = tr;
    }
    if (n_file_normals == 0) {
        /* no vn in the file: every vertex gets the normalised sum of the unit normals of its faces, faces taken
         * in file order (src/entity.cu:110-133; the order fixes the fp32 sums) */
        for (size_t f = 0; f < this->n_triangles; f++) {
            Vertex *corner_v[3] = {&vertices[triangles[f].idx_a], &vertices[triangles[f].idx_b], &vertices[triangles[f].idx_c]};
            const vec3 face = cross(corner_v[1]->position - corner_v[0]->position, corner_v[2]->position - corner_v[0]->position).normalized();
            for (Vertex *v : corner_v) v->normal = v->normal + face;
        }
        for (size_t v = 0; v < this->n_vertices; v++) this->vertices[v].normal.normalize();
    }

    this->aabb = AABB(ext.lo, ext.hi);
    this->center = (ext.lo + ext.hi) / 2;           /* midpoint of the unpadded box (src/entity.cu:139-143) */
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

void Entity::construct_octree(int depth) {
    if (depth < 1 || depth > OCTREE_DEPTH_LIMIT) {
        std::cerr << "construct_octree: depth must be in [1, " << OCTREE_DEPTH_LIMIT << "]" << std::endl;
        exit(1);
    }
    this->requested_octree_depth = depth;
}

void Entity::print() {      /* src/entity.cu:175-180: centre, then the padded extent per axis */
    std::cout << "center: " << this->center << std::endl;
    const float lo[3] = {aabb.min.x, aabb.min.y, aabb.min.z}, hi[3] = {aabb.max.x, aabb.max.y, aabb.max.z};
    for (int axis = 0; axis < 3; axis++) std::cout << "xyz"[axis] << ": " << lo[axis] << " to " << hi[axis] << std::endl;
}

/* applies `f` to every vertex of a mesh and refreshes its bounding box */
template <typename F>
void Entity::transform_vertices(F f) {
    for (size_t i = 0; i < this->n_vertices; i++) f(this->vertices[i]);
    this->aabb.recalculate(this->vertices, (int)this->n_vertices);
}

void Entity::scale(float factor) {      /* src/entity.cu:185-202: about the centre; normals keep their direction */
    if (this->shape == SPHERE) {
        this->radius *= factor;
        return;
    }
    const vec3 c = this->center;
    transform_vertices([&](Vertex &v) { v.position = ((v.position - c) * factor) + c; });
}

void Entity::translate(vec3 delta) {    /* src/entity.cu:207-222: the centre moves along */
    this->center = this->center + delta;
    if (this->shape == TRIANGLE_MESH) transform_vertices([&](Vertex &v) { v.position = v.position + delta; });
}

void Entity::rotate(vec3 rot) {         /* src/entity.cu:227-258: about the centre, normals turn with the mesh */
    if (this->shape == SPHERE) return;
    const Rows3 R = euler_zyx(rot);
    const vec3 c = this->center;
    transform_vertices([&](Vertex &v) {
        v.position = R.apply(v.position - c) + c;
        v.normal = R.apply(v.normal);
    });
}

namespace {
/* positions / normals of a mesh entity as the flat float arrays the C ABI takes */
void flatten_vertices(const Entity &e, std::vector<float> &pos, std::vector<float> &nrm) {
    const size_t nv = e.vertex_count();
    pos.resize(3 * nv);
    nrm.resize(3 * nv);
    for (size_t i = 0; i < nv; i++) {
        const Vertex &v = e.vertex_data()[i];
        pos[3 * i] = v.position.x; pos[3 * i + 1] = v.position.y; pos[3 * i + 2] = v.position.z;
        nrm[3 * i] = v.normal.x; nrm[3 * i + 1] = v.normal.y; nrm[3 * i + 2] = v.normal.z;
    }
}
static_assert(sizeof(Triangle) == 6 * sizeof(int), "Triangle is six ints");
static_assert(sizeof(vec2) == 2 * sizeof(float), "vec2 is two floats");
}  // namespace

Entity::OctreeStatistics Entity::octree_statistics() const {
    OctreeStatistics st;
    if (this->shape != TRIANGLE_MESH || this->requested_octree_depth == 0) return st;
    pt_scene *host_only = nullptr;
    PT_CHECK(pt_scene_create(-1, &host_only));
    pt_material_desc md;
    memset(&md, 0, sizeof md);
    int mat = -1, ent = -1;
    PT_CHECK(pt_scene_add_material(host_only, &md, &mat));
    std::vector<float> pos, nrm;
    flatten_vertices(*this, pos, nrm);
    PT_CHECK(pt_scene_add_mesh(host_only, pos.data(), nrm.data(), (int)n_vertices, (const int32_t *)triangles, (int)n_triangles,
                               (const float *)uvs, (int)n_uvs, mat, requested_octree_depth, &ent));
    pt_octree_info info;
    PT_CHECK(pt_scene_octree_info(host_only, ent, &info));
    st.nodes_per_depth.assign(info.nodes_per_depth, info.nodes_per_depth + info.max_depth + 1);
    st.n_nodes = info.n_nodes;
    st.n_leaf_refs = info.n_leaf_refs;
    st.max_leaf = info.max_leaf;
    PT_CHECK(pt_scene_destroy(host_only));
    return st;
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
pt_scene *Scene::build_on(int dev) const {
    pt_scene *h = nullptr;
    PT_CHECK(pt_scene_create(dev, &h));
    std::map<const Material *, int> mat_ids;      /* materials may be shared between entities (src/entity.cu:363) */
    for (const Entity *e : this->entities) {
        const Material *m = e->material;
        if (mat_ids.count(m)) continue;
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
        PT_CHECK(pt_scene_add_material(h, &d, &id));
        mat_ids[m] = id;
    }
    std::vector<float> pos, nrm;
    for (const Entity *e : this->entities) {
        const int mat = mat_ids[e->material];
        if (e->shape == SPHERE) {
            const vec3 c = e->get_center();
            const float cc[3] = {c.x, c.y, c.z};
            PT_CHECK(pt_scene_add_sphere(h, cc, e->get_radius(), mat, nullptr));
        } else {
            flatten_vertices(*e, pos, nrm);
            PT_CHECK(pt_scene_add_mesh(h, pos.data(), nrm.data(), (int)e->vertex_count(), (const int32_t *)e->triangle_data(),
                                       (int)e->triangle_count(), (const float *)e->uv_data(), (int)e->uv_count(), mat, e->octree_depth(),
                                       nullptr));
        }
    }
    if (has_hdri)
        PT_CHECK(pt_scene_set_hdri(h, this->hdri.data(), this->hdri_width, this->hdri_height, this->hdri_exposure, this->hdri_contrast,
                                   this->hdri_rot_offset, this->use_smooth_hdri ? 1 : 0));
    PT_CHECK(pt_scene_commit(h));
    return h;
}

void Scene::copy_to_device() {
    if (this->on_device) this->free_from_device();
    this->handle = this->build_on(this->device);
    for (Entity *e : this->entities) {
        e->on_device = true;
        e->material->on_device = true;
    }
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

namespace {

pt_camera to_abi(const Camera &camera) {
    pt_camera cam;
    cam.position[0] = camera.position.x; cam.position[1] = camera.position.y; cam.position[2] = camera.position.z;
    cam.direction[0] = camera.direction.x; cam.direction[1] = camera.direction.y; cam.direction[2] = camera.direction.z;
    cam.up[0] = camera.up.x; cam.up[1] = camera.up.y; cam.up[2] = camera.up.z;
    cam.res_x = camera.resolution.x; cam.res_y = camera.resolution.y;
    cam.focal_length = camera.focal_length; cam.aperture = camera.aperture; cam.focus_distance = camera.focus_distance;
    return cam;
}

/* one GPU's share of a render: samples [begin, end) of every pixel into fresh split buffers, compounded into
 * *d_sum_out (device memory on `dev`, owned by the caller afterwards) */
/* returns 0, or the failed call's status with its message in `err` (the C ABI's error string is per thread: a worker
 * thread reports through the string, and the caller exits after joining every thread) */
int render_share(pt_scene *scene, int dev, const pt_camera &cam, unsigned seed, int begin, int end, int n_split, void **d_sum_out, std::string &err) {
    const int64_t n_pixels = (int64_t)cam.res_x * cam.res_y;
    const size_t img_bytes = (size_t)n_pixels * sizeof(vec3);
    void *d_split = nullptr, *d_sum = nullptr;
    int rc = 0;
#define PT_TRY(call) if (!rc && (rc = (call)) != 0) err = std::string(#call) + " failed: " + pt_last_error()
    PT_TRY(pt_malloc(dev, img_bytes * n_split, &d_split));
    PT_TRY(pt_malloc(dev, img_bytes, &d_sum));
    PT_TRY(pt_memset(dev, d_split, 0, img_bytes * n_split, nullptr));    /* the reference never clears its buffer */
    PT_TRY(pt_render_accumulate(scene, &cam, seed, begin, end, n_split, (float *)d_split, nullptr, nullptr));
    PT_TRY(pt_compound_buffers(dev, (const float *)d_split, (float *)d_sum, n_pixels, n_split, nullptr));
    PT_TRY(pt_stream_synchronize(dev, nullptr));
    if (d_split) (void)pt_free(dev, d_split);
    *d_sum_out = d_sum;
    return rc;
}

}  // namespace

/* src/renderer.cu:192-312: upload, trace n_split * n_blocks * n_samples_per_pass samples per pixel into n_split
 * split buffers, compound, then the fixed post chain -- all device resident here, on one GPU or several. */
Image Renderer::render(const Camera &camera, Scene &scene) {
    const int64_t n_pixels = (int64_t)camera.resolution.x * camera.resolution.y;
    const size_t img_bytes = (size_t)n_pixels * sizeof(vec3);
    const unsigned int spp = this->n_split_buffers * this->n_blocks_per_tile * this->n_samples_per_pass;
    const int n_split = (int)this->n_split_buffers;
    const pt_camera cam = to_abi(camera);
    std::vector<int> devs = this->devices;
    if (devs.empty()) devs.push_back(scene.device);
    const int n_dev = (int)devs.size(), root = devs[0];

    std::cout << "copying scene to device..." << std::endl;
    std::vector<pt_scene *> handles(n_dev, nullptr);
    if (n_dev == 1 && root == scene.device) {
        scene.copy_to_device();
        handles[0] = scene.device_scene();
    }
    std::cout << "done!" << std::endl;
    char name[256];
    for (int d : devs)
        if (pt_device_name(d, name, sizeof name) == 0) printf("Device %d: \"%s\"\n", d, name);

    std::cout << "start render" << std::endl;
    std::vector<void *> d_sums(n_dev, nullptr);
    std::vector<std::string> errors(n_dev);
    std::vector<int> status(n_dev, 0);
    if (handles[0]) {
        status[0] = render_share(handles[0], root, cam, this->seed, 0, (int)spp, n_split, &d_sums[0], errors[0]);
    } else {
        /* The scene is flattened and its octrees are built ONCE, for the first GPU; every other GPU gets a clone that
         * shares that host-side description (pt_scene_clone) and only uploads it.  One host thread per GPU: upload,
         * render the GPU's contiguous share of the samples.  A thread that fails records why; nothing exits while
         * other threads are inside the runtime. */
        handles[0] = scene.build_on(devs[0]);
        std::vector<std::thread> workers;
        for (int r = 0; r < n_dev; r++)
            workers.emplace_back([&, r]() {
                if (r > 0) {
                    status[r] = pt_scene_clone(handles[0], devs[r], &handles[r]);
                    if (!status[r]) status[r] = pt_scene_commit(handles[r]);
                    if (status[r]) { errors[r] = std::string("copying the scene to device failed: ") + pt_last_error(); return; }
                }
                const int begin = (int)(((long long)spp * r) / n_dev), end = (int)(((long long)spp * (r + 1)) / n_dev);
                status[r] = render_share(handles[r], devs[r], cam, this->seed, begin, end, n_split, &d_sums[r], errors[r]);
            });
        for (std::thread &w : workers) w.join();
    }
    for (int r = 0; r < n_dev; r++)
        if (status[r]) {
            fprintf(stderr, "ptamd: device %d: %s\n", devs[r], errors[r].c_str());
            exit(status[r]);
        }
    /* root += the other GPUs' sums, in list order (fixed order: reproducible bits) */
    if (n_dev > 1)
        PT_CHECK(pt_reduce_sum(root, (float *)d_sums[0], n_dev - 1, devs.data() + 1, (const float *const *)(d_sums.data() + 1), n_pixels * 3));

    void *d_out = nullptr;
    PT_CHECK(pt_malloc(root, img_bytes, &d_out));
    PT_CHECK(pt_post_chain(root, (const float *)d_sums[0], (float *)d_out, camera.resolution.x, camera.resolution.y, spp, nullptr));
    std::vector<vec3> result_pixels((size_t)n_pixels);
    PT_CHECK(pt_download(root, result_pixels.data(), d_out, img_bytes, nullptr));
    this->last_hdr_sum.resize((size_t)n_pixels);
    PT_CHECK(pt_download(root, this->last_hdr_sum.data(), d_sums[0], img_bytes, nullptr));
    this->last_samples_per_pixel = spp;
    PT_CHECK(pt_free(root, d_out));
    for (int r = 0; r < n_dev; r++) PT_CHECK(pt_free(devs[r], d_sums[r]));

    std::cout << "freeing scene from device..." << std::endl;
    if (n_dev == 1 && root == scene.device) {
        scene.free_from_device();
    } else {
        for (pt_scene *h : handles) PT_CHECK(pt_scene_destroy(h));
    }
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
int g_post_device = 0;      /* GPU of the free post-processing functions */

/* runs one device image operation on a host vector: upload, op, download (the reference's
 * apply_filter does the same round trip, src/post_processing.cu:105-131) */
template <typename F>
void on_device(std::vector<vec3> &buf, F op) {
    const int dev = g_post_device;
    const size_t bytes = buf.size() * sizeof(vec3);
    void *d = nullptr;
    PT_CHECK(pt_malloc(dev, bytes, &d));
    PT_CHECK(pt_upload(dev, d, buf.data(), bytes, nullptr));
    op(dev, (float *)d);
    PT_CHECK(pt_download(dev, buf.data(), d, bytes, nullptr));
    PT_CHECK(pt_free(dev, d));
}

/* device copy of a host image on the post-processing GPU; freed by the destructor */
struct DeviceImage {
    int dev;
    void *p = nullptr;
    DeviceImage(const void *host, size_t bytes) : dev(g_post_device) {
        PT_CHECK(pt_malloc(dev, bytes, &p));
        if (host) PT_CHECK(pt_upload(dev, p, host, bytes, nullptr));
    }
    ~DeviceImage() { (void)pt_free(dev, p); }
    float *f() const { return (float *)p; }
};
} // namespace

void set_post_processing_device(int device) { g_post_device = device; }
int post_processing_device() { return g_post_device; }

void apply_filter(std::vector<vec3> &buf, ivec2 resolution, const Kernel &kernel, FilterType filter_type) {
    const size_t bytes = buf.size() * sizeof(vec3);
    DeviceImage in(buf.data(), bytes), out(nullptr, bytes);
    PT_CHECK(pt_apply_filter(in.dev, in.f(), out.f(), resolution.x, resolution.y, kernel.taps.data(), (int)kernel.size,
                             filter_type == BILATERAL ? 1 : 0, nullptr));
    PT_CHECK(pt_download(out.dev, buf.data(), out.p, bytes, nullptr));
}

void compound_buffers(std::vector<vec3> &out_image, const std::vector<vec3> &in_buf, unsigned int n_split_buffers) {
    const size_t n = out_image.size();      /* the reference uses capacity(); size() is what it means */
    assert(in_buf.size() >= n * n_split_buffers);
    DeviceImage in(in_buf.data(), n * n_split_buffers * sizeof(vec3)), out(nullptr, n * sizeof(vec3));
    PT_CHECK(pt_compound_buffers(in.dev, in.f(), out.f(), (int64_t)n, (int)n_split_buffers, nullptr));
    PT_CHECK(pt_download(out.dev, out_image.data(), out.p, n * sizeof(vec3), nullptr));
}

void normalize_image(std::vector<vec3> &buf, unsigned int n_samples_per_pixel) {
    on_device(buf, [&](int dev, float *d) { PT_CHECK(pt_normalize_image(dev, d, (int64_t)buf.size(), n_samples_per_pixel, nullptr)); });
}
void gamma_correct(std::vector<vec3> &buf, float gamma) {
    on_device(buf, [&](int dev, float *d) { PT_CHECK(pt_gamma_correct(dev, d, (int64_t)buf.size(), gamma, nullptr)); });
}
void apply_aces(std::vector<vec3> &buf) {
    on_device(buf, [&](int dev, float *d) { PT_CHECK(pt_apply_aces(dev, d, (int64_t)buf.size(), nullptr)); });
}
void image_add(std::vector<vec3> &buf, const std::vector<vec3> &layer, float opacity) {
    assert(buf.size() == layer.size());
    DeviceImage l(layer.data(), layer.size() * sizeof(vec3));
    on_device(buf, [&](int dev, float *d) { PT_CHECK(pt_image_add(dev, d, l.f(), (int64_t)buf.size(), opacity, nullptr)); });
}
std::vector<vec3> apply_threshold(const std::vector<vec3> &in_buf, float threshold) {
    std::vector<vec3> result(in_buf.size());
    const size_t bytes = in_buf.size() * sizeof(vec3);
    DeviceImage in(in_buf.data(), bytes), out(nullptr, bytes);
    PT_CHECK(pt_apply_threshold(in.dev, in.f(), out.f(), (int64_t)in_buf.size(), threshold, nullptr));
    PT_CHECK(pt_download(out.dev, result.data(), out.p, bytes, nullptr));
    return result;
}

/* ================================================================== io */
void save_ppm(const std::string &filepath, const Image &img) {      /* src/io.cu:15-39: binary P6, min(x, 0.999) * 255 truncated */
    const size_t n_pixels = (size_t)img.resolution.x * img.resolution.y;
    std::vector<unsigned char> bytes(3 * n_pixels);
    for (size_t i = 0; i < n_pixels; i++) {
        const float rgb[3] = {img.buf[i].x, img.buf[i].y, img.buf[i].z};
        for (int c = 0; c < 3; c++) bytes[3 * i + c] = (unsigned char)(std::fmin(rgb[c], 0.999f) * 255);
    }
    FILE *fp = fopen(filepath.c_str(), "wb");
    if (!fp) {
        fprintf(stderr, "cannot write %s\n", filepath.c_str());
        exit(1);
    }
    fprintf(fp, "P6\n# Generated by ptamd\n%d %d\n255\n", img.resolution.x, img.resolution.y);
    fwrite(bytes.data(), 1, bytes.size(), fp);
    fclose(fp);
}
