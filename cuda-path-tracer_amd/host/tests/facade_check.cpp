/*
 * facade_check -- drives the host C++ API (path_tracer.h) from the command line and dumps what it
 * produced as raw binary, so that the Python tests can compare it with the same scene prepared
 * through the ctypes binding.
 *
 *   facade_check mesh  <in.obj> <out.bin> [ops]      ops: s<f>  t<x>,<y>,<z>  r<x>,<y>,<z>
 *   facade_check hdr   <in.hdr> <out.bin>
 *   facade_check pnm   <in.ppm|pgm> <channels> <out.bin>
 *   facade_check teapot <utah.obj> <floor.obj> <w> <h> <spp> <out.bin> [octree depth [devices, e.g. 0,0]]   (needs a GPU)
 *   facade_check octree <in.obj> <depth>                                     (prints the tree's shape)
 *   facade_check ptree <in.obj> <depth> <out.bin>                            (the Octree class's pointer tree, dumped in pre-order)
 *   facade_check post  <w> <h> <out.bin>                                     (needs a GPU)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "path_tracer.h"
#include "ptamd.h"

static void put(FILE *f, const void *p, size_t n) {
    if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(2); }
}

static void apply_ops(Entity &e, int argc, char **argv, int first) {
    for (int i = first; i < argc; i++) {
        float a = 0, b = 0, c = 0;
        const char *s = argv[i];
        if (s[0] == 's' && sscanf(s + 1, "%f", &a) == 1) e.scale(a);
        else if (s[0] == 't' && sscanf(s + 1, "%f,%f,%f", &a, &b, &c) == 3) e.translate(vec3(a, b, c));
        else if (s[0] == 'r' && sscanf(s + 1, "%f,%f,%f", &a, &b, &c) == 3) e.rotate(vec3(a, b, c));
        else { fprintf(stderr, "bad op %s\n", s); exit(2); }
    }
}

static int cmd_mesh(int argc, char **argv) {
    Material m;
    Entity e(argv[2], &m);
    apply_ops(e, argc, argv, 4);
    FILE *f = fopen(argv[3], "wb");
    const int32_t hdr[3] = {(int32_t)e.vertex_count(), (int32_t)e.triangle_count(), (int32_t)e.uv_count()};
    put(f, hdr, sizeof hdr);
    put(f, e.vertex_data(), e.vertex_count() * sizeof(Vertex));
    put(f, e.triangle_data(), e.triangle_count() * sizeof(Triangle));
    if (e.uv_count()) put(f, e.uv_data(), e.uv_count() * sizeof(vec2));
    const AABB &bb = e.bounds();
    const vec3 c = e.get_center();
    const float tail[9] = {bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z, c.x, c.y, c.z};
    put(f, tail, sizeof tail);
    fclose(f);
    return 0;
}

static int cmd_hdr(char **argv) {
    std::vector<float> rgb;
    int w = 0, h = 0;
    if (!load_hdr_image(argv[2], rgb, w, h)) return 1;
    FILE *f = fopen(argv[3], "wb");
    const int32_t hdr[2] = {w, h};
    put(f, hdr, sizeof hdr);
    put(f, rgb.data(), rgb.size() * sizeof(float));
    fclose(f);
    return 0;
}

static int cmd_pnm(char **argv) {
    const int ch = atoi(argv[3]);
    ColorTexture c;
    GrayscaleTexture g;
    Texture *t = ch == 3 ? (Texture *)&c : (Texture *)&g;
    if (ch == 3) c.set(argv[2]); else g.set(argv[2]);
    if (!t->is_set()) return 1;
    FILE *f = fopen(argv[4], "wb");
    const int32_t hdr[2] = {t->resolution.x, t->resolution.y};
    put(f, hdr, sizeof hdr);
    put(f, t->data, (size_t)t->resolution.x * t->resolution.y * ch);
    fclose(f);
    return 0;
}

/* the teapot scene of the Python scene library, written against the class API */
static int cmd_teapot(int argc, char **argv) {
    Material m_floor(vec3(0.8f, 0.8f, 0.8f), vec3(0.0f), 0, 0, false);
    Material m_pot(vec3(0.8f, 0.6f, 0.3f), vec3(0.3f, 0.3f, 0.3f), 0, 0.5f, true);
    Material m_light(vec3(1, 1, 1), vec3(0.0f), 10, 0, false);
    Entity pot(argv[2], &m_pot);
    Entity floor(argv[3], &m_floor);
    Entity light(vec3(-150.0f, 160.0f, -40.0f), 40.0f, &m_light);
    floor.scale(20.0f);
    floor.translate(vec3(-205.3 - 27.5, -36.4, 82.5 - 28.5));
    if (argc > 8) pot.construct_octree(atoi(argv[8]));
    else pot.construct_octree();
    Scene scene;
    scene.add_entity(&pot);
    scene.add_entity(&floor);
    scene.add_entity(&light);
    const int w = atoi(argv[4]), h = atoi(argv[5]);
    Camera camera(vec3(-205.3f, 40.0f, -150.0f), vec3(0, -0.12, 1).normalized(), ivec2(w, h));
    camera.focal_length = 1.6f;
    Renderer renderer;
    renderer.set_samples_per_pixel((unsigned)atoi(argv[6]));
    if (argc > 9)
        for (const char *p = argv[9]; *p; p++)
            if (*p != ',') renderer.devices.push_back(*p - '0');
    Image img = renderer.render(camera, scene);
    FILE *f = fopen(argv[7], "wb");
    const int32_t hdr[3] = {w, h, (int32_t)renderer.last_samples_per_pixel};
    put(f, hdr, sizeof hdr);
    put(f, renderer.last_hdr_sum.data(), renderer.last_hdr_sum.size() * sizeof(vec3));
    put(f, img.buf.data(), img.buf.size() * sizeof(vec3));
    fclose(f);
    save_ppm(std::string(argv[7]) + ".ppm", img);
    uint64_t builds = 0;
    pt_octree_builds(&builds);
    printf("octrees built by the library: %llu\n", (unsigned long long)builds);
    return 0;
}

/* the post-processing functions one by one, in the order Renderer::render of the reference chains
 * them (src/renderer.cu:277-297), on a deterministic synthetic HDR image */
static int cmd_post(char **argv) {
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    std::vector<vec3> img((size_t)w * h);
    uint32_t s = 12345u;
    for (auto &p : img) {
        float c[3];
        for (int k = 0; k < 3; k++) {
            s = s * 1664525u + 1013904223u;
            c[k] = (float)(s >> 8) * (1.0f / 16777216.0f);
        }
        const bool hot = (s >> 5) % 97 == 0;
        p = vec3(c[0] * (hot ? 400.0f : 40.0f), c[1] * (hot ? 400.0f : 40.0f), c[2] * (hot ? 400.0f : 40.0f));
    }
    FILE *f = fopen(argv[4], "wb");
    const int32_t hdr[2] = {w, h};
    put(f, hdr, sizeof hdr);
    put(f, img.data(), img.size() * sizeof(vec3));
    normalize_image(img, 32);
    std::vector<vec3> bloom = apply_threshold(img, 3.0f);
    Kernel gaussian, tiny;
    gaussian.make_gaussian(32, 10.0f);
    apply_filter(bloom, ivec2(w, h), gaussian);
    image_add(img, bloom, 0.5f);
    apply_aces(img);
    gamma_correct(img, 2.2f);
    tiny.make_gaussian(2, 0.7f);
    apply_filter(img, ivec2(w, h), tiny, BILATERAL);
    put(f, img.data(), img.size() * sizeof(vec3));
    fclose(f);
    return 0;
}

static int cmd_octree(char **argv) {
    Material m;
    Entity e(argv[2], &m);
    e.construct_octree(atoi(argv[3]));
    const Entity::OctreeStatistics st = e.octree_statistics();
    printf("depth %d nodes %lld leaf_refs %lld max_leaf %lld per_depth", e.octree_depth(), st.n_nodes, st.n_leaf_refs, st.max_leaf);
    for (long long n : st.nodes_per_depth) printf(" %lld", n);
    printf("\n");
    return 0;
}

/* pointer octree built through the Octree class (root = the entity's box at depth 0, as Entity::construct_octree
 * roots it in the reference, src/entity.cu:167-170), written in the pre-order word format of pt_scene_octree_dump:
 * per node {depth, slot, child mask (inner) or 0, 0 (inner) or triangle count, triangle indices...} */
static void dump_pointer_tree(const Octree *n, int slot, int leaf_depth, std::vector<int32_t> &out) {
    int mask = 0;
    for (int i = 0; i < 8; i++) mask |= n->children[i] ? 1 << i : 0;
    out.push_back(n->depth); out.push_back(slot);
    if (n->depth == leaf_depth) {
        out.push_back(0); out.push_back((int32_t)n->triangle_indices.size());
        for (int t : n->triangle_indices) out.push_back(t);
        return;
    }
    out.push_back(mask); out.push_back(0);
    for (int i = 0; i < 8; i++)
        if (n->children[i]) dump_pointer_tree(n->children[i], i, leaf_depth, out);
}

static int cmd_ptree(char **argv) {
    Material m;
    Entity e(argv[2], &m);
    const int leaf_depth = atoi(argv[3]);
    Octree root(e.bounds(), 0, leaf_depth);
    root.insert_triangles(const_cast<Vertex *>(e.vertex_data()), const_cast<Triangle *>(e.triangle_data()), e.triangle_count());
    std::vector<int32_t> words;
    dump_pointer_tree(&root, 0, leaf_depth, words);
    FILE *f = fopen(argv[4], "wb");
    if (!f) return 1;
    put(f, words.data(), words.size() * sizeof(int32_t));
    fclose(f);
    AABB unit(vec3(0.0f), vec3(1.0f));          /* AABB::intersects_triangle on its own: one overlap, one miss */
    const bool in = unit.intersects_triangle(vec3(0.2f, 0.2f, 0.2f), vec3(0.8f, 0.3f, 0.2f), vec3(0.3f, 0.9f, 0.4f));
    const bool out = unit.intersects_triangle(vec3(2.0f, 2.0f, 2.0f), vec3(3.0f, 2.0f, 2.0f), vec3(2.0f, 3.0f, 2.0f));
    printf("words %zu overlap %d miss %d\n", words.size(), (int)in, (int)out);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 5 && !strcmp(argv[1], "ptree")) return cmd_ptree(argv);
    if (argc >= 4 && !strcmp(argv[1], "mesh")) return cmd_mesh(argc, argv);
    if (argc == 4 && !strcmp(argv[1], "hdr")) return cmd_hdr(argv);
    if (argc == 5 && !strcmp(argv[1], "pnm")) return cmd_pnm(argv);
    if (argc >= 8 && argc <= 10 && !strcmp(argv[1], "teapot")) return cmd_teapot(argc, argv);
    if (argc == 4 && !strcmp(argv[1], "octree")) return cmd_octree(argv);
    if (argc == 5 && !strcmp(argv[1], "post")) return cmd_post(argv);
    fprintf(stderr, "usage: see the header of facade_check.cpp\n");
    return 2;
}
