/* Builds the octree of an OBJ mesh with the library's host-side builder under a host sanitizer
 * (the test compiles this file together with csrc/octree_build.cpp, once with
 * -fsanitize=address,undefined and once with -fsanitize=thread) and prints the structure counts.
 *   octree_sanitize <mesh.obj> <depth> */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "octree_build.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    std::ifstream in(argv[1]);
    std::vector<float> pos;
    std::vector<int32_t> tri;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        ss >> tag;
        if (tag == "v") {
            float x, y, z;
            ss >> x >> y >> z;
            pos.insert(pos.end(), {x, y, z});
        } else if (tag == "f") {
            for (int k = 0; k < 3; k++) {
                std::string c;
                ss >> c;
                tri.push_back(std::atoi(c.c_str()) - 1);
            }
            tri.insert(tri.end(), {-1, -1, -1});
        }
    }
    const int64_t nt = (int64_t)tri.size() / 6;
    float mn[3], mx[3];
    ptamd::padded_aabb(pos.data(), (int64_t)pos.size() / 3, mn, mx);
    std::vector<float> normals(pos.size());
    ptamd::compute_vertex_normals(pos.data(), (int64_t)pos.size() / 3, tri.data(), nt, normals.data());
    ptamd::FlatOctree t;
    ptamd::build_octree(pos.data(), tri.data(), nt, mn, mx, std::atoi(argv[2]), t);
    std::vector<int32_t> dump;
    ptamd::dump_octree(t, dump);
    printf("%lld %lld %lld", (long long)t.nodes.size(), (long long)t.leaf_tris.size(), (long long)t.max_leaf);
    for (int64_t n : t.nodes_per_depth) printf(" %lld", (long long)n);
    printf(" %zu\n", dump.size());
    return 0;
}
