/*
 * North-star scene rendered through the host C++ API (cuda-path-tracer_amd/host/path_tracer.h):
 * a textured statue on its own, lit by an environment map, seen through a thin-lens camera.
 * The class API is the reference's (src/main.cu uses the same calls), so this file also shows what
 * a scene script looks like after the move: only the CUDA includes go away and texture files are
 * PPM/PGM.  Assets: `python -m ptamd.export_assets examples` writes procedural stand-ins.
 *
 *   pt_example [samples_per_pixel [width height [output.ppm [asset_dir]]]]
 */
#include <cstdlib>
#include <iostream>
#include <string>

#include "renderer.cuh"
#include "io.cuh"

namespace {

struct Options {
    unsigned spp = 2048;
    int width = 1920, height = 1080;
    std::string output = "output.ppm";
    std::string assets = "examples";
};

Options parse(int argc, char **argv) {
    Options o;
    if (argc > 1) o.spp = (unsigned)std::atoi(argv[1]);
    if (argc > 3) { o.width = std::atoi(argv[2]); o.height = std::atoi(argv[3]); }
    if (argc > 4) o.output = argv[4];
    if (argc > 5) o.assets = argv[5];
    return o;
}

}  // namespace

int main(int argc, char **argv) {
    const Options opt = parse(argc, argv);
    const std::string statue_dir = opt.assets + "/hcandersen/";

    /* surface: glossy dielectric with all three texture maps */
    Material bronze;
    bronze.specular = vec3(0.2f);
    bronze.smoothness = 1.0f;
    bronze.albedo_map.set(statue_dir + "textures/diff.ppm");
    bronze.normal_map.set(statue_dir + "textures/normal.ppm");
    bronze.smoothness_map.set(statue_dir + "textures/gloss.pgm");

    /* geometry: stand the statue upright, lift it, accelerate it */
    Entity statue(statue_dir + "source/80k.obj", &bronze);
    const float quarter_turn = 3.14159265f / 2.0f;
    statue.rotate(vec3(-quarter_turn, quarter_turn, 0.0f));
    statue.translate(vec3(0.0f, 22.65f, 0.0f));
    statue.construct_octree();
    statue.print();

    Scene world;
    world.add_entity(&statue);
    world.set_hdri(opt.assets + "/hdris/quattro_canti_4k.hdr");
    world.use_hdri_smoothing(true);
    world.rotate_hdri(0.85f);

    Camera eye(vec3(40.5f, 30.5f, -50.0f), vec3(-0.5f, 0.0f, 1.0f).normalized(), ivec2(opt.width, opt.height));
    eye.focal_length = 2.5f;
    eye.aperture = 1.0f;
    eye.focus_distance = 85.0f;

    Renderer renderer;
    renderer.set_samples_per_pixel(opt.spp);
    const Image picture = renderer.render(eye, world);
    std::cout << "rendered " << picture.resolution << " at " << renderer.last_samples_per_pixel << " spp" << std::endl;
    save_ppm(opt.output, picture);
    return 0;
}
