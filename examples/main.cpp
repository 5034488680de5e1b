/*
 * The scene script of the reference (src/main.cu:7-69) against this library's host API: the same
 * statements, minus the two CUDA includes.  The reference does not ship the statue, its texture
 * maps or the HDRI; `python -m ptamd.export_assets examples` writes procedural stand-ins under the
 * same paths (OBJ, binary PPM/PGM instead of JPEG, Radiance .hdr).
 *
 *   pt_example [samples_per_pixel [width height [output.ppm]]]
 */
#include <cstdlib>
#include <iostream>
#include "renderer.cuh"
#include "io.cuh"

int main(int argc, char **argv) {
    const unsigned spp = argc > 1 ? (unsigned)atoi(argv[1]) : 2048;
    const int width = argc > 3 ? atoi(argv[2]) : 1920;
    const int height = argc > 3 ? atoi(argv[3]) : 1080;
    const char *out = argc > 4 ? argv[4] : "output.ppm";

    // materials
    Material m_default;
    Material m_hc;
    m_hc.specular = vec3(0.2f);
    m_hc.smoothness = 1.0f;
    m_hc.albedo_map.set("examples/hcandersen/textures/diff.ppm");
    m_hc.smoothness_map.set("examples/hcandersen/textures/gloss.pgm");
    m_hc.normal_map.set("examples/hcandersen/textures/normal.ppm");

    // entities
    Entity hcandersen("examples/hcandersen/source/80k.obj", &m_hc);

    // transforms
    hcandersen.rotate(vec3(-3.14159265f / 2.0f, 3.14159265f / 2.0f, 0));
    hcandersen.translate(vec3(0, 22.65, 0));

    // construct octree
    hcandersen.construct_octree();

    // debug
    hcandersen.print();

    // add to scene
    Scene scene;
    scene.add_entity(&hcandersen);

    scene.use_hdri_smoothing(true);
    scene.set_hdri("examples/hdris/quattro_canti_4k.hdr");
    scene.rotate_hdri(0.85f);

    // camera
    ivec2 res = ivec2(width, height);
    Camera camera(vec3(40.5f, 30.5f, -50), vec3(-0.5, 0, 1).normalized(), res);
    camera.focal_length = 2.5f;
    camera.aperture = 1;
    camera.focus_distance = 85;

    // render
    Renderer renderer;
    renderer.set_samples_per_pixel(spp);
    Image image = renderer.render(camera, scene);
    std::cout << image.resolution << std::endl;
    save_ppm(out, image);
}
