// main.cpp — the `rtx` command line: src/main.rs:63-175 on top of the MI355X trace loop.
// Usage: rtx scene_config.json   (the reference's JSON schema, sample-config.json runs unmodified)
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "rt.h"

using namespace rt;

int main(int argc, char **argv) {
    auto start = std::chrono::steady_clock::now();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count(); };
    if (argc != 2) {                                                     // main.rs:51-61
        std::fprintf(stderr, "Usage: %s scene_config.json\n", argv[0]);
        return 1;
    }
    std::ifstream in(argv[1]);
    if (!in) {
        std::fprintf(stderr, "No such file or directory (os error 2)\n");
        return 1;
    }
    std::stringstream ss;
    ss << in.rdbuf();
    JobConfig config;
    std::string err = parse_job_config(ss.str(), &config);
    if (!err.empty()) {                                                  // main.rs:87-101
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    std::printf("Job started...\nLoading scene...\n");
    auto scene_config = scene_by_name(config.name, config.asset_root);
    if (!scene_config) {                                                 // main.rs:105-111
        std::fprintf(stderr, "unknown scene ``%s''\n", config.name.c_str());
        return 1;
    }
    std::shared_ptr<const Scene> shared_scene = scene_config->get_scene();
    Camera camera = config.animating ? scene_config->get_animation_camera(config.size[0], config.size[1], config.fov)
                                     : scene_config->get_camera(config.size[0], config.size[1], config.fov);
    double scene_time = secs();
    std::printf("Scene loaded (%.3fs)...\n", scene_time);
    Renderer renderer;
    renderer.options = RenderOptions{config.reflect_depth, config.refract_depth, config.shadow_samples, config.gloss_samples,
                                     config.pixel_samples};
    renderer.rng_mode = config.rng_mode == "center" ? RTX_RNG_CENTER : RTX_RNG_PHILOX;
    renderer.seed = config.seed;
    renderer.device = config.device;
    try {
        if (config.animating) {                                          // main.rs:141-158
            Animator animator;
            animator.fps = config.fps;
            animator.animate_from = config.time_slice[0];
            animator.animate_to = config.time_slice[1];
            animator.starting_frame_number = config.starting_frame_number;
            animator.renderer = renderer;
            std::printf("Animating - FPS: %g, start: %gs, end:%gs, starting frame: %u\n", animator.fps, animator.animate_from,
                        animator.animate_to, animator.starting_frame_number);
            animator.animate(camera, shared_scene, config.output_file);
            std::printf("Render done (%.3fs)\n", secs() - scene_time);
        } else {                                                         // main.rs:159-174
            std::printf("Rendering on HIP device %d...\n", config.device);
            Surface image = renderer.render(camera, shared_scene);
            double render_time = secs();
            const rtx_stats &st = renderer.stats.last;
            std::printf("Render done (%.3fs; device %.2f ms, %.1f Mrays/s)...\nWriting file...\n", render_time - scene_time,
                        st.ms_device, renderer.stats.rays_total / (st.ms_device * 1e3));
            std::string out_file = config.output_file + ".ppm";
            if (!util::to_ppm(image, out_file, &err)) {
                std::fprintf(stderr, "ppm write failure: %s\n", err.c_str());
                return 1;
            }
            std::printf("Write done (%.3fs). Written to %s\nTotal: %.3fs\n", secs() - render_time, config.output_file.c_str(), secs());
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
