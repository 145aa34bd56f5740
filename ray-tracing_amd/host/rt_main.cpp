// rt_main.cpp - command-line renderer: what the reference's main.cpp does (banner, scene,
// render, timing), with the frame settings as options instead of #defines.
//
//   rt_main [--width W] [--height H] [--lights N] [--seed S] [--depth D] [--device G] [--gpus N] [--bounces B]
//           [--out FILE.png|.ppm]
//
// Run it from a directory that holds meshs/bunny.off and meshs/dino.off (scene.h:95-99).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/rtx_dropin.hpp"

int main(int argc, char **argv) {
    std::printf("RAY TRACING on MI355X\n---------------------\n");
    rtx_host::Settings &cfg = rtx_host::settings();
    for (int k = 1; k + 1 < argc; k += 2) {
        const char *a = argv[k], *v = argv[k + 1];
        if (!std::strcmp(a, "--width")) cfg.frame.width = (uint32_t)std::atoi(v);
        else if (!std::strcmp(a, "--height")) cfg.frame.height = (uint32_t)std::atoi(v);
        else if (!std::strcmp(a, "--lights")) cfg.frame.n_lights = (uint32_t)std::atoi(v);
        else if (!std::strcmp(a, "--seed")) cfg.frame.seed = (uint32_t)std::atoi(v);
        else if (!std::strcmp(a, "--depth")) cfg.box_depth = std::atoi(v);
        else if (!std::strcmp(a, "--device")) cfg.device = std::atoi(v);
        else if (!std::strcmp(a, "--gpus")) {  // split the frame over GPUs 0 .. N-1
            cfg.devices.clear();
            for (int g = 0; g < std::atoi(v); g++) cfg.devices.push_back(g);
        }
        else if (!std::strcmp(a, "--bounces")) cfg.frame.max_depth = (uint32_t)std::atoi(v);
        else if (!std::strcmp(a, "--out")) cfg.output = v;
        else {
            std::fprintf(stderr, "unknown option %s\n", a);
            return 2;
        }
    }
    Light light;
    Boxs boxs;
    if (!init_scene(light, boxs)) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    std::printf("Rendering image... 0 %%\r");
    if (!render_image(light, boxs)) return 1;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Compute time: %.3f s (%u x %u, %u light samples)\n", s, cfg.frame.width, cfg.frame.height,
                cfg.frame.n_lights);
    return 0;
}
