// rt_main.cpp - command-line renderer: what the reference's main.cpp does, line by line
// (banner main.cpp:9, init_scene :14, "Start time" :17-19, "Rendering image... 0 %" :21,
// render_image :23 with its progress and "Image successfully saved!" lines
// raytracing.h:132-141, "Finish time" :25-27, "Compute time" :29-31), with the frame settings
// as options instead of #defines.  Two deliberate differences in the text: the compute time
// is the elapsed time as h:m:s (the reference pushes the difference of two time stamps through
// localtime() and subtracts one hour, which is only right in a UTC+1 time zone), and one more
// line gives the elapsed time in milliseconds (the reference's clock ticks in seconds).
//
//   rt_main [--width W] [--height H] [--lights N] [--seed S] [--depth D] [--device G] [--gpus N] [--bounces B]
//           [--out FILE.png|.ppm]
//
// Run it from a directory that holds meshs/bunny.off and meshs/dino.off (scene.h:95-99).
#include <chrono>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>

#include "../../include/rtx_dropin.hpp"

int main(int argc, char **argv) {
    std::printf("RAY TRACING by Romain Roy\n-------------------------\n"); // main.cpp:9
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
    if (!init_scene(light, boxs)) return 1; // main.cpp:14
    const std::time_t t1 = std::time(nullptr);
    const std::tm start = *std::localtime(&t1);
    std::printf("Start time: %d:%d:%d\n", start.tm_hour, start.tm_min, start.tm_sec); // main.cpp:19
    const auto c0 = std::chrono::steady_clock::now();
    std::printf("Rendering image... 0 %%\r"); // main.cpp:21
    std::fflush(stdout);
    if (!render_image(light, boxs)) return 1; // main.cpp:23
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
    const std::time_t t2 = std::time(nullptr);
    const std::tm finish = *std::localtime(&t2);
    std::printf("Finish time: %d:%d:%d\n", finish.tm_hour, finish.tm_min, finish.tm_sec); // main.cpp:27
    const long dt = (long)(t2 - t1);
    std::printf("Compute time: %ld:%ld:%ld\n", dt / 3600, (dt / 60) % 60, dt % 60); // main.cpp:31
    std::printf("Compute time (ms): %.3f (%u x %u, %u light samples, %u GPU(s))\n", wall * 1e3, cfg.frame.width,
                cfg.frame.height, cfg.frame.n_lights, cfg.devices.empty() ? 1u : (unsigned)cfg.devices.size());
    rtx_host::release_replicas();
    return 0;
}
