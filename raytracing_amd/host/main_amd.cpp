// Drop-in for the reference's main.cpp: the same scene and camera construction through the same
// class surface (main.cpp:109-116), the pixel loop (main.cpp:118-146) replaced by one GPU render
// call, and output.ppm written in the same ASCII P3 format.
//
//   raytracing_amd [nx ny ns [output.ppm [seed [gpus]]]]      defaults: 400 300 5 output.ppm 0 1  (main.cpp:87-89)
//
// gpus > 1 (or 0 = every visible GPU) splits the rows over that many GPUs of the node; the image does not depend on it.
//
// Differences from the reference, on purpose: the output file is created/truncated (the reference
// opens it with std::fstream's default mode and silently writes nothing when it does not exist);
// sizes come from the command line instead of being compiled in; the uniform stream is Philox
// keyed on (pixel, sample, bounce) instead of rand() plus an unseedable lens generator, so the
// image is reproducible.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "gpu_render.h"
#include "scene.h"

int main(int argc, char const* argv[])
{
    int nx = 400, ny = 300, ns = 5;
    std::string path = "output.ppm";
    unsigned long long seed = 0;
    if (argc >= 4) { nx = std::atoi(argv[1]); ny = std::atoi(argv[2]); ns = std::atoi(argv[3]); }
    if (argc >= 5) path = argv[4];
    if (argc >= 6) seed = std::strtoull(argv[5], nullptr, 10);
    int gpus = 1;
    if (argc >= 7) gpus = std::atoi(argv[6]);

    Hittable* world = RandomScene();
    const Vec3 lookFrom(13.0f, 2.0f, 3.0f), lookAt(0.0f, 0.0f, 0.0f);
    const float distToFocus = 10.0f, aperture = 0.1f;
    const Camera camera(lookFrom, lookAt, Vec3(0.0f, 1.0f, 0.0f), 20.0f, float(nx) / float(ny), aperture, distToFocus);

    try {
        const auto t0 = std::chrono::steady_clock::now();
        const rtamd::RenderResult img = gpus == 1 ? rtamd::RenderOnGpu(world, camera, nx, ny, ns, seed) : rtamd::RenderOnGpus(world, camera, nx, ny, ns, seed, gpus);
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rt_write_ppm(path.c_str(), img.irgb.data(), nx, ny) != RT_OK) { std::fprintf(stderr, "%s\n", rt_last_error(nullptr)); return 1; }
        std::printf("Rays traced: %llu\nKernel time: %.3f ms\nRays per second: %.0f\nWall (setup + render + copy): %.3f s\n",
                    (unsigned long long)img.rays, img.kernel_ms, img.rays / (img.kernel_ms * 1e-3), wall);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "render failed: %s\n", e.what());
        return 1;
    }
    DestroyScene(world);
    return 0;
}
