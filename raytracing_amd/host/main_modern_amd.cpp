// Drop-in for the reference's main_modern.cpp: the same banner, scene and camera (main_modern.cpp:135-168),
// the thread pool over 64 x 64 tiles (main_modern.cpp:197-219) replaced by GPU renders of 64-row bands
// with each band read back as soon as it is finished, the same statistics report (main_modern.cpp:221-224)
// and the same output.ppm.
//
//   raytracing_modern_amd [nx ny ns [output.ppm [seed [tile_size]]]]
//       defaults: 2560 1440 10 output.ppm 0 64 (main_modern.cpp:136-138, 198); tile_size 0 = one call
//
// The statistics cover what the reference's timer covers: scene construction, rendering and read-back
// (main_modern.cpp:155, 221), not the PPM write.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>
#include "gpu_render.h"
#include "performance.h"
#include "scene.h"

int main(int argc, char const* argv[])
{
    int nx = 2560, ny = 1440, ns = 10, tile_size = 64;
    std::string path = "output.ppm";
    unsigned long long seed = 0;
    if (argc >= 4) { nx = std::atoi(argv[1]); ny = std::atoi(argv[2]); ns = std::atoi(argv[3]); }
    if (argc >= 5) path = argv[4];
    if (argc >= 6) seed = std::strtoull(argv[5], nullptr, 10);
    if (argc >= 7) tile_size = std::atoi(argv[6]);
    if (nx <= 0 || ny <= 0 || ns <= 0 || tile_size < 0) { std::fprintf(stderr, "bad size\n"); return 2; }

    std::cout << "=== Modern Raytracer ===" << std::endl;
    std::cout << "Resolution: " << nx << "x" << ny << std::endl;
    std::cout << "Samples per pixel: " << ns << std::endl;
    int gpus = 0;
    if (rt_device_count(&gpus) != RT_OK || gpus < 1) { std::fprintf(stderr, "%s\n", rt_last_error(nullptr)); return 1; }
    std::cout << "GPUs visible: " << gpus << " (rendering on device 0)" << std::endl;

    PerformanceTimer total_timer;
    RenderStats final_stats;
    total_timer.begin();

    Hittable* world = RandomScene();
    const Vec3 lookFrom(13.0f, 2.0f, 3.0f), lookAt(0.0f, 0.0f, 0.0f);
    const float distToFocus = 10.0f, aperture = 0.1f;
    const Camera camera(lookFrom, lookAt, Vec3(0.0f, 1.0f, 0.0f), 20.0f, float(nx) / float(ny), aperture, distToFocus);

    std::vector<int32_t> pixels((size_t)nx * ny * 3);
    float kernel_ms = 0.0f;
    std::cout << "Rendering..." << std::endl;
    try {
        rtamd::GpuRenderer gpu(world, camera, 0);
        rt_counters counters{};
        if (tile_size == 0) {
            kernel_ms = gpu.Render(rtamd::GpuRenderer::Params(nx, ny, ns, seed), pixels.data(), &counters);
        } else {
            const std::vector<RenderTile> tiles = create_tiles(nx, ny, ns, tile_size);
            const int tiles_across = (nx + tile_size - 1) / tile_size;
            size_t tiles_done = 0;
            kernel_ms = gpu.RenderProgressive(nx, ny, ns, seed, tile_size, &counters, [&](int first_row, int rows, const int32_t* band) {
                std::copy(band, band + (size_t)rows * nx * 3, pixels.begin() + (size_t)first_row * nx * 3);
                tiles_done += (size_t)tiles_across;
                std::fprintf(stderr, "\rtiles %zu / %zu", tiles_done, tiles.size());
            });
            std::fprintf(stderr, "\n");
        }
        final_stats.rays_traced += counters.rays_traced;
        final_stats.intersection_tests += counters.intersection_tests;
        final_stats.material_evaluations += counters.material_evaluations;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "render failed: %s\n", e.what());
        return 1;
    }
    final_stats.total_time_ms = total_timer.elapsed_ms();

    std::cout << "Rendering complete!" << std::endl;
    final_stats.print_stats();
    std::printf("GPU kernel time: %.3f ms (%.0f rays per second inside the kernels)\n", kernel_ms, final_stats.rays_traced.load() / (kernel_ms * 1e-3));

    std::cout << "Writing output file..." << std::endl;
    if (rt_write_ppm(path.c_str(), pixels.data(), nx, ny) != RT_OK) { std::fprintf(stderr, "%s\n", rt_last_error(nullptr)); return 1; }
    std::cout << "Output written to " << path << std::endl;
    DestroyScene(world);
    return 0;
}
