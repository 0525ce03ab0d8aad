// Caller-side bookkeeping of the multi-threaded variant of the reference (performance.h:15-60, 133-155),
// kept under the same names so that code written against it compiles unchanged: a wall-clock timer, the
// three render counters with the report main_modern.cpp:224 prints, and the tile list.  The thread pool
// of performance.h:63-130 has no counterpart here: the GPU renderer takes its place.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <vector>

class PerformanceTimer {
public:
    void begin() { m_start = clock::now(); }
    double elapsed_ms() const { return std::chrono::duration<double, std::milli>(clock::now() - m_start).count(); }
    double elapsed_seconds() const { return elapsed_ms() * 1e-3; }

private:
    using clock = std::chrono::steady_clock;
    clock::time_point m_start = clock::now();
};

// rays_traced: Color() invocations; intersection_tests: invocations whose world->Hit() was true;
// material_evaluations: invocations whose Scatter() was true below the depth cap (main_modern.cpp:23-61).
struct RenderStats {
    std::atomic<size_t> rays_traced{ 0 };
    std::atomic<size_t> intersection_tests{ 0 };
    std::atomic<size_t> material_evaluations{ 0 };
    double total_time_ms = 0.0;
    double intersection_time_ms = 0.0;
    double shading_time_ms = 0.0;

    void reset()
    {
        rays_traced = 0; intersection_tests = 0; material_evaluations = 0;
        total_time_ms = intersection_time_ms = shading_time_ms = 0.0;
    }

    // Same lines, same order, same number formats as performance.h:51-59.
    void print_stats() const
    {
        const size_t rays = rays_traced.load();
        std::printf("=== Render Statistics ===\n");
        std::printf("Total time: %.2f ms (%.2f seconds)\n", total_time_ms, total_time_ms / 1000.0);
        std::printf("Rays traced: %zu\n", rays);
        std::printf("Intersection tests: %zu\n", intersection_tests.load());
        std::printf("Material evaluations: %zu\n", material_evaluations.load());
        std::printf("Rays per second: %.0f\n", rays / (total_time_ms / 1000.0));
        std::printf("========================\n");
    }
};

struct RenderTile {
    int x_start, y_start;
    int width, height;
    int samples_per_pixel;
    RenderTile(int x, int y, int w, int h, int samples) : x_start(x), y_start(y), width(w), height(h), samples_per_pixel(samples) {}
};

// Row-major list of tile_size x tile_size tiles (edge tiles clipped), y ascending then x ascending.
inline std::vector<RenderTile> create_tiles(int image_width, int image_height, int samples, int tile_size = 64)
{
    std::vector<RenderTile> tiles;
    if (tile_size <= 0) return tiles;
    const int across = (image_width + tile_size - 1) / tile_size, down = (image_height + tile_size - 1) / tile_size;
    tiles.reserve((size_t)std::max(across, 0) * (size_t)std::max(down, 0));
    for (int ty = 0; ty < down; ++ty)
        for (int tx = 0; tx < across; ++tx) {
            const int x = tx * tile_size, y = ty * tile_size;
            tiles.emplace_back(x, y, std::min(tile_size, image_width - x), std::min(tile_size, image_height - y), samples);
        }
    return tiles;
}
