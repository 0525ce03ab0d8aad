#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "rt_amd.h"
int main()
{
    std::mt19937 g(1);
    // formatter: exact-size buffers, odd values
    for (int rep = 0; rep < 200; ++rep) {
        const int w = 1 + g() % 40, h = 1 + g() % 30;
        std::vector<int32_t> img((size_t)w * h * 3);
        for (auto& v : img) { const unsigned r = g() % 20; v = r == 0 ? INT32_MIN : r == 1 ? INT32_MAX : r == 2 ? -(int)(g() % 100000) : (int)(g() % 256); }
        const int64_t need = -rt_format_ppm(img.data(), w, h, nullptr, 0);
        std::vector<char> buf((size_t)need);
        const int64_t n = rt_format_ppm(img.data(), w, h, buf.data(), need);
        if (n != need) { std::printf("size mismatch %lld %lld\n", (long long)n, (long long)need); return 1; }
        std::vector<char> small((size_t)need - 1);
        if (rt_format_ppm(img.data(), w, h, small.data(), need - 1) != -need) { std::printf("short buffer not reported\n"); return 1; }
    }
    // scene grid and hierarchy
    for (int half : { 1, 2, 11, 30 }) {
        const int cap = 4 * half * half + 4;
        std::vector<float> table((size_t)cap * 9); int n = 0;
        if (rt_scene_grid(7, half, 1 << 30, table.data(), cap, &n) != RT_OK) { std::printf("scene_grid failed\n"); return 1; }
        std::vector<float> cx(n), cy(n), cz(n), r(n);
        for (int i = 0; i < n; ++i) { cx[i] = table[9 * i]; cy[i] = table[9 * i + 1]; cz[i] = table[9 * i + 2]; r[i] = table[9 * i + 3]; }
        std::vector<float> nodes((size_t)(2 * n + 2) * 8); std::vector<uint32_t> leaf(n + 1), big(16); int32_t counts[4];
        if (rt_build_hierarchy(n, cx.data(), cy.data(), cz.data(), r.data(), nodes.data(), 2 * n + 2, leaf.data(), big.data(), counts) != RT_OK) { std::printf("hierarchy failed: %s\n", rt_last_error(nullptr)); return 1; }
        std::printf("half %d: %d spheres, %d nodes, depth %d, big %d\n", half, n, counts[0], counts[3], counts[2]);
    }
    // error paths
    if (rt_scene_grid(7, 11, 100, nullptr, 0, nullptr) == RT_OK) return 1;
    if (rt_write_ppm("/nonexistent_dir/x.ppm", nullptr, 1, 1) == RT_OK) return 1;
    std::printf("asan driver ok\n");
    return 0;
}
