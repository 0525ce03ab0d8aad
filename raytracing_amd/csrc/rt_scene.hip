// Host-side scene generation (untimed setup): RandomScene() of main.cpp:41-83 generalised, with an explicit draw order.
#include "rt_internal.hpp"
#include "rt_bvh.hpp"

extern "C" {

// ---------------------------------------------------------------------------- scene setup
static void philox_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

int rt_scene_grid(uint64_t seed, int half_extent, int max_small, float* table9, int cap, int* n_out)
{
    if (!table9 || !n_out || half_extent <= 0 || half_extent > 1024 || max_small < 0 || cap < 4) return fail(nullptr, RT_ERR_INVALID, "rt_scene_grid: bad argument");
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    int n = 0;
    auto put = [&](float x, float y, float z, float r, int m, float ar, float ag, float ab, float p) {
        float* row = table9 + (size_t)n * 9;
        row[0] = x; row[1] = y; row[2] = z; row[3] = r; row[4] = (float)m; row[5] = ar; row[6] = ag; row[7] = ab; row[8] = p; ++n;
    };
    auto u01h = [](uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); };
    put(0.0f, -1000.0f, 0.0f, 1000.0f, RT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f);
    const int side = 2 * half_extent;
    int kept = 0;
    for (int a = -half_extent; a < half_extent && kept < max_small; ++a) {
        for (int b = -half_extent; b < half_extent && kept < max_small; ++b) {
            const uint32_t t = (uint32_t)((a + half_extent) * side + (b + half_extent));
            uint32_t w0[4], w1[4], w2[4];
            philox_host(t, 0, 0, 0, k0, k1, w0);
            const float choose = u01h(w0[0]);
            const float x = (float)a + 0.9f * u01h(w0[1]), z = (float)b + 0.9f * u01h(w0[2]);
            const float dx = x - 4.0f, dy = 0.2f - 0.2f, dz = z - 0.0f;
            if (!(sqrtf(dx * dx + dy * dy + dz * dz) > 0.9f)) continue;
            if (n + 4 > cap) return fail(nullptr, RT_ERR_TOO_LARGE, "rt_scene_grid: table capacity %d too small", cap);
            philox_host(t, 1, 0, 0, k0, k1, w1);
            if (choose < 0.8f) {
                philox_host(t, 2, 0, 0, k0, k1, w2);
                put(x, 0.2f, z, 0.2f, RT_LAMBERTIAN, u01h(w1[0]) * u01h(w1[1]), u01h(w1[2]) * u01h(w1[3]), u01h(w2[0]) * u01h(w2[1]), 0.0f);
            } else if (choose < 0.95f) {
                put(x, 0.2f, z, 0.2f, RT_METAL, 0.5f * (1.0f + u01h(w1[0])), 0.5f * (1.0f + u01h(w1[1])), 0.5f * (1.0f + u01h(w1[2])), 0.5f * u01h(w1[3]));
            } else {
                put(x, 0.2f, z, 0.2f, RT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
            }
            ++kept;
        }
    }
    if (n + 3 > cap) return fail(nullptr, RT_ERR_TOO_LARGE, "rt_scene_grid: table capacity %d too small", cap);
    put(0.0f, 1.0f, 0.0f, 1.0f, RT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
    put(-4.0f, 1.0f, 0.0f, 1.0f, RT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f);
    put(4.0f, 1.0f, 0.0f, 1.0f, RT_METAL, 0.7f, 0.6f, 0.5f, 0.0f);
    *n_out = n;
    return RT_OK;
}

// The hierarchy RT_MODE_BVH walks, built on the host exactly as rt_upload_scene builds it, copied out for inspection.
int rt_build_hierarchy(int n, const float* cx, const float* cy, const float* cz, const float* radius,
                       float* nodes8, int node_cap, uint32_t* leaf_index, uint32_t* big, int32_t counts[4])
{
    if (n <= 0 || !cx || !cy || !cz || !radius || !nodes8 || !leaf_index || !big || !counts) return fail(nullptr, RT_ERR_INVALID, "rt_build_hierarchy: bad argument");
    BvhHost tree;
    if (!build_bvh(n, cx, cy, cz, radius, &tree)) return fail(nullptr, RT_ERR_TOO_LARGE, "rt_build_hierarchy: hierarchy deeper than %d levels", kBvhStack - 2);
    if (tree.node_count() > node_cap) return fail(nullptr, RT_ERR_TOO_LARGE, "rt_build_hierarchy: %d nodes, capacity %d", tree.node_count(), node_cap);
    memcpy(nodes8, tree.nodes.data(), tree.nodes.size() * sizeof(float4));
    memcpy(leaf_index, tree.leaf_index.data(), tree.leaf_index.size() * sizeof(uint32_t));
    if (!tree.big.empty()) memcpy(big, tree.big.data(), tree.big.size() * sizeof(uint32_t));
    counts[0] = tree.node_count(); counts[1] = (int32_t)tree.leaf_index.size(); counts[2] = (int32_t)tree.big.size(); counts[3] = tree.depth;
    return RT_OK;
}

}  // extern "C"
