// Host-side construction of the sphere hierarchy used by RT_MODE_BVH (the step beyond the brute-force sweep of
// hittablelist.cpp:13-31: same closest hit, found without visiting every sphere).
//
// The hierarchy is built over the sphere CENTRES: every node stores the bounding box of the centres below it
// and (a hair more than) the largest |radius| below it.  The traversal inflates that box per ray (radius plus an
// allowance for the rounding of the reference's discriminant, which grows with the distance of the ray origin), so
// nothing about a particular ray is baked in here.  Spheres that are large next to the rest (the r = 1000 ground of the final
// scene) would blow every ancestor's box up; they are kept in a short list that every ray tests directly.
//
// Layout: node i = two float4, nodes[2i] = (cmin.xyz, rmax plus 1e-6 (|cmin| + |cmax|) for the slab test's rounding),
// nodes[2i+1] = (cmax.xyz, link as int bits).
// link >= 0: inner node, children at link and link + 1.  link < 0: leaf, -(link + 1) = first * 8 + (count - 1),
// sphere indices leaf_index[first .. first + count), ascending inside a leaf.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#include <hip/hip_runtime.h>

namespace rt {

#ifndef RT_BVH_LEAF
#define RT_BVH_LEAF 4
#endif
constexpr int kBvhLeafSize = RT_BVH_LEAF;     // at most 8 (the leaf code keeps count - 1 in three bits)
constexpr int kBvhMaxBig = 16;
constexpr int kBvhStack = 24;       // traversal stack entries per lane; the builder refuses deeper trees

struct BvhHost {
    std::vector<float4> nodes;          // 2 per node
    std::vector<uint32_t> leaf_index;   // sphere indices in leaf order
    std::vector<uint32_t> big;          // spheres tested directly by every ray (ascending)
    int depth = 0;
    int node_count() const { return (int)(nodes.size() / 2); }
};

namespace bvh_detail {
struct Builder {
    const float *cx, *cy, *cz, *radius;
    BvhHost* out;
    std::vector<uint32_t> order;
    float centre(uint32_t i, int axis) const { return axis == 0 ? cx[i] : axis == 1 ? cy[i] : cz[i]; }

    // Fills node `self` for order[lo, hi); children are appended behind the current end of the node array.
    void build(int self, int lo, int hi, int depth)
    {
        out->depth = std::max(out->depth, depth);
        float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY }, rmax = 0.0f;
        for (int k = lo; k < hi; ++k) {
            const uint32_t i = order[k];
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], centre(i, a)); mx[a] = std::max(mx[a], centre(i, a)); }
            rmax = std::max(rmax, std::fabs(radius[i]));
        }
        int link;
        const int count = hi - lo;
        if (count <= kBvhLeafSize) {
            const int first = (int)out->leaf_index.size();
            std::vector<uint32_t> ids(order.begin() + lo, order.begin() + hi);
            std::sort(ids.begin(), ids.end());
            out->leaf_index.insert(out->leaf_index.end(), ids.begin(), ids.end());
            link = -(first * 8 + (count - 1)) - 1;
        } else {
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
            const int mid = lo + count / 2;
            std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                             [&](uint32_t p, uint32_t q) { const float a = centre(p, axis), b = centre(q, axis); return a < b || (a == b && p < q); });
            link = out->node_count();
            out->nodes.resize(out->nodes.size() + 4);       // two adjacent children
            build(link, lo, mid, depth + 1);
            build(link + 1, mid, hi, depth + 1);
        }
        float linkf; std::memcpy(&linkf, &link, sizeof linkf);
        // the radius the traversal inflates the box by, with the box's share of the slab test's absolute margin
        float margin = 0.0f;
        for (int a = 0; a < 3; ++a) margin += std::fabs(mn[a]) + std::fabs(mx[a]);
        const float rpad = std::nextafter(rmax + 1e-6f * margin, INFINITY);
        out->nodes[2 * (size_t)self] = make_float4(mn[0], mn[1], mn[2], rpad);
        out->nodes[2 * (size_t)self + 1] = make_float4(mx[0], mx[1], mx[2], linkf);
    }
};
}  // namespace bvh_detail

// Returns false when the tree would be deeper than the traversal stack allows (cannot happen for n <= 65535 with
// median splits, kept as a guard).
inline bool build_bvh(int n, const float* cx, const float* cy, const float* cz, const float* radius, BvhHost* out)
{
    *out = BvhHost();
    // A sphere whose centre or radius is not finite is hit by no ray in the reference's arithmetic (its discriminant comes out
    // NaN, or its roots infinite, sphere.cpp:17-50), so it takes no part in the hierarchy at all -- and NaNs must not reach the
    // comparators below, which would no longer be strict weak orderings.
    std::vector<char> usable(n, 0);
    std::vector<float> mags(n, 0.0f), sorted;
    for (int i = 0; i < n; ++i) {
        usable[i] = std::isfinite(cx[i]) && std::isfinite(cy[i]) && std::isfinite(cz[i]) && std::isfinite(radius[i]);
        if (usable[i]) { mags[i] = std::fabs(radius[i]); sorted.push_back(mags[i]); }
    }
    float median = 0.0f;
    if (!sorted.empty()) { std::nth_element(sorted.begin(), sorted.begin() + sorted.size() / 2, sorted.end()); median = sorted[sorted.size() / 2]; }
    bvh_detail::Builder b{ cx, cy, cz, radius, out, {} };
    // "large next to the rest": more than eight median radii; the largest ones first, at most kBvhMaxBig
    std::vector<uint32_t> candidates;
    for (int i = 0; i < n; ++i) if (usable[i] && mags[i] > 8.0f * median) candidates.push_back((uint32_t)i);
    std::sort(candidates.begin(), candidates.end(), [&](uint32_t p, uint32_t q) { return mags[p] > mags[q] || (mags[p] == mags[q] && p < q); });
    if ((int)candidates.size() > kBvhMaxBig) candidates.resize(kBvhMaxBig);
    std::vector<char> is_big(n, 0);
    for (uint32_t i : candidates) is_big[i] = 1;
    for (int i = 0; i < n; ++i) if (usable[i]) (is_big[i] ? out->big : b.order).push_back((uint32_t)i);
    out->nodes.resize(2);
    if (b.order.empty()) {
        // nothing left for the tree: an empty leaf that no ray can enter
        const int link = -1 - 0; float linkf; std::memcpy(&linkf, &link, sizeof linkf);
        out->nodes[0] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
        out->nodes[1] = make_float4(-INFINITY, -INFINITY, -INFINITY, linkf);
        out->leaf_index.push_back(out->big.empty() ? 0u : out->big[0]);     // a valid index; the inverted box is never entered
        return true;
    }
    b.build(0, 0, (int)b.order.size(), 0);
    return out->depth + 2 <= kBvhStack;
}

}  // namespace rt
