// The kd-tree container in 1, 2, 3 and more dimensions (the reference's basic_linked_kdtree_t is generic in K:
// include/pcp/kdtree/linked_kdtree.hpp:64-65): k nearest neighbours and box ranges against brute force on the host.
// K <= 3: missing axes travel as 0 to the device index, so every distance and every containment test is the K-dimensional one;
// K > 3: the exhaustive search of pcpx_kd_* (include/pcpx.h).
#include "pcp/kdtree/linked_kdtree.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

template <std::size_t K>
int run(unsigned seed)
{
    using coords = std::array<float, K>;
    struct element { coords c; int id; };
    auto const map = [](element const& e) { return e.c; };
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> u(-5.f, 5.f);
    std::vector<element> pts(3000);
    for (std::size_t i = 0; i < pts.size(); ++i)
    {
        for (std::size_t a = 0; a < K; ++a) pts[i].c[a] = u(gen);
        pts[i].id = static_cast<int>(i);
    }
    pcp::basic_linked_kdtree_t<element, K, decltype(map)> tree(pts.begin(), pts.end(), map);
    int bad = 0;
    auto d2 = [](coords const& a, coords const& b) {
        float s = 0.f;  // the reference's order: x, y, z products summed left to right
        for (std::size_t i = 0; i < K; ++i) s = i == 0 ? (a[i] - b[i]) * (a[i] - b[i]) : s + (a[i] - b[i]) * (a[i] - b[i]);
        return s;
    };
    for (int q = 0; q < 40; ++q)
    {
        coords target;
        for (std::size_t a = 0; a < K; ++a) target[a] = u(gen);
        std::size_t const k = 1 + static_cast<std::size_t>(q % 12);
        auto const got = tree.nearest_neighbours(target, k);
        std::vector<std::pair<float, int>> all;
        for (auto const& e : pts)
        {
            float far = 0.f;  // the reference skips points within eps of the target on every axis (vector3d_queries.hpp:47-64)
            for (std::size_t i = 0; i < K; ++i) far = std::max(far, std::fabs(e.c[i] - target[i]));
            if (far >= 1e-5f) all.push_back({d2(e.c, target), e.id});
        }
        std::sort(all.begin(), all.end());
        if (got.size() != k) { ++bad; std::printf("  query %d: %zu neighbours for k = %zu\n", q, got.size(), k); continue; }
        for (std::size_t j = 0; j < k; ++j)
            if (d2(got[j].c, target) != all[j].first)  // (ids may differ only where distances tie exactly)
            {
                ++bad;
                std::printf("  query %d: neighbour %zu at d2 %.9g, brute force %.9g\n", q, j, d2(got[j].c, target), all[j].first);
            }
        // a box around the target
        pcp::kd_axis_aligned_bounding_box_t<float, K> box;
        for (std::size_t a = 0; a < K; ++a) { box.min[a] = target[a] - 0.4f; box.max[a] = target[a] + 0.4f; }
        auto const in = tree.range_search(box);
        std::size_t expect = 0;
        for (auto const& e : pts) expect += box.contains(e.c) ? 1u : 0u;
        if (in.size() != expect) { ++bad; std::printf("  query %d: %zu in the box, brute force %zu\n", q, in.size(), expect); }
        for (auto const& e : in) if (!box.contains(e.c)) ++bad;
    }
    {  // the batched form gives the rows of the single calls
        std::vector<element> some(pts.begin(), pts.begin() + 25);
        auto const rows = tree.nearest_neighbours_batch(some.begin(), some.end(), 7);
        for (std::size_t i = 0; i < some.size(); ++i)
        {
            auto const one = tree.nearest_neighbours(some[i], 7);
            if (rows[i].size() != one.size()) { ++bad; continue; }
            for (std::size_t j = 0; j < one.size(); ++j)
                if (d2(rows[i][j].c, some[i].c) != d2(one[j].c, some[i].c)) ++bad;
        }
    }
    auto const bb = tree.aabb();
    for (auto const& e : pts) if (!bb.contains(e.c)) ++bad;
    std::printf("K = %zu: %d mismatches\n", K, bad);
    return bad;
}

int main()
{
    int bad = run<1>(11) + run<2>(12) + run<3>(13);
    bad += run<4>(14) + run<5>(15) + run<8>(18) + run<16>(26);  // K > 3: pcpx_kd_* (exhaustive search on the GPU)
    std::printf(bad == 0 ? "ok\n" : "FAILED\n");
    return bad == 0 ? 0 : 1;
}
