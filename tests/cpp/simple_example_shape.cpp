// The reference's README program (examples/simple_example.cpp:6-110) against this repository's headers, call for call:
// read_ply -> point views -> octree of views -> per-point range_search (density) under std::execution::par ->
// algorithm::estimate_normals with a PLAIN LAMBDA knn map -> write_ply.  Only the range-v3 view of the original
// (`points | views::transform`, a third-party dependency this image lacks) is replaced by a std::vector of views.
// The caller does nothing GPU-specific: the containers notice the per-point call pattern and answer from one batched
// launch (pcp/gpu/device_index.hpp).  Results are checked against the committed golden rows of the bunny
// (tests/golden/bunny_k15.npz, exported as raw arrays by the test driver); each phase is timed.
// usage: simple_example_shape <in.ply> <out.ply> <golden dir>
#include <pcp/pcp.hpp>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <execution>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <vector>

template <class T>
static std::vector<T> slurp(std::filesystem::path const& f)
{
    std::ifstream is(f, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    std::vector<T> out(raw.size() / sizeof(T));
    std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
    return out;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    using point_type      = pcp::point_t;
    using point_view_type = pcp::point_view_t;
    using normal_type     = pcp::normal_t;

    double t0 = now();
    std::filesystem::path input_ply{argv[1]};
    auto [points, normals] = pcp::io::read_ply<point_type, normal_type>(input_ply);
    double const t_read = now() - t0;
    if (points.empty()) { std::printf("could not read %s\n", argv[1]); return 1; }

    std::vector<point_view_type> point_views;
    point_views.reserve(points.size());
    for (auto& point : points) point_views.push_back(point_view_type{&point});

    auto const point_view_map = [](point_view_type const& p) { return p; };

    t0 = now();
    using octree_type = pcp::basic_linked_octree_t<point_view_type>;
    octree_type octree{point_views.begin(), point_views.end(), point_view_map};
    double const t_tree = now() - t0;

    t0 = now();
    std::vector<float> density(points.size(), 0.f);
    std::vector<std::size_t> in_range(points.size(), 0u);
    std::transform(std::execution::par, point_views.begin(), point_views.end(), density.begin(), [&](auto const& p) {
        pcp::sphere_t<pcp::point_t> sphere{};
        sphere.radius              = 0.01f;
        sphere.position            = point_type{p};
        auto const points_in_range = octree.range_search(sphere, point_view_map);
        in_range[static_cast<std::size_t>(&p - point_views.data())] = points_in_range.size();
        auto const pi              = 3.14159f;
        auto const r3              = sphere.radius * sphere.radius * sphere.radius;
        auto const volume          = 4.f / 3.f * pi * r3;
        return static_cast<float>(points_in_range.size()) / volume;
    });
    double const t_density = now() - t0;

    normals.resize(points.size());
    auto const knn = [&](auto const& p) { return octree.nearest_neighbours(p, 15u, point_view_map); };

    t0 = now();
    pcp::algorithm::estimate_normals(std::execution::par, point_views.begin(), point_views.end(), normals.begin(), point_view_map, knn,
                                     pcp::algorithm::default_normal_transform<point_view_type, normal_type>);
    double const t_normals = now() - t0;

    t0 = now();
    pcp::io::write_ply(std::filesystem::path{argv[2]}, points, normals, pcp::io::ply_format_t::binary_little_endian);
    double const t_write = now() - t0;

    // ---- checks against the golden rows ----
    std::filesystem::path const g{argv[3]};
    auto const qi    = slurp<std::int64_t>(g / "query_index.bin");
    auto const gidx  = slurp<std::uint32_t>(g / "knn_idx.bin");
    auto const gnrm  = slurp<float>(g / "normals.bin");
    auto const grc   = slurp<std::uint32_t>(g / "range_count_r001.bin");
    int bad = 0;
    for (std::size_t r = 0; r < qi.size(); ++r)
    {
        std::size_t const i = static_cast<std::size_t>(qi[r]);
        auto const nn = octree.nearest_neighbours(point_views[i], 15u, point_view_map);
        if (nn.size() != 15u) { ++bad; continue; }
        for (std::size_t j = 0; j < 15u; ++j)
        {
            auto const& e = points[gidx[r * 15 + j]];
            if (nn[j].x() != e.x() || nn[j].y() != e.y() || nn[j].z() != e.z()) ++bad;
        }
        double const dot = double(normals[i].nx()) * gnrm[3 * r] + double(normals[i].ny()) * gnrm[3 * r + 1] + double(normals[i].nz()) * gnrm[3 * r + 2];
        if (1.0 - std::abs(dot) > 1e-4) ++bad;
        if (in_range[i] != grc[r]) ++bad;
    }
    auto [rp, rn] = pcp::io::read_ply<point_type, normal_type>(std::filesystem::path{argv[2]});
    if (rp.size() != points.size() || rn.size() != normals.size()) ++bad;
    for (std::size_t i = 0; i < rn.size() && i < normals.size(); i += 997)
        if (rn[i].nx() != normals[i].nx() || rn[i].nz() != normals[i].nz()) ++bad;

    std::printf("{\"points\": %zu, \"read_ply_ms\": %.3f, \"octree_ctor_ms\": %.3f, \"density_loop_ms\": %.3f, \"estimate_normals_ms\": %.3f, "
                "\"write_ply_ms\": %.3f, \"golden_rows_checked\": %zu, \"mismatches\": %d}\n",
                points.size(), t_read * 1e3, t_tree * 1e3, t_density * 1e3, t_normals * 1e3, t_write * 1e3, qi.size(), bad);
    return bad == 0 ? 0 : 1;
}
