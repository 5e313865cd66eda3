// Drop-in check for the reference's third canonical caller, examples/filter_point_cloud_noise_by_density.cpp:15-133.
// This is not that program: it is a test that uses each pcp call SHAPE the example relies on -- the same overloads with
// the same argument types -- so that a maintainer knows the example builds and behaves against these headers:
//   io::read_ply<point_t, normal_t>(path) with structured bindings; an octree of point_view_t built from a view range and an
//   identity point-view map; octree.nearest_neighbours(point_t, k, map) inside std::transform(std::execution::par, ...);
//   common::norm(point_t(view) - point) inside std::accumulate; sphere_t<point_t>{centre, radius} and
//   octree.range_search(ball, map).size() inside std::remove_if(std::execution::par, ...); io::write_ply(path, points,
//   normals, ply_format_t::binary_little_endian); common::basic_timer_t's register_op / start / stop / ops.
// usage: density_filter_shape <in.ply> <out.ply> [min points per ball = 5] [radius multiplier = 1] [k = 15]
// prints one JSON object: cloud size before and after, the ball radius, the time of every registered phase
#include <pcp/common/normals/normal.hpp>
#include <pcp/common/points/point.hpp>
#include <pcp/common/points/point_view.hpp>
#include <pcp/common/sphere.hpp>
#include <pcp/common/timer.hpp>
#include <pcp/common/vector3d_queries.hpp>
#include <pcp/io/ply.hpp>
#include <pcp/octree/octree.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <execution>
#include <filesystem>
#include <numeric>
#include <vector>

namespace {

using cloud_t  = std::vector<pcp::point_t>;
using view_t   = pcp::point_view_t;
using octree_t = pcp::basic_linked_octree_t<view_t>;
auto const identity = [](view_t const& v) { return v; };

struct phase_t  // one timed phase: registered on construction, stopped on destruction
{
    pcp::common::basic_timer_t& t;
    phase_t(pcp::common::basic_timer_t& timer, char const* name) : t(timer)
    {
        t.register_op(name);
        t.start();
    }
    ~phase_t() { t.stop(); }
};

// mean over the cloud of (mean distance from a point to its k nearest neighbours)
float ball_radius(cloud_t const& cloud, octree_t const& tree, std::size_t k)
{
    std::vector<float> per_point(cloud.size());
    std::transform(std::execution::par, cloud.cbegin(), cloud.cend(), per_point.begin(), [&](pcp::point_t const& centre) {
        auto const& nearest = tree.nearest_neighbours(centre, k, identity);
        float const total   = std::accumulate(nearest.cbegin(), nearest.cend(), 0.f, [&centre](float acc, view_t const& other) {
            return acc + pcp::common::norm(pcp::point_t(other) - centre);
        });
        return total / static_cast<float>(nearest.size());
    });
    return std::reduce(std::execution::par, per_point.cbegin(), per_point.cend(), 0.f) / static_cast<float>(per_point.size());
}

// removes, in place, every point whose ball of the given radius holds fewer than `at_least` points
void drop_sparse_points(cloud_t& cloud, octree_t const& tree, float radius, std::size_t at_least)
{
    auto const first_dropped = std::remove_if(std::execution::par, cloud.begin(), cloud.end(), [&](pcp::point_t const& centre) {
        pcp::sphere_t<pcp::point_t> ball{centre, radius};
        return tree.range_search(ball, identity).size() < at_least;
    });
    cloud.erase(first_dropped, cloud.end());
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::size_t const at_least = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 5u;
    float const multiplier     = argc > 4 ? std::strtof(argv[4], nullptr) : 1.f;
    std::size_t const k        = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 15u;
    pcp::common::basic_timer_t timer;

    cloud_t cloud;
    std::vector<pcp::normal_t> no_normals;
    {
        phase_t phase(timer, "read");
        auto [points, normals] = pcp::io::read_ply<pcp::point_t, pcp::normal_t>(std::filesystem::path{argv[1]});
        cloud                  = std::move(points);
        no_normals             = std::move(normals);
    }
    if (cloud.empty()) return 1;
    std::size_t const before = cloud.size();

    std::vector<view_t> views;
    for (auto& p : cloud) views.push_back(view_t{&p});
    timer.register_op("tree");
    timer.start();
    octree_t tree{views.begin(), views.end(), identity};
    timer.stop();

    float radius = 0.f;
    {
        phase_t phase(timer, "radius");
        radius = ball_radius(cloud, tree, k);
    }
    {
        phase_t phase(timer, "filter");
        drop_sparse_points(cloud, tree, radius * multiplier, at_least);
    }
    {
        phase_t phase(timer, "write");
        pcp::io::write_ply(std::filesystem::path{argv[2]}, cloud, no_normals, pcp::io::ply_format_t::binary_little_endian);
    }

    std::printf("{\"points_before\": %zu, \"points_after\": %zu, \"radius\": %.9g", before, cloud.size(), static_cast<double>(radius));
    for (auto const& [name, took] : timer.ops) std::printf(", \"%s_ms\": %.2f", name.c_str(), std::chrono::duration<double, std::milli>(took).count());
    std::printf("}\n");
    return 0;
}
