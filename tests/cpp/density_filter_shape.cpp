// The reference's third canonical caller (examples/filter_point_cloud_noise_by_density.cpp:15-133) against this
// repository's headers, call for call: read_ply -> point views -> octree of views -> per point, under std::execution::par,
// the mean distance to its k nearest neighbours -> their mean is the ball radius -> std::remove_if of the points whose
// ball holds fewer than `density threshold` points -> write_ply; phases timed with pcp::common::basic_timer_t.
// Only range-v3's transform view (a third-party dependency this image lacks) is replaced by a std::vector of views.
// usage: density_filter_shape <in.ply> <out.ply> [density threshold = 5] [radius multiplier = 1] [k = 15]
// prints one JSON line: points before / after, the radius, the phase times
#include <algorithm>
#include <cstdio>
#include <execution>
#include <filesystem>
#include <numeric>
#include <pcp/common/normals/normal.hpp>
#include <pcp/common/points/point.hpp>
#include <pcp/common/points/point_view.hpp>
#include <pcp/common/sphere.hpp>
#include <pcp/common/timer.hpp>
#include <pcp/common/vector3d_queries.hpp>
#include <pcp/io/ply.hpp>
#include <pcp/octree/octree.hpp>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::filesystem::path input_ply{argv[1]};
    std::filesystem::path output_ply{argv[2]};
    std::size_t const density_threshold = argc >= 4 ? std::stoull(argv[3]) : 5u;
    float const radius_multiplier       = argc >= 5 ? std::stof(argv[4]) : 1.f;
    std::size_t const k                 = argc >= 6 ? std::stoull(argv[5]) : 15u;

    pcp::common::basic_timer_t timer;
    timer.register_op("parse ply point cloud");
    timer.start();
    auto [points, _] = pcp::io::read_ply<pcp::point_t, pcp::normal_t>(input_ply);
    timer.stop();
    if (points.empty()) { std::printf("could not read %s\n", argv[1]); return 1; }
    std::size_t const before = points.size();

    timer.register_op("setup octree");
    timer.start();
    std::vector<pcp::point_view_t> point_views;
    point_views.reserve(points.size());
    for (auto& p : points) point_views.push_back(pcp::point_view_t{&p});
    auto const point_view_map = [](pcp::point_view_t const& p) { return p; };
    pcp::basic_linked_octree_t<pcp::point_view_t> octree{point_views.begin(), point_views.end(), point_view_map};
    timer.stop();

    timer.register_op("compute k neighborhood average radius");
    timer.start();
    std::vector<float> mean_distances(points.size(), 0.f);
    std::transform(std::execution::par, points.cbegin(), points.cend(), mean_distances.begin(), [&](pcp::point_t const& p) {
        auto const& neighbours = octree.nearest_neighbours(p, k, point_view_map);
        float const sum        = std::accumulate(neighbours.cbegin(), neighbours.cend(), 0.f, [&p](float val, pcp::point_view_t const& neighbour) {
            auto const distance = pcp::common::norm(pcp::point_t(neighbour) - p);
            return val + distance;
        });
        return sum / static_cast<float>(neighbours.size());
    });
    float const radius = std::reduce(std::execution::par, mean_distances.cbegin(), mean_distances.cend(), 0.f) / static_cast<float>(mean_distances.size());
    timer.stop();

    timer.register_op("remove points by density threshold");
    timer.start();
    auto it = std::remove_if(std::execution::par, points.begin(), points.end(), [&](pcp::point_t const& p) {
        pcp::sphere_t<pcp::point_t> ball{p, radius * radius_multiplier};
        auto const& points_in_ball = octree.range_search(ball, point_view_map);
        auto const density         = points_in_ball.size();
        return density < density_threshold;
    });
    points.erase(it, points.end());
    timer.stop();

    timer.register_op("write filtered point cloud to ply");
    timer.start();
    pcp::io::write_ply(output_ply, points, _, pcp::io::ply_format_t::binary_little_endian);
    timer.stop();

    std::printf("{\"points_before\": %zu, \"points_after\": %zu, \"radius\": %.9g", before, points.size(), static_cast<double>(radius));
    for (auto const& [operation, duration] : timer.ops)
        std::printf(", \"%s ms\": %.2f", operation.c_str(), std::chrono::duration<double, std::milli>(duration).count());
    std::printf("}\n");
    return 0;
}
