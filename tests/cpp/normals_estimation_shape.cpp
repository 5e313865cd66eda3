// Drop-in check for the reference's second canonical caller, examples/normals_estimation.cpp:12-148.  Not that program:
// a test that uses each pcp call SHAPE the example relies on, with the same argument types --
//   io::read_ply<point_t, normal_t>; a basic_linked_kdtree_t<uint64_t, 3, CoordinateMap> over index elements built with
//   construction_params_t{compute_max_depth}; kdtree.nearest_neighbours(element, k) wrapped in a plain lambda that serves
//   as the KnnMap of algorithm::estimate_normals (both execution policies, output written THROUGH the transform into the
//   caller's vector while the algorithm's own output range is the index range), of propagate_normal_orientations
//   (index map, knn map, point map, normal map, transform) and of average_distance_to_neighbors; bilateral::params_t and
//   bilateral_filter_normals whose output iterator is the begin() of the very vector its normal map reads; io::write_ply.
// usage: normals_estimation_shape <in.ply> <out.ply> <k> <par|seq> [<filter iterations> <sigmaf / mean distance> <sigmag / mean distance>]
// prints one JSON object (sizes, the mean neighbour distance, phase times)
#include <pcp/algorithm/algorithm.hpp>
#include <pcp/common/normals/normal.hpp>
#include <pcp/common/points/point.hpp>
#include <pcp/io/ply.hpp>
#include <pcp/kdtree/kdtree.hpp>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execution>
#include <filesystem>
#include <numeric>
#include <vector>

namespace {
struct stopwatch
{
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap_ms()
    {
        auto const t1   = std::chrono::steady_clock::now();
        double const ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0              = t1;
        return ms;
    }
};
} // namespace

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    using id_t          = std::uint64_t;
    std::uint64_t const k = std::strtoull(argv[3], nullptr, 10);
    bool const use_par    = std::strcmp(argv[4], "par") == 0;
    bool const filter     = argc >= 8;
    std::size_t const filter_rounds = filter ? std::strtoull(argv[5], nullptr, 10) : 0u;
    double const f_scale            = filter ? std::strtod(argv[6], nullptr) : 0.;
    double const g_scale            = filter ? std::strtod(argv[7], nullptr) : 0.;

    auto [cloud, unused] = pcp::io::read_ply<pcp::point_t, pcp::normal_t>(std::filesystem::path{argv[1]});
    (void)unused;
    if (cloud.empty()) return 1;
    std::vector<pcp::normal_t> field(cloud.size());
    std::vector<id_t> ids(cloud.size());
    std::iota(ids.begin(), ids.end(), id_t{0});

    auto const id_of     = [](id_t const& i) { return i; };
    auto const point_of  = [&](id_t const& i) { return cloud[i]; };
    auto const normal_of = [&](id_t const& i) { return field[i]; };
    auto const coords_of = [&](id_t const& i) { return std::array<float, 3u>{cloud[i].x(), cloud[i].y(), cloud[i].z()}; };

    stopwatch watch;
    pcp::kdtree::construction_params_t tree_params;
    tree_params.compute_max_depth = true;
    pcp::basic_linked_kdtree_t<id_t, 3u, decltype(coords_of)> tree{ids.begin(), ids.end(), coords_of, tree_params};
    double const tree_ms = watch.lap_ms();

    auto const neighbours_of = [&](id_t const& i) { return tree.nearest_neighbours(i, k); };
    auto const store         = [&](id_t const& i, pcp::normal_t const& n) {
        field[i] = n;
        return i;
    };

    if (use_par)
        pcp::algorithm::estimate_normals(std::execution::par, ids.begin(), ids.end(), ids.begin(), point_of, neighbours_of, store);
    else
        pcp::algorithm::estimate_normals(std::execution::seq, ids.begin(), ids.end(), ids.begin(), point_of, neighbours_of, store);
    double const normals_ms = watch.lap_ms();

    pcp::algorithm::propagate_normal_orientations(ids.begin(), ids.end(), id_of, neighbours_of, point_of, normal_of, store);
    double const orient_ms = watch.lap_ms();

    float mean_distance = 0.f;
    double filter_ms    = 0.;
    if (filter)
    {
        mean_distance = pcp::algorithm::average_distance_to_neighbors(ids.begin(), ids.end(), point_of, neighbours_of);
        pcp::algorithm::bilateral::params_t bp;
        bp.sigmaf = f_scale * static_cast<double>(mean_distance);
        bp.sigmag = g_scale * static_cast<double>(mean_distance);
        bp.K      = filter_rounds;
        pcp::algorithm::bilateral_filter_normals(ids.begin(), ids.end(), field.begin(), point_of, normal_of, bp);
        filter_ms = watch.lap_ms();
    }

    pcp::io::write_ply(std::filesystem::path{argv[2]}, cloud, field, pcp::io::ply_format_t::binary_little_endian);

    std::size_t unit = 0;
    for (auto const& n : field)
    {
        double const len = std::sqrt(double(n.nx()) * n.nx() + double(n.ny()) * n.ny() + double(n.nz()) * n.nz());
        if (std::abs(len - 1.) < 1e-5) ++unit;
    }
    std::printf("{\"points\": %zu, \"unit_normals\": %zu, \"mean_distance\": %.9g, \"tree_ms\": %.1f, \"estimate_normals_ms\": %.1f, "
                "\"orientation_ms\": %.1f, \"bilateral_ms\": %.1f}\n",
                field.size(), unit, static_cast<double>(mean_distance), tree_ms, normals_ms, orient_ms, filter_ms);
    return 0;
}
