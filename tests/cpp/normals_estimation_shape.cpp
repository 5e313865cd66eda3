// The reference's second example (examples/normals_estimation.cpp:12-148) against this repository's headers, call for
// call: read_ply -> index elements -> kd-tree over a coordinate map -> algorithm::estimate_normals with a plain lambda
// knn map and a transform that stores into the caller's vector -> propagate_normal_orientations over the same lambda ->
// average_distance_to_neighbors -> bilateral_filter_normals writing into the vector its own normal map reads ->
// write_ply.  Only range-v3's iota view (a third-party dependency this image lacks) is replaced by std::iota.
// usage: normals_estimation_shape <in.ply> <out.ply> <k> <parallel|seq> [bilateral <iterations> <sigmaf mult> <sigmag mult>]
// prints one JSON line (sizes, the mean neighbour distance, how far the filter turned the normals, per-phase times)
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
#include <execution>
#include <filesystem>
#include <numeric>
#include <string>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::uint64_t const k = argc >= 4 ? std::stoull(argv[3]) : 10u;
    bool const parallel   = argc >= 5 ? std::string(argv[4]) == "parallel" : false;
    bool const bilateral  = argc >= 6 ? std::string(argv[5]) == "bilateral" : false;
    if (bilateral && argc < 9) return 2;
    std::size_t const bk           = bilateral ? std::stoull(argv[6]) : 0u;
    double const sigmaf_multiplier = bilateral ? std::stod(argv[7]) : 0.;
    double const sigmag_multiplier = bilateral ? std::stod(argv[8]) : 0.;
    std::filesystem::path ply_point_cloud = argv[1];

    using index_type  = std::uint64_t;
    using point_type  = pcp::point_t;
    using normal_type = pcp::normal_t;

    auto [p, n] = pcp::io::read_ply<point_type, normal_type>(ply_point_cloud);
    std::vector<point_type> points = std::move(p);
    if (points.empty()) { std::printf("could not read %s\n", argv[1]); return 1; }
    std::vector<normal_type> normals(points.size());
    std::vector<index_type> indices(points.size());
    std::iota(indices.begin(), indices.end(), index_type{0});

    auto const index_map      = [](index_type const& i) { return i; };
    auto const point_map      = [&](index_type const& i) { return points[i]; };
    auto const normal_map     = [&](index_type const& i) { return normals[i]; };
    auto const coordinate_map = [&](index_type const& i) { return std::array<float, 3u>{points[i].x(), points[i].y(), points[i].z()}; };

    double t0 = now();
    pcp::kdtree::construction_params_t params;
    params.compute_max_depth = true;
    pcp::basic_linked_kdtree_t<index_type, 3u, decltype(coordinate_map)> kdtree{indices.begin(), indices.end(), coordinate_map, params};
    double const t_tree = now() - t0;

    auto const knn          = [&](index_type const& i) { return kdtree.nearest_neighbours(i, k); };
    auto const transform_op = [&](index_type const& i, pcp::normal_t const& nrm) {
        normals[i] = nrm;
        return i;
    };

    t0 = now();
    if (parallel)
        pcp::algorithm::estimate_normals(std::execution::par, indices.begin(), indices.end(), indices.begin(), point_map, knn, transform_op);
    else
        pcp::algorithm::estimate_normals(std::execution::seq, indices.begin(), indices.end(), indices.begin(), point_map, knn, transform_op);
    double const t_normals = now() - t0;

    t0 = now();
    pcp::algorithm::propagate_normal_orientations(indices.begin(), indices.end(), index_map, knn, point_map, normal_map, transform_op);
    double const t_orient = now() - t0;

    std::vector<normal_type> const before = normals;
    float avg = 0.f;
    double t_filter = 0.;
    if (bilateral)
    {
        t0  = now();
        avg = pcp::algorithm::average_distance_to_neighbors(indices.begin(), indices.end(), point_map, knn);
        pcp::algorithm::bilateral::params_t bparams;
        bparams.K      = bk;
        bparams.sigmaf = sigmaf_multiplier * static_cast<double>(avg);
        bparams.sigmag = sigmag_multiplier * static_cast<double>(avg);
        pcp::algorithm::bilateral_filter_normals(indices.begin(), indices.end(), normals.begin(), point_map, normal_map, bparams);
        t_filter = now() - t0;
    }

    pcp::io::write_ply(std::filesystem::path(argv[2]), points, normals, pcp::io::ply_format_t::binary_little_endian);

    double worst_len = 0., mean_cos = 0.;
    std::size_t finite = 0;
    for (std::size_t i = 0; i < normals.size(); ++i)
    {
        auto const& a = normals[i];
        auto const& b = before[i];
        double const len = std::sqrt(double(a.nx()) * a.nx() + double(a.ny()) * a.ny() + double(a.nz()) * a.nz());
        if (std::isfinite(len)) ++finite;
        worst_len = std::max(worst_len, std::abs(len - 1.));
        mean_cos += double(a.nx()) * b.nx() + double(a.ny()) * b.ny() + double(a.nz()) * b.nz();
    }
    mean_cos /= static_cast<double>(normals.size());
    std::printf("{\"points\": %zu, \"finite_normals\": %zu, \"worst_length_error\": %.3g, \"mean_distance\": %.9g, "
                "\"mean_cos_to_unfiltered\": %.6f, \"tree_ms\": %.1f, \"estimate_normals_ms\": %.1f, \"orientation_ms\": %.1f, "
                "\"bilateral_ms\": %.1f}\n",
                normals.size(), finite, worst_len, static_cast<double>(avg), mean_cos, t_tree * 1e3, t_normals * 1e3, t_orient * 1e3,
                t_filter * 1e3);
    return 0;
}
