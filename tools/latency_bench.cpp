// tools/latency_bench.cpp -- the shape of the reference's own kNN benchmark
// (benchmark/spatial_data_structures_benchmark.cpp:243-264, bm_linked_octree_knn_search): build an octree over N points
// U(-100, 100)^3 on the voxel grid [-100, 100]^3, then ONE random query per iteration, k = 10, through the drop-in
// header (pcp::linked_octree_t::nearest_neighbours).  Prints microseconds per query; arbitrary targets never hit the
// per-point cache, so this is the single-query latency path (pcpx_few.hip) end to end.  Also times pcp::estimate_normal
// on the returned neighbourhood and a batch of 256 queries through nearest_neighbours_batch.
// Round 3 adds the reference's other two benchmark shapes through the same headers:
//   * construction from host points (:108-148, bm_linked_octree_construction / bm_linked_kdtree_construction): container from the
//     point vector + the device index (the containers here build it on first use; .index() forces it, as size() is used there),
//   * one random axis-aligned range per iteration (:169-213, bm_linked_octree_range_search / bm_linked_kdtree_range_search): a box
//     of half-width up to 1 around a random centre, results returned as elements.
// build: g++ -std=c++17 -O2 -I include tools/latency_bench.cpp -L point-cloud-processing_amd -lpcpx -Wl,-rpath,... -pthread
#include <pcp/pcp.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    std::uint64_t const n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1ull << 20);
    int const iters       = argc > 2 ? std::atoi(argv[2]) : 2000;
    std::size_t const k   = 10;
    float const min = -100.f, max = 100.f;
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> coord(min, max);
    std::vector<pcp::point_t> points;
    points.reserve(n);
    for (std::uint64_t i = 0; i < n; ++i) points.push_back(pcp::point_t{coord(gen), coord(gen), coord(gen)});
    auto const point_map = [](pcp::point_t const& p) { return p; };

    pcp::octree_parameters_t<pcp::point_t> params;
    params.voxel_grid = pcp::axis_aligned_bounding_box_t<pcp::point_t>{{min, min, min}, {max, max, max}};
    double t0 = now();
    pcp::linked_octree_t octree(points.cbegin(), points.cend(), point_map, params);
    (void)octree.nearest_neighbours(pcp::point_t{0.f, 0.f, 0.f}, k, point_map);  // first query builds the device index
    double const t_build = now() - t0;

    std::vector<pcp::point_t> refs;
    for (int i = 0; i < iters; ++i) refs.push_back(pcp::point_t{coord(gen), coord(gen), coord(gen)});
    std::size_t got = 0;
    for (int i = 0; i < 200 && i < iters; ++i) got += octree.nearest_neighbours(refs[static_cast<std::size_t>(i)], k, point_map).size();  // warm-up
    t0 = now();
    for (auto const& r : refs) got += octree.nearest_neighbours(r, k, point_map).size();
    double const t_knn = (now() - t0) / iters;

    // + the PCA normal of each neighbourhood (the per-point shape of estimate_normals' body)
    float acc = 0.f;
    int const niter2 = iters < 500 ? iters : 500;
    for (int i = 0; i < 20 && i < niter2; ++i)  // warm-up: the first estimate_normal of a process sets up the stream it runs on (8 ms on this stack)
    {
        auto const nn = octree.nearest_neighbours(refs[static_cast<std::size_t>(i)], k, point_map);
        acc += pcp::estimate_normal(nn.begin(), nn.end(), point_map).nx();
    }
    t0 = now();
    for (int i = 0; i < niter2; ++i)
    {
        auto const nn = octree.nearest_neighbours(refs[static_cast<std::size_t>(i)], k, point_map);
        auto const nrm = pcp::estimate_normal(nn.begin(), nn.end(), point_map);
        acc += nrm.nx();
    }
    double const t_knn_normal = (now() - t0) / niter2;

    t0 = now();
    int const nb = 20;
    for (int b = 0; b < nb; ++b)
    {
        auto const rows = octree.nearest_neighbours_batch(refs.begin(), refs.begin() + (iters < 256 ? iters : 256), point_map, k);
        got += rows.size();
    }
    double const t_batch = (now() - t0) / nb;

    // ---- range search: one random box per iteration (get_range, benchmark:46-63) ----
    std::uniform_real_distribution<float> centre(min + 1.f, max - 1.f), lo_b(-1.f, 0.f), hi_b(0.f, 1.f);
    auto const kd_map = [](pcp::point_t const& p) { return std::array<float, 3u>{p.x(), p.y(), p.z()}; };
    pcp::kdtree::construction_params_t kd_params;
    kd_params.compute_max_depth = true;
    t0 = now();
    pcp::basic_linked_kdtree_t<pcp::point_t, 3u, decltype(kd_map)> kdtree{points.begin(), points.end(), kd_map, kd_params};
    (void)kdtree.index();
    double const t_kd_build = now() - t0;
    std::vector<pcp::axis_aligned_bounding_box_t<pcp::point_t>> boxes;
    for (int i = 0; i < iters; ++i)
    {
        float const x = centre(gen), y = centre(gen), z = centre(gen);
        boxes.push_back({pcp::point_t{lo_b(gen) + x, lo_b(gen) + y, lo_b(gen) + z}, pcp::point_t{hi_b(gen) + x, hi_b(gen) + y, hi_b(gen) + z}});
    }
    std::size_t found = 0;
    for (int i = 0; i < 100 && i < iters; ++i) found += octree.range_search(boxes[static_cast<std::size_t>(i)], point_map).size();
    t0 = now();
    for (auto const& b : boxes) found += octree.range_search(b, point_map).size();
    double const t_range_oct = (now() - t0) / iters;
    t0 = now();
    for (auto const& b : boxes)
    {
        pcp::kd_axis_aligned_bounding_box_t<float, 3u> kb;
        kb.min = {b.min.x(), b.min.y(), b.min.z()};
        kb.max = {b.max.x(), b.max.y(), b.max.z()};
        found += kdtree.range_search(kb).size();
    }
    double const t_range_kd = (now() - t0) / iters;

    // ---- construction from the host vector, per iteration (benchmark:108-148) ----
    int const nbuild = 5;
    t0 = now();
    for (int i = 0; i < nbuild; ++i)
    {
        pcp::linked_octree_t oc(points.cbegin(), points.cend(), point_map, params);
        (void)oc.index();
        got += oc.size();
    }
    double const t_oct_build = (now() - t0) / nbuild;
    t0 = now();
    for (int i = 0; i < nbuild; ++i)
    {
        pcp::basic_linked_kdtree_t<pcp::point_t, 3u, decltype(kd_map)> kd{points.begin(), points.end(), kd_map, kd_params};
        (void)kd.index();
        got += kd.size();
    }
    double const t_kd_build_loop = (now() - t0) / nbuild;

    std::printf("{\"points\": %llu, \"k\": %zu, \"iterations\": %d, \"first_query_incl_index_build_ms\": %.3f, \"knn_single_query_us\": %.2f, "
                "\"knn_plus_estimate_normal_us\": %.2f, \"batch_of_256_queries_us\": %.2f, \"octree_range_search_us\": %.2f, "
                "\"kdtree_range_search_us\": %.2f, \"mean_points_per_range\": %.2f, \"octree_construction_ms\": %.3f, \"kdtree_construction_ms\": %.3f, "
                "\"kdtree_first_construction_ms\": %.3f, \"checksum\": %zu, \"acc\": %g}\n",
                static_cast<unsigned long long>(n), k, iters, t_build * 1e3, t_knn * 1e6, t_knn_normal * 1e6, t_batch * 1e6, t_range_oct * 1e6,
                t_range_kd * 1e6, static_cast<double>(found) / (2.0 * iters + 100.0), t_oct_build * 1e3, t_kd_build_loop * 1e3, t_kd_build * 1e3, got,
                static_cast<double>(acc));
    return 0;
}
