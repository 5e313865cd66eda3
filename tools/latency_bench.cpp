// tools/latency_bench.cpp -- the shape of the reference's own kNN benchmark
// (benchmark/spatial_data_structures_benchmark.cpp:243-264, bm_linked_octree_knn_search): build an octree over N points
// U(-100, 100)^3 on the voxel grid [-100, 100]^3, then ONE random query per iteration, k = 10, through the drop-in
// header (pcp::linked_octree_t::nearest_neighbours).  Prints microseconds per query; arbitrary targets never hit the
// per-point cache, so this is the single-query latency path (pcpx_few.hip) end to end.  Also times pcp::estimate_normal
// on the returned neighbourhood and a batch of 256 queries through nearest_neighbours_batch.
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
    t0 = now();
    float acc = 0.f;
    int const niter2 = iters < 500 ? iters : 500;
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

    std::printf("{\"points\": %llu, \"k\": %zu, \"iterations\": %d, \"first_query_incl_index_build_ms\": %.3f, \"knn_single_query_us\": %.2f, "
                "\"knn_plus_estimate_normal_us\": %.2f, \"batch_of_256_queries_us\": %.2f, \"checksum\": %zu, \"acc\": %g}\n",
                static_cast<unsigned long long>(n), k, iters, t_build * 1e3, t_knn * 1e6, t_knn_normal * 1e6, t_batch * 1e6, got, static_cast<double>(acc));
    return 0;
}
