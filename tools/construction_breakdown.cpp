// Where the time of a container construction from a host vector goes (benchmark/spatial_data_structures_benchmark.cpp:108-148 shape):
// the constructor (property map evaluated over the range, coordinates captured, elements copied) and the device index it builds on
// first use (.index(): upload + build).  usage: construction_breakdown [n] [repeats]
// build: g++ -std=c++17 -O2 -I include tools/construction_breakdown.cpp -L point-cloud-processing_amd -lpcpx -Wl,-rpath,... -pthread
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
    int const reps        = argc > 2 ? std::atoi(argv[2]) : 5;
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> coord(-100.f, 100.f);
    std::vector<pcp::point_t> points;
    points.reserve(n);
    for (std::uint64_t i = 0; i < n; ++i) points.push_back(pcp::point_t{coord(gen), coord(gen), coord(gen)});
    auto const point_map = [](pcp::point_t const& p) { return p; };
    auto const kd_map    = [](pcp::point_t const& p) { return std::array<float, 3u>{p.x(), p.y(), p.z()}; };
    pcp::octree_parameters_t<pcp::point_t> params;
    params.voxel_grid = {pcp::point_t{-100.f, -100.f, -100.f}, pcp::point_t{100.f, 100.f, 100.f}};
    pcp::kdtree::construction_params_t kd_params;
    kd_params.compute_max_depth = true;
    {
        pcp::linked_octree_t warm(points.cbegin(), points.cend(), point_map, params);  // (first touch of the device, the library, the allocator)
        (void)warm.index();
    }
    double oc_ctor = 0, oc_index = 0, oc_dtor = 0, kd_ctor = 0, kd_index = 0, kd_dtor = 0;
    std::size_t got = 0;
    for (int r = 0; r < reps; ++r)
    {
        double t0 = now(), t1, t2;
        {
            pcp::linked_octree_t oc(points.cbegin(), points.cend(), point_map, params);
            t1 = now();
            (void)oc.index();
            t2 = now();
            got += oc.size();
        }
        double t3 = now();
        oc_ctor += t1 - t0, oc_index += t2 - t1, oc_dtor += t3 - t2;
        t0 = now();
        {
            pcp::basic_linked_kdtree_t<pcp::point_t, 3u, decltype(kd_map)> kd{points.begin(), points.end(), kd_map, kd_params};
            t1 = now();
            (void)kd.index();
            t2 = now();
            got += kd.size();
        }
        t3 = now();
        kd_ctor += t1 - t0, kd_index += t2 - t1, kd_dtor += t3 - t2;
    }
    double const s = 1e3 / reps;
    std::printf("{\"points\": %llu, \"repeats\": %d, \"octree_constructor_ms\": %.3f, \"octree_device_index_ms\": %.3f, \"octree_destructor_ms\": %.3f, "
                "\"kdtree_constructor_ms\": %.3f, \"kdtree_device_index_ms\": %.3f, \"kdtree_destructor_ms\": %.3f, \"host_threads\": %u, \"checksum\": %zu}\n",
                static_cast<unsigned long long>(n), reps, oc_ctor * s, oc_index * s, oc_dtor * s, kd_ctor * s, kd_index * s, kd_dtor * s,
                pcp::gpu::capture_threads(n), got);
    return 0;
}
