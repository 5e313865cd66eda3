// Container construction from a large range (pcp/gpu/host_capture.hpp: the property map evaluated on several threads) gives what
// the reference's one-by-one insertion gives (include/pcp/octree/linked_octree.hpp:83-121, linked_octree_node.hpp:143-175;
// include/pcp/kdtree/linked_kdtree.hpp:100-135): the same elements in the same order, points outside the voxel grid skipped,
// the same coordinates, the same bounding boxes.  Host only: no query is made, so no device index is built.
#include <pcp/pcp.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <list>
#include <random>
#include <vector>

static int failures = 0;
#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

template <class A, class B>
static bool same_floats(A const& a, B const& b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
}

int main()
{
    std::size_t const n = 300000;  // well above pcp::gpu::parallel_capture_threshold
    static_assert(300000 > pcp::gpu::parallel_capture_threshold, "the large-range path");
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> coord(-1.f, 1.f);
    std::vector<pcp::point_t> points;
    for (std::size_t i = 0; i < n; ++i) points.push_back(pcp::point_t{coord(gen), coord(gen), coord(gen)});
    auto const point_map = [](pcp::point_t const& p) { return p; };

    // ---- octree, explicit voxel grid smaller than the cloud: about half of the points are outside ----
    pcp::octree_parameters_t<pcp::point_t> params;
    params.voxel_grid = {pcp::point_t{-0.8f, -0.7f, -0.9f}, pcp::point_t{0.9f, 0.6f, 0.8f}};
    // a point exactly on the grid's faces is inside (inclusive test), the first and the last point are outside
    points[1]     = pcp::point_t{-0.8f, 0.6f, 0.8f};
    points[0]     = pcp::point_t{-0.9f, 0.f, 0.f};
    points[n - 1] = pcp::point_t{0.f, 0.f, 0.85f};
    pcp::linked_octree_t many(points.cbegin(), points.cend(), point_map, params);
    pcp::linked_octree_t one(params);
    std::size_t inserted = 0;
    for (auto const& p : points) inserted += one.insert(p, point_map) ? 1u : 0u;
    CHECK(inserted < n && inserted > n / 4);
    CHECK(many.size() == inserted && one.size() == inserted);
    CHECK(same_floats(many.coordinates(), one.coordinates()));
    {
        auto a = many.cbegin();
        auto b = one.cbegin();
        bool same = true;
        for (; a != many.cend() && b != one.cend(); ++a, ++b) same = same && a->x() == b->x() && a->y() == b->y() && a->z() == b->z();
        CHECK(same && a == many.cend() && b == one.cend());
    }
    CHECK(many.cbegin()->x() == -0.8f);  // points[1]: points[0] was dropped
    // insert() of a large range into a container that already holds elements appends, and reports what it inserted
    pcp::linked_octree_t grown(params);
    CHECK(grown.insert(points[1], point_map));
    CHECK(grown.insert(points.cbegin(), points.cend(), point_map) == inserted);
    CHECK(grown.size() == inserted + 1);
    CHECK(std::memcmp(grown.coordinates().data() + 3, one.coordinates().data(), one.coordinates().size() * sizeof(float)) == 0);
    // nothing outside: the common case
    pcp::octree_parameters_t<pcp::point_t> wide;
    wide.voxel_grid = {pcp::point_t{-1.f, -1.f, -1.f}, pcp::point_t{1.f, 1.f, 1.f}};
    pcp::linked_octree_t all(points.cbegin(), points.cend(), point_map, wide);
    CHECK(all.size() == n);
    // a range that is not random access takes the one-by-one loop and gives the same container
    std::list<pcp::point_t> as_list(points.begin(), points.end());
    pcp::linked_octree_t from_list(as_list.cbegin(), as_list.cend(), point_map, params);
    CHECK(from_list.size() == inserted && same_floats(from_list.coordinates(), one.coordinates()));
    // a random-access range that is not contiguous
    std::deque<pcp::point_t> as_deque(points.begin(), points.end());
    pcp::linked_octree_t from_deque(as_deque.cbegin(), as_deque.cend(), point_map, params);
    CHECK(from_deque.size() == inserted && same_floats(from_deque.coordinates(), one.coordinates()));

    // ---- octree, bounding box of the range as the voxel grid (linked_octree.hpp:103-121) ----
    pcp::linked_octree_t boxed(points.cbegin(), points.cend(), point_map);
    auto const ref_box = pcp::bounding_box<std::vector<pcp::point_t>::const_iterator, pcp::point_t>(points.cbegin(), points.cend());
    CHECK(boxed.size() == n);
    CHECK(boxed.voxel_grid().min.x() == ref_box.min.x() && boxed.voxel_grid().min.y() == ref_box.min.y() && boxed.voxel_grid().min.z() == ref_box.min.z());
    CHECK(boxed.voxel_grid().max.x() == ref_box.max.x() && boxed.voxel_grid().max.y() == ref_box.max.y() && boxed.voxel_grid().max.z() == ref_box.max.z());

    // ---- elements that are views of caller memory (point_view_t) and indices (the shape of examples/normals_estimation.cpp) ----
    std::vector<std::uint64_t> ids(n);
    for (std::size_t i = 0; i < n; ++i) ids[i] = i;
    auto const id_map = [&points](std::uint64_t i) { return points[i]; };
    pcp::basic_linked_octree_t<std::uint64_t> by_id(ids.cbegin(), ids.cend(), id_map, params);
    CHECK(by_id.size() == inserted && same_floats(by_id.coordinates(), one.coordinates()));
    CHECK(*by_id.cbegin() == 1u);

    // ---- kd-tree: stored elements, coordinates, aabb() ----
    auto const kd_map = [](pcp::point_t const& p) { return std::array<float, 3u>{p.x(), p.y(), p.z()}; };
    pcp::basic_linked_kdtree_t<pcp::point_t, 3u, decltype(kd_map)> kd{points.begin(), points.end(), kd_map};
    pcp::basic_linked_kdtree_t<pcp::point_t, 3u, decltype(kd_map)> kd_list{as_list.begin(), as_list.end(), kd_map};
    auto const kd_box = pcp::kd_bounding_box<float, 3u, decltype(kd_map), std::vector<pcp::point_t>::const_iterator>(points.cbegin(), points.cend(), kd_map);
    CHECK(kd.size() == n && kd_list.size() == n);
    CHECK(same_floats(kd.coordinates(), kd_list.coordinates()));
    CHECK(kd.aabb().min == kd_box.min && kd.aabb().max == kd_box.max);
    CHECK(kd_list.aabb().min == kd_box.min && kd_list.aabb().max == kd_box.max);
    {
        bool same   = true;
        std::size_t i = 0;
        for (auto it = kd.cbegin(); it != kd.cend(); ++it, ++i) same = same && it->x() == points[i].x() && it->y() == points[i].y() && it->z() == points[i].z();
        CHECK(same && i == n);
        CHECK(kd.coordinates()[3 * 12345 + 1] == points[12345].y());
    }
    auto const xy_map = [](pcp::point_t const& p) { return std::array<float, 2u>{p.x(), p.y()}; };
    pcp::basic_linked_kdtree_t<pcp::point_t, 2u, decltype(xy_map)> kd2{points.begin(), points.end(), xy_map};
    CHECK(kd2.size() == n && kd2.coordinates()[3 * 777 + 2] == 0.f && kd2.coordinates()[3 * 777] == points[777].x());
    CHECK(kd2.aabb().min[0] == kd_box.min[0] && kd2.aabb().max[1] == kd_box.max[1]);

    // ---- a property map that throws reaches the caller as it does from the one-by-one loop (not std::terminate from a thread) ----
    {
        struct refused {};
        auto const throwing_map = [&points](std::uint64_t i) {
            if (i == 222222u) throw refused{};
            return points[i];
        };
        bool caught = false;
        try
        {
            pcp::basic_linked_octree_t<std::uint64_t> never(ids.cbegin(), ids.cend(), throwing_map, params);
            CHECK(never.size() == 0u && false);
        }
        catch (refused const&)
        {
            caught = true;
        }
        CHECK(caught);
        // ... and insert() of such a range leaves the container as it was
        pcp::basic_linked_octree_t<std::uint64_t> kept(params);
        CHECK(kept.insert(ids[1], throwing_map));
        caught = false;
        try
        {
            kept.insert(ids.cbegin(), ids.cend(), throwing_map);
        }
        catch (refused const&)
        {
            caught = true;
        }
        CHECK(caught && kept.size() == 1u && kept.coordinates().size() == 3u && *kept.cbegin() == 1u);
    }

    if (failures) return 1;
    std::printf("host capture: ok\n");
    return 0;
}
