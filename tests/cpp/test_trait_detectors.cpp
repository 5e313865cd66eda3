// Compiled (syntax only) by tests/test_cpp_api.py when /root/reference is present: the REFERENCE's own concept
// detectors (include/pcp/traits/*.hpp, std-only headers, included from the reference tree where they lie -- nothing is
// copied) are instantiated on THIS repository's drop-in types.  SURVEY.md section 2 row 9: the build must satisfy
// is_point_view_v, is_point_v, is_normal_v, is_knn_map_v, is_point_view_map_v, is_coordinate_map_v, is_range_v,
// is_vector3d_v, is_plane_v.  Include order: -I <repo>/include comes first, so "pcp/common/..." is this repository's
// header and "pcp/traits/..." (which this repository does not have) is the reference's.
#include "pcp/traits/coordinate_map.hpp"
#include "pcp/traits/knn_map.hpp"
#include "pcp/traits/normal_traits.hpp"
#include "pcp/traits/plane_traits.hpp"
#include "pcp/traits/point_map.hpp"
#include "pcp/traits/point_traits.hpp"
#include "pcp/traits/range_traits.hpp"
#include "pcp/traits/vector3d_traits.hpp"

#include "pcp/pcp.hpp"

#include <array>
#include <cstdint>
#include <vector>

namespace t = pcp::traits;

// value types
static_assert(t::is_point_view_v<pcp::point_t>, "point_t is a PointView");
static_assert(t::is_point_v<pcp::point_t>, "point_t is a Point (reference point_traits.hpp:50-75)");
static_assert(t::is_point_view_v<pcp::point_view_t>, "point_view_t is a PointView");
static_assert(t::is_point_view_v<pcp::vertex_t>, "vertex_t is a PointView");
static_assert(t::is_normal_v<pcp::normal_t>, "normal_t is a Normal");
static_assert(t::is_vector3d_v<pcp::common::vector3d_t>, "vector3d_t is a Vector3d");
static_assert(t::is_vector3d_v<pcp::normal_t>, "a normal is usable as a Vector3d (point + normal)");
static_assert(t::is_plane_v<pcp::common::plane3d_t>, "plane3d_t is a Plane");
static_assert(std::is_same_v<decltype(std::declval<pcp::point_t&>() - std::declval<pcp::point_t&>()), pcp::common::vector3d_t>,
              "point - point is a vector (reference points/point.hpp:73-80)");

// ranges
using aabb_t    = pcp::axis_aligned_bounding_box_t<pcp::point_t>;
using kd_aabb_t = pcp::kd_axis_aligned_bounding_box_t<float, 3>;
static_assert(t::is_range_v<pcp::sphere_t<pcp::point_t>, pcp::point_t>, "sphere_t is a Range");
static_assert(t::is_range_v<aabb_t, pcp::point_t>, "axis_aligned_bounding_box_t is a Range");
static_assert(t::is_range_v<kd_aabb_t, std::array<float, 3>>, "kd_axis_aligned_bounding_box_t is a Range");
// (sphere_a is not a Range by the reference's own detector either: the reference has no intersects(sphere_a, sphere_a),
//  include/pcp/common/intersections.hpp:113-147 only pairs it with kd boxes)
static_assert(!t::is_range_v<pcp::sphere_a<float>, std::array<float, 3>>, "as in the reference");

// property maps of the two canonical call sequences (examples/simple_example.cpp, examples/normals_estimation.cpp)
struct point_view_map_t
{
    pcp::point_view_t operator()(pcp::point_view_t const& p) const { return p; }
};
struct coordinate_map_t
{
    std::vector<pcp::point_t> const* points;
    std::array<float, 3> operator()(std::uint64_t i) const { return {(*points)[i].x(), (*points)[i].y(), (*points)[i].z()}; }
};
static_assert(t::is_point_view_map_v<point_view_map_t, pcp::point_view_t>, "PointViewMap");
static_assert(t::is_coordinate_map_v<coordinate_map_t, std::uint64_t, float, 3>, "CoordinateMap");

using octree_t = pcp::basic_linked_octree_t<pcp::point_view_t>;
using kdtree_t = pcp::basic_linked_kdtree_t<std::uint64_t, 3, coordinate_map_t>;
static_assert(t::is_knn_map_v<pcp::gpu::knn_map_t<octree_t, point_view_map_t>, pcp::point_view_t>, "gpu::knn_map_t is a KnnMap");
static_assert(t::is_knn_map_v<pcp::gpu::self_knn_map_t<octree_t>, pcp::point_view_t>, "gpu::self_knn_map_t is a KnnMap");
static_assert(t::is_knn_map_v<pcp::gpu::self_knn_map_t<kdtree_t>, std::uint64_t>, "gpu::self_knn_map_t over a kd-tree is a KnnMap");

int main() { return 0; }
