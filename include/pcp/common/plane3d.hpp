// pcp::common::basic_plane3d_t / plane3d_t / tangent_plane -- drop-in for include/pcp/common/plane3d.hpp
// (:28-75 plane type, :93-105 tangent_plane) and center_of_geometry (vector3d_queries.hpp:77-99).
#ifndef PCP_COMMON_PLANE3D_HPP
#define PCP_COMMON_PLANE3D_HPP

#include "pcp/common/normals/normal.hpp"
#include "pcp/common/normals/normal_estimation.hpp"
#include "pcp/common/points/point.hpp"
#include "pcp/common/vector3d_queries.hpp"

#include <iterator>

namespace pcp {
namespace common {

// sum from the zero point in sequence order, then divide by n
template <class ForwardIter, class PointMap>
auto center_of_geometry(ForwardIter begin, ForwardIter end, PointMap const& point_map)
{
    using point_type = std::decay_t<decltype(point_map(*begin))>;
    using T          = typename point_type::coordinate_type;
    T sx = T(0), sy = T(0), sz = T(0);
    std::size_t n = 0;
    for (; begin != end; ++begin, ++n)
    {
        auto const p = point_map(*begin);
        sx = sx + p.x();
        sy = sy + p.y();
        sz = sz + p.z();
    }
    T const np = static_cast<T>(n);
    return pcp::basic_point_t<T>{sx / np, sy / np, sz / np};
}

template <class Point, class Normal>
class basic_plane3d_t
{
  public:
    using point_type     = Point;
    using normal_type    = Normal;
    using component_type = typename point_type::component_type;

    basic_plane3d_t() noexcept = default;
    basic_plane3d_t(point_type const& p, normal_type const& n) : point_(p), normal_(n) {}

    normal_type const& normal() const { return normal_; }
    point_type const& point() const { return point_; }
    void normal(normal_type const& n) { normal_ = n; }
    void point(point_type const& p) { point_ = p; }

    template <class PointView>
    component_type signed_distance_to(PointView const& p) const
    {
        component_type const dx = p.x() - point_.x(), dy = p.y() - point_.y(), dz = p.z() - point_.z();
        return normal_.x() * dx + normal_.y() * dy + normal_.z() * dz;
    }
    template <class PointView>
    bool contains(PointView const& p, component_type eps = static_cast<component_type>(1e-5)) const
    {
        return floating_point_equals(signed_distance_to(p), static_cast<component_type>(0.0), eps);
    }

  private:
    point_type point_;
    normal_type normal_;
};

using plane3d_t = basic_plane3d_t<pcp::point_t, pcp::normal_t>;

template <class ForwardIter, class PointMap, class Plane = plane3d_t>
Plane tangent_plane(ForwardIter begin, ForwardIter end, PointMap const& point_map)
{
    using normal_type = typename Plane::normal_type;
    using point_type  = typename Plane::point_type;
    normal_type const normal = pcp::estimate_normal<ForwardIter, PointMap, normal_type>(begin, end, point_map);
    auto const c             = center_of_geometry(begin, end, point_map);
    return Plane(point_type{c.x(), c.y(), c.z()}, normal);
}

} // namespace common
} // namespace pcp

#endif
