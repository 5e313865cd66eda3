// pcp::intersections::intersects -- drop-in for include/pcp/common/intersections.hpp.
// NOTE: the reference's box/sphere test compares the squared distance with `radius`, not radius^2
// (:101, :129).  For radius <= 1 that only prunes less; for radius > 1 the reference's range_search can
// miss points depending on its tree shape.  This overload implements the geometric predicate
// (d2 <= radius^2); the GPU range search returns exactly the points with contains() == true.
#ifndef PCP_COMMON_INTERSECTIONS_HPP
#define PCP_COMMON_INTERSECTIONS_HPP

#include "pcp/common/axis_aligned_bounding_box.hpp"
#include "pcp/common/sphere.hpp"

namespace pcp {
namespace intersections {

template <class Point>
inline bool intersects(axis_aligned_bounding_box_t<Point> const& a, axis_aligned_bounding_box_t<Point> const& b)
{
    return (a.max.x() >= b.min.x() && a.max.y() >= b.min.y() && a.max.z() >= b.min.z()) &&
           (a.min.x() <= b.max.x() && a.min.y() <= b.max.y() && a.min.z() <= b.max.z());
}
template <class T, std::size_t K>
inline bool intersects(kd_axis_aligned_bounding_box_t<T, K> const& a, kd_axis_aligned_bounding_box_t<T, K> const& b)
{
    for (std::size_t i = 0; i < K; ++i)
        if (!(a.max[i] >= b.min[i] && a.min[i] <= b.max[i])) return false;
    return true;
}
template <class Point>
inline bool intersects(sphere_t<Point> const& a, sphere_t<Point> const& b)
{
    auto const r = a.radius + b.radius;
    return common::squared_distance(a.center(), b.center()) <= r * r;
}
template <class Point>
inline bool intersects(axis_aligned_bounding_box_t<Point> const& b, sphere_t<Point> const& s)
{
    Point const c = s.center();
    return common::squared_distance(b.nearest_point_from(c), c) <= s.radius * s.radius;
}
template <class T>
inline bool intersects(kd_axis_aligned_bounding_box_t<T, 3> const& b, sphere_a<T> const& s)
{
    auto const c = s.center();
    return common::squared_distance(b.nearest_point_from(c), c) <= s.radius * s.radius;
}
template <class T>
inline bool intersects(sphere_a<T> const& s, kd_axis_aligned_bounding_box_t<T, 3> const& b)
{
    return intersects(b, s);
}
template <class Point>
inline bool intersects(sphere_t<Point> const& s, axis_aligned_bounding_box_t<Point> const& b)
{
    return intersects(b, s);
}

} // namespace intersections
} // namespace pcp

#endif
