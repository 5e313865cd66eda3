// pcp::sphere_t / pcp::sphere_a -- drop-in for include/pcp/common/sphere.hpp:19-57:
// contains(p) <=> squared_distance(center, p) <= radius * radius.
#ifndef PCP_COMMON_SPHERE_HPP
#define PCP_COMMON_SPHERE_HPP

#include "pcp/common/norm.hpp"

#include <array>

namespace pcp {

template <class Point>
struct sphere_t
{
    Point position{0.f, 0.f, 0.f};
    typename Point::coordinate_type radius = static_cast<typename Point::coordinate_type>(0.);
    Point center() const { return position; }
    bool contains(Point const& p) const { return common::squared_distance(position, p) <= radius * radius; }
};

template <class T>
struct sphere_a
{
    using point_type = std::array<T, 3>;
    point_type position;
    T radius;
    point_type center() const { return position; }
    bool contains(point_type const& p) const { return common::squared_distance(position, p) <= radius * radius; }
};

} // namespace pcp

#endif
