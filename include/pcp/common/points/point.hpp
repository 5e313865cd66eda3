// pcp::basic_point_t / pcp::point_t -- drop-in for include/pcp/common/points/point.hpp:21-93 of the
// reference: a 3-component point with x()/y()/z() accessors (the PointView concept every container and
// algorithm of the hot path is written against).  12 bytes for float, array-of-structures.
#ifndef PCP_COMMON_POINTS_POINT_HPP
#define PCP_COMMON_POINTS_POINT_HPP

#include "pcp/common/vector3d.hpp"

#include <utility>

namespace pcp {

template <class T>
class basic_point_t
{
  public:
    using component_type  = T;
    using coordinate_type = T;
    using self_type       = basic_point_t<T>;

    constexpr basic_point_t() noexcept = default;
    constexpr basic_point_t(T x, T y, T z) noexcept : c_{x, y, z} {}
    // from anything with x(), y(), z() (a view, a vertex, another point type)
    template <class PointView, class = decltype(std::declval<PointView const&>().x())>
    basic_point_t(PointView const& v) noexcept : c_{v.x(), v.y(), v.z()}
    {
    }

    T const& x() const { return c_[0]; }
    T const& y() const { return c_[1]; }
    T const& z() const { return c_[2]; }
    void x(T const& v) { c_[0] = v; }
    void y(T const& v) { c_[1] = v; }
    void z(T const& v) { c_[2] = v; }

    friend self_type operator*(T k, self_type const& p) noexcept { return {k * p.c_[0], k * p.c_[1], k * p.c_[2]}; }
    friend self_type operator/(self_type const& p, T k) noexcept { return {p.c_[0] / k, p.c_[1] / k, p.c_[2] / k}; }
    template <class Vector3d>
    self_type operator+(Vector3d const& v) const noexcept
    {
        return {c_[0] + v.x(), c_[1] + v.y(), c_[2] + v.z()};
    }
    // displacement between two points (reference point.hpp:73-80): point - point view = vector
    template <class PointView, class Vector3d = common::basic_vector3d_t<coordinate_type>>
    Vector3d operator-(PointView const& other) const noexcept
    {
        return Vector3d{c_[0] - other.x(), c_[1] - other.y(), c_[2] - other.z()};
    }
    self_type operator-() const noexcept { return {-c_[0], -c_[1], -c_[2]}; }

  private:
    T c_[3] = {T(0), T(0), T(0)};
};

using point_t = basic_point_t<float>;

} // namespace pcp

#endif
