// pcp::common::basic_vector3d_t / vector3d_t -- drop-in for include/pcp/common/vector3d.hpp:19-97 of the reference:
// the displacement type that `point - point` yields (include/pcp/common/points/point.hpp:74), with x()/y()/z()
// accessors and setters (the Vector3d concept, include/pcp/traits/vector3d_traits.hpp).
#ifndef PCP_COMMON_VECTOR3D_HPP
#define PCP_COMMON_VECTOR3D_HPP

#include <utility>

namespace pcp {
namespace common {

template <class T>
class basic_vector3d_t
{
  public:
    using component_type = T;
    using self_type      = basic_vector3d_t<T>;

    constexpr basic_vector3d_t() noexcept = default;
    constexpr basic_vector3d_t(T x, T y, T z) noexcept : c_{x, y, z} {}
    // from anything with x(), y(), z() (another vector type, a normal)
    template <class Vector3d, class = decltype(std::declval<Vector3d const&>().x())>
    basic_vector3d_t(Vector3d const& v) : c_{v.x(), v.y(), v.z()}
    {
    }

    T x() const { return c_[0]; }
    T y() const { return c_[1]; }
    T z() const { return c_[2]; }
    void x(T v) { c_[0] = v; }
    void y(T v) { c_[1] = v; }
    void z(T v) { c_[2] = v; }

    friend self_type operator*(T k, self_type const& v) { return {k * v.c_[0], k * v.c_[1], k * v.c_[2]}; }
    friend self_type operator/(self_type const& v, T k) { return {v.c_[0] / k, v.c_[1] / k, v.c_[2] / k}; }
    self_type operator+(self_type const& v) const { return {c_[0] + v.c_[0], c_[1] + v.c_[1], c_[2] + v.c_[2]}; }
    self_type operator-(self_type const& v) const { return {c_[0] - v.c_[0], c_[1] - v.c_[1], c_[2] - v.c_[2]}; }

  private:
    T c_[3] = {T(0), T(0), T(0)};
};

using vector3d_t = basic_vector3d_t<float>;

} // namespace common
} // namespace pcp

#endif
