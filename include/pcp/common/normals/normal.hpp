// pcp::basic_normal_t / pcp::normal_t -- drop-in for include/pcp/common/normals/normal.hpp:19-94.
// operator== on floating components is the reference's 1e-5 per-component comparison (:55-70).
#ifndef PCP_COMMON_NORMALS_NORMAL_HPP
#define PCP_COMMON_NORMALS_NORMAL_HPP

#include <cmath>
#include <type_traits>

namespace pcp {

template <class T>
struct basic_normal_t
{
    using component_type = T;

    basic_normal_t() = default;
    basic_normal_t(T x, T y, T z) : c_{x, y, z} {}

    T const& x() const { return c_[0]; }
    T const& y() const { return c_[1]; }
    T const& z() const { return c_[2]; }
    void x(T v) { c_[0] = v; }
    void y(T v) { c_[1] = v; }
    void z(T v) { c_[2] = v; }
    T const& nx() const { return c_[0]; }
    T const& ny() const { return c_[1]; }
    T const& nz() const { return c_[2]; }
    void nx(T v) { c_[0] = v; }
    void ny(T v) { c_[1] = v; }
    void nz(T v) { c_[2] = v; }

    friend basic_normal_t operator*(T k, basic_normal_t n) { return {k * n.c_[0], k * n.c_[1], k * n.c_[2]}; }
    friend basic_normal_t operator/(basic_normal_t n, T k) { return {n.c_[0] / k, n.c_[1] / k, n.c_[2] / k}; }
    basic_normal_t operator+(basic_normal_t const& o) const { return {c_[0] + o.c_[0], c_[1] + o.c_[1], c_[2] + o.c_[2]}; }
    basic_normal_t operator-(basic_normal_t const& o) const { return {c_[0] - o.c_[0], c_[1] - o.c_[1], c_[2] - o.c_[2]}; }
    basic_normal_t operator-() const { return {-c_[0], -c_[1], -c_[2]}; }

    bool operator==(basic_normal_t const& o) const
    {
        if constexpr (std::is_integral_v<T>)
            return c_[0] == o.c_[0] && c_[1] == o.c_[1] && c_[2] == o.c_[2];
        else
        {
            T const e = static_cast<T>(1e-5);
            return std::abs(c_[0] - o.c_[0]) < e && std::abs(c_[1] - o.c_[1]) < e && std::abs(c_[2] - o.c_[2]) < e;
        }
    }
    bool operator!=(basic_normal_t const& o) const { return !(*this == o); }

  private:
    T c_[3] = {T(0), T(0), T(0)};
};

using normal_t = basic_normal_t<float>;

} // namespace pcp

#endif
