// pcp::common::inner_product / norm / squared_distance -- drop-in for include/pcp/common/norm.hpp.
// squared_distance is evaluated exactly as the reference does (:102-112, :123-141): d = p2 - p1 per
// axis, dx*dx + dy*dy + dz*dz left to right; the GPU kernels use the same expression without FMA.
#ifndef PCP_COMMON_NORM_HPP
#define PCP_COMMON_NORM_HPP

#include <array>
#include <cmath>
#include <cstddef>

namespace pcp {
namespace common {

struct l2
{
};

template <class V1, class V2>
inline typename V1::component_type inner_product(V1 const& a, V2 const& b)
{
    return b.x() * a.x() + b.y() * a.y() + b.z() * a.z();
}

template <class V, class Norm = l2>
typename V::component_type norm(V const& v, Norm const& = Norm{})
{
    return std::sqrt(v.x() * v.x() + v.y() * v.y() + v.z() * v.z());
}

template <class P1, class P2>
inline typename P1::coordinate_type squared_distance(P1 const& p1, P2 const& p2)
{
    auto const dx = p2.x() - p1.x();
    auto const dy = p2.y() - p1.y();
    auto const dz = p2.z() - p1.z();
    return dx * dx + dy * dy + dz * dz;
}

template <class T, std::size_t K>
inline T squared_distance(std::array<T, K> const& p1, std::array<T, K> const& p2)
{
    T d = T{0};
    for (std::size_t i = 0; i < K; ++i)
    {
        T const c = p2[i] - p1[i];
        d = d + c * c;
    }
    return d;
}

} // namespace common
} // namespace pcp

#endif
