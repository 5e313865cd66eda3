// pcp::common::floating_point_equals / are_vectors_equal -- drop-in for
// include/pcp/common/vector3d_queries.hpp:30-35 and :47-64: strict |a-b| < eps per component.  This is
// the test the kNN kernels apply to exclude points coincident with the query.
#ifndef PCP_COMMON_VECTOR3D_QUERIES_HPP
#define PCP_COMMON_VECTOR3D_QUERIES_HPP

#include <cmath>

namespace pcp {
namespace common {

template <class T>
bool floating_point_equals(T a, T b, T eps = static_cast<T>(1e-5))
{
    return std::abs(a - b) < eps;
}

template <class V1, class V2>
bool are_vectors_equal(V1 const& a, V2 const& b,
                       typename V1::component_type eps = static_cast<typename V1::component_type>(1e-5))
{
    return floating_point_equals(a.x(), b.x(), eps) && floating_point_equals(a.y(), b.y(), eps) &&
           floating_point_equals(a.z(), b.z(), eps);
}

} // namespace common
} // namespace pcp

#endif
