// pcp::estimate_normal -- drop-in for include/pcp/common/normals/normal_estimation.hpp:32-78.
// PCA normal of one neighbourhood: row mean, centred scatter matrix V'V'^T (not divided by n), eigenvector
// of the smallest eigenvalue from the same float32 tridiagonal-QL iteration Eigen 3.3.8's
// SelfAdjointEigenSolver<Matrix3f>::compute runs; sign arbitrary.  Evaluated on the GPU through
// pcpx_estimate_normal (one small launch: the batched path is algorithm::estimate_normals).
#ifndef PCP_COMMON_NORMALS_NORMAL_ESTIMATION_HPP
#define PCP_COMMON_NORMALS_NORMAL_ESTIMATION_HPP

#include "pcp/common/normals/normal.hpp"
#include "pcp/gpu/device_index.hpp"

#include <iterator>
#include <vector>

namespace pcp {

template <class ForwardIter, class PointViewMap, class Normal = pcp::normal_t>
Normal estimate_normal(ForwardIter it, ForwardIter end, PointViewMap const& point_map)
{
    std::vector<float> xyz;
    for (; it != end; ++it)
    {
        auto const p = point_map(*it);
        xyz.push_back(static_cast<float>(p.x()));
        xyz.push_back(static_cast<float>(p.y()));
        xyz.push_back(static_cast<float>(p.z()));
    }
    float n[3] = {0.f, 0.f, 0.f};
    gpu::check(pcpx_estimate_normal(xyz.data(), xyz.size() / 3, 0, n), "pcpx_estimate_normal");
    using T = typename Normal::component_type;
    return Normal{static_cast<T>(n[0]), static_cast<T>(n[1]), static_cast<T>(n[2])};
}

} // namespace pcp

#endif
