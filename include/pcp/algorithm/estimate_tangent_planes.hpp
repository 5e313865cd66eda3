// pcp::algorithm::estimate_tangent_planes -- drop-in for include/pcp/algorithm/estimate_tangent_planes.hpp
// (:50-98 execution-policy overload, :116-176 sequential overload): plane of an element = (centre of
// geometry of its k neighbours, their PCA normal).  With pcp::gpu::self_knn_map the whole loop is the fused
// GPU kernel (pcpx_tangent_planes_knn_self); any other KnnMap takes the reference's per-element path.
#ifndef PCP_ALGORITHM_ESTIMATE_TANGENT_PLANES_HPP
#define PCP_ALGORITHM_ESTIMATE_TANGENT_PLANES_HPP

#include "pcp/algorithm/estimate_normals.hpp"
#include "pcp/common/plane3d.hpp"

#include <iterator>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace pcp {
namespace algorithm {

template <class Input, class Plane>
inline auto const default_plane_transform = [](Input const&, Plane const& plane) { return plane; };

namespace detail {
template <class ForwardIter1, class PointMap, class KnnMap, class Plane, class Emit>
void estimate_tangent_planes_impl(ForwardIter1 begin, ForwardIter1 end, PointMap const& point_map, KnnMap const& knn, Emit&& emit)
{
    using knn_type    = std::remove_cv_t<std::remove_reference_t<KnnMap>>;
    using point_type  = typename Plane::point_type;
    using normal_type = typename Plane::normal_type;
    if constexpr (gpu::is_self_knn_map<knn_type>::value)
    {
        std::size_t const n = static_cast<std::size_t>(std::distance(begin, end));
        if (n != knn.tree->size())
            throw std::invalid_argument("self_knn_map: the range must be the container's own element sequence");
        std::vector<float> cen, nrm;
        knn.tree->index().tangent_planes_self(static_cast<std::uint32_t>(knn.k), knn.eps, n, cen, nrm);
        std::size_t i = 0;
        using T = typename point_type::coordinate_type;
        for (; begin != end; ++begin, ++i)
            emit(*begin, Plane(point_type{static_cast<T>(cen[3 * i]), static_cast<T>(cen[3 * i + 1]), static_cast<T>(cen[3 * i + 2])},
                               make_normal<normal_type>(nrm.data() + 3 * i)));
    }
    else
    {
        for (; begin != end; ++begin)
        {
            auto const neighbours = knn(*begin);
            emit(*begin, pcp::common::tangent_plane<decltype(std::begin(neighbours)), PointMap, Plane>(
                             std::begin(neighbours), std::end(neighbours), point_map));
        }
    }
}
} // namespace detail

template <class ExecutionPolicy, class ForwardIter1, class ForwardIter2, class PointMap, class KnnMap, class TransformOp,
          class Plane = pcp::common::plane3d_t,
          class = std::enable_if_t<std::is_invocable_v<TransformOp, typename std::iterator_traits<ForwardIter1>::value_type, Plane>>>
void estimate_tangent_planes(ExecutionPolicy&&, ForwardIter1 begin, ForwardIter1 end, ForwardIter2 out_begin,
                             PointMap const& point_map, KnnMap&& knn_map, TransformOp&& op)
{
    using value_type = typename std::iterator_traits<ForwardIter1>::value_type;
    detail::estimate_tangent_planes_impl<ForwardIter1, PointMap, KnnMap, Plane>(
        begin, end, point_map, knn_map, [&](value_type const& v, Plane const& p) { *out_begin++ = op(v, p); });
}

template <class ForwardIter1, class ForwardIter2, class PointMap, class KnnMap, class TransformOp,
          class Plane = pcp::common::plane3d_t,
          class = std::enable_if_t<std::is_invocable_v<TransformOp, typename std::iterator_traits<ForwardIter1>::value_type, Plane>>>
void estimate_tangent_planes(ForwardIter1 begin, ForwardIter1 end, ForwardIter2 out_begin, PointMap const& point_map,
                             KnnMap&& knn_map, TransformOp&& op)
{
    using value_type = typename std::iterator_traits<ForwardIter1>::value_type;
    detail::estimate_tangent_planes_impl<ForwardIter1, PointMap, KnnMap, Plane>(
        begin, end, point_map, knn_map, [&](value_type const& v, Plane const& p) { *out_begin++ = op(v, p); });
}

} // namespace algorithm
} // namespace pcp

#endif
