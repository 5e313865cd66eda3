// pcp::algorithm::estimate_normals -- drop-in for include/pcp/algorithm/estimate_normals.hpp:50-93
// (execution-policy overload) and :116-164 (sequential overload, supports back_inserter).
//
// For an arbitrary KnnMap callable both overloads do what the reference does: one knn_map call and one
// pcp::estimate_normal per element.  The GPU is used well when knn_map is a pcp::gpu::knn_map_t -- a
// KnnMap bound to one of this tree's containers: then ONE batched kNN launch answers every element of
// [begin, end) and ONE launch computes all normals; `op` is applied on the host in input order.
// If the range is the container's own element sequence, pcp::gpu::self_knn_map_t selects the fused
// kernel (kNN + PCA normal per point, neighbour lists never leave the GPU).
//
// propagate_normal_orientations (:187-302) is a sequential BFS over the kNN graph and is not part of
// the data-parallel hot path.
#ifndef PCP_ALGORITHM_ESTIMATE_NORMALS_HPP
#define PCP_ALGORITHM_ESTIMATE_NORMALS_HPP

#include "pcp/algorithm/common.hpp"
#include "pcp/common/normals/normal.hpp"
#include "pcp/common/normals/normal_estimation.hpp"
#include "pcp/gpu/device_index.hpp"

#include <algorithm>
#include <cstddef>
#include <iterator>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace pcp {
namespace gpu {

// KnnMap over a container of this tree (basic_linked_octree_t / basic_linked_kdtree_t): callable per
// element like any knn_map, and recognised by algorithm::estimate_normals for batching.
// QueryPointMap: element -> something with x(), y(), z() (where to search from).
template <class Tree, class QueryPointMap>
struct knn_map_t
{
    using element_type = typename Tree::element_type;
    Tree const* tree;
    QueryPointMap query_point;
    std::size_t k;
    float eps = 1e-5f;

    std::vector<element_type> operator()(element_type const& e) const
    {
        auto const p = query_point(e);
        float const q[3] = {static_cast<float>(p.x()), static_cast<float>(p.y()), static_cast<float>(p.z())};
        auto const r = tree->index().knn(q, 1, static_cast<std::uint32_t>(k), eps);
        std::vector<element_type> out;
        for (std::uint32_t i = 0; i < r.count[0]; ++i) out.push_back(tree->element(r.idx[i]));
        return out;
    }
};
template <class Tree, class QueryPointMap>
knn_map_t<Tree, QueryPointMap> knn_map(Tree const& tree, QueryPointMap query_point, std::size_t k, float eps = 1e-5f)
{
    return knn_map_t<Tree, QueryPointMap>{&tree, query_point, k, eps};
}

// Same, with the promise that the queried range is exactly the container's elements in insertion order.
template <class Tree>
struct self_knn_map_t
{
    using element_type = typename Tree::element_type;
    Tree const* tree;
    std::size_t k;
    float eps = 1e-5f;
};
template <class Tree>
self_knn_map_t<Tree> self_knn_map(Tree const& tree, std::size_t k, float eps = 1e-5f)
{
    return self_knn_map_t<Tree>{&tree, k, eps};
}

template <class T>
struct is_knn_map : std::false_type
{
};
template <class Tree, class Q>
struct is_knn_map<knn_map_t<Tree, Q>> : std::true_type
{
};
template <class T>
struct is_self_knn_map : std::false_type
{
};
template <class Tree>
struct is_self_knn_map<self_knn_map_t<Tree>> : std::true_type
{
};

} // namespace gpu

namespace algorithm {
namespace detail {

template <class Normal>
Normal make_normal(float const* n)
{
    using T = typename Normal::component_type;
    return Normal{static_cast<T>(n[0]), static_cast<T>(n[1]), static_cast<T>(n[2])};
}

// shared body: `emit(value, normal)` receives the results in input order
template <class ForwardIter1, class PointViewMap, class KnnMap, class Normal, class Emit>
void estimate_normals_impl(ForwardIter1 begin, ForwardIter1 end, PointViewMap const& point_map, KnnMap const& knn, Emit&& emit)
{
    using knn_type = std::remove_cv_t<std::remove_reference_t<KnnMap>>;
    if constexpr (gpu::is_self_knn_map<knn_type>::value)
    {
        std::size_t const n = static_cast<std::size_t>(std::distance(begin, end));
        if (n != knn.tree->size())
            throw std::invalid_argument("self_knn_map: the range must be the container's own element sequence");
        std::vector<float> const nrm = knn.tree->index().normals_self(static_cast<std::uint32_t>(knn.k), knn.eps, n);
        std::size_t i = 0;
        for (; begin != end; ++begin, ++i) emit(*begin, make_normal<Normal>(nrm.data() + 3 * i));
    }
    else if constexpr (gpu::is_knn_map<knn_type>::value)
    {
        std::vector<float> q;
        for (ForwardIter1 it = begin; it != end; ++it)
        {
            auto const p = knn.query_point(*it);
            q.push_back(static_cast<float>(p.x()));
            q.push_back(static_cast<float>(p.y()));
            q.push_back(static_cast<float>(p.z()));
        }
        auto const& ix  = knn.tree->index();
        auto const rows = ix.knn(q.data(), q.size() / 3, static_cast<std::uint32_t>(knn.k), knn.eps);
        std::vector<float> const nrm = ix.normals_from_knn(rows);
        std::size_t i = 0;
        for (; begin != end; ++begin, ++i) emit(*begin, make_normal<Normal>(nrm.data() + 3 * i));
    }
    else
    {
        for (; begin != end; ++begin)
        {
            auto const neighbours = knn(*begin);
            using iterator_type   = decltype(std::begin(neighbours));
            emit(*begin, pcp::estimate_normal<iterator_type, PointViewMap, Normal>(std::begin(neighbours), std::end(neighbours), point_map));
        }
    }
}

} // namespace detail

// execution-policy overload (the policy is accepted for signature parity: the parallelism is the GPU's)
template <class ExecutionPolicy, class ForwardIter1, class ForwardIter2, class PointViewMap, class KnnMap, class TransformOp,
          class Normal = pcp::normal_t,
          class = std::enable_if_t<!std::is_same_v<std::decay_t<TransformOp>, void> &&
                                   std::is_invocable_v<TransformOp, typename std::iterator_traits<ForwardIter1>::value_type, Normal>>>
void estimate_normals(ExecutionPolicy&&, ForwardIter1 begin, ForwardIter1 end, ForwardIter2 out_begin,
                      PointViewMap const& point_map, KnnMap&& knn_map, TransformOp&& op)
{
    using value_type = typename std::iterator_traits<ForwardIter1>::value_type;
    detail::estimate_normals_impl<ForwardIter1, PointViewMap, KnnMap, Normal>(
        begin, end, point_map, knn_map, [&](value_type const& v, Normal const& n) { *out_begin++ = op(v, n); });
}

// sequential overload; out_begin may be a std::back_inserter
template <class ForwardIter1, class ForwardIter2, class PointViewMap, class KnnMap, class TransformOp,
          class Normal = pcp::normal_t,
          class = std::enable_if_t<std::is_invocable_v<TransformOp, typename std::iterator_traits<ForwardIter1>::value_type, Normal>>>
void estimate_normals(ForwardIter1 begin, ForwardIter1 end, ForwardIter2 out_begin, PointViewMap const& point_map,
                      KnnMap&& knn_map, TransformOp&& op)
{
    using value_type = typename std::iterator_traits<ForwardIter1>::value_type;
    detail::estimate_normals_impl<ForwardIter1, PointViewMap, KnnMap, Normal>(
        begin, end, point_map, knn_map, [&](value_type const& v, Normal const& n) { *out_begin++ = op(v, n); });
}

} // namespace algorithm
} // namespace pcp

#endif
