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
// propagate_normal_orientations (:187-302) keeps the reference's signature and result: a breadth-first
// propagation over the directed kNN graph.  The visit order decides which parent orients a vertex, so the
// search runs on the host like the reference's for arbitrary callables; with a pcp::gpu::knn_map_t all
// neighbour rows come from ONE batched launch instead of one tree query per vertex, and with a
// self_knn_map_t (range = the container's own sequence, ids = positions) the whole pass runs on the GPU as
// a level-synchronous search that reproduces the sequential visit order (csrc/pcpx_orient.hip).
#ifndef PCP_ALGORITHM_ESTIMATE_NORMALS_HPP
#define PCP_ALGORITHM_ESTIMATE_NORMALS_HPP

#include "pcp/algorithm/common.hpp"
#include "pcp/common/normals/normal.hpp"
#include "pcp/common/normals/normal_estimation.hpp"
#include "pcp/common/norm.hpp"
#include "pcp/common/vector3d_queries.hpp"
#include "pcp/gpu/device_index.hpp"

#include <algorithm>
#include <cmath>
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
        auto const row = tree->index().knn_one(q, static_cast<std::uint32_t>(k), eps);
        std::vector<element_type> out;
        out.reserve(row.size());
        for (std::uint32_t i : row) out.push_back(tree->element(i));
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

    // per element it is an ordinary KnnMap (reference traits/knn_map.hpp:22-45); algorithm::estimate_normals never
    // calls this: it recognises the type and runs the whole range as one fused launch
    std::vector<element_type> operator()(element_type const& e) const { return tree->nearest_neighbours_of(e, k, eps); }
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
        // an arbitrary (user) knn_map: it is called per element, in order, like the reference's sequential overload; the
        // neighbourhoods' coordinates are collected (CSR) and their PCA normals computed by ONE launch per block of
        // elements (pcpx_estimate_normals_batch) instead of one pcp::estimate_normal round trip each
        constexpr std::size_t block = std::size_t(1) << 16;
        std::vector<float> xyz;
        std::vector<std::uint64_t> off;
        std::vector<float> nrm;
        while (begin != end)
        {
            ForwardIter1 const first = begin;
            xyz.clear();
            off.assign(1, 0);
            for (; begin != end && off.size() <= block; ++begin)
            {
                auto const neighbours = knn(*begin);
                for (auto const& nb : neighbours)
                {
                    auto const p = point_map(nb);
                    xyz.push_back(static_cast<float>(p.x()));
                    xyz.push_back(static_cast<float>(p.y()));
                    xyz.push_back(static_cast<float>(p.z()));
                }
                off.push_back(xyz.size() / 3);
            }
            std::size_t const rows = off.size() - 1;
            nrm.assign(rows * 3, 0.f);
            gpu::check(pcpx_estimate_normals_batch(xyz.data(), off.data(), rows, gpu::default_device().load(), nrm.data()), "pcpx_estimate_normals_batch");
            std::size_t i = 0;
            for (ForwardIter1 it = first; it != begin; ++it, ++i) emit(*it, make_normal<Normal>(nrm.data() + 3 * i));
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

// Orients the normals of [begin, end) consistently (include/pcp/algorithm/estimate_normals.hpp:187-302):
// directed kNN graph (vertex -> its neighbours, in neighbour order; the sequence must be sorted by
// index_map, ids 0..N-1, as the reference requires: knn_adjacency_list.hpp:100-103), root = first element of
// largest z whose normal becomes (0,0,1), then a breadth-first search (graph/search.hpp:36-85) in which a
// vertex reached for the first time gets its normal flipped -- op(v, -normal_map(v)) -- iff its inner
// product with the (already oriented) normal of the vertex it was reached from is negative and not within
// 1e-5 of zero.
template <class ForwardIter1, class IndexMap, class KnnMap, class PointViewMap, class NormalMap, class TransformOp>
void propagate_normal_orientations(ForwardIter1 begin, ForwardIter1 end, IndexMap const& index_map, KnnMap&& knn_map,
                                   PointViewMap&& point_map, NormalMap& normal_map, TransformOp&& op)
{
    using element_type = typename std::iterator_traits<ForwardIter1>::value_type;
    using normal_type  = std::remove_cv_t<std::remove_reference_t<std::invoke_result_t<NormalMap, element_type>>>;
    using scalar_type  = typename normal_type::component_type;
    using knn_type     = std::remove_cv_t<std::remove_reference_t<KnnMap>>;
    static_assert(std::is_invocable_v<TransformOp, element_type, normal_type>, "op must be callable as op(element, normal)");

    std::vector<element_type> const vertices(begin, end);
    std::size_t const n = vertices.size();
    if (n == 0) return;

    // ---- whole pass on the GPU when the range is the container's own sequence and ids are positions ----
    if constexpr (gpu::is_self_knn_map<knn_type>::value)
    {
        if (n != knn_map.tree->size())
            throw std::invalid_argument("self_knn_map: the range must be the container's own element sequence");
        bool ids_are_positions = true;
        for (std::size_t i = 0; i < n && ids_are_positions; ++i) ids_are_positions = static_cast<std::size_t>(index_map(vertices[i])) == i;
        // the device search compares z of the indexed coordinates; point_map must describe the same points
        if (ids_are_positions)
        {
            std::vector<float> nrm(3 * n);
            for (std::size_t i = 0; i < n; ++i)
            {
                auto const nn = normal_map(vertices[i]);
                nrm[3 * i] = static_cast<float>(nn.x());
                nrm[3 * i + 1] = static_cast<float>(nn.y());
                nrm[3 * i + 2] = static_cast<float>(nn.z());
            }
            std::vector<float> const before = nrm;
            knn_map.tree->index().orient_normals_self(static_cast<std::uint32_t>(knn_map.k), knn_map.eps, nrm);
            for (std::size_t i = 0; i < n; ++i)  // report what the search changed (the root's (0,0,1) and every flip)
                if (nrm[3 * i] != before[3 * i] || nrm[3 * i + 1] != before[3 * i + 1] || nrm[3 * i + 2] != before[3 * i + 2] ||
                    std::signbit(nrm[3 * i]) != std::signbit(before[3 * i]) || std::signbit(nrm[3 * i + 1]) != std::signbit(before[3 * i + 1]) ||
                    std::signbit(nrm[3 * i + 2]) != std::signbit(before[3 * i + 2]))
                    op(vertices[i], normal_type{static_cast<scalar_type>(nrm[3 * i]), static_cast<scalar_type>(nrm[3 * i + 1]),
                                                static_cast<scalar_type>(nrm[3 * i + 2])});
            return;
        }
    }

    // ---- the graph: out-edges of vertex i = ids of its neighbours, CSR ----
    std::vector<std::size_t> first(n + 1, 0);
    std::vector<std::size_t> target;
    if constexpr (gpu::is_self_knn_map<knn_type>::value)
    {
        auto const rows = knn_map.tree->index().knn_self(static_cast<std::uint32_t>(knn_map.k), knn_map.eps, n);
        target.reserve(n * knn_map.k);
        for (std::size_t i = 0; i < n; ++i)
        {
            std::uint32_t const* row = rows.row(i);
            for (std::uint32_t j = 0; j < rows.size_of_row(i); ++j)
                target.push_back(static_cast<std::size_t>(index_map(knn_map.tree->element(row[j]))));
            first[i + 1] = target.size();
        }
    }
    else if constexpr (gpu::is_knn_map<knn_type>::value)
    {
        std::vector<float> q;
        q.reserve(3 * n);
        for (auto const& v : vertices)
        {
            auto const p = knn_map.query_point(v);
            q.push_back(static_cast<float>(p.x()));
            q.push_back(static_cast<float>(p.y()));
            q.push_back(static_cast<float>(p.z()));
        }
        auto const rows = knn_map.tree->index().knn(q.data(), n, static_cast<std::uint32_t>(knn_map.k), knn_map.eps);
        target.reserve(n * knn_map.k);
        for (std::size_t i = 0; i < n; ++i)
        {
            for (std::uint32_t j = 0; j < rows.count[i]; ++j)
                target.push_back(static_cast<std::size_t>(index_map(knn_map.tree->element(rows.idx[i * knn_map.k + j]))));
            first[i + 1] = target.size();
        }
    }
    else
    {
        for (std::size_t i = 0; i < n; ++i)
        {
            auto const neighbours = knn_map(vertices[i]);
            for (auto const& nb : neighbours) target.push_back(static_cast<std::size_t>(index_map(nb)));
            first[i + 1] = target.size();
        }
    }

    // ---- root: the first element of largest z ----
    std::size_t root = 0;
    for (std::size_t i = 1; i < n; ++i)
        if (point_map(vertices[root]).z() < point_map(vertices[i]).z()) root = i;
    op(vertices[root], normal_type{static_cast<scalar_type>(0.0), static_cast<scalar_type>(0.0), static_cast<scalar_type>(1.0)});

    // ---- breadth-first propagation ----
    std::vector<bool> visited(n, false);
    std::vector<std::size_t> order;
    order.reserve(n + 1);
    order.push_back(root);
    scalar_type const zero = static_cast<scalar_type>(0.0);
    for (std::size_t head = 0; head < order.size(); ++head)
    {
        std::size_t const u = order[head];
        for (std::size_t e = first[u]; e < first[u + 1]; ++e)
        {
            std::size_t const v = target[e];
            if (visited[v]) continue;
            auto const n1   = normal_map(vertices[u]);
            auto const n2   = normal_map(vertices[v]);
            auto const prod = common::inner_product(n1, n2);
            if (prod < zero && !common::floating_point_equals(prod, zero)) op(vertices[v], -n2);
            visited[v] = true;
            order.push_back(v);
        }
        visited[u] = true;
    }
}

} // namespace algorithm
} // namespace pcp

#endif
