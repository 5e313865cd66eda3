// pcp::algorithm::average_distances_to_neighbors / average_distance_to_neighbors -- drop-in for
// include/pcp/algorithm/average_distance_to_neighbors.hpp (:32-73, :89-113): per element, the mean
// Euclidean distance to the elements its knn_map returns; then the mean of those.  With
// pcp::gpu::self_knn_map the per-element means come from the fused GPU kernel.
#ifndef PCP_ALGORITHM_AVERAGE_DISTANCE_TO_NEIGHBORS_HPP
#define PCP_ALGORITHM_AVERAGE_DISTANCE_TO_NEIGHBORS_HPP

#include "pcp/algorithm/estimate_normals.hpp"

#include <cmath>
#include <iterator>
#include <numeric>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace pcp {
namespace algorithm {

template <class RandomAccessIter, class PointMap, class KnnMap,
          class ScalarType = typename std::invoke_result_t<PointMap, typename std::iterator_traits<RandomAccessIter>::value_type>::coordinate_type>
std::vector<ScalarType> average_distances_to_neighbors(RandomAccessIter begin, RandomAccessIter end, PointMap const& point_map,
                                                       KnnMap const& knn_map)
{
    using knn_type = std::remove_cv_t<std::remove_reference_t<KnnMap>>;
    std::size_t const n = static_cast<std::size_t>(std::distance(begin, end));
    std::vector<ScalarType> mean_distances(n);
    if constexpr (gpu::is_self_knn_map<knn_type>::value)
    {
        if (n != knn_map.tree->size())
            throw std::invalid_argument("self_knn_map: the range must be the container's own element sequence");
        std::vector<float> const m = knn_map.tree->index().mean_knn_distance_self(static_cast<std::uint32_t>(knn_map.k), knn_map.eps, n);
        for (std::size_t i = 0; i < n; ++i) mean_distances[i] = static_cast<ScalarType>(m[i]);
    }
    else
    {
        std::size_t i = 0;
        for (; begin != end; ++begin, ++i)
        {
            auto const neighbours = knn_map(*begin);
            auto const pi         = point_map(*begin);
            ScalarType sum        = static_cast<ScalarType>(0.);
            for (auto const& j : neighbours)
            {
                auto const pj     = point_map(j);
                ScalarType const x = pi.x() - pj.x(), y = pi.y() - pj.y(), z = pi.z() - pj.z();
                sum               = sum + std::sqrt(x * x + y * y + z * z);
            }
            mean_distances[i] = sum / static_cast<ScalarType>(neighbours.size());
        }
    }
    return mean_distances;
}

template <class RandomAccessIter, class PointMap, class KnnMap,
          class ScalarType = typename std::invoke_result_t<PointMap, typename std::iterator_traits<RandomAccessIter>::value_type>::coordinate_type>
ScalarType average_distance_to_neighbors(RandomAccessIter begin, RandomAccessIter end, PointMap const& point_map, KnnMap const& knn_map)
{
    std::vector<ScalarType> const m = average_distances_to_neighbors<RandomAccessIter, PointMap, KnnMap, ScalarType>(begin, end, point_map, knn_map);
    float const sum = std::accumulate(m.begin(), m.end(), static_cast<ScalarType>(0.));
    float const mu  = sum / static_cast<ScalarType>(m.size());
    return mu;
}

} // namespace algorithm
} // namespace pcp

#endif
