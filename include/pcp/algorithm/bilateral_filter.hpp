// pcp::algorithm::bilateral_filter_points / bilateral_filter_normals -- drop-in for
// include/pcp/algorithm/bilateral_filter.hpp (:303-428 and :460-574; params_t :29-34).  Same signatures, same
// meaning: K rounds of the 3d bilateral filter of Jones, Durand, Zwicker (2004) over the range of radius 2 * sigmaf
// around every point -- on the points (a new range tree over the moved points every round) or on the normals (the
// Jacobian of the filter applied to the normal).  The reference materialises every point's range from a kd-tree and
// loops over it under std::execution::par; here the whole K-round loop is one call into libpcpx (pcpx_bilateral_filter_*:
// the per-neighbour arithmetic fused into the GPU range walk, nothing but the result crossing PCIe).
// Results agree with the reference to float rounding, not bit for bit: the order in which a range is summed is the
// tree's, which the reference's interface leaves open.
#ifndef PCP_ALGORITHM_BILATERAL_FILTER_HPP
#define PCP_ALGORITHM_BILATERAL_FILTER_HPP

#include "pcp/gpu/device_index.hpp"
#include "pcp/traits/output_iterator_traits.hpp"

#include <cstddef>
#include <iterator>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace pcp {
namespace algorithm {
namespace bilateral {

struct params_t
{
    double sigmaf = 1.;  ///< Standard deviation of the spatial weight f (the range is 2 * sigmaf wide)
    double sigmag = 0.1; ///< Standard deviation of the influence weight g
    std::size_t K = 1u;  ///< Number of iterations
};

namespace detail {

// flattens the element range through its point / normal maps into the n x 3 float arrays the C ABI takes
template <class RandomAccessIter, class PointMap, class NormalMap>
void flatten(RandomAccessIter begin, RandomAccessIter end, PointMap const& point_map, NormalMap const& normal_map,
             std::vector<float>& xyz, std::vector<float>& nrm)
{
    std::size_t const n = static_cast<std::size_t>(std::distance(begin, end));
    xyz.reserve(3 * n);
    nrm.reserve(3 * n);
    for (; begin != end; ++begin)
    {
        auto const p = point_map(*begin);
        auto const v = normal_map(*begin);
        xyz.push_back(static_cast<float>(p.x()));
        xyz.push_back(static_cast<float>(p.y()));
        xyz.push_back(static_cast<float>(p.z()));
        nrm.push_back(static_cast<float>(v.nx()));
        nrm.push_back(static_cast<float>(v.ny()));
        nrm.push_back(static_cast<float>(v.nz()));
    }
}

inline void check_params(params_t const& params, std::size_t n)
{
    // the reference asserts these (:327-330)
    if (params.K == 0u || n == 0u || !(params.sigmaf > 0.) || !(params.sigmag > 0.))
        throw std::invalid_argument("bilateral filter: K, the range and both sigmas must be positive");
}

} // namespace detail
} // namespace bilateral

template <class RandomAccessIter, class OutputIter, class PointMap, class NormalMap>
OutputIter bilateral_filter_points(RandomAccessIter begin, RandomAccessIter end, OutputIter out_begin, PointMap const& point_map,
                                   NormalMap const& normal_map, bilateral::params_t const& params)
{
    using output_point_type = typename xstd::output_iterator_traits<OutputIter>::value_type;
    using T                 = typename output_point_type::coordinate_type;
    std::vector<float> xyz, nrm;
    bilateral::detail::flatten(begin, end, point_map, normal_map, xyz, nrm);
    std::size_t const n = xyz.size() / 3u;
    bilateral::detail::check_params(params, n);
    std::vector<float> out(3u * n);
    gpu::check(pcpx_bilateral_filter_points(xyz.data(), nrm.data(), n, params.sigmaf, params.sigmag, params.K, 0, out.data()),
               "pcpx_bilateral_filter_points");
    for (std::size_t i = 0; i < n; ++i)
        *out_begin++ = output_point_type{static_cast<T>(out[3 * i]), static_cast<T>(out[3 * i + 1]), static_cast<T>(out[3 * i + 2])};
    return out_begin;
}

template <class RandomAccessIter, class OutputIter, class PointMap, class NormalMap>
OutputIter bilateral_filter_normals(RandomAccessIter begin, RandomAccessIter end, OutputIter out_begin, PointMap const& point_map,
                                    NormalMap const& normal_map, bilateral::params_t const& params)
{
    using output_normal_type = typename xstd::output_iterator_traits<OutputIter>::value_type;
    using T                  = typename output_normal_type::component_type;
    std::vector<float> xyz, nrm;
    bilateral::detail::flatten(begin, end, point_map, normal_map, xyz, nrm);
    std::size_t const n = xyz.size() / 3u;
    bilateral::detail::check_params(params, n);
    std::vector<float> out(3u * n);
    gpu::check(pcpx_bilateral_filter_normals(xyz.data(), nrm.data(), n, params.sigmaf, params.sigmag, params.K, 0, out.data()),
               "pcpx_bilateral_filter_normals");
    for (std::size_t i = 0; i < n; ++i)
        *out_begin++ = output_normal_type{static_cast<T>(out[3 * i]), static_cast<T>(out[3 * i + 1]), static_cast<T>(out[3 * i + 2])};
    return out_begin;
}

} // namespace algorithm
} // namespace pcp

#endif
