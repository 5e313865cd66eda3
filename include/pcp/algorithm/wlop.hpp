// pcp::algorithm::wlop::wlop -- drop-in for include/pcp/algorithm/wlop.hpp (:287-428; params_t :238-247): weighted
// locally optimal projection (Huang et al. 2009) of I resampled points onto the input cloud, k solver iterations with
// radial support h, repulsion mu; uniform = false is plain LOP.  Same signature and meaning.  As in the reference the I
// seed points are a random draw from the input (std::random_device + std::shuffle, :331-343), so two runs differ -- there
// as here; the overload with an explicit `sample` (indices of the seed points) is the reproducible form the tests use.
// The solver itself is one call into libpcpx (pcpx_wlop): per iteration a GPU range tree over the current samples, the
// sample densities, and the median + repulsion update fused into the range walks.
#ifndef PCP_ALGORITHM_WLOP_HPP
#define PCP_ALGORITHM_WLOP_HPP

#include "pcp/gpu/device_index.hpp"
#include "pcp/traits/output_iterator_traits.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <numeric>
#include <random>
#include <stdexcept>
#include <vector>

namespace pcp {
namespace algorithm {
namespace wlop {

struct params_t
{
    std::size_t I = 0u;   ///< Size of the resampled point cloud
    double mu     = 0.45; ///< Repulsion coefficient, in [0, 0.5]
    double h      = 0.;   ///< Radial functions' support
    std::size_t k = 10u;  ///< Number of solver iterations
    bool uniform  = true; ///< Weigh by local densities (WLOP); false = LOP
};

// reproducible form: x starts at the points named by `sample` (params.I is ignored, the sample's size counts)
template <class RandomAccessIter, class OutputIter, class PointMap>
OutputIter wlop(RandomAccessIter begin, RandomAccessIter end, OutputIter out_begin, PointMap point_map, params_t const& params,
                std::vector<std::uint64_t> const& sample)
{
    using output_point_type = typename xstd::output_iterator_traits<OutputIter>::value_type;
    using T                 = typename output_point_type::coordinate_type;
    std::size_t const J     = static_cast<std::size_t>(std::distance(begin, end));
    // the reference asserts I > 0 && J >= I and 0 <= mu <= 0.5 (:309-311)
    if (sample.empty() || sample.size() > J || !(params.mu >= 0. && params.mu <= .5))
        throw std::invalid_argument("wlop: need 0 < I <= J and mu in [0, 0.5]");
    std::vector<float> xyz;
    xyz.reserve(3 * J);
    for (; begin != end; ++begin)
    {
        auto const p = point_map(*begin);
        xyz.push_back(static_cast<float>(p.x()));
        xyz.push_back(static_cast<float>(p.y()));
        xyz.push_back(static_cast<float>(p.z()));
    }
    std::vector<float> out(3u * sample.size());
    gpu::check(pcpx_wlop(xyz.data(), J, sample.data(), sample.size(), params.mu, params.h, params.k, params.uniform ? 1 : 0, 0, out.data()),
               "pcpx_wlop");
    for (std::size_t i = 0; i < sample.size(); ++i)
        *out_begin++ = output_point_type{static_cast<T>(out[3 * i]), static_cast<T>(out[3 * i + 1]), static_cast<T>(out[3 * i + 2])};
    return out_begin;
}

template <class RandomAccessIter, class OutputIter, class PointMap>
OutputIter wlop(RandomAccessIter begin, RandomAccessIter end, OutputIter out_begin, PointMap point_map, params_t const& params)
{
    std::size_t const J = static_cast<std::size_t>(std::distance(begin, end));
    if (params.I == 0u || params.I > J) throw std::invalid_argument("wlop: need 0 < I <= J");
    // the last I entries of a shuffled index sequence, as in the reference
    std::vector<std::uint64_t> js(J);
    std::iota(js.begin(), js.end(), std::uint64_t{0});
    std::random_device rd{};
    std::mt19937 generator{rd()};
    std::shuffle(js.begin(), js.end(), generator);
    std::vector<std::uint64_t> const sample(js.end() - static_cast<std::ptrdiff_t>(params.I), js.end());
    return wlop(begin, end, out_begin, point_map, params, sample);
}

} // namespace wlop
} // namespace algorithm
} // namespace pcp

#endif
