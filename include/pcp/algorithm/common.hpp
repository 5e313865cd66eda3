// pcp::algorithm::default_normal_transform -- drop-in for include/pcp/algorithm/common.hpp:31-34.
#ifndef PCP_ALGORITHM_COMMON_HPP
#define PCP_ALGORITHM_COMMON_HPP

namespace pcp {
namespace algorithm {

template <class Input, class Normal>
inline auto const default_normal_transform = [](Input const&, Normal const& n) { return n; };

} // namespace algorithm
} // namespace pcp

#endif
