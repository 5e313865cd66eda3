// pcp::common::basic_timer_t -- drop-in for include/pcp/common/timer.hpp:22-40, the named-interval stopwatch the
// reference's example programs print their phase times with (examples/filter_point_cloud_noise_by_density.cpp:30-120):
// register_op(name) opens an entry, start() / stop() bracket it, `ops` holds (name, duration) in registration order.
#ifndef PCP_COMMON_TIMER_HPP
#define PCP_COMMON_TIMER_HPP

#include <chrono>
#include <string>
#include <utility>
#include <vector>

namespace pcp {
namespace common {

template <class Clock>
struct stopwatch_t
{
    using time_type     = typename Clock::time_point;
    using duration_type = typename Clock::duration;
    using entry_type    = std::pair<std::string, duration_type>;

    // a new interval; its duration stays zero until stop() closes it
    void register_op(std::string const& op_name) { ops.emplace_back(op_name, duration_type::zero()); }
    void start() { begin = Clock::now(); }
    // closes the most recently registered interval (nothing to close: the reading is taken and dropped)
    void stop()
    {
        end = Clock::now();
        if (!ops.empty()) ops.back().second = end - begin;
    }

    time_type begin{};
    time_type end{};
    std::vector<entry_type> ops;
};

using basic_timer_t = stopwatch_t<std::chrono::high_resolution_clock>;

} // namespace common
} // namespace pcp

#endif
