// drop-in for include/pcp/algorithm/algorithm.hpp (hot-path subset)
#ifndef PCP_ALGORITHM_ALGORITHM_HPP
#define PCP_ALGORITHM_ALGORITHM_HPP
#include "pcp/algorithm/common.hpp"
#include "pcp/algorithm/average_distance_to_neighbors.hpp"
#include "pcp/algorithm/bilateral_filter.hpp"
#include "pcp/algorithm/estimate_normals.hpp"
#include "pcp/algorithm/estimate_tangent_planes.hpp"
#include "pcp/algorithm/wlop.hpp"
#endif
