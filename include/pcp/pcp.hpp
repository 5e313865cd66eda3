// drop-in for include/pcp/pcp.hpp: the hot-path subset of the library (containers, geometry, normals).
#ifndef PCP_PCP_HPP
#define PCP_PCP_HPP
#include "pcp/algorithm/algorithm.hpp"
#include "pcp/common/common.hpp"
#include "pcp/io/io.hpp"
#include "pcp/kdtree/kdtree.hpp"
#include "pcp/octree/octree.hpp"
#endif
