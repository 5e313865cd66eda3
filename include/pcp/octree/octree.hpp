// drop-in for include/pcp/octree/octree.hpp
#ifndef PCP_OCTREE_OCTREE_HPP
#define PCP_OCTREE_OCTREE_HPP
#include "pcp/octree/linked_octree.hpp"
#endif
