// drop-in for include/pcp/kdtree/kdtree.hpp
#ifndef PCP_KDTREE_KDTREE_HPP
#define PCP_KDTREE_KDTREE_HPP
#include "pcp/kdtree/linked_kdtree.hpp"
#endif
