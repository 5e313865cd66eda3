// drop-in for include/pcp/io/io.hpp: the point-cloud PLY subset (OBJ and mesh PLY are outside the hot path's scope).
#ifndef PCP_IO_IO_HPP
#define PCP_IO_IO_HPP
#include "pcp/io/ply.hpp"
#endif
