// drop-in for include/pcp/common/common.hpp (hot-path subset)
#ifndef PCP_COMMON_COMMON_HPP
#define PCP_COMMON_COMMON_HPP
#include "pcp/common/axis_aligned_bounding_box.hpp"
#include "pcp/common/intersections.hpp"
#include "pcp/common/norm.hpp"
#include "pcp/common/normals/normal.hpp"
#include "pcp/common/normals/normal_estimation.hpp"
#include "pcp/common/plane3d.hpp"
#include "pcp/common/points/point.hpp"
#include "pcp/common/points/point_view.hpp"
#include "pcp/common/points/vertex.hpp"
#include "pcp/common/sphere.hpp"
#include "pcp/common/vector3d.hpp"
#include "pcp/common/vector3d_queries.hpp"
#endif
