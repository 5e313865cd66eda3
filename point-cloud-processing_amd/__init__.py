"""pcp-mi355x: MI355X-native kNN / radius search / PCA normals behind the `pcp` API.

The compute path is libpcpx.so (hand-written HIP for gfx950, C ABI in include/pcpx.h); importing the
package does not load it, using any compute entry point does and fails loudly if it is missing.
"""
from . import ply, synthetic  # noqa: F401
from .filters import bilateral_filter_normals, bilateral_filter_points, wlop  # noqa: F401
from .index import (Index, KdTreeK, LinkedKdTree, LinkedOctree, PcpxError, bounding_box, device_count,  # noqa: F401
                    estimate_normal, estimate_normals, propagate_normal_orientations, propagate_normal_orientations_dev, shard_range, shard_cuts_by_cost)

__all__ = ["Index", "LinkedOctree", "LinkedKdTree", "KdTreeK", "PcpxError", "bounding_box", "device_count", "estimate_normal",
           "estimate_normals", "propagate_normal_orientations", "propagate_normal_orientations_dev", "shard_range", "shard_cuts_by_cost", "ply", "synthetic",
           "bilateral_filter_points", "bilateral_filter_normals", "wlop"]
