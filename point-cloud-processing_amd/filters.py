"""Host-side mirror of the reference's two sphere-range consumers, over the C ABI (libpcpx.so):

  bilateral_filter_points / bilateral_filter_normals   include/pcp/algorithm/bilateral_filter.hpp:303-428, :460-574
  wlop                                                 include/pcp/algorithm/wlop.hpp:287-428

Same argument meaning as the reference's params_t structs (sigmaf, sigmag, K; I, mu, h, k, uniform).  No CPU path: without
the HIP library these raise.
"""
import ctypes as C

import numpy as np

from . import _capi
from .index import _f32, _vp, check


def bilateral_filter_points(points, normals, sigmaf=1.0, sigmag=0.1, K=1, device=0):
    """K rounds of the bilateral filter on the points (the normals stay); returns the n x 3 filtered points."""
    pts, nrm = _f32(points, 3), _f32(normals, 3)
    if pts.shape != nrm.shape:
        raise ValueError("points and normals must both be n x 3")
    out = np.empty_like(pts)
    check(_capi.load().pcpx_bilateral_filter_points(_vp(pts), _vp(nrm), len(pts), C.c_double(sigmaf), C.c_double(sigmag), int(K), device, _vp(out)))
    return out


def bilateral_filter_normals(points, normals, sigmaf=1.0, sigmag=0.1, K=1, device=0):
    """K rounds of the bilateral normal improvement (the points stay); returns the n x 3 normals."""
    pts, nrm = _f32(points, 3), _f32(normals, 3)
    if pts.shape != nrm.shape:
        raise ValueError("points and normals must both be n x 3")
    out = np.empty_like(nrm)
    check(_capi.load().pcpx_bilateral_filter_normals(_vp(pts), _vp(nrm), len(pts), C.c_double(sigmaf), C.c_double(sigmag), int(K), device, _vp(out)))
    return out


def wlop(points, I=None, mu=0.45, h=0.0, k=10, uniform=True, sample=None, seed=None, device=0):
    """WLOP resampling to I points.  The reference draws its seed points with std::random_device (wlop.hpp:331-343); here
    `sample` names them (indices into points), or I of them are drawn with numpy's default_rng(seed)."""
    pts = _f32(points, 3)
    if sample is None:
        if I is None:
            raise ValueError("give I or sample")
        if not 0 < int(I) <= len(pts):  # (the reference asserts J >= I, wlop.hpp:300)
            raise ValueError("I must be between 1 and the number of points")
        sample = np.random.default_rng(seed).permutation(len(pts))[len(pts) - int(I):]
    sample = np.ascontiguousarray(sample, dtype=np.uint64)
    out = np.empty((len(sample), 3), np.float32)
    check(_capi.load().pcpx_wlop(_vp(pts), len(pts), _vp(sample), len(sample), C.c_double(mu), C.c_double(h), int(k), int(bool(uniform)),
                                 device, _vp(out)))
    return out
