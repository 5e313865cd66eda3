// pcp::axis_aligned_bounding_box_t / kd_axis_aligned_bounding_box_t / bounding_box / kd_bounding_box --
// drop-in for include/pcp/common/axis_aligned_bounding_box.hpp: inclusive contains (:111-125, :45-75),
// center = (min+max)/2.f (:130), clamp-based nearest_point_from (:138-148, :81-90), bounding boxes from
// +-max with strict comparisons (:214-251, :164-201).
#ifndef PCP_COMMON_AXIS_ALIGNED_BOUNDING_BOX_HPP
#define PCP_COMMON_AXIS_ALIGNED_BOUNDING_BOX_HPP

#include <algorithm>
#include <array>
#include <cstddef>
#include <iterator>
#include <limits>

namespace pcp {

template <class T, std::size_t K>
struct kd_axis_aligned_bounding_box_t
{
    using scalar_type = T;
    using point_type  = std::array<T, K>;
    point_type min{}, max{};

    bool contains(point_type const& p) const
    {
        for (std::size_t i = 0; i < K; ++i)
            if (!(p[i] >= min[i] && p[i] <= max[i])) return false;
        return true;
    }
    point_type nearest_point_from(point_type const& p) const
    {
        point_type q = p;
        for (std::size_t i = 0; i < K; ++i) q[i] = std::clamp(q[i], min[i], max[i]);
        return q;
    }
};

template <class Point>
struct axis_aligned_bounding_box_t
{
    using point_type = Point;
    Point min{0., 0., 0.}, max{0., 0., 0.};

    template <class P>
    bool contains(P const& p) const
    {
        return (p.x() >= min.x() && p.y() >= min.y() && p.z() >= min.z()) &&
               (p.x() <= max.x() && p.y() <= max.y() && p.z() <= max.z());
    }
    Point center() const { return (min + max) / 2.f; }
    template <class P>
    Point nearest_point_from(P const& p) const
    {
        return Point{std::clamp(p.x(), min.x(), max.x()), std::clamp(p.y(), min.y(), max.y()),
                     std::clamp(p.z(), min.z(), max.z())};
    }
};

template <class T, std::size_t K, class CoordinateMap, class ForwardIter>
inline kd_axis_aligned_bounding_box_t<T, K> kd_bounding_box(ForwardIter begin, ForwardIter end, CoordinateMap const& cm)
{
    kd_axis_aligned_bounding_box_t<T, K> b;
    for (std::size_t i = 0; i < K; ++i)
    {
        b.min[i] = std::numeric_limits<T>::max();
        b.max[i] = std::numeric_limits<T>::lowest();
    }
    for (; begin != end; ++begin)
    {
        auto const p = cm(*begin);
        for (std::size_t i = 0; i < K; ++i)
        {
            if (p[i] < b.min[i]) b.min[i] = p[i];
            if (p[i] > b.max[i]) b.max[i] = p[i];
        }
    }
    return b;
}

template <class ForwardIter, class Point, class AABB = axis_aligned_bounding_box_t<Point>>
inline AABB bounding_box(ForwardIter begin, ForwardIter end)
{
    using T = typename Point::coordinate_type;
    T const hi = std::numeric_limits<T>::max(), lo = std::numeric_limits<T>::lowest();
    AABB b;
    b.min = Point{hi, hi, hi};
    b.max = Point{lo, lo, lo};
    for (; begin != end; ++begin)
    {
        auto const& p = *begin;
        if (p.x() < b.min.x()) b.min.x(p.x());
        if (p.y() < b.min.y()) b.min.y(p.y());
        if (p.z() < b.min.z()) b.min.z(p.z());
        if (p.x() > b.max.x()) b.max.x(p.x());
        if (p.y() > b.max.y()) b.max.y(p.y());
        if (p.z() > b.max.z()) b.max.z(p.z());
    }
    return b;
}

} // namespace pcp

#endif
