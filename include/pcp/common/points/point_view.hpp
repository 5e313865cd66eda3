// pcp::basic_point_view_t / pcp::point_view_t -- drop-in for
// include/pcp/common/points/point_view.hpp:23-80: a non-owning view (one pointer) over a point.
#ifndef PCP_COMMON_POINTS_POINT_VIEW_HPP
#define PCP_COMMON_POINTS_POINT_VIEW_HPP

#include "pcp/common/points/point.hpp"

namespace pcp {

template <class Point>
class basic_point_view_t
{
  public:
    using point_type      = Point;
    using self_type       = basic_point_view_t<Point>;
    using component_type  = typename Point::component_type;
    using coordinate_type = typename Point::coordinate_type;

    basic_point_view_t() noexcept = default;
    explicit basic_point_view_t(point_type* p) noexcept : p_(p) {}

    coordinate_type const& x() const { return p_->x(); }
    coordinate_type const& y() const { return p_->y(); }
    coordinate_type const& z() const { return p_->z(); }
    void x(coordinate_type v) { p_->x(v); }
    void y(coordinate_type v) { p_->y(v); }
    void z(coordinate_type v) { p_->z(v); }

    void point(point_type* p) noexcept { p_ = p; }
    point_type const* point() const noexcept { return p_; }
    point_type* point() noexcept { return p_; }

  private:
    point_type* p_ = nullptr;
};

using point_view_t = basic_point_view_t<point_t>;

} // namespace pcp

#endif
