// pcp::basic_point_view_vertex_t / pcp::vertex_t -- drop-in for
// include/pcp/common/points/vertex.hpp:29-117: a point view that also carries a 64-bit identifier
// (an Element type the containers are commonly instantiated with).
#ifndef PCP_COMMON_POINTS_VERTEX_HPP
#define PCP_COMMON_POINTS_VERTEX_HPP

#include "pcp/common/points/point_view.hpp"

#include <cstdint>

namespace pcp {

template <class Point>
class basic_point_view_vertex_t : public basic_point_view_t<Point>
{
  public:
    using id_type     = std::uint64_t;
    using parent_type = basic_point_view_t<Point>;
    using point_type  = Point;

    basic_point_view_vertex_t() = default;
    explicit basic_point_view_vertex_t(Point* p) : parent_type(p) {}
    explicit basic_point_view_vertex_t(id_type id) : id_(id) {}
    basic_point_view_vertex_t(Point* p, id_type id) : parent_type(p), id_(id) {}

    id_type id() const { return id_; }
    void id(id_type v) { id_ = v; }
    bool operator==(basic_point_view_vertex_t const& o) const { return id_ == o.id_; }
    bool operator!=(basic_point_view_vertex_t const& o) const { return id_ != o.id_; }

  private:
    id_type id_ = 0u;
};

using vertex_t = basic_point_view_vertex_t<point_t>;

} // namespace pcp

#endif
