// pcp::octree_parameters_t / pcp::basic_linked_octree_t / pcp::linked_octree_t -- drop-in for
// include/pcp/octree/linked_octree.hpp:40-283 and linked_octree_node.hpp:31-40 of the reference, for the
// data-parallel part of the container: construction from a range, insert, size/empty/clear/voxel_grid,
// nearest_neighbours and range_search.
//
// Same template parameters, constructors, member names, argument meaning and result conventions
// (std::vector<Element> by value, kNN ascending, eps-coincident points excluded, points outside the voxel
// grid silently not inserted).  What differs is underneath: elements and their float32 coordinates are
// kept in flat host arrays and queries run on the GPU index of libpcpx.so (pcp/gpu/device_index.hpp),
// which is (re)built lazily on the first query after a mutation.  node_capacity and max_depth are kept
// for source compatibility; they shaped the reference's pointer tree and have no observable effect on
// query results.  find/erase and the post-order iterator are not provided (dynamic-update API, out of the
// hot path's scope); begin()/end() iterate the inserted elements in insertion order.
//
// Additive, batched entry points (one launch for many queries -- the only way to use the GPU well):
// nearest_neighbours_batch, range_count_batch, range_search_batch, and pcp::gpu::knn_map_t
// (pcp/algorithm/estimate_normals.hpp) for algorithm::estimate_normals.
#ifndef PCP_OCTREE_LINKED_OCTREE_HPP
#define PCP_OCTREE_LINKED_OCTREE_HPP

#include "pcp/common/axis_aligned_bounding_box.hpp"
#include "pcp/common/points/point.hpp"
#include "pcp/common/sphere.hpp"
#include "pcp/gpu/device_index.hpp"
#include "pcp/gpu/host_capture.hpp"

#include <array>
#include <cassert>
#include <cstdint>
#include <functional>
#include <iterator>
#include <limits>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

namespace pcp {

template <class Point>
struct octree_parameters_t
{
    using point_type = Point;
    using aabb_type  = axis_aligned_bounding_box_t<Point>;

    std::uint32_t node_capacity = 32u;
    std::uint8_t max_depth      = 21u;
    aabb_type voxel_grid{};
};

template <class Element, class ParamsType = octree_parameters_t<pcp::point_t>>
class basic_linked_octree_t
{
  public:
    using element_type    = Element;
    using params_type     = ParamsType;
    using aabb_type       = typename ParamsType::aabb_type;
    using aabb_point_type = typename aabb_type::point_type;
    using value_type      = element_type;
    using reference       = value_type&;
    using const_reference = value_type const&;
    using iterator        = typename gpu::element_storage_t<element_type>::iterator;
    using const_iterator  = typename gpu::element_storage_t<element_type>::const_iterator;
    using self_type       = basic_linked_octree_t<element_type, params_type>;

    basic_linked_octree_t(self_type&&) = default;
    self_type& operator=(self_type&&) = default;

    explicit basic_linked_octree_t(params_type const& params) : params_(params) { check_params(); }

    template <class ForwardIter, class PointViewMap>
    explicit basic_linked_octree_t(ForwardIter begin, ForwardIter end, PointViewMap const& point_view,
                                   params_type const& params)
        : params_(params)
    {
        check_params();
        insert(begin, end, point_view);
    }

    // voxel grid = tight bounding box of the range (reference: linked_octree.hpp:103-121)
    template <class ForwardIter, class PointViewMap>
    explicit basic_linked_octree_t(ForwardIter begin, ForwardIter end, PointViewMap const& point_view)
    {
        using T = typename aabb_point_type::coordinate_type;
        T const hi = std::numeric_limits<T>::max(), lo = std::numeric_limits<T>::lowest();
        params_.voxel_grid.min = aabb_point_type{hi, hi, hi};
        params_.voxel_grid.max = aabb_point_type{lo, lo, lo};
        auto const widen = [&point_view](aabb_type& b, ForwardIter first, ForwardIter last) {
            for (ForwardIter it = first; it != last; ++it)
            {
                auto const p = point_view(*it);
                if (p.x() < b.min.x()) b.min.x(p.x());
                if (p.y() < b.min.y()) b.min.y(p.y());
                if (p.z() < b.min.z()) b.min.z(p.z());
                if (p.x() > b.max.x()) b.max.x(p.x());
                if (p.y() > b.max.y()) b.max.y(p.y());
                if (p.z() > b.max.z()) b.max.z(p.z());
            }
        };
        if constexpr (is_random_access<ForwardIter>)
        {
            // the box of a range is the box of its pieces' boxes (min / max by strict comparison: the same box in any order)
            std::size_t const n   = static_cast<std::size_t>(end - begin);
            unsigned const pieces = gpu::capture_threads(n);
            std::vector<aabb_type> part(pieces, params_.voxel_grid);
            gpu::parallel_chunks(n, pieces, [&](std::size_t a, std::size_t b, unsigned c) {
                using diff_t = typename std::iterator_traits<ForwardIter>::difference_type;
                widen(part[c], begin + static_cast<diff_t>(a), begin + static_cast<diff_t>(b));
            });
            for (auto const& pb : part)
            {
                auto& b = params_.voxel_grid;
                if (pb.min.x() < b.min.x()) b.min.x(pb.min.x());
                if (pb.min.y() < b.min.y()) b.min.y(pb.min.y());
                if (pb.min.z() < b.min.z()) b.min.z(pb.min.z());
                if (pb.max.x() > b.max.x()) b.max.x(pb.max.x());
                if (pb.max.y() > b.max.y()) b.max.y(pb.max.y());
                if (pb.max.z() > b.max.z()) b.max.z(pb.max.z());
            }
        }
        else widen(params_.voxel_grid, begin, end);
        insert(begin, end, point_view);
    }

    std::size_t size() const { return elements_.size(); }
    bool empty() const { return elements_.empty(); }
    void clear()
    {
        elements_.clear();
        xyz_.clear();
        dirty_ = true;
    }
    aabb_type const& voxel_grid() const { return params_.voxel_grid; }

    iterator begin() { return elements_.begin(); }
    iterator end() { return elements_.end(); }
    const_iterator cbegin() const { return elements_.cbegin(); }
    const_iterator cend() const { return elements_.cend(); }

    // returns the number of elements inserted; elements whose point lies outside the voxel grid are
    // skipped (linked_octree_node.hpp:174-175)
    template <class ForwardIter, class PointViewMap>
    std::size_t insert(ForwardIter begin, ForwardIter end, PointViewMap const& point_view)
    {
        if constexpr (is_random_access<ForwardIter>)
        {
            std::size_t const n = static_cast<std::size_t>(end - begin);
            if (n >= gpu::parallel_capture_threshold) return insert_many(begin, n, point_view);
        }
        std::size_t inserted = 0;
        for (; begin != end; ++begin) inserted += insert(*begin, point_view) ? 1u : 0u;
        return inserted;
    }
    template <class PointViewMap>
    bool insert(element_type const& e, PointViewMap const& point_view)
    {
        auto const p = point_view(e);
        if (!params_.voxel_grid.contains(p)) return false;
        if (!point_of_)  // remembered for nearest_neighbours_of (the callable is copied, like the elements)
            point_of_ = [point_view](element_type const& el) {
                auto const q = point_view(el);
                return std::array<float, 3>{static_cast<float>(q.x()), static_cast<float>(q.y()), static_cast<float>(q.z())};
            };
        elements_.push_back(e);
        xyz_.push_back(static_cast<float>(p.x()));
        xyz_.push_back(static_cast<float>(p.y()));
        xyz_.push_back(static_cast<float>(p.z()));
        dirty_ = true;
        return true;
    }

    // k nearest neighbours of target, nearest first; points within eps of the target on all three axes
    // are not returned (linked_octree_node.hpp:540).  point_view is accepted for signature parity: the
    // coordinates were captured at insertion.
    template <class TPointView, class PointViewMap>
    std::vector<element_type> nearest_neighbours(TPointView const& target, std::size_t k, PointViewMap const&,
                                                 double eps = 1e-5) const
    {
        if (k == 0 || elements_.empty()) return {};
        float const q[3] = {static_cast<float>(target.x()), static_cast<float>(target.y()), static_cast<float>(target.z())};
        auto const row = index().knn_one(q, static_cast<std::uint32_t>(k), static_cast<float>(eps));
        return gather(row.data(), row.size());
    }

    // k nearest neighbours of one of the container's own elements, located through the point-view map the
    // elements were inserted with (what pcp::gpu::self_knn_map_t calls per element)
    std::vector<element_type> nearest_neighbours_of(element_type const& e, std::size_t k, double eps = 1e-5) const
    {
        if (k == 0 || elements_.empty() || !point_of_) return {};
        auto const q = point_of_(e);
        auto const row = index().knn_one(q.data(), static_cast<std::uint32_t>(k), static_cast<float>(eps));
        return gather(row.data(), row.size());
    }

    // every element whose point satisfies range.contains(point)
    template <class Range, class PointViewMap>
    std::vector<element_type> range_search(Range const& range, PointViewMap const& point_view) const
    {
        if (elements_.empty()) return {};
        std::vector<std::uint64_t> off;
        std::vector<std::uint32_t> idx;
        if constexpr (std::is_same_v<Range, sphere_t<aabb_point_type>>)
        {
            float const c[3] = {static_cast<float>(range.position.x()), static_cast<float>(range.position.y()),
                                static_cast<float>(range.position.z())};
            idx = index().range_sphere_one(c, static_cast<float>(range.radius));
        }
        else if constexpr (std::is_same_v<Range, aabb_type>)
        {
            float const b[6] = {static_cast<float>(range.min.x()), static_cast<float>(range.min.y()),
                                static_cast<float>(range.min.z()), static_cast<float>(range.max.x()),
                                static_cast<float>(range.max.y()), static_cast<float>(range.max.z())};
            index().range_boxes(b, 1, off, idx);
        }
        else
        {
            // user-defined Range: only contains() is known, so test every element on the host
            std::vector<element_type> out;
            for (auto const& e : elements_)
                if (range.contains(point_view(e))) out.push_back(e);
            return out;
        }
        return gather(idx.data(), idx.size());
    }

    // ---- batched additions --------------------------------------------------------------------
    // kNN of many targets in one launch: row q of the result = neighbours of *(begin + q)
    template <class ForwardIter, class TargetPointMap>
    std::vector<std::vector<element_type>> nearest_neighbours_batch(ForwardIter begin, ForwardIter end,
                                                                    TargetPointMap const& target_point, std::size_t k,
                                                                    double eps = 1e-5) const
    {
        std::vector<float> q;
        for (; begin != end; ++begin)
        {
            auto const p = target_point(*begin);
            q.push_back(static_cast<float>(p.x()));
            q.push_back(static_cast<float>(p.y()));
            q.push_back(static_cast<float>(p.z()));
        }
        std::size_t const nq = q.size() / 3;
        std::vector<std::vector<element_type>> rows(nq);
        if (k == 0 || elements_.empty() || nq == 0) return rows;
        auto const r = index().knn(q.data(), nq, static_cast<std::uint32_t>(k), static_cast<float>(eps));
        for (std::size_t i = 0; i < nq; ++i) rows[i] = gather(r.idx.data() + i * k, r.count[i]);
        return rows;
    }
    // number of elements inside the sphere of `radius` around each target (what a density filter needs)
    template <class ForwardIter, class TargetPointMap>
    std::vector<std::uint32_t> range_count_batch(ForwardIter begin, ForwardIter end, TargetPointMap const& target_point,
                                                 float radius) const
    {
        std::vector<float> q;
        for (; begin != end; ++begin)
        {
            auto const p = target_point(*begin);
            q.push_back(static_cast<float>(p.x()));
            q.push_back(static_cast<float>(p.y()));
            q.push_back(static_cast<float>(p.z()));
        }
        if (elements_.empty()) return std::vector<std::uint32_t>(q.size() / 3, 0u);
        return index().range_count(q.data(), q.size() / 3, radius);
    }

    // the device index over the inserted elements (built on demand); index i <-> *(begin() + i)
    gpu::device_index_t const& index() const
    {
        std::lock_guard<std::mutex> lock(*mutex_);  // queries are const and may come from many threads
        if (dirty_ || !index_.valid())
        {
            auto const& g = params_.voxel_grid;
            float const grid[6] = {static_cast<float>(g.min.x()), static_cast<float>(g.min.y()), static_cast<float>(g.min.z()),
                                   static_cast<float>(g.max.x()), static_cast<float>(g.max.y()), static_cast<float>(g.max.z())};
            index_.build(xyz_.data(), elements_.size(), grid);
            dirty_ = false;
        }
        return index_;
    }
    gpu::coord_buffer_t const& coordinates() const { return xyz_; }
    element_type const& element(std::size_t i) const { return elements_[i]; }

  private:
    template <class Iter>
    static constexpr bool is_random_access =
        std::is_base_of_v<std::random_access_iterator_tag, typename std::iterator_traits<Iter>::iterator_category>;

    // insert(begin, begin + n) for a large random-access range: the property map is evaluated and the coordinates are stored
    // by several threads, each on a contiguous piece, while the calling thread copies the elements; the result is what the
    // one-by-one loop gives (elements in range order, those outside the voxel grid skipped).
    template <class RandomIter, class PointViewMap>
    std::size_t insert_many(RandomIter begin, std::size_t n, PointViewMap const& point_view)
    {
        using diff_t = typename std::iterator_traits<RandomIter>::difference_type;
        if (!point_of_)
            point_of_ = [point_view](element_type const& el) {
                auto const q = point_view(el);
                return std::array<float, 3>{static_cast<float>(q.x()), static_cast<float>(q.y()), static_cast<float>(q.z())};
            };
        std::size_t const old = elements_.size();
        xyz_.resize(3 * (old + n));  // (default-initialised: the pieces below are first touched by the threads that fill them)
        constexpr bool in_pieces = gpu::constructible_in_pieces<element_type>;
        if constexpr (in_pieces) elements_.resize(old + n);  // claims the slots; every one is constructed below
        [[maybe_unused]] element_type* const slot = elements_.data() + old;
        unsigned const pieces = gpu::capture_threads(n);
        std::vector<std::vector<std::size_t>> outside(pieces);  // per piece: offsets of the points outside the grid, ascending
        auto const& grid = params_.voxel_grid;
        float* const dst = xyz_.data() + 3 * old;
        try
        {
        gpu::parallel_chunks(
            n, pieces,
            [&](std::size_t a, std::size_t b, unsigned c) {
                RandomIter it = begin + static_cast<diff_t>(a);
                for (std::size_t i = a; i < b; ++i, ++it)
                {
                    if constexpr (in_pieces) ::new (static_cast<void*>(slot + i)) element_type(*it);
                    auto const p = point_view(*it);
                    if (!grid.contains(p)) outside[c].push_back(i);
                    dst[3 * i]     = static_cast<float>(p.x());
                    dst[3 * i + 1] = static_cast<float>(p.y());
                    dst[3 * i + 2] = static_cast<float>(p.z());
                }
            },
            [&] {
                if constexpr (!in_pieces) elements_.insert(elements_.end(), begin, begin + static_cast<diff_t>(n));
            });
        }
        catch (...)  // (a property map or an element's copy threw: the container is what it was before the call)
        {
            if (elements_.size() > old) elements_.erase(elements_.begin() + static_cast<std::ptrdiff_t>(old), elements_.end());
            xyz_.resize(3 * old);
            throw;
        }
        dirty_ = true;
        std::vector<std::size_t> drop;
        for (auto const& o : outside) drop.insert(drop.end(), o.begin(), o.end());
        if (drop.empty()) return n;
        // some points lie outside the grid: close the gaps (elements and coordinates alike), in range order
        std::size_t w = drop[0];
        for (std::size_t d = 0; d < drop.size(); ++d)
        {
            std::size_t const next = d + 1 < drop.size() ? drop[d + 1] : n;
            for (std::size_t i = drop[d] + 1; i < next; ++i, ++w)
            {
                elements_[old + w] = std::move(elements_[old + i]);
                dst[3 * w]         = dst[3 * i];
                dst[3 * w + 1]     = dst[3 * i + 1];
                dst[3 * w + 2]     = dst[3 * i + 2];
            }
        }
        elements_.erase(elements_.begin() + static_cast<std::ptrdiff_t>(old + w), elements_.end());
        xyz_.resize(3 * (old + w));
        return w;
    }

    void check_params() const
    {
        assert(params_.node_capacity > 0u);
        assert(params_.max_depth > 0u);
    }
    std::vector<element_type> gather(std::uint32_t const* idx, std::size_t n) const
    {
        std::vector<element_type> out;
        out.reserve(n);
        for (std::size_t i = 0; i < n; ++i) out.push_back(elements_[idx[i]]);
        return out;
    }

    params_type params_{};
    gpu::element_storage_t<element_type> elements_;
    gpu::coord_buffer_t xyz_;  // point i of elements_ at [3 i, 3 i + 3)
    std::function<std::array<float, 3>(element_type const&)> point_of_;
    mutable gpu::device_index_t index_;
    mutable bool dirty_ = true;
    mutable std::unique_ptr<std::mutex> mutex_ = std::make_unique<std::mutex>();
};

using linked_octree_t = pcp::basic_linked_octree_t<pcp::point_t>;

} // namespace pcp

#endif
