// pcp::kdtree::construction_params_t / pcp::basic_linked_kdtree_t -- drop-in for
// include/pcp/kdtree/linked_kdtree.hpp:24-33 and :64-560 of the reference, for construction,
// nearest_neighbours (coordinates and element overloads), range_search, aabb(), size/empty/clear and
// begin()/end().
//
// Same template parameters <Element, K, CoordinateMap>, constructor, member names and result conventions
// (ascending kNN, eps-coincident points skipped per coordinate, range results unordered and including the
// query point).  K = 1, 2 or 3 goes through the device index (three-dimensional: missing axes are carried as 0, which changes
// no distance, no eps-box test and no box containment); 4 <= K <= 16 is answered by exhaustive search on the GPU (pcpx_kd_*,
// include/pcpx.h: same rows; nearest_neighbours and range_search with the tree's own box type).  The
// construction parameters are accepted for source compatibility: they shaped the reference's median-split
// tree (depth, leaf size), not the query results.  Storage keeps the elements in input order (the
// reference permutes its storage with nth_element; that order was never specified).
#ifndef PCP_KDTREE_LINKED_KDTREE_HPP
#define PCP_KDTREE_LINKED_KDTREE_HPP

#include "pcp/common/axis_aligned_bounding_box.hpp"
#include "pcp/common/sphere.hpp"
#include "pcp/gpu/device_index.hpp"
#include "pcp/gpu/host_capture.hpp"

#include <array>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <limits>
#include <new>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

namespace pcp {

namespace kdtree {
enum class construction_t { nth_element, presort };

struct construction_params_t
{
    std::size_t max_depth                           = 12u;
    construction_t construction                     = construction_t::nth_element;
    std::size_t min_element_count_for_parallel_exec = 32'768;
    bool compute_max_depth                          = false;
    std::size_t max_elements_per_leaf               = 64u;
};
} // namespace kdtree

template <class Element, std::size_t K, class CoordinateMap>
class basic_linked_kdtree_t
{
    static_assert(K >= 1 && K <= PCPX_KD_MAX_DIMS, "1 ... 16 coordinates per point");
    static constexpr bool wide = K > 3;  // (beyond the device index's three coordinates: pcpx_kd_*)
    // coordinate a of a K-dimensional point as the device sees it (0 beyond K)
    template <class C>
    static float axis(C const& c, std::size_t a)
    {
        return a < K ? static_cast<float>(c[a]) : 0.f;
    }

  public:
    using self_type        = basic_linked_kdtree_t;
    using element_type     = Element;
    using coordinates_type = std::invoke_result_t<CoordinateMap, Element>;
    using coordinate_type  = typename coordinates_type::value_type;
    using aabb_type        = kd_axis_aligned_bounding_box_t<coordinate_type, K>;
    using iterator         = typename gpu::element_storage_t<element_type>::iterator;
    using const_iterator   = typename gpu::element_storage_t<element_type>::const_iterator;
    using value_type       = element_type;
    using reference        = value_type&;
    using const_reference  = value_type const&;
    using size_type        = typename gpu::element_storage_t<element_type>::size_type;

    template <class ForwardIter>
    basic_linked_kdtree_t(ForwardIter begin, ForwardIter end, CoordinateMap coordinate_map = CoordinateMap{},
                          kdtree::construction_params_t params = kdtree::construction_params_t{})
        : coordinate_map_(coordinate_map), params_(params)
    {
        // One walk gives the stored copies of the elements, the device's coordinates and the bounding box (kd_bounding_box's
        // box: +-max to start with, strict comparisons -- the box of the pieces' boxes is the same box), on several threads
        // when the range is large (pcp/gpu/host_capture.hpp).
        constexpr bool random_access =
            std::is_base_of_v<std::random_access_iterator_tag, typename std::iterator_traits<ForwardIter>::iterator_category>;
        bool copied = false;
        if constexpr (random_access && gpu::constructible_in_pieces<element_type>)
        {
            std::size_t const count = static_cast<std::size_t>(end - begin);
            if (count >= gpu::parallel_capture_threshold)
            {
                storage_.resize(count);  // claims the slots; the pieces below construct every one
                using diff_t = typename std::iterator_traits<ForwardIter>::difference_type;
                gpu::parallel_chunks(count, gpu::capture_threads(count), [&](std::size_t first, std::size_t last, unsigned) {
                    ForwardIter it = begin + static_cast<diff_t>(first);
                    for (std::size_t i = first; i < last; ++i, ++it) ::new (static_cast<void*>(storage_.data() + i)) element_type(*it);
                });
                copied = true;
            }
        }
        if (!copied) storage_.assign(begin, end);
        std::size_t const n = storage_.size();
        xyz_.resize((wide ? K : 3) * n);  // (K > 3: rows of K coordinates)
        unsigned const pieces = n >= gpu::parallel_capture_threshold ? gpu::capture_threads(n) : 1u;
        aabb_type none;
        for (std::size_t a = 0; a < K; ++a)
        {
            none.min[a] = std::numeric_limits<coordinate_type>::max();
            none.max[a] = std::numeric_limits<coordinate_type>::lowest();
        }
        std::vector<aabb_type> part(pieces, none);
        gpu::parallel_chunks(n, pieces, [&](std::size_t first, std::size_t last, unsigned piece) {
            aabb_type b = none;
            for (std::size_t i = first; i < last; ++i)
            {
                auto const c   = coordinate_map_(storage_[i]);
                if constexpr (wide)
                {
                    for (std::size_t a = 0; a < K; ++a) xyz_[K * i + a] = static_cast<float>(c[a]);
                }
                else
                {
                    xyz_[3 * i]     = axis(c, 0);
                    xyz_[3 * i + 1] = axis(c, 1);
                    xyz_[3 * i + 2] = axis(c, 2);
                }
                for (std::size_t a = 0; a < K; ++a)
                {
                    if (c[a] < b.min[a]) b.min[a] = c[a];
                    if (c[a] > b.max[a]) b.max[a] = c[a];
                }
            }
            part[piece] = b;
        });
        aabb_ = none;
        for (auto const& b : part)
            for (std::size_t a = 0; a < K; ++a)
            {
                if (b.min[a] < aabb_.min[a]) aabb_.min[a] = b.min[a];
                if (b.max[a] > aabb_.max[a]) aabb_.max[a] = b.max[a];
            }
    }
    basic_linked_kdtree_t(self_type&&) = default;
    self_type& operator=(self_type&&) = default;

    bool empty() const { return storage_.empty(); }
    std::size_t size() const { return storage_.size(); }
    void clear()
    {
        storage_.clear();
        xyz_.clear();
        index_.reset();
        wide_index_.reset();
    }
    iterator begin() { return storage_.begin(); }
    iterator end() { return storage_.end(); }
    const_iterator cbegin() const { return storage_.cbegin(); }
    const_iterator cend() const { return storage_.cend(); }
    aabb_type const& aabb() const { return aabb_; }

    std::vector<element_type> nearest_neighbours(coordinates_type const& target, std::size_t k,
                                                 coordinate_type eps = static_cast<coordinate_type>(1e-5)) const
    {
        if (k == 0 || storage_.empty()) return {};
        if constexpr (wide)
        {
            float q[K];
            for (std::size_t a = 0; a < K; ++a) q[a] = static_cast<float>(target[a]);
            std::vector<std::uint32_t> idx, count;
            wide_index().knn(q, 1, static_cast<std::uint32_t>(k), static_cast<float>(eps), idx, count);
            return gather(idx.data(), count[0]);
        }
        else
        {
            float const q[3] = {axis(target, 0), axis(target, 1), axis(target, 2)};
            auto const row = index().knn_one(q, static_cast<std::uint32_t>(k), static_cast<float>(eps));
            return gather(row.data(), row.size());
        }
    }
    std::vector<element_type> nearest_neighbours(element_type const& element_target, std::size_t k,
                                                 coordinate_type eps = static_cast<coordinate_type>(1e-5)) const
    {
        return nearest_neighbours(coordinate_map_(element_target), k, eps);
    }

    std::vector<element_type> nearest_neighbours_of(element_type const& e, std::size_t k, double eps = 1e-5) const
    {
        return nearest_neighbours(e, k, static_cast<coordinate_type>(eps));
    }

    template <class Range>
    std::vector<element_type> range_search(Range const& range) const
    {
        if (storage_.empty()) return {};
        std::vector<std::uint64_t> off;
        std::vector<std::uint32_t> idx;
        if constexpr (wide && std::is_same_v<Range, aabb_type>)
        {
            float b[2 * K];
            for (std::size_t a = 0; a < K; ++a)
            {
                b[a]     = static_cast<float>(range.min[a]);
                b[K + a] = static_cast<float>(range.max[a]);
            }
            wide_index().range_boxes(b, 1, off, idx);
        }
        else if constexpr (wide)
        {
            std::vector<element_type> out;
            for (auto const& e : storage_)
                if (range.contains(coordinate_map_(e))) out.push_back(e);
            return out;
        }
        else if constexpr (K == 3 && std::is_same_v<Range, sphere_a<coordinate_type>>)
        {
            float const c[3] = {axis(range.position, 0), axis(range.position, 1), axis(range.position, 2)};
            idx = index().range_sphere_one(c, static_cast<float>(range.radius));
        }
        else if constexpr (std::is_same_v<Range, aabb_type>)
        {
            float const b[6] = {axis(range.min, 0), axis(range.min, 1), axis(range.min, 2),
                                axis(range.max, 0), axis(range.max, 1), axis(range.max, 2)};
            index().range_boxes(b, 1, off, idx);
        }
        else
        {
            std::vector<element_type> out;
            for (auto const& e : storage_)
                if (range.contains(coordinate_map_(e))) out.push_back(e);
            return out;
        }
        return gather(idx.data(), idx.size());
    }

    // ---- batched additions --------------------------------------------------------------------
    template <class ForwardIter>
    std::vector<std::vector<element_type>> nearest_neighbours_batch(ForwardIter begin, ForwardIter end, std::size_t k,
                                                                    coordinate_type eps = static_cast<coordinate_type>(1e-5)) const
    {
        std::vector<float> q;
        for (; begin != end; ++begin)
        {
            auto const c = coordinate_map_(*begin);
            if constexpr (wide)
            {
                for (std::size_t a = 0; a < K; ++a) q.push_back(static_cast<float>(c[a]));
            }
            else
            {
                q.push_back(axis(c, 0));
                q.push_back(axis(c, 1));
                q.push_back(axis(c, 2));
            }
        }
        std::size_t const nq = q.size() / (wide ? K : 3);
        std::vector<std::vector<element_type>> rows(nq);
        if (k == 0 || storage_.empty() || nq == 0) return rows;
        if constexpr (wide)
        {
            std::vector<std::uint32_t> idx, count;
            wide_index().knn(q.data(), nq, static_cast<std::uint32_t>(k), static_cast<float>(eps), idx, count);
            for (std::size_t i = 0; i < nq; ++i) rows[i] = gather(idx.data() + i * k, count[i]);
        }
        else
        {
            auto const r = index().knn(q.data(), nq, static_cast<std::uint32_t>(k), static_cast<float>(eps));
            for (std::size_t i = 0; i < nq; ++i) rows[i] = gather(r.idx.data() + i * k, r.count[i]);
        }
        return rows;
    }
    std::vector<std::uint32_t> range_count_self(float radius) const
    {
        static_assert(!wide, "sphere ranges are three-dimensional (pcp::sphere_a)");
        if (storage_.empty()) return {};
        return index().range_count(xyz_.data(), storage_.size(), radius);
    }

    gpu::kd_wide_index_t const& wide_index() const
    {
        static_assert(wide, "K <= 3 goes through index()");
        std::lock_guard<std::mutex> lock(*mutex_);
        if (!wide_index_.valid()) wide_index_.build(xyz_.data(), storage_.size(), static_cast<std::uint32_t>(K));
        return wide_index_;
    }
    gpu::device_index_t const& index() const
    {
        static_assert(!wide, "K > 3 goes through wide_index()");
        std::lock_guard<std::mutex> lock(*mutex_);
        if (!index_.valid()) index_.build(xyz_.data(), storage_.size());
        return index_;
    }
    gpu::coord_buffer_t const& coordinates() const { return xyz_; }
    element_type const& element(std::size_t i) const { return storage_[i]; }
    CoordinateMap const& coordinate_map() const { return coordinate_map_; }

  private:
    std::vector<element_type> gather(std::uint32_t const* idx, std::size_t n) const
    {
        std::vector<element_type> out;
        out.reserve(n);
        for (std::size_t i = 0; i < n; ++i) out.push_back(storage_[idx[i]]);
        return out;
    }

    gpu::element_storage_t<element_type> storage_;
    gpu::coord_buffer_t xyz_;  // element i of storage_ at [3 i, 3 i + 3); K > 3: at [K i, K i + K)
    CoordinateMap coordinate_map_;
    kdtree::construction_params_t params_;
    aabb_type aabb_{};
    mutable gpu::device_index_t index_;
    mutable gpu::kd_wide_index_t wide_index_;  // K > 3
    mutable std::unique_ptr<std::mutex> mutex_ = std::make_unique<std::mutex>();
};

} // namespace pcp

#endif
