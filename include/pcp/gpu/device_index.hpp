// pcp::gpu::device_index_t -- RAII owner of a pcpx_index (include/pcpx.h): the device-resident
// Morton-sorted implicit AABB tree every container of this tree delegates to.  Not part of the
// reference API; the containers in pcp/octree and pcp/kdtree hold one of these instead of a node tree.
// Errors from the C ABI become std::runtime_error (there is no CPU fallback to fall back to).
#ifndef PCP_GPU_DEVICE_INDEX_HPP
#define PCP_GPU_DEVICE_INDEX_HPP

#include "pcpx.h"

#include <array>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pcp {
namespace gpu {

inline void check(int status, char const* what)
{
    if (status != PCPX_OK)
        throw std::runtime_error(std::string(what) + ": pcpx status " + std::to_string(status) + ": " + pcpx_last_error());
}

// The device the most recently built index lives on: what the entry points WITHOUT a handle use when they are called on
// behalf of a tree they cannot see (estimate_normals with an arbitrary knn_map lambda).
inline std::atomic<int>& default_device()
{
    static std::atomic<int> d{0};
    return d;
}

// k-nearest-neighbour rows of a batch: row q = idx[q*k .. q*k + count[q])
// (self queries of the whole index may come back in the index's curve order -- that form overlaps the copies with the
//  kernels, pcpx_normals_knn_self_curve_order -- with the table that says where the row of input point i is)
struct knn_result_t
{
    std::uint32_t k = 0;
    std::vector<std::uint32_t> idx;
    std::vector<std::uint32_t> count;
    std::vector<std::uint32_t> position_of;  // empty: row i belongs to query / input point i
    std::size_t row_of(std::size_t i) const { return position_of.empty() ? i : position_of[i]; }
    std::uint32_t const* row(std::size_t i) const { return idx.data() + row_of(i) * k; }
    std::uint32_t size_of_row(std::size_t i) const { return count[row_of(i)]; }
};

// A typed block of device memory (hipMalloc behind the C ABI: this header does not need a HIP toolchain).
template <class T>
class device_array_t
{
  public:
    device_array_t() = default;
    explicit device_array_t(std::size_t count, int device = 0) : n_(count), device_(device)
    {
        check(pcpx_device_malloc(static_cast<std::uint64_t>(count) * sizeof(T), device, &p_), "pcpx_device_malloc");
    }
    device_array_t(device_array_t const&)            = delete;
    device_array_t& operator=(device_array_t const&) = delete;
    device_array_t(device_array_t&& o) noexcept : p_(std::exchange(o.p_, nullptr)), n_(std::exchange(o.n_, 0)), device_(o.device_) {}
    device_array_t& operator=(device_array_t&& o) noexcept
    {
        if (this != &o)
        {
            pcpx_device_free(p_, device_);
            p_      = std::exchange(o.p_, nullptr);
            n_      = std::exchange(o.n_, 0);
            device_ = o.device_;
        }
        return *this;
    }
    ~device_array_t() { pcpx_device_free(p_, device_); }
    T* data() const { return static_cast<T*>(p_); }
    std::size_t size() const { return n_; }
    void upload(T const* src, std::size_t count) { check(pcpx_device_upload(p_, src, count * sizeof(T), device_, nullptr), "pcpx_device_upload"); }
    // elements [first, first + count) to the host
    std::vector<T> download(std::size_t first, std::size_t count) const
    {
        std::vector<T> out(count);
        check(pcpx_device_download(out.data(), data() + first, count * sizeof(T), device_, nullptr), "pcpx_device_download");
        return out;
    }
    std::vector<T> download() const { return download(0, n_); }

  private:
    void* p_       = nullptr;
    std::size_t n_ = 0;
    int device_    = 0;
};

// Device-resident results of the self queries (row i belongs to input point i): nothing crosses PCIe until asked for.
struct device_rows_t
{
    std::uint32_t k = 0;
    std::uint32_t pitch = 0;              // entries between rows of idx: k rounded up to 8 / 16 / 32 when that is k or k + 1 (a row is then
                                          // one aligned piece written with 16-byte stores: pcpx_knn_self_strided_dev), else k
    device_array_t<std::uint32_t> idx;    // rows x pitch; row i = idx[i * pitch .. i * pitch + count[i])
    device_array_t<std::uint32_t> count;  // rows
    device_array_t<float> normals;        // rows x 3 (empty unless normals were requested)
};

class device_index_t
{
  public:
    device_index_t() = default;
    device_index_t(device_index_t const&)            = delete;
    device_index_t& operator=(device_index_t const&) = delete;
    device_index_t(device_index_t&& o) noexcept { *this = std::move(o); }
    device_index_t& operator=(device_index_t&& o) noexcept
    {
        if (this != &o)
        {
            reset();
            h_      = std::exchange(o.h_, nullptr);
            n_      = std::exchange(o.n_, 0);
            xyz_    = std::exchange(o.xyz_, nullptr);
            n_in_   = std::exchange(o.n_in_, 0);
            lookup_ = std::move(o.lookup_);
            knn_    = std::move(o.knn_);
            range_  = std::move(o.range_);
        }
        return *this;
    }
    ~device_index_t() { reset(); }

    void reset()
    {
        if (h_) pcpx_index_destroy(h_);
        h_ = nullptr;
        n_ = 0;
        forget();
    }
    bool valid() const { return h_ != nullptr; }
    std::uint64_t size() const { return n_; }

    // xyz: n x 3 floats; grid6 (optional): {min, max} used as voxel grid / quantisation box
    void build(float const* xyz, std::uint64_t n, float const* grid6 = nullptr, int device = 0)
    {
        pcpx_build_params p{};
        p.struct_size = sizeof(p);
        if (grid6)
        {
            p.flags = PCPX_BUILD_USE_GRID;
            for (int a = 0; a < 3; ++a)
            {
                p.grid_min[a] = grid6[a];
                p.grid_max[a] = grid6[3 + a];
            }
        }
        if (h_) check(pcpx_index_rebuild(h_, xyz, n, grid6 ? &p : nullptr), "pcpx_index_rebuild");
        else
        {
            check(pcpx_index_create(xyz, n, grid6 ? &p : nullptr, device, &h_), "pcpx_index_create");
            default_device().store(device);
        }
        check(pcpx_index_size(h_, &n_), "pcpx_index_size");
        forget();
        xyz_  = xyz;  // the owner keeps the coordinates alive and rebuilds after any change (see the containers)
        n_in_ = n;
    }

    // ---- device-resident form: one fused launch, rows and normals stay in HBM (the full rate) ----
    device_rows_t knn_self_device(std::uint32_t k, float eps, std::uint64_t rows, bool with_normals, int device = 0) const
    {
        device_rows_t r;
        r.k = k;
        std::uint32_t const cap = k <= 8u ? 8u : k <= 16u ? 16u : 32u;
        r.pitch = (k <= 32u && k + 1u >= cap) ? cap : k;
        r.idx   = device_array_t<std::uint32_t>(static_cast<std::size_t>(rows) * r.pitch, device);
        r.count = device_array_t<std::uint32_t>(static_cast<std::size_t>(rows), device);
        std::uint32_t const stride = r.pitch == k ? 0u : r.pitch;
        if (with_normals)
        {
            r.normals = device_array_t<float>(static_cast<std::size_t>(rows) * 3, device);
            check(pcpx_normals_knn_self_strided_dev(h_, k, eps, 0, UINT64_MAX, stride, r.normals.data(), r.idx.data(), r.count.data()), "pcpx_normals_knn_self_strided_dev");
        }
        else check(pcpx_knn_self_strided_dev(h_, k, eps, 0, UINT64_MAX, stride, r.idx.data(), r.count.data(), nullptr), "pcpx_knn_self_strided_dev");
        check(pcpx_index_synchronize(h_), "pcpx_index_synchronize");
        return r;
    }

    // ---- per-point calls of an unchanged reference caller -------------------------------------------------
    // `nearest_neighbours(p, k)` inside a loop over the cloud's own points (examples/simple_example.cpp:83-99) is one
    // GPU round trip per point if taken literally.  After `batch_after` single-query calls with the same (k, eps) the
    // k nearest neighbours of EVERY indexed point are computed in one launch and kept on the host; a later call whose
    // target has exactly the coordinates of an indexed point is answered from those rows (the row of a point depends on
    // its coordinates only: the eps-box test excludes by position, not by index), any other target takes the
    // single-query path.  Same distances either way; which of several points tying EXACTLY with the k-th distance is
    // returned is unspecified (the cached rows come from the throughput kernel, an uncached call from the latency
    // kernel).  The same is done for sphere ranges of one radius.  If the one-off batch fill fails (out of memory, a device
    // error) the cache is marked failed and every later call takes the single-query path: the failure is not retried.
    static constexpr std::size_t single_range_room = 256;  // indices a single-range call offers up front
    static constexpr unsigned batch_after         = 16;
    static constexpr std::uint64_t max_cache_ints = 1ull << 28;  // 1 GiB of indices at most

    // row of k nearest neighbours of q (indices into the input array)
    std::vector<std::uint32_t> knn_one(float const* q, std::uint32_t k, float eps) const
    {
        std::vector<std::uint32_t> out;
        if (!h_ || k == 0) return out;
        {
            std::lock_guard<std::mutex> lock(*mu_);
            if (!(knn_.ready && knn_.k == k && knn_.eps == eps))
            {
                if (knn_.k != k || knn_.eps != eps) knn_ = knn_cache_t{};
                knn_.k   = k;
                knn_.eps = eps;
                if (!knn_.failed && ++knn_.calls >= batch_after && xyz_ && n_in_ * k <= max_cache_ints)
                {
                    knn_.failed = true;  // (cleared on success: a throw below downgrades to the single-query path for good)
                    try
                    {
                        build_lookup();
                        knn_.rows   = knn_self(k, eps, n_in_);
                        knn_.ready  = true;
                        knn_.failed = false;
                    }
                    catch (...)
                    {
                        knn_.rows = {};
                    }
                }
            }
            if (knn_.ready && knn_.k == k && knn_.eps == eps)
            {
                std::int64_t const i = find(q);
                if (i >= 0)
                {
                    auto const* row = knn_.rows.row(static_cast<std::size_t>(i));
                    out.assign(row, row + knn_.rows.size_of_row(static_cast<std::size_t>(i)));
                    return out;
                }
            }
        }
        out.resize(k);
        std::uint32_t count = 0;
        check(pcpx_knn_batch(h_, q, 1, k, eps, out.data(), &count, nullptr), "pcpx_knn_batch");
        out.resize(count);
        return out;
    }

    // indices of the points inside the sphere (centre c, radius r)
    std::vector<std::uint32_t> range_sphere_one(float const* c, float r) const
    {
        std::vector<std::uint32_t> out;
        if (!h_) return out;
        {
            std::lock_guard<std::mutex> lock(*mu_);
            if (!(range_.ready && range_.radius == r))
            {
                if (range_.radius != r) range_ = range_cache_t{};
                range_.radius = r;
                if (!range_.failed && ++range_.calls >= batch_after && xyz_ && n_in_ > 0)
                {
                    range_.failed = true;  // (cleared on success)
                    try
                    {
                        build_lookup();
                        std::vector<std::uint32_t> counts(static_cast<std::size_t>(n_in_));
                        check(pcpx_range_count_batch(h_, xyz_, n_in_, r, counts.data()), "pcpx_range_count_batch");
                        std::uint64_t total = 0;
                        for (auto v : counts) total += v;
                        if (total <= max_cache_ints)
                        {
                            range_.off.assign(static_cast<std::size_t>(n_in_) + 1, 0);
                            range_.idx.resize(static_cast<std::size_t>(total));
                            check(pcpx_range_sphere_batch(h_, xyz_, nullptr, r, n_in_, range_.off.data(), range_.idx.data(), total),
                                  "pcpx_range_sphere_batch");
                            range_.ready  = true;
                            range_.failed = false;
                        }
                    }
                    catch (...)
                    {
                        range_.off = {};
                        range_.idx = {};
                    }
                }
            }
            if (range_.ready && range_.radius == r)
            {
                std::int64_t const i = find(c);
                if (i >= 0)
                {
                    out.assign(range_.idx.begin() + static_cast<std::ptrdiff_t>(range_.off[static_cast<std::size_t>(i)]),
                               range_.idx.begin() + static_cast<std::ptrdiff_t>(range_.off[static_cast<std::size_t>(i) + 1]));
                    return out;
                }
            }
        }
        std::vector<std::uint64_t> off;
        range_spheres(c, &r, 1, off, out);
        return out;
    }

    std::array<float, 6> bbox() const
    {
        std::array<float, 6> b{};
        check(pcpx_index_bbox(h_, b.data()), "pcpx_index_bbox");
        return b;
    }

    knn_result_t knn(float const* q_xyz, std::uint64_t nq, std::uint32_t k, float eps) const
    {
        knn_result_t r;
        r.k = k;
        r.idx.resize(static_cast<std::size_t>(nq) * k);
        r.count.assign(static_cast<std::size_t>(nq), 0u);
        if (nq && k) check(pcpx_knn_batch(h_, q_xyz, nq, k, eps, r.idx.data(), r.count.data(), nullptr), "pcpx_knn_batch");
        return r;
    }
    knn_result_t knn_self(std::uint32_t k, float eps, std::uint64_t rows) const
    {
        knn_result_t r;
        r.k = k;
        r.idx.resize(static_cast<std::size_t>(rows) * k);
        r.count.assign(static_cast<std::size_t>(rows), 0u);
        if (!(rows && k)) return r;
        if (n_ == n_in_ && rows == n_in_)
        {
            // every input point is indexed: rows in curve order + the table of positions; the rows reach the host while later
            // slices are still being computed (access through row(i) / size_of_row(i))
            r.position_of.resize(static_cast<std::size_t>(rows));
            check(pcpx_normals_knn_self_curve_order(h_, k, eps, nullptr, r.idx.data(), r.count.data(), nullptr, r.position_of.data()),
                  "pcpx_normals_knn_self_curve_order");
        }
        else check(pcpx_knn_self(h_, k, eps, r.idx.data(), r.count.data(), nullptr), "pcpx_knn_self");
        return r;
    }
    // CSR lists of the points inside each sphere (per-sphere radii)
    void range_spheres(float const* centers, float const* radii, std::uint64_t n, std::vector<std::uint64_t>& off,
                       std::vector<std::uint32_t>& idx) const
    {
        off.assign(static_cast<std::size_t>(n) + 1, 0);
        idx.assign(n == 1 ? single_range_room : 0, 0u);  // (one range: room for a typical result, so that one call -- one launch -- does it)
        int st = pcpx_range_sphere_batch(h_, centers, radii, 0.f, n, off.data(), idx.empty() ? nullptr : idx.data(), idx.size());
        if (st == PCPX_OK) idx.resize(static_cast<std::size_t>(off[n]));
        if (st == PCPX_ERR_CAPACITY)
        {
            idx.resize(static_cast<std::size_t>(off[n]));
            st = pcpx_range_sphere_batch(h_, centers, radii, 0.f, n, off.data(), idx.data(), idx.size());
        }
        check(st, "pcpx_range_sphere_batch");
    }
    void range_boxes(float const* boxes6, std::uint64_t n, std::vector<std::uint64_t>& off, std::vector<std::uint32_t>& idx) const
    {
        off.assign(static_cast<std::size_t>(n) + 1, 0);
        idx.assign(n == 1 ? single_range_room : 0, 0u);
        int st = pcpx_range_aabb_batch(h_, boxes6, n, off.data(), idx.empty() ? nullptr : idx.data(), idx.size());
        if (st == PCPX_OK) idx.resize(static_cast<std::size_t>(off[n]));
        if (st == PCPX_ERR_CAPACITY)
        {
            idx.resize(static_cast<std::size_t>(off[n]));
            st = pcpx_range_aabb_batch(h_, boxes6, n, off.data(), idx.data(), idx.size());
        }
        check(st, "pcpx_range_aabb_batch");
    }
    std::vector<std::uint32_t> range_count(float const* centers, std::uint64_t n, float radius) const
    {
        std::vector<std::uint32_t> c(static_cast<std::size_t>(n), 0u);
        if (n) check(pcpx_range_count_batch(h_, centers, n, radius, c.data()), "pcpx_range_count_batch");
        return c;
    }
    // fused kNN + PCA normal of every indexed point; rows = number of input points
    std::vector<float> normals_self(std::uint32_t k, float eps, std::uint64_t rows) const
    {
        std::vector<float> nrm(static_cast<std::size_t>(rows) * 3, 0.f);
        if (rows) check(pcpx_normals_knn_self(h_, k, eps, nrm.data(), nullptr, nullptr), "pcpx_normals_knn_self");
        return nrm;
    }
    // fused kNN + (centroid, PCA normal) of every indexed point: the tangent planes
    void tangent_planes_self(std::uint32_t k, float eps, std::uint64_t rows, std::vector<float>& centroids,
                             std::vector<float>& normals) const
    {
        centroids.assign(static_cast<std::size_t>(rows) * 3, 0.f);
        normals.assign(static_cast<std::size_t>(rows) * 3, 0.f);
        if (rows) check(pcpx_tangent_planes_knn_self(h_, k, eps, centroids.data(), normals.data()), "pcpx_tangent_planes_knn_self");
    }
    // fused kNN + mean Euclidean distance to the k neighbours of every indexed point
    std::vector<float> mean_knn_distance_self(std::uint32_t k, float eps, std::uint64_t rows) const
    {
        std::vector<float> m(static_cast<std::size_t>(rows), 0.f);
        if (rows) check(pcpx_mean_knn_distance_self(h_, k, eps, m.data()), "pcpx_mean_knn_distance_self");
        return m;
    }
    // propagate_normal_orientations over the index's own kNN graph, all on the GPU; normals: rows x 3, in place
    std::uint64_t orient_normals_self(std::uint32_t k, float eps, std::vector<float>& normals) const
    {
        std::uint64_t reached = 0;
        if (!normals.empty()) check(pcpx_orient_normals_knn_self(h_, k, eps, normals.data(), &reached), "pcpx_orient_normals_knn_self");
        return reached;
    }
    std::vector<float> normals_from_knn(knn_result_t const& r) const
    {
        std::vector<float> nrm(r.count.size() * 3, 0.f);
        if (!r.count.empty())
            check(pcpx_normals_from_knn(h_, r.idx.data(), r.count.data(), r.count.size(), r.k, nrm.data(), nullptr),
                  "pcpx_normals_from_knn");
        if (r.position_of.empty()) return nrm;
        // rows in curve order (knn_self): normal i of the result belongs to input point i, like r.row(i)
        std::vector<float> by_input(r.position_of.size() * 3, 0.f);
        for (std::size_t i = 0; i < r.position_of.size(); ++i)
        {
            std::size_t const row = r.position_of[i];
            if (row >= r.count.size()) continue;  // a point outside the voxel grid has no row
            by_input[3 * i]     = nrm[3 * row];
            by_input[3 * i + 1] = nrm[3 * row + 1];
            by_input[3 * i + 2] = nrm[3 * row + 2];
        }
        return by_input;
    }

    pcpx_index* handle() const { return h_; }

  private:
    struct knn_cache_t
    {
        std::uint32_t k = 0;
        float eps       = 0.f;
        unsigned calls  = 0;
        bool ready = false, failed = false;
        knn_result_t rows;
    };
    struct range_cache_t
    {
        float radius   = -1.f;
        unsigned calls = 0;
        bool ready = false, failed = false;
        std::vector<std::uint64_t> off;
        std::vector<std::uint32_t> idx;
    };

    void forget()
    {
        xyz_  = nullptr;
        n_in_ = 0;
        lookup_.clear();
        knn_   = knn_cache_t{};
        range_ = range_cache_t{};
    }
    static std::uint64_t hash3(float const* p)
    {
        std::uint32_t b[3];
        std::memcpy(b, p, sizeof b);
        std::uint64_t h = (static_cast<std::uint64_t>(b[0]) * 0x9E3779B97F4A7C15ull) ^ (static_cast<std::uint64_t>(b[1]) * 0xC2B2AE3D27D4EB4Full) ^
                          (static_cast<std::uint64_t>(b[2]) * 0x165667B19E3779F9ull);
        return h ^ (h >> 29);
    }
    // coordinates (bit patterns) -> first index holding them: open addressing, built once per index
    void build_lookup() const
    {
        if (!lookup_.empty() || !xyz_) return;
        std::size_t cap = 16;
        while (cap < 2 * static_cast<std::size_t>(n_in_)) cap <<= 1;
        lookup_.assign(cap, 0xFFFFFFFFu);
        for (std::uint64_t i = 0; i < n_in_; ++i)
        {
            std::size_t s = static_cast<std::size_t>(hash3(xyz_ + 3 * i)) & (cap - 1);
            while (lookup_[s] != 0xFFFFFFFFu && std::memcmp(xyz_ + 3 * lookup_[s], xyz_ + 3 * i, 12) != 0) s = (s + 1) & (cap - 1);
            if (lookup_[s] == 0xFFFFFFFFu) lookup_[s] = static_cast<std::uint32_t>(i);
        }
    }
    std::int64_t find(float const* q) const
    {
        if (lookup_.empty()) return -1;
        std::size_t const cap = lookup_.size();
        std::size_t s         = static_cast<std::size_t>(hash3(q)) & (cap - 1);
        while (lookup_[s] != 0xFFFFFFFFu)
        {
            if (std::memcmp(xyz_ + 3 * lookup_[s], q, 12) == 0) return lookup_[s];
            s = (s + 1) & (cap - 1);
        }
        return -1;
    }

    pcpx_index* h_   = nullptr;
    std::uint64_t n_ = 0;
    float const* xyz_     = nullptr;  // coordinates the index was built from (owned by the container)
    std::uint64_t n_in_   = 0;
    mutable std::vector<std::uint32_t> lookup_;
    mutable knn_cache_t knn_;
    mutable range_cache_t range_;
    mutable std::unique_ptr<std::mutex> mu_ = std::make_unique<std::mutex>();
};

// RAII owner of a pcpx_kd_index (include/pcpx.h: kd-tree queries for K > 3 coordinates, exhaustive search on the GPU).
class kd_wide_index_t
{
  public:
    kd_wide_index_t() = default;
    kd_wide_index_t(kd_wide_index_t const&)            = delete;
    kd_wide_index_t& operator=(kd_wide_index_t const&) = delete;
    kd_wide_index_t(kd_wide_index_t&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    kd_wide_index_t& operator=(kd_wide_index_t&& o) noexcept
    {
        if (this != &o)
        {
            reset();
            h_ = std::exchange(o.h_, nullptr);
        }
        return *this;
    }
    ~kd_wide_index_t() { reset(); }
    void reset()
    {
        if (h_) pcpx_kd_destroy(h_);
        h_ = nullptr;
    }
    bool valid() const { return h_ != nullptr; }
    // rows: n x dims floats
    void build(float const* rows, std::uint64_t n, std::uint32_t dims, int device = 0)
    {
        reset();
        check(pcpx_kd_create(rows, n, dims, device, &h_), "pcpx_kd_create");
    }
    // idx: nq x k (0xFFFFFFFF beyond count[q]); nearest first
    void knn(float const* queries, std::uint64_t nq, std::uint32_t k, float eps, std::vector<std::uint32_t>& idx,
             std::vector<std::uint32_t>& count) const
    {
        idx.assign(nq * k, 0xFFFFFFFFu);
        count.assign(nq, 0u);
        if (nq == 0 || k == 0) return;
        check(pcpx_kd_knn_batch(h_, queries, nq, k, eps, idx.data(), count.data(), nullptr), "pcpx_kd_knn_batch");
    }
    // boxes: nb x 2 dims (min then max); CSR offsets (nb + 1) and indices
    void range_boxes(float const* boxes, std::uint64_t nb, std::vector<std::uint64_t>& off, std::vector<std::uint32_t>& idx) const
    {
        off.assign(nb + 1, 0);
        idx.clear();
        int st = pcpx_kd_range_aabb_batch(h_, boxes, nb, off.data(), nullptr, 0);
        if (st == PCPX_OK) return;
        if (st != PCPX_ERR_CAPACITY) check(st, "pcpx_kd_range_aabb_batch");
        idx.resize(off[nb]);
        check(pcpx_kd_range_aabb_batch(h_, boxes, nb, off.data(), idx.data(), idx.size()), "pcpx_kd_range_aabb_batch");
    }

  private:
    pcpx_kd_index* h_ = nullptr;
};

} // namespace gpu
} // namespace pcp

#endif
