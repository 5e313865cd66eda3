// pcp::gpu::device_index_t -- RAII owner of a pcpx_index (include/pcpx.h): the device-resident
// Morton-sorted implicit AABB tree every container of this tree delegates to.  Not part of the
// reference API; the containers in pcp/octree and pcp/kdtree hold one of these instead of a node tree.
// Errors from the C ABI become std::runtime_error (there is no CPU fallback to fall back to).
#ifndef PCP_GPU_DEVICE_INDEX_HPP
#define PCP_GPU_DEVICE_INDEX_HPP

#include "pcpx.h"

#include <array>
#include <cstdint>
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

// k-nearest-neighbour rows of a batch: row q = idx[q*k .. q*k + count[q])
struct knn_result_t
{
    std::uint32_t k = 0;
    std::vector<std::uint32_t> idx;
    std::vector<std::uint32_t> count;
};

class device_index_t
{
  public:
    device_index_t() = default;
    device_index_t(device_index_t const&)            = delete;
    device_index_t& operator=(device_index_t const&) = delete;
    device_index_t(device_index_t&& o) noexcept : h_(std::exchange(o.h_, nullptr)), n_(o.n_) {}
    device_index_t& operator=(device_index_t&& o) noexcept
    {
        if (this != &o)
        {
            reset();
            h_ = std::exchange(o.h_, nullptr);
            n_ = o.n_;
        }
        return *this;
    }
    ~device_index_t() { reset(); }

    void reset()
    {
        if (h_) pcpx_index_destroy(h_);
        h_ = nullptr;
        n_ = 0;
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
        else check(pcpx_index_create(xyz, n, grid6 ? &p : nullptr, device, &h_), "pcpx_index_create");
        check(pcpx_index_size(h_, &n_), "pcpx_index_size");
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
        if (rows && k) check(pcpx_knn_self(h_, k, eps, r.idx.data(), r.count.data(), nullptr), "pcpx_knn_self");
        return r;
    }
    // CSR lists of the points inside each sphere (per-sphere radii)
    void range_spheres(float const* centers, float const* radii, std::uint64_t n, std::vector<std::uint64_t>& off,
                       std::vector<std::uint32_t>& idx) const
    {
        off.assign(static_cast<std::size_t>(n) + 1, 0);
        idx.clear();
        int st = pcpx_range_sphere_batch(h_, centers, radii, 0.f, n, off.data(), nullptr, 0);
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
        idx.clear();
        int st = pcpx_range_aabb_batch(h_, boxes6, n, off.data(), nullptr, 0);
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
        return nrm;
    }

    pcpx_index* handle() const { return h_; }

  private:
    pcpx_index* h_   = nullptr;
    std::uint64_t n_ = 0;
};

} // namespace gpu
} // namespace pcp

#endif
