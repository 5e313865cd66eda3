// pcpx_normals.hip -- pcp::estimate_normal over explicit neighbourhoods (rows of indices, or one point set); the
// fused form lives in the kNN kernel (pcpx_query.hip), both use the solver of pcpx_eig3.h.
#include "pcpx_eig3.h"

namespace pcpx {

namespace {

// thread i handles neighbourhood row = rowmap ? rowmap[first+i] : first+i
__global__ __launch_bounds__(256) void k_normals(const float* __restrict__ xyz, const u32* __restrict__ nbr,
                                                 const u32* __restrict__ cnt, const u32* __restrict__ rowmap, u64 first,
                                                 u64 count, u32 k, float* __restrict__ out, float* __restrict__ evals,
                                                 float* __restrict__ centroids, float* __restrict__ meandist, u32 row_bias)
{
    u64 i = blockIdx.x * static_cast<u64>(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    u64 row = rowmap ? rowmap[first + i] : static_cast<u32>(static_cast<u32>(first + i) + row_bias);
    u32 n = cnt ? cnt[row] : k;
    const u32* nb = nbr + row * k;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (u32 j = 0; j < n; ++j) {
        u64 id = nb[j];
        float x = xyz[3 * id], y = xyz[3 * id + 1], z = xyz[3 * id + 2];
        if (j == 0) { sx = x; sy = y; sz = z; }
        else { sx += x; sy += y; sz += z; }
    }
    float fn = static_cast<float>(n);
    float mx = sx / fn, my = sy / fn, mz = sz / fn;
    if (centroids) {  // center_of_geometry of the row: the tangent plane's point
        centroids[3 * row] = mx;
        centroids[3 * row + 1] = my;
        centroids[3 * row + 2] = mz;
    }
    if (meandist) {  // average_distances_to_neighbors; the row belongs to indexed point `row` (self queries only)
        const float qx = xyz[3 * row], qy = xyz[3 * row + 1], qz = xyz[3 * row + 2];
        float sum = 0.f;
        for (u32 j = 0; j < n; ++j) {
            u64 id = nb[j];
            float dx = xyz[3 * id] - qx, dy = xyz[3 * id + 1] - qy, dz = xyz[3 * id + 2] - qz;
            sum += sqrtf(sq3(dx, dy, dz));
        }
        meandist[row] = sum / fn;
    }
    if (!out) return;
    float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
    for (u32 j = 0; j < n; ++j) {
        u64 id = nb[j];
        float vx = xyz[3 * id] - mx, vy = xyz[3 * id + 1] - my, vz = xyz[3 * id + 2] - mz;
        c00 += vx * vx;
        c10 += vy * vx;
        c11 += vy * vy;
        c20 += vz * vx;
        c21 += vz * vy;
        c22 += vz * vz;
    }
    float nrm[3], ev[3];
    eig3_smallest(c00, c10, c20, c11, c21, c22, nrm, ev);
    out[3 * row] = nrm[0];
    out[3 * row + 1] = nrm[1];
    out[3 * row + 2] = nrm[2];
    if (evals) {
        evals[3 * row] = ev[0];
        evals[3 * row + 1] = ev[1];
        evals[3 * row + 2] = ev[2];
    }
}

// estimate_normal over an explicit point set (m x 3): a single thread, the set is tiny in practice
__global__ void k_normal_single(const float* __restrict__ xyz, u64 m, float* __restrict__ out3)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (u64 j = 0; j < m; ++j) {
        float x = xyz[3 * j], y = xyz[3 * j + 1], z = xyz[3 * j + 2];
        if (j == 0) { sx = x; sy = y; sz = z; }
        else { sx += x; sy += y; sz += z; }
    }
    float fn = static_cast<float>(m);
    float mx = sx / fn, my = sy / fn, mz = sz / fn;
    float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
    for (u64 j = 0; j < m; ++j) {
        float vx = xyz[3 * j] - mx, vy = xyz[3 * j + 1] - my, vz = xyz[3 * j + 2] - mz;
        c00 += vx * vx; c10 += vy * vx; c11 += vy * vy; c20 += vz * vx; c21 += vz * vy; c22 += vz * vz;
    }
    float ev[3];
    eig3_smallest(c00, c10, c20, c11, c21, c22, out3, ev);
}

// The same for up to NORMAL_ARG_POINTS points that travel in the kernel arguments (the per-call shape of the reference:
// examples/simple_example.cpp:83-99 -- k nearest neighbours, then estimate_normal of them, point by point): one wavefront, lane j
// fetches point j (so the argument block's cache lines are requested together, not one after the other across PCIe), the sums
// are then formed in index order exactly as above from the lanes' values, the normal goes to `out3` (pinned host memory) and
// the host polls `done_flag` for `epoch`.
struct NormalPointsArg {
    float v[3 * NORMAL_ARG_POINTS];
};
__global__ __launch_bounds__(64) void k_normal_args(NormalPointsArg pts, u32 m, float* __restrict__ out3, u32* __restrict__ done_flag,
                                                    u32 epoch)
{
    const u32 lane = threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    if (lane < m) {
        x = pts.v[3 * lane];
        y = pts.v[3 * lane + 1];
        z = pts.v[3 * lane + 2];
    }
    auto at = [](float v, u32 j) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), j)); };
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (u32 j = 0; j < m; ++j) {
        const float xj = at(x, j), yj = at(y, j), zj = at(z, j);
        if (j == 0) { sx = xj; sy = yj; sz = zj; }
        else { sx += xj; sy += yj; sz += zj; }
    }
    const float fn = static_cast<float>(m);
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
    for (u32 j = 0; j < m; ++j) {
        const float vx = at(x, j) - mx, vy = at(y, j) - my, vz = at(z, j) - mz;
        c00 += vx * vx; c10 += vy * vx; c11 += vy * vy; c20 += vz * vx; c21 += vz * vy; c22 += vz * vz;
    }
    float nrm[3], ev[3];
    eig3_smallest(c00, c10, c20, c11, c21, c22, nrm, ev);
    if (lane == 0) {
        out3[0] = nrm[0];
        out3[1] = nrm[1];
        out3[2] = nrm[2];
    }
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(done_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// estimate_normal over many explicit point sets at once: row r = points [offsets[r], offsets[r+1]) of xyz (relative to
// offsets[0]); one thread per row, same arithmetic and order as k_normal_single
__global__ __launch_bounds__(256) void k_normals_csr(const float* __restrict__ xyz, const u64* __restrict__ offsets, u64 nrows,
                                                     float* __restrict__ out)
{
    const u64 r = blockIdx.x * static_cast<u64>(blockDim.x) + threadIdx.x;
    if (r >= nrows) return;
    const u64 base = offsets[0], a = offsets[r] - base, b = offsets[r + 1] - base;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (u64 j = a; j < b; ++j) {
        float x = xyz[3 * j], y = xyz[3 * j + 1], z = xyz[3 * j + 2];
        if (j == a) { sx = x; sy = y; sz = z; }
        else { sx += x; sy += y; sz += z; }
    }
    float fn = static_cast<float>(b - a);
    float mx = sx / fn, my = sy / fn, mz = sz / fn;
    float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
    for (u64 j = a; j < b; ++j) {
        float vx = xyz[3 * j] - mx, vy = xyz[3 * j + 1] - my, vz = xyz[3 * j + 2] - mz;
        c00 += vx * vx; c10 += vy * vx; c11 += vy * vy; c20 += vz * vx; c21 += vz * vy; c22 += vz * vz;
    }
    float nrm[3], ev[3];
    eig3_smallest(c00, c10, c20, c11, c21, c22, nrm, ev);
    out[3 * r] = nrm[0];
    out[3 * r + 1] = nrm[1];
    out[3 * r + 2] = nrm[2];
}

}  // namespace

int launch_normals_csr(const float* d_xyz, const u64* d_offsets, u64 nrows, float* d_out, hipStream_t s)
{
    if (nrows == 0) return PCPX_OK;
    k_normals_csr<<<static_cast<u32>((nrows + 255) / 256), 256, 0, s>>>(d_xyz, d_offsets, nrows, d_out);
    return check_hip(hipGetLastError(), "k_normals_csr launch", __FILE__, __LINE__);
}

int launch_normals(Index& ix, const u32* d_nbr, const u32* d_cnt, const u32* d_rowmap, u64 first, u64 count, u32 k,
                   float* d_out, float* d_evals, float* d_centroids, float* d_meandist, u32 row_bias)
{
    if (count == 0) return PCPX_OK;
    ProfileScope prof(ix, PCPX_K_NORMALS);
    hipStream_t s = ix.stream;
    const float* d_xyz = ix.shard.on ? ix.shard.cloud : ix.d_xyz;  // (rows hold indices into the whole cloud)
    k_normals<<<static_cast<u32>((count + 255) / 256), 256, 0, s>>>(d_xyz, d_nbr, d_cnt, d_rowmap, first, count, k, d_out,
                                                                     d_evals, d_centroids, d_meandist, row_bias);
    return check_hip(hipGetLastError(), "k_normals launch", __FILE__, __LINE__);
}

// xyz: on the host, m <= NORMAL_ARG_POINTS points; out3 and done_flag: pinned host memory the device writes in place
int launch_normal_args(const float* xyz, u32 m, float* out3, u32* done_flag, u32 epoch, hipStream_t s)
{
    NormalPointsArg pts;
    std::memcpy(pts.v, xyz, static_cast<size_t>(m) * 3 * sizeof(float));
    k_normal_args<<<1, 64, 0, s>>>(pts, m, out3, done_flag, epoch);
    return check_hip(hipGetLastError(), "k_normal_args launch", __FILE__, __LINE__);
}

int launch_normal_single(const float* d_xyz, u64 m, float* d_out3, hipStream_t s)
{
    k_normal_single<<<1, 64, 0, s>>>(d_xyz, m, d_out3);
    return check_hip(hipGetLastError(), "k_normal_single launch", __FILE__, __LINE__);
}

}  // namespace pcpx
