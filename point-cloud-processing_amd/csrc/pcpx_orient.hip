// pcpx_orient.hip -- pcp::algorithm::propagate_normal_orientations (include/pcp/algorithm/estimate_normals.hpp:187-302)
// on the GPU, with the result of the reference's SEQUENTIAL breadth-first search.
//
// The reference pops vertices from a FIFO queue; a vertex v is oriented against the vertex u it is first reached
// from, i.e. the earliest (queue position of u, edge index j in u's neighbour row) with edge u -> v among the not
// yet visited v, and joins the queue in that order (include/pcp/graph/search.hpp:36-85).  All vertices discovered
// while the vertices of BFS level d are popped form level d+1, in order of their winning (position, edge) pair.
// That is a level-synchronous formulation with a deterministic tie rule:
//   claim    every edge (p, j) of the frontier proposes key = p << 32 | j to its unvisited target: atomicMin
//   resolve  the edge whose key survived orients the target against its source (whose normal is final: it was
//            settled one level earlier), stamps the target with the new level, and is counted per source
//   scan     exclusive prefix sum of the per-source win counts = where each source's discoveries start in the
//            next frontier
//   emit     winners are written to the next frontier in (p, j) order
// Edge order inside a row and frontier order are exactly the reference's, so the flips are bit-identical with the
// host pass (pcpx_propagate_normal_orientations) -- tests/test_gpu_parity.py checks that.
#include "pcpx_internal.h"

#include <cmath>

#pragma clang fp contract(off)

namespace pcpx {

namespace {

constexpr u32 NO_LEVEL = 0xFFFFFFFFu;
constexpr int OB = 256;           // threads per block
constexpr int SCAN_ITEMS = 8;     // per thread in the block scan
constexpr int SCAN_TILE = OB * SCAN_ITEMS;

// ---- root: the FIRST point of largest z (std::max_element with `a.z < b.z`, estimate_normals.hpp:224-232) ----
// ordered key of z: larger z = larger key; NaN never wins a `>` comparison = lowest key
__device__ __forceinline__ u32 z_key(float z)
{
    if (z != z) return 0u;
    u32 b = z == 0.f ? 0u : __float_as_uint(z);  // -0 and +0 compare equal: one key
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // >= 1 for every non-NaN (~0xFFFFFFFF cannot occur: that is a NaN)
}

__global__ __launch_bounds__(OB) void k_orient_maxz(const float* __restrict__ xyz, u32 n, u32* __restrict__ best_key)
{
    u32 m = 0;
    for (u32 i = blockIdx.x * OB + threadIdx.x; i < n; i += gridDim.x * OB) {
        u32 key = z_key(xyz[3ull * i + 2]);
        m = key > m ? key : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        u32 o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(best_key, m);
}

__global__ __launch_bounds__(OB) void k_orient_root(const float* __restrict__ xyz, u32 n, const u32* __restrict__ best_key,
                                                    u32* __restrict__ root)
{
    const u32 want = *best_key;
    u32 r = NO_LEVEL;
    for (u32 i = blockIdx.x * OB + threadIdx.x; i < n; i += gridDim.x * OB)
        if (z_key(xyz[3ull * i + 2]) == want) {
            r = i;
            break;  // this thread's smallest index
        }
    for (int off = 32; off > 0; off >>= 1) {
        u32 o = __shfl_xor(r, off);
        r = o < r ? o : r;
    }
    if ((threadIdx.x & 63) == 0 && r != NO_LEVEL) atomicMin(root, r);
}

// level 0: the root, normal (0, 0, 1).  A NaN z at index 0 makes every comparison of the reference's search
// false: the root is then index 0 whatever follows.
__global__ void k_orient_start(const float* __restrict__ xyz, u32* __restrict__ root, u32* __restrict__ level,
                               u32* __restrict__ frontier, float* __restrict__ normals)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float z0 = xyz[2];
    u32 r = (z0 != z0 || *root == NO_LEVEL) ? 0u : *root;
    *root = r;
    level[r] = 0;
    frontier[0] = r;
    normals[3ull * r] = 0.f;
    normals[3ull * r + 1] = 0.f;
    normals[3ull * r + 2] = 1.f;
}

// The frontier size lives in device memory (fsize_dev): several levels are enqueued per host synchronisation, and a
// level whose frontier turned out empty is a no-op.
__global__ __launch_bounds__(OB) void k_orient_claim(const u32* __restrict__ frontier, const u32* __restrict__ fsize_dev,
                                                     const u32* __restrict__ nbr, const u32* __restrict__ cnt, u32 k,
                                                     const u32* __restrict__ level, u64* __restrict__ claim)
{
    const u32 fsize = *fsize_dev;
    for (u32 p = blockIdx.x * OB + threadIdx.x; p < fsize; p += gridDim.x * OB) {
        const u32 u = frontier[p];
        const u32 c = cnt ? cnt[u] : k;
        const u32* row = nbr + static_cast<u64>(u) * k;
        for (u32 j = 0; j < c; ++j) {
            const u32 v = row[j];
            if (level[v] == NO_LEVEL) atomicMin(reinterpret_cast<unsigned long long*>(&claim[v]), (static_cast<u64>(p) << 32) | j);
        }
    }
}

__global__ __launch_bounds__(OB) void k_orient_resolve(const u32* __restrict__ frontier, const u32* __restrict__ fsize_dev,
                                                       const u32* __restrict__ nbr, const u32* __restrict__ cnt, u32 k,
                                                       u32* __restrict__ level, const u64* __restrict__ claim, u32 next_level,
                                                       float* __restrict__ normals, u32* __restrict__ wins)
{
    const u32 fsize = *fsize_dev;
    for (u32 p = blockIdx.x * OB + threadIdx.x; p < fsize; p += gridDim.x * OB) {
        const u32 u = frontier[p];
        const u32 c = cnt ? cnt[u] : k;
        const u32* row = nbr + static_cast<u64>(u) * k;
        const float ax = normals[3ull * u], ay = normals[3ull * u + 1], az = normals[3ull * u + 2];  // settled one level ago
        u32 w = 0;
        for (u32 j = 0; j < c; ++j) {
            const u32 v = row[j];
            // the winner of v is unique (one surviving key per v): only one thread ever passes this test for v.  A
            // target settled in an earlier level keeps its old claim, which a (p, j) of this level may equal by
            // coincidence, hence the level test comes first.
            if (level[v] == NO_LEVEL && claim[v] == ((static_cast<u64>(p) << 32) | j)) {
                float* b = normals + 3ull * v;
                const float bx = b[0], by = b[1], bz = b[2];
                const float xx = bx * ax, yy = by * ay, zz = bz * az;  // inner_product(n1, n2): norm.hpp:34-45
                const float dot = xx + yy + zz;
                if (dot < 0.f && !(fabsf(dot - 0.f) < 1e-5f)) {  // estimate_normals.hpp:290-298, vector3d_queries.hpp:31-35
                    b[0] = -bx;
                    b[1] = -by;
                    b[2] = -bz;
                }
                level[v] = next_level;
                ++w;
            }
        }
        wins[p] = w;
    }
}

// ---- exclusive prefix sum of u32 (tile sums, one block over the tile sums, add back) --------------------
__global__ __launch_bounds__(OB) void k_scan_tiles(const u32* __restrict__ in, const u32* __restrict__ n_dev, u32* __restrict__ out,
                                                   u32* __restrict__ tile_sum)
{
    __shared__ u32 part[OB];
    const u32 n = *n_dev;
    const u32 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // block-uniform trip count
        const u32 base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
        u32 v[SCAN_ITEMS], s = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            v[i] = base + i < n ? in[base + i] : 0u;
            s += v[i];
        }
        part[threadIdx.x] = s;
        __syncthreads();
        for (int off = 1; off < OB; off <<= 1) {  // Hillis-Steele over the 256 thread sums
            u32 t = threadIdx.x >= static_cast<u32>(off) ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        u32 run = part[threadIdx.x] - s;  // exclusive
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            if (base + i < n) out[base + i] = run;
            run += v[i];
        }
        if (threadIdx.x == OB - 1) tile_sum[tile] = part[OB - 1];
        __syncthreads();
    }
}

// one block: exclusive scan of the tile sums in place; the grand total is the next frontier's size.  Also keeps
// the running totals of the search: counters[0] += vertices discovered, counters[1] += 1 for a non-empty level.
__global__ __launch_bounds__(OB) void k_scan_tile_sums(u32* __restrict__ tile_sum, const u32* __restrict__ n_dev,
                                                       u32* __restrict__ total, unsigned long long* __restrict__ counters)
{
    __shared__ u32 part[OB];
    __shared__ u32 carry;
    const u32 ntiles = (*n_dev + SCAN_TILE - 1) / SCAN_TILE;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < ntiles; base += OB) {
        const u32 i = base + threadIdx.x;
        const u32 x = i < ntiles ? tile_sum[i] : 0u;
        part[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < OB; off <<= 1) {
            u32 t = threadIdx.x >= static_cast<u32>(off) ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < ntiles) tile_sum[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 0) carry += part[OB - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry;
        counters[0] += carry;
        counters[1] += *n_dev > 0 ? 1ull : 0ull;
    }
}

__global__ __launch_bounds__(OB) void k_orient_emit(const u32* __restrict__ frontier, const u32* __restrict__ fsize_dev,
                                                    const u32* __restrict__ nbr, const u32* __restrict__ cnt, u32 k,
                                                    const u32* __restrict__ level, const u64* __restrict__ claim, u32 next_level,
                                                    const u32* __restrict__ offs, const u32* __restrict__ tile_sum,
                                                    u32* __restrict__ next_frontier)
{
    const u32 fsize = *fsize_dev;
    for (u32 p = blockIdx.x * OB + threadIdx.x; p < fsize; p += gridDim.x * OB) {
        const u32 u = frontier[p];
        const u32 c = cnt ? cnt[u] : k;
        const u32* row = nbr + static_cast<u64>(u) * k;
        u32 at = offs[p] + tile_sum[p / SCAN_TILE];
        for (u32 j = 0; j < c; ++j) {
            const u32 v = row[j];
            if (level[v] == next_level && claim[v] == ((static_cast<u64>(p) << 32) | j)) next_frontier[at++] = v;
        }
    }
}

__global__ __launch_bounds__(OB) void k_orient_validate(const u32* __restrict__ nbr, const u32* __restrict__ cnt, u64 n, u32 k,
                                                        u32* __restrict__ bad)
{
    for (u64 i = blockIdx.x * static_cast<u64>(OB) + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * OB) {
        const u32 c = cnt ? cnt[i] : k;
        if (c > k) {
            atomicOr(bad, 1u);
            continue;
        }
        for (u32 j = 0; j < c; ++j)
            if (nbr[i * k + j] >= n) atomicOr(bad, 2u);
    }
}

inline u32 blocks_for(u64 n, u32 cap = 4096)
{
    u64 b = (n + OB - 1) / OB;
    return static_cast<u32>(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

int orient_normals_device(const float* d_xyz, u64 n, const u32* d_nbr, const u32* d_cnt, u32 k, float* d_normals, hipStream_t s,
                          u64* out_reached, u32* out_levels)
{
    if (out_reached) *out_reached = 0;
    if (out_levels) *out_levels = 0;
    if (n == 0) return PCPX_OK;
    if (n >= NO_LEVEL) {
        set_error("pcpx: normal orientation is limited to 2^32 - 2 vertices");
        return PCPX_ERR_UNSUPPORTED;
    }
    const u32 n32 = static_cast<u32>(n);
    const u32 ntiles = (n32 + SCAN_TILE - 1) / SCAN_TILE;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_claim = 0, o_level = o_claim + al(n * 8), o_f0 = o_level + al(n * 4), o_f1 = o_f0 + al(n * 4),
                 o_wins = o_f1 + al(n * 4), o_offs = o_wins + al(n * 4), o_tiles = o_offs + al(n * 4),
                 o_scal = o_tiles + al(static_cast<size_t>(ntiles) * 4), total_bytes = o_scal + 256;
    char* base = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), total_bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for the orientation pass failed: %s", total_bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return PCPX_ERR_ALLOC;
    }
    struct Free {
        char* p;
        ~Free() { (void)hipFree(p); }
    } guard{base};
    u64* claim = reinterpret_cast<u64*>(base + o_claim);
    u32* level = reinterpret_cast<u32*>(base + o_level);
    u32* fr[2] = {reinterpret_cast<u32*>(base + o_f0), reinterpret_cast<u32*>(base + o_f1)};
    u32* wins = reinterpret_cast<u32*>(base + o_wins);
    u32* offs = reinterpret_cast<u32*>(base + o_offs);
    u32* tiles = reinterpret_cast<u32*>(base + o_tiles);
    u32* scal = reinterpret_cast<u32*>(base + o_scal);  // [0] max z key, [1] root, [3] bad rows, [4..5] frontier sizes, [8..11] counters

    PCPX_HIP(hipMemsetAsync(claim, 0xFF, n * 8, s));
    PCPX_HIP(hipMemsetAsync(level, 0xFF, n * 4, s));
    PCPX_HIP(hipMemsetAsync(scal, 0, 64, s));
    PCPX_HIP(hipMemsetAsync(scal + 1, 0xFF, 4, s));
    k_orient_validate<<<blocks_for(n), OB, 0, s>>>(d_nbr, d_cnt, n, k, scal + 3);
    u32 bad = 0;
    PCPX_HIP(hipMemcpyAsync(&bad, scal + 3, sizeof(u32), hipMemcpyDeviceToHost, s));
    PCPX_HIP(hipStreamSynchronize(s));
    if (bad) {
        set_error("pcpx: neighbour rows for the orientation pass hold %s", (bad & 2u) ? "an index >= n" : "a count > k");
        return PCPX_ERR_INVALID;
    }
    k_orient_maxz<<<blocks_for(n, 1024), OB, 0, s>>>(d_xyz, n32, scal);
    k_orient_root<<<blocks_for(n, 1024), OB, 0, s>>>(d_xyz, n32, scal, scal + 1);
    k_orient_start<<<1, 64, 0, s>>>(d_xyz, scal + 1, level, fr[0], d_normals);
    PCPX_HIP(hipGetLastError());

    // fs[0], fs[1]: frontier sizes of even / odd levels; counters: {vertices discovered, non-empty levels}
    u32* fs = scal + 4;
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(scal + 8);
    const u32 one = 1;
    PCPX_HIP(hipMemcpyAsync(fs, &one, sizeof(u32), hipMemcpyHostToDevice, s));
    const u32 grid = blocks_for(n, 2048);
    constexpr u32 LEVELS_PER_SYNC = 8;  // an empty frontier makes the remaining levels of a batch no-ops
    u32 depth = 0;
    for (;;) {
        for (u32 i = 0; i < LEVELS_PER_SYNC; ++i, ++depth) {
            const int cur = depth & 1;
            k_orient_claim<<<grid, OB, 0, s>>>(fr[cur], fs + cur, d_nbr, d_cnt, k, level, claim);
            k_orient_resolve<<<grid, OB, 0, s>>>(fr[cur], fs + cur, d_nbr, d_cnt, k, level, claim, depth + 1, d_normals, wins);
            k_scan_tiles<<<grid, OB, 0, s>>>(wins, fs + cur, offs, tiles);
            k_scan_tile_sums<<<1, OB, 0, s>>>(tiles, fs + cur, fs + (cur ^ 1), counters);
            k_orient_emit<<<grid, OB, 0, s>>>(fr[cur], fs + cur, d_nbr, d_cnt, k, level, claim, depth + 1, offs, tiles, fr[cur ^ 1]);
        }
        u32 next = 0;
        PCPX_HIP(hipMemcpyAsync(&next, fs + (depth & 1), sizeof(u32), hipMemcpyDeviceToHost, s));
        PCPX_HIP(hipStreamSynchronize(s));
        PCPX_HIP(hipGetLastError());
        if (next == 0) break;
        if (depth > n32 + LEVELS_PER_SYNC) {  // cannot happen: every level settles at least one vertex
            set_error("pcpx: orientation search did not terminate");
            return PCPX_ERR_DEVICE;
        }
    }
    unsigned long long totals[2] = {0, 0};
    PCPX_HIP(hipMemcpy(totals, counters, sizeof(totals), hipMemcpyDeviceToHost));
    const u64 reached = 1 + totals[0];
    depth = static_cast<u32>(totals[1]);
    if (out_reached) *out_reached = reached;
    if (out_levels) *out_levels = depth;
    return PCPX_OK;
}

}  // namespace pcpx
