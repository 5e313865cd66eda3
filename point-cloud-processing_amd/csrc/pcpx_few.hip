// pcpx_few.hip -- the LATENCY path: k nearest neighbours of a handful of query points.
//
// The reference's own benchmarks time ONE query per iteration (benchmark/spatial_data_structures_benchmark.cpp:243-264),
// and unchanged callers ask per point (examples/simple_example.cpp:83-85).  The throughput kernel (pcpx_query.hip) gives
// a query one lane and walks the tree step by dependent step: ~190 dependent record fetches for a lone query, and a sort
// of the batch in front of it.  Here a whole wavefront serves one query and the parallelism is INSIDE the query:
//
//   1. beam descent: the (up to 64) nodes of tree level 3 are tested by one lane each, then two levels per step -- the 16
//      grandchildren of a node are contiguous in the heap layout, so the 4 nearest nodes give 64 boxes, one per lane --
//      down to the 8 nearest leaves = 64 points, one per lane; tau0 = k-th smallest distance among them: an upper bound of
//      the true k-th distance;
//   2. pruned breadth-first sweep, again from level 3 (with the box distances step 1 already has) and two levels per
//      step: every grandchild of the frontier is tested by its own lane against tau0, the survivors are compacted (ballot +
//      prefix count) into the next frontier in LDS;
//   3. the points of the surviving leaves, one per lane, with d2 <= tau0 outside the eps-box are the candidates (a
//      superset of the answer); the k smallest by (d2, index) are found by counting ranks and written in order.
//
// About depth + 3 dependent memory round trips per query (rounds 1-2: 2 x depth + 3) instead of ~190, no sort, no allocation;
// the query and the row can live in pinned host memory (the host-pointer entry points do exactly that: no copy is issued),
// and a single query travels in the kernel arguments (one PCIe read less).
// Arithmetic is the reference's (d = p - q, dx*dx + dy*dy + dz*dz in float32 without FMA, eps-box exclusion), rows are
// ascending in (d2, index): identical to the throughput kernel's rows except for WHICH of several points tied exactly at
// the k-th distance is kept -- unspecified in the reference too (linked_octree_node.hpp:479-489).
// Capacity: k <= 32; a frontier of more than FRONTIER nodes or more than CANDS candidates (a query far outside a huge
// cloud, hundreds of exact ties) sets the query's flag and the caller re-runs the batch through the general path.
#include "pcpx_device.h"

namespace pcpx {

namespace {

constexpr int BEAM = 8;
constexpr int FRONTIER = 1024;
constexpr int CANDS = 768;

__device__ __forceinline__ float lane_value(float v, int j)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), j));
}

// rank of this lane's value among the first `width` lanes (ties broken by lane): counting, no data movement
template <int WIDTH>
__device__ __forceinline__ u32 rank_among(float v, u32 lane)
{
    u32 r = 0;
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) {
        const float o = lane_value(v, j);
        r += (o < v || (o == v && static_cast<u32>(j) < lane)) ? 1u : 0u;
    }
    return r;
}

__device__ __forceinline__ u32 level_base(int d) { return d == 0 ? 0u : (0x55555555u >> (32 - 2 * d)); }

// smallest value over the wave (DPP within rows of 16, then the four rows through scalar registers)
__device__ __forceinline__ u32 wave_min_u32(u32 v)
{
    auto step = [](u32 x, auto ctrl) {
        const u32 o = static_cast<u32>(__builtin_amdgcn_update_dpp(static_cast<int>(x), static_cast<int>(x), decltype(ctrl)::value, 0xF, 0xF, false));
        return o < x ? o : x;
    };
    v = step(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
    v = step(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
    v = step(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane of a row holds the row's minimum
    const u32 a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
              d = __builtin_amdgcn_readlane(v, 48);
    const u32 ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// The `keep` lanes with the smallest finite non-negative `v` write `node` to out[0 .. count) in ascending order of v (ties and
// the lowest 6 mantissa bits: by lane -- this only steers the beam); returns count.  Wave-uniform control flow.
__device__ __forceinline__ u32 keep_nearest(float v, bool valid, u32 node, u32 keep, u32* __restrict__ out, u32 lane)
{
    const float inf = std::numeric_limits<float>::infinity();
    u32 key = (valid && v < inf) ? ((__float_as_uint(v) & ~63u) | lane) : 0xFFFFFFFFu;
    u32 count = 0;
    for (; count < keep; ++count) {
        const u32 m = wave_min_u32(key);
        if (m == 0xFFFFFFFFu) break;
        if (lane == (m & 63u)) {
            out[count] = node;
            key = 0xFFFFFFFFu;
        }
    }
    return count;
}

// q_in_args != 0 (a single query): the query is q0x, q0y, q0z and `queries` is not read
__global__ __launch_bounds__(64) void k_knn_few(TreeView t, const float* __restrict__ queries, u32 nq, u32 q_in_args, float q0x, float q0y, float q0z, u32 k,
                                                float eps, u32* __restrict__ out_idx, u32* __restrict__ out_cnt, float* __restrict__ out_d2,
                                                u32* __restrict__ flags, u32* __restrict__ done_count, u32* __restrict__ done_flag,
                                                u32 epoch)
{
    __shared__ u32 front[2][FRONTIER];
    __shared__ u64 cand[CANDS];
    const u32 lane = threadIdx.x;
    const u32 qi = blockIdx.x;
    if (qi >= nq) return;
    const float inf = std::numeric_limits<float>::infinity();
    // Completion signal for a host that polls instead of waiting on the stream: when the last block of the launch has
    // made its row visible system-wide it stores the launch's epoch into *done_flag (pinned host memory).
    auto signal_done = [&]() {
        __threadfence_system();  // this block's row, count and flag are visible to the host before the counter moves
        if (nq == 1u) {  // the only block: no counter to consult (one device-memory atomic round trip less on the way out)
            if (lane == 0 && done_flag) __hip_atomic_store(done_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (lane == 0 && done_count) {
            const u32 prev = atomicAdd(done_count, 1u);
            if (prev == nq - 1u) {
                *done_count = 0u;  // ready for the next launch on this stream
                __threadfence_system();
                __hip_atomic_store(done_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    float qx = q0x, qy = q0y, qz = q0z;
    if (!q_in_args) {
        qx = queries[3ull * qi];
        qy = queries[3ull * qi + 1];
        qz = queries[3ull * qi + 2];
    }
    u32* row_idx = out_idx + static_cast<u64>(qi) * k;
    float* row_d2 = out_d2 ? out_d2 + static_cast<u64>(qi) * k : nullptr;
    if (lane == 0) flags[qi] = 0u;
    if (t.nleaves == 0) {
        if (lane < k) {
            row_idx[lane] = INVALID_ID;
            if (row_d2) row_d2[lane] = inf;
        }
        if (lane == 0) out_cnt[qi] = 0u;
        signal_done();
        return;
    }

    // real nodes of tree level l (the ones the build writes: the children of a real node need not be real)
    // (the tree's bottom level is UNITS of UNIT_LEAVES leaf records)
    auto nreal = [&](int l) { return (t.nunits() + (1u << (2 * (t.depth - l))) - 1u) >> (2 * (t.depth - l)); };
    // box distance of node `node` of level l, +inf if it does not exist or is padding (NaN)
    auto box_at = [&](int l, u32 node, bool mine) {
        float v = inf;
        if (mine && node < nreal(l)) {
            const NodeBox b = t.nodes[level_base(l) + node];
            const float w = box_d2(b, qx, qy, qz);
            v = w == w ? w : inf;
        }
        return v;
    };

    // ---- 1. beam descent ----
    const int l0 = t.depth < 3 ? t.depth : 3;  // first level looked at: at most 64 nodes, lane = node
    const float bd0 = box_at(l0, lane, lane < (1u << (2 * l0)));
    int cur = 0;
    constexpr u32 BEAM_UNITS = BEAM / UNIT_LEAVES;  // the beam's last step keeps 64 points' worth of units
    u32 nbeam = keep_nearest(bd0, true, lane, l0 == t.depth ? BEAM_UNITS : 4u, front[0], lane);
    __syncthreads();
    for (int d = l0; d < t.depth;) {
        const int s = t.depth - d >= 2 ? 2 : 1;  // levels this step goes down: 16 (or 4) descendants per beam node
        const u32 fan = 1u << (2 * s);
        const bool mine = lane < nbeam * fan;
        const u32 node = mine ? front[cur][lane >> (2 * s)] * fan + (lane & (fan - 1u)) : 0u;
        const float bd = box_at(d + s, node, mine);
        d += s;
        nbeam = keep_nearest(bd, mine, node, d == t.depth ? BEAM_UNITS : 4u, front[cur ^ 1], lane);
        cur ^= 1;
        __syncthreads();
    }
    // the beam's leaves: up to 64 points, one per lane
    float tau0 = inf;
    {
        float d2 = inf;
        if (lane < static_cast<u32>(UNIT_POINTS) * nbeam) {
            const u32 leaf = front[cur][lane / UNIT_POINTS] * UNIT_LEAVES + (lane % UNIT_POINTS) / LEAF;
            if (leaf < t.nleaves) {
                const Leaf& lf = t.leaves[leaf];
                const float dx = lf.x[lane & 7u] - qx, dy = lf.y[lane & 7u] - qy, dz = lf.z[lane & 7u] - qz;
                const float v = sq3(dx, dy, dz);
                const float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
                if (v == v && m >= eps) d2 = v;  // NaN: padding slot of the last leaf
            }
        }
        const u32 r = rank_among<64>(d2, lane);
        const u64 kth = __builtin_amdgcn_ballot_w64(r == k - 1u && d2 < inf);
        if (kth != 0ull) tau0 = lane_value(d2, __builtin_ctzll(kth));
    }

    // ---- 2. pruned breadth-first sweep, from level l0 (whose box distances are in bd0) ----
    __syncthreads();
    u32 m = 0;  // frontier size at level d
    bool overflow = false;
    {
        const bool need = bd0 <= tau0;  // (+inf: no such node; tau0 = +inf keeps every real one)
        const bool keep0 = need && bd0 < inf;
        const u64 mask = __builtin_amdgcn_ballot_w64(keep0);
        const u32 below = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mask), 0u));
        if (keep0) front[0][below] = lane;
        m = static_cast<u32>(__builtin_popcountll(mask));
    }
    cur = 0;
    __syncthreads();
    for (int d = l0; d < t.depth && m > 0;) {
        const int s = t.depth - d >= 2 ? 2 : 1;
        const u32 fan = 1u << (2 * s);
        u32 next = 0;
        for (u32 c0 = 0; c0 < fan * m; c0 += 64u) {
            const u32 c = c0 + lane;
            const bool mine = c < fan * m;
            const u32 node = mine ? front[cur][c >> (2 * s)] * fan + (c & (fan - 1u)) : 0u;
            const bool need = box_at(d + s, node, mine) <= tau0 && mine && node < nreal(d + s);
            const u64 mask = __builtin_amdgcn_ballot_w64(need);
            const u32 below = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mask), 0u));
            const u32 slot = next + below;
            if (need && slot < FRONTIER) front[cur ^ 1][slot] = node;
            next += static_cast<u32>(__builtin_popcountll(mask));
        }
        if (next > FRONTIER) {
            overflow = true;
            next = FRONTIER;
        }
        d += s;
        m = next;
        cur ^= 1;
        __syncthreads();
    }

    // ---- 3. candidates: every point of the surviving leaves within tau0 and outside the eps-box ----
    u32 nc = 0;
    for (u32 c0 = 0; c0 < static_cast<u32>(UNIT_POINTS) * m; c0 += 64u) {
        const u32 c = c0 + lane;
        bool take = false;
        u64 key = 0;
        if (c < static_cast<u32>(UNIT_POINTS) * m) {
            const u32 leaf = front[cur][c / UNIT_POINTS] * UNIT_LEAVES + (c % UNIT_POINTS) / LEAF;
            if (leaf < t.nleaves) {
                const Leaf& lf = t.leaves[leaf];
                const u32 s = c & 7u;
                const float dx = lf.x[s] - qx, dy = lf.y[s] - qy, dz = lf.z[s] - qz;
                const float v = sq3(dx, dy, dz);
                const float mm = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
                take = v <= tau0 && mm >= eps;  // NaN padding fails v <= tau0
                key = (static_cast<u64>(__float_as_uint(v)) << 32) | lf.id[s];
            }
        }
        const u64 mask = __builtin_amdgcn_ballot_w64(take);
        const u32 below = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mask), 0u));
        const u32 slot = nc + below;
        if (take && slot < CANDS) cand[slot] = key;
        nc += static_cast<u32>(__builtin_popcountll(mask));
    }
    if (nc > CANDS) {
        overflow = true;
        nc = CANDS;
    }
    __syncthreads();

    // ---- the k smallest keys, by counting ranks (keys are distinct: the low word is the point's index) ----
    const u32 found = nc < k ? nc : k;
    for (u32 i = lane; i < nc; i += 64u) {
        const u64 mine = cand[i];
        u32 r = 0;
        for (u32 j = 0; j < nc; ++j) r += cand[j] < mine ? 1u : 0u;
        if (r < k) {
            row_idx[r] = static_cast<u32>(mine);
            if (row_d2) row_d2[r] = __uint_as_float(static_cast<u32>(mine >> 32));
        }
    }
    for (u32 j = found + lane; j < k; j += 64u) {
        row_idx[j] = INVALID_ID;
        if (row_d2) row_d2[j] = inf;
    }
    if (lane == 0) {
        out_cnt[qi] = found;
        if (overflow) flags[qi] = 1u;
    }
    signal_done();
}

}  // namespace

// queries (nq x 3), rows and flags may be device memory or pinned host memory mapped into the device
int launch_knn_few(Index& ix, const float* q_aos, const float* q_host, u32 nq, u32 k, float eps, u32* out_idx, u32* out_cnt, float* out_d2,
                   u32* flags, u32* done_count, u32* done_flag, u32 epoch)
{
    if (nq == 0) return PCPX_OK;
    ProfileScope prof(ix, PCPX_K_KNN);
    // a single query travels in the kernel arguments when the caller can read it (q_host: the same query on the host side)
    const bool in_args = nq == 1 && q_host != nullptr;
    k_knn_few<<<nq, 64, 0, ix.stream>>>(ix.view(), q_aos, nq, in_args ? 1u : 0u, in_args ? q_host[0] : 0.f, in_args ? q_host[1] : 0.f,
                                        in_args ? q_host[2] : 0.f, k, sanitize_eps(eps), out_idx, out_cnt, out_d2, flags, done_count, done_flag,
                                        epoch);
    return check_hip(hipGetLastError(), "k_knn_few launch", __FILE__, __LINE__);
}

}  // namespace pcpx
