// pcpx_shard.hip -- the RANK-LOCAL index of the multi-GPU path (PCPX_BUILD_SHARD).
//
// One process per GPU, the cloud replicated, the queries sharded along the curve (pcpx_shard_range): with the whole-cloud index
// every rank sorts and boxes all n points to answer n / world queries, and in BASELINE configs[4] (rebuild + kNN every
// iteration) that replicated build caps the 8-rank speed-up at 5.2x.  Here a rank indexes only what its shard can reach:
//
//   keys        one sort word per point of the WHOLE cloud (k_codes, as before) + the number of points in every level-4 cell
//               of the curve (4096 cells, counted in LDS on the same sweep);
//   plan        the cells that hold curve positions [first, first + count) of the whole cloud's order are the CORE
//               (k_shard_plan: a scan of the 4096 counts; the curve order of the cells is the order of the points);
//   selection   core + a HALO of `h` cells of the 64^3 selection grid around it, as a bitmap over curve prefixes
//               (k_sel_*: mark on the xyz grid, dilate per axis, pack); cells an earlier coverage check asked for stay in;
//   compaction  the selected words, in input order (count per tile, scan, stable compaction), each with its point's record
//               {x, y, z, index in the cloud};
//   local tree  the same radix sort, leaf fill and box sweep as the whole-cloud build, over the selected points only.
//               Their relative order is the whole-cloud order (same key bits, ties by input order), so every tie between
//               equal distances is broken as the whole-cloud index breaks it: results are bit-identical.
//
// A query is exact if every point within its k-th distance is in the tree: k_knn leaves the k-th squared distance of every
// answered position behind (KnnOutputs::tau) and k_shard_verify checks that every selection cell the ball touches is selected.
// Queries that fail (points in sparse places whose ball leaves the halo) are collected, the cells they miss are added to
// the selection, the local tree is rebuilt over the larger selection and only those queries are answered again, as a batch of
// arbitrary queries whose rows go where the failed rows were -- their balls are now covered, so one round settles it.  The
// handle keeps the added cells: a static index pays for them once, a streaming one (whose cloud moves a little per
// iteration) keeps them across rebuilds.
//
// Nothing is exchanged between ranks: every rank derives its shard from its own copy of the cloud and the common grid, so
// BASELINE north_star's "an RCCL all-gather of per-rank bounding boxes and nothing else" holds.
#include "pcpx_curve.h"
#include "pcpx_device.h"

#include <algorithm>
#include <cmath>

namespace pcpx {

namespace {

constexpr int SEL_LEVEL = 6;                      // selection cells per axis: 2^6
constexpr u32 SEL_AXIS = 1u << SEL_LEVEL;         // 64
constexpr u32 SEL_CELLS = 1u << (3 * SEL_LEVEL);  // 262144
constexpr u32 SEL_WORDS = SEL_CELLS / 32;         // bitmap words: 8192 (32 KB, fits LDS)
constexpr int SEL_SHIFT = 64 - 3 * SEL_LEVEL;     // sort word -> curve prefix of its selection cell
constexpr int CORE_SHIFT = 52;                    // sort word -> level-4 cell (12 bits)
constexpr u32 CORE_CELLS = 4096;

// device scalars of a rank-local build (Index::Shard::d_plan)
enum Plan : u32 {
    P_CLO = 0,      // first level-4 cell of the core
    P_CHI = 1,      // last one (P_CLO > P_CHI: the shard is empty)
    P_G0 = 2,       // global curve position of the core's first point
    P_CORE = 3,     // points in the core's cells
    P_FIRST = 4,    // the shard: first global position ...
    P_COUNT = 5,    // ... and count
    P_NVALID = 6,   // inserted points of the whole cloud
    P_M = 7,        // selected points
    P_L0 = 8,       // local position of the core's first point
    P_LCORE = 9,    // core points found in the local order (= P_CORE)
    P_NSEL = 10,    // selected cells
    P_ALL = 11,     // a coverage check asked for (nearly) everything
    P_WORDS = 16
};

__device__ __forceinline__ bool word_is_outside(u64 w, int idx_bits) { return (w >> idx_bits) == (~0ull >> idx_bits); }

// curve prefix (3 * SEL_LEVEL bits) of the selection cell (cx, cy, cz)
__device__ __forceinline__ u32 sel_prefix(u32 cx, u32 cy, u32 cz, const u32* __restrict__ htab)
{
    constexpr int S = CURVE_BITS - SEL_LEVEL;
    return static_cast<u32>(hilbert_index_table(cx << S, cy << S, cz << S, htab) >> (3 * S));
}

// ---- plan: which cells of the curve hold the shard -------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_shard_plan(const u32* __restrict__ hist12, u32 rank, u32 world, u32 explicit_range, u64 range_first,
                                                     u64 range_count, u32* __restrict__ plan)
{
    __shared__ u32 wsum[16];
    __shared__ u32 res[4];
    const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
    u32 c[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = hist12[4 * t + j];
        s += c[j];
    }
    u32 incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 up = __shfl_up(incl, off);
        if (lane >= static_cast<u32>(off)) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    if (t < 4) res[t] = t == 0 ? 1u : 0u;  // c_lo = 1 > c_hi = 0: empty
    __syncthreads();
    u32 before = 0, total = 0;
    for (u32 i = 0; i < 16; ++i) {
        before += i < w ? wsum[i] : 0u;
        total += wsum[i];
    }
    u32 prefix = before + incl - s;  // points in the cells before cell 4t
    // the shard of pcpx_shard_range over the inserted points
    const u64 nvalid = total;
    const u64 groups = (nvalid + GROUP - 1) / GROUP;
    u64 first = groups * rank / world * GROUP, end = groups * (static_cast<u64>(rank) + 1) / world * GROUP;
    if (explicit_range) {  // (PCPX_BUILD_SHARD_RANGE: a cut by work, pcpx_shard_cuts_by_cost)
        first = range_first;
        end = range_count > nvalid - (first < nvalid ? first : nvalid) ? nvalid : first + range_count;
    }
    if (first > nvalid) first = nvalid;
    if (end > nvalid) end = nvalid;
    if (end > first) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c[j] != 0u) {
                if (prefix <= first && first < static_cast<u64>(prefix) + c[j]) {
                    res[0] = 4 * t + j;
                    res[2] = prefix;
                }
                if (prefix <= end - 1 && end - 1 < static_cast<u64>(prefix) + c[j]) {
                    res[1] = 4 * t + j;
                    res[3] = prefix + c[j];
                }
            }
            prefix += c[j];
        }
    }
    __syncthreads();
    if (t == 0) {
        plan[P_CLO] = res[0];
        plan[P_CHI] = res[1];
        plan[P_G0] = end > first ? res[2] : static_cast<u32>(first);
        plan[P_CORE] = end > first ? res[3] - res[2] : 0u;
        plan[P_FIRST] = static_cast<u32>(first);
        plan[P_COUNT] = static_cast<u32>(end - first);
        plan[P_NVALID] = static_cast<u32>(nvalid);
        plan[P_NSEL] = 0;
        plan[P_ALL] = 0;
    }
}

// ---- selection: core cells on the xyz grid, dilated per axis, packed into a bitmap over curve prefixes ---------------------
__global__ __launch_bounds__(1024) void k_sel_mark(const u32* __restrict__ plan, unsigned char* __restrict__ grid)
{
    __shared__ u32 htab[HILBERT_TABLE_WORDS];
    hilbert_table_to_lds(htab);
    const u32 cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= SEL_CELLS) return;
    const u32 cx = cell & (SEL_AXIS - 1), cy = (cell >> SEL_LEVEL) & (SEL_AXIS - 1), cz = cell >> (2 * SEL_LEVEL);
    const u32 c12 = sel_prefix(cx, cy, cz, htab) >> (3 * SEL_LEVEL - 12);
    grid[cell] = (plan[P_CLO] <= c12 && c12 <= plan[P_CHI]) ? 1 : 0;
}
__global__ __launch_bounds__(1024) void k_sel_dilate(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int axis, int h)
{
    const u32 cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= SEL_CELLS) return;
    const int sh = axis * SEL_LEVEL;
    const int c = static_cast<int>((cell >> sh) & (SEL_AXIS - 1));
    const u32 rest = cell & ~((SEL_AXIS - 1) << sh);
    const int lo = c - h < 0 ? 0 : c - h, hi = c + h > static_cast<int>(SEL_AXIS) - 1 ? static_cast<int>(SEL_AXIS) - 1 : c + h;
    unsigned char v = 0;
    for (int j = lo; j <= hi; ++j) v |= in[rest | (static_cast<u32>(j) << sh)];
    out[cell] = v;
}
// sel = need (cells earlier coverage checks asked for) | the dilated core; counts the selected cells
__global__ __launch_bounds__(1024) void k_sel_pack(const unsigned char* __restrict__ grid, const u32* __restrict__ need, u32* __restrict__ sel,
                                                   u32* __restrict__ plan)
{
    __shared__ u32 htab[HILBERT_TABLE_WORDS];
    hilbert_table_to_lds(htab);
    const u32 cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= SEL_CELLS) return;
    const u32 cx = cell & (SEL_AXIS - 1), cy = (cell >> SEL_LEVEL) & (SEL_AXIS - 1), cz = cell >> (2 * SEL_LEVEL);
    const u32 h = sel_prefix(cx, cy, cz, htab);
    const bool on = grid[cell] != 0 || ((need[h >> 5] >> (h & 31u)) & 1u) != 0;
    // (only cells that were not selected before are counted: the range path runs this again over a selection that already holds
    //  the earlier halo, and P_NSEL >= SEL_CELLS means "everything is selected: nothing to verify")
    const bool fresh = on && (atomicOr(&sel[h >> 5], 1u << (h & 31u)) & (1u << (h & 31u))) == 0u;
    const u64 b = __builtin_amdgcn_ballot_w64(fresh);
    if ((threadIdx.x & 63u) == 0 && b) atomicAdd(&plan[P_NSEL], static_cast<u32>(__builtin_popcountll(b)));
}
__global__ __launch_bounds__(1024) void k_sel_or(u32* __restrict__ sel, const u32* __restrict__ need, u32* __restrict__ plan)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SEL_WORDS) return;
    const u32 before = sel[i], after = before | need[i];
    sel[i] = after;
    const u32 added = static_cast<u32>(__builtin_popcount(after ^ before));
    if (added) atomicAdd(&plan[P_NSEL], added);
}
__global__ __launch_bounds__(1024) void k_fill_u32(u32* __restrict__ p, u32 n, u32 v)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- compaction of the selected words, in input order ------------------------------------------------------------------------
constexpr int CMP_BLOCK = 256;
constexpr int CMP_ITEMS = SORT_TILE_WORDS / CMP_BLOCK;  // 16 rows of 256 words per tile

__device__ __forceinline__ void sel_to_lds(u32* lds_sel, const u32* __restrict__ sel)
{
    for (u32 i = threadIdx.x; i < SEL_WORDS; i += blockDim.x) lds_sel[i] = sel[i];
    __syncthreads();
}
__device__ __forceinline__ bool word_selected(u64 w, int idx_bits, const u32* lds_sel)
{
    const u32 h = static_cast<u32>(w >> SEL_SHIFT);
    return !word_is_outside(w, idx_bits) && ((lds_sel[h >> 5] >> (h & 31u)) & 1u) != 0;
}

__global__ __launch_bounds__(CMP_BLOCK) void k_shard_count(const u64* __restrict__ codes, u32 n, int idx_bits, const u32* __restrict__ sel,
                                                            u32 ntiles, u32* __restrict__ tile_cnt)
{
    __shared__ u32 lds_sel[SEL_WORDS];
    __shared__ u32 wcnt[CMP_BLOCK / 64];
    sel_to_lds(lds_sel, sel);
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u32 lo = tile * SORT_TILE_WORDS;
        u32 mine = 0;
#pragma unroll
        for (int it = 0; it < CMP_ITEMS; ++it) {
            const u32 i = lo + it * CMP_BLOCK + threadIdx.x;
            if (i < n) mine += word_selected(codes[i], idx_bits, lds_sel) ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if ((threadIdx.x & 63u) == 0) wcnt[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) tile_cnt[tile] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}

// exclusive scan of the tiles' counts in place (one block; a few thousand tiles), total -> plan[P_M]
__global__ __launch_bounds__(1024) void k_shard_scan(u32* __restrict__ tile_cnt, u32 ntiles, u32* __restrict__ plan)
{
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s;
    const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < ntiles; base += 1024) {
        const u32 i = base + t;
        const u32 v = i < ntiles ? tile_cnt[i] : 0u;
        u32 incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 up = __shfl_up(incl, off);
            if (lane >= static_cast<u32>(off)) incl += up;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        u32 before = carry_s, total = 0;
        for (u32 j = 0; j < 16; ++j) {
            before += j < w ? wsum[j] : 0u;
            total += wsum[j];
        }
        if (i < ntiles) tile_cnt[i] = before + incl - v;
        __syncthreads();
        if (t == 0) carry_s += total;
        __syncthreads();
    }
    if (t == 0) plan[P_M] = carry_s;
}

// the selected words of a tile go to positions tile_off[tile] ... in input order: word = key bits | position, record =
// {x, y, z, index in the cloud} at the same position
__global__ __launch_bounds__(CMP_BLOCK) void k_shard_compact(const u64* __restrict__ codes, const float* __restrict__ cloud, u32 n, int idx_bits,
                                                              const u32* __restrict__ sel, u32 ntiles, const u32* __restrict__ tile_off,
                                                              u64* __restrict__ words, float4* __restrict__ rec_in)
{
    __shared__ u32 lds_sel[SEL_WORDS];
    __shared__ u32 rowcnt[CMP_ITEMS * (CMP_BLOCK / 64)];  // [row][wave], in output order
    sel_to_lds(lds_sel, sel);
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u64 low = (1ull << idx_bits) - 1ull;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u32 lo = tile * SORT_TILE_WORDS;
        u64 word[CMP_ITEMS];
        u32 flags = 0;
#pragma unroll
        for (int it = 0; it < CMP_ITEMS; ++it) {
            const u32 i = lo + it * CMP_BLOCK + threadIdx.x;
            word[it] = codes[i < n ? i : n - 1u];
            const bool on = i < n && word_selected(word[it], idx_bits, lds_sel);
            flags |= on ? (1u << it) : 0u;
            const u64 b = __builtin_amdgcn_ballot_w64(on);
            if (lane == 0) rowcnt[it * (CMP_BLOCK / 64) + w] = static_cast<u32>(__builtin_popcountll(b));
        }
        __syncthreads();
        if (w == 0) {  // exclusive scan of the 64 (row, wave) counts by one wave
            const u32 v = rowcnt[lane];
            u32 incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 up = __shfl_up(incl, off);
                if (lane >= static_cast<u32>(off)) incl += up;
            }
            rowcnt[lane] = incl - v;
        }
        __syncthreads();
        const u32 base = tile_off[tile];
#pragma unroll
        for (int it = 0; it < CMP_ITEMS; ++it) {
            const bool on = (flags >> it) & 1u;
            const u64 b = __builtin_amdgcn_ballot_w64(on);
            if (on) {
                const u32 i = lo + it * CMP_BLOCK + threadIdx.x;
                const u32 below = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(b >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(b), 0u));
                const u32 dst = base + rowcnt[it * (CMP_BLOCK / 64) + w] + below;
                words[dst] = (word[it] & ~low) | dst;
                rec_in[dst] = make_float4(cloud[3ull * i], cloud[3ull * i + 1], cloud[3ull * i + 2], __uint_as_float(i));
            }
        }
        __syncthreads();
    }
}

// where the core sits in the local order (one thread: two binary searches over the sorted words)
__global__ void k_shard_locate(const u64* __restrict__ sorted, u32* __restrict__ plan)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u32 m = plan[P_M], clo = plan[P_CLO], chi = plan[P_CHI];
    auto lower = [&](u32 cell) {  // first position whose level-4 cell is >= cell
        u32 lo = 0, hi = m;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (static_cast<u32>(sorted[mid] >> CORE_SHIFT) < cell) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    if (clo > chi) {
        plan[P_L0] = 0;
        plan[P_LCORE] = 0;
        return;
    }
    const u32 a = lower(clo), b = lower(chi + 1u);
    plan[P_L0] = a;
    plan[P_LCORE] = b - a;
}

// ---- coverage check ----------------------------------------------------------------------------------------------------------
// Position p of the local order (a self query that has just been answered) is exact iff every point of the cloud within its
// k-th distance is in the tree, and it is if every selection cell that the axis-aligned box around its search ball touches is
// selected.  The box is taken a little larger than the ball (the float arithmetic of the distance, of q +- r, and the
// quantisation of a coordinate are all monotone; one ulp outwards and a relative 1e-5 cover their rounding).  Cells the
// check misses are noted in `need`; positions that fail are listed.  Most queries never look at the bitmap: a query's own cell
// is a core cell, and every cell within `halo` cells of a core cell is selected by construction, so a box that reaches no
// farther than that from the query's cell is covered.  A box of more than BIG_BOX cells (a straggler far from
// everything) is not walked by its own thread: it goes on a list that k_shard_verify_big works off with a block per box; a
// box of more than half the grid, or more such boxes than the list holds, asks for everything.
constexpr u32 BIG_BOX = 512, BIG_CAP = 8192, BIG_WORDS = 8;

__global__ __launch_bounds__(256) void k_shard_verify(TreeView t, const float* __restrict__ box6, const float* __restrict__ tau, u32 pos_lo, u32 pos_hi,
                                                       const u32* __restrict__ sel, u32* __restrict__ need, u32* __restrict__ fail, u32* __restrict__ plan,
                                                       u32* __restrict__ big, u32 halo)
{
    __shared__ u32 htab[HILBERT_TABLE_WORDS];
    hilbert_table_to_lds(htab);
    const u32 p = pos_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pos_hi) return;
    const CurveGrid g = curve_grid(box6[0], box6[1], box6[2], box6[3], box6[4], box6[5]);
    const Leaf& lf = t.leaves[p / LEAF];
    const float q[3] = {lf.x[p % LEAF], lf.y[p % LEAF], lf.z[p % LEAF]};
    const float tk = tau[p];
    bool bad = false;
    if (!(tk < std::numeric_limits<float>::infinity())) {
        bad = true;  // fewer than k neighbours in the tree: only the whole cloud can tell
        atomicExch(&plan[P_ALL], 1u);
    } else {
        const float r = sqrtf(tk) * 1.00001f + 1e-37f;
        u32 c0[3], c1[3];
        constexpr int S = CURVE_BITS - SEL_LEVEL;
        u32 reach = 0;  // cells the box extends from the query's own cell
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = nextafterf(q[a] - r, -std::numeric_limits<float>::infinity());
            const float hi = nextafterf(q[a] + r, std::numeric_limits<float>::infinity());
            c0[a] = curve_cell(lo, g.lo[a], g.scale[a]) >> S;
            c1[a] = curve_cell(hi, g.lo[a], g.scale[a]) >> S;
            const u32 cq = curve_cell(q[a], g.lo[a], g.scale[a]) >> S;
            reach = max(reach, max(cq - c0[a], c1[a] - cq));
        }
        const u32 cells = (c1[0] - c0[0] + 1u) * (c1[1] - c0[1] + 1u) * (c1[2] - c0[2] + 1u);
        if (reach <= halo) {
            // covered by construction
        } else if (cells > SEL_CELLS / 2) {
            bad = true;
            atomicExch(&plan[P_ALL], 1u);
        } else if (cells > BIG_BOX) {
            const u32 slot = atomicAdd(&big[0], 1u);
            if (slot < BIG_CAP) {
                u32* e = big + 1 + static_cast<size_t>(slot) * BIG_WORDS;
                e[0] = p;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    e[1 + a] = c0[a];
                    e[4 + a] = c1[a];
                }
            } else {
                bad = true;
                atomicExch(&plan[P_ALL], 1u);
            }
        } else {
            for (u32 cz = c0[2]; cz <= c1[2]; ++cz)
                for (u32 cy = c0[1]; cy <= c1[1]; ++cy)
                    for (u32 cx = c0[0]; cx <= c1[0]; ++cx) {
                        const u32 h = sel_prefix(cx, cy, cz, htab);
                        if (((sel[h >> 5] >> (h & 31u)) & 1u) == 0u) {
                            bad = true;
                            atomicOr(&need[h >> 5], 1u << (h & 31u));
                        }
                    }
        }
    }
    if (bad) fail[1u + atomicAdd(&fail[0], 1u)] = p;
}
// one block per listed box
__global__ __launch_bounds__(256) void k_shard_verify_big(const u32* __restrict__ big, const u32* __restrict__ sel, u32* __restrict__ need,
                                                           u32* __restrict__ fail)
{
    __shared__ u32 htab[HILBERT_TABLE_WORDS];
    __shared__ u32 bad_s;
    hilbert_table_to_lds(htab);
    const u32 listed = big[0] < BIG_CAP ? big[0] : BIG_CAP;
    for (u32 b = blockIdx.x; b < listed; b += gridDim.x) {
        const u32* e = big + 1 + static_cast<size_t>(b) * BIG_WORDS;
        if (threadIdx.x == 0) bad_s = 0;
        __syncthreads();
        const u32 nx = e[4] - e[1] + 1u, ny = e[5] - e[2] + 1u, nz = e[6] - e[3] + 1u;
        bool bad = false;
        for (u32 c = threadIdx.x; c < nx * ny * nz; c += blockDim.x) {
            const u32 cx = e[1] + c % nx, cy = e[2] + (c / nx) % ny, cz = e[3] + c / (nx * ny);
            const u32 h = sel_prefix(cx, cy, cz, htab);
            if (((sel[h >> 5] >> (h & 31u)) & 1u) == 0u) {
                bad = true;
                atomicOr(&need[h >> 5], 1u << (h & 31u));
            }
        }
        if (bad) bad_s = 1;
        __syncthreads();
        if (threadIdx.x == 0 && bad_s) fail[1u + atomicAdd(&fail[0], 1u)] = e[0];
        __syncthreads();
    }
}

// the failed queries as a batch of arbitrary queries: coordinates and the rows their answers go to
__global__ __launch_bounds__(256) void k_shard_collect(TreeView t, const u32* __restrict__ fail, u32 nfail, u32 by_position, u32 pos_bias,
                                                        float* __restrict__ q, u32* __restrict__ dest)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfail) return;
    const u32 p = fail[1u + i];
    const Leaf& lf = t.leaves[p / LEAF];
    q[3ull * i] = lf.x[p % LEAF];
    q[3ull * i + 1] = lf.y[p % LEAF];
    q[3ull * i + 2] = lf.z[p % LEAF];
    dest[i] = by_position ? p + pos_bias : lf.id[p % LEAF];
}
__global__ __launch_bounds__(256) void k_shard_remap_rows(u32* __restrict__ row, u32 nq, const u32* __restrict__ dest)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) row[i] = dest[row[i]];
}

// perm / position table of the core at global positions
__global__ __launch_bounds__(256) void k_shard_perm(const u32* __restrict__ perm, u32 l0, u32 count, u32 g0, u32* __restrict__ out_perm,
                                                     u32* __restrict__ out_pos)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 id = perm[l0 + i];
    if (out_perm) out_perm[g0 + i] = id;
    if (out_pos) out_pos[id] = g0 + i;
}

template <class T>
int shard_alloc(T*& p, size_t count)
{
    p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) for the rank-local index failed: %s", count * sizeof(T), hipGetErrorString(e));
        return PCPX_ERR_ALLOC;
    }
    return PCPX_OK;
}

// halo width in selection cells: 1.6 x the k-th neighbour distance of a uniform cloud of n points in the grid's box (the k-th
// distance of a Poisson cloud is tightly concentrated: 1.6 x is (1.6)^3 = 4 x the expected volume), at least one cell
u32 halo_cells_for(u64 n, u32 k)
{
    if (n == 0) return 1;
    const double rk = std::cbrt(3.0 * static_cast<double>(k) / (4.0 * 3.14159265358979 * static_cast<double>(n)));
    const double h = std::ceil(1.6 * rk * SEL_AXIS);
    return h < 1.0 ? 1u : h > SEL_AXIS ? SEL_AXIS : static_cast<u32>(h);
}

// selection -> compacted words and records -> local sort -> tree.  The selection bitmap is final when this is called.
int shard_local_build(Index& ix)
{
    Index::Shard& sh = ix.shard;
    hipStream_t s = ix.stream;
    const u64 n = ix.n_in;
    const u32 ntiles = static_cast<u32>((n + SORT_TILE_WORDS - 1) / SORT_TILE_WORDS);
    int st;
    if (ntiles > sh.tile_cap) {
        PCPX_HIP(hipStreamSynchronize(s));
        (void)hipFree(sh.d_tile_cnt);
        sh.tile_cap = 0;
        if ((st = shard_alloc(sh.d_tile_cnt, static_cast<size_t>(ntiles) + 64)) != PCPX_OK) return st;
        sh.tile_cap = static_cast<u64>(ntiles) + 64;
    }
    const u32 cblocks = ntiles < 1024u ? (ntiles ? ntiles : 1u) : 1024u;
    if (n > 0) k_shard_count<<<cblocks, CMP_BLOCK, 0, s>>>(ix.d_codes[0], static_cast<u32>(n), ix.idx_bits, sh.d_sel, ntiles, sh.d_tile_cnt);
    k_shard_scan<<<1, 1024, 0, s>>>(sh.d_tile_cnt, n > 0 ? ntiles : 0u, sh.d_plan);
    u32 plan[P_WORDS];
    PCPX_HIP(hipMemcpyAsync(plan, sh.d_plan, sizeof(plan), hipMemcpyDeviceToHost, s));
    PCPX_HIP(hipStreamSynchronize(s));
    const u32 m = plan[P_M];
    // what describes the shard is published at the end, when the tree exists: a build that fails on the way leaves a handle that
    // answers nothing (n_glob = 0: every slice is empty), not one whose ranges point into a tree that is not there
    const u64 new_n_glob = plan[P_NVALID], new_g_first = plan[P_FIRST], new_g_count = plan[P_COUNT], new_core_g0 = plan[P_G0];
    const bool new_everything = plan[P_NSEL] >= SEL_CELLS;
    sh.n_glob = sh.g_first = sh.g_count = sh.core_g0 = sh.core_l0 = sh.core_count = 0;
    sh.verified_k = 0;
    // local arrays, with some room: a streaming cloud's selection changes a little from rebuild to rebuild
    if (m > sh.cap_loc || !sh.d_words) {
        const u64 cap = static_cast<u64>(m) + m / 8 + 4096;
        PCPX_HIP(hipStreamSynchronize(s));
        (void)hipFree(sh.d_words);
        (void)hipFree(sh.d_tau);
        (void)hipFree(sh.d_fail);
        sh.d_words = nullptr;
        sh.d_tau = nullptr;
        sh.d_fail = nullptr;
        sh.cap_loc = 0;
        if ((st = shard_alloc(sh.d_words, cap)) != PCPX_OK) return st;
        if ((st = shard_alloc(sh.d_tau, cap + GROUP)) != PCPX_OK) return st;
        if ((st = shard_alloc(sh.d_fail, cap + 1)) != PCPX_OK) return st;
        if ((st = build_grow_tree_arrays(ix, cap)) != PCPX_OK) return st;
        sh.cap_loc = cap;
    } else if ((st = build_grow_tree_arrays(ix, m)) != PCPX_OK) {
        return st;
    }
    ix.n_in = n;  // (build_grow_tree_arrays empties the handle when it grows)
    for (int attempt = 0;; ++attempt) {
        if (m > 0) {
            // records travel through the leaf buffer: the sort's first pass has consumed them before the leaf fill writes it
            float4* rec_in = reinterpret_cast<float4*>(ix.d_leaves);
            k_shard_compact<<<cblocks, CMP_BLOCK, 0, s>>>(ix.d_codes[0], sh.cloud, static_cast<u32>(n), ix.idx_bits, sh.d_sel, ntiles, sh.d_tile_cnt,
                                                          sh.d_words, rec_in);
            PCPX_HIP(hipGetLastError());
            if ((st = sort_for_build(ix, sh.d_words, m, nullptr, rec_in, nullptr)) != PCPX_OK) return st;
        } else {
            ix.finish_words = nullptr;
        }
        // (the words are ordered on their top 16 bits at least wherever they are: enough for the core's cells)
        k_shard_locate<<<1, 64, 0, s>>>(ix.finish_words ? ix.finish_words : ix.d_codes[1], sh.d_plan);
        if ((st = build_tree_from_sorted(ix, m, false)) != PCPX_OK) return st;
        u32 sc[10], redo[8];  // (sc[9] = d_scalars[15]: tiles of the sort's lowest pass)
        PCPX_HIP(hipMemcpyAsync(sc, ix.d_scalars + 6, sizeof(sc), hipMemcpyDeviceToHost, s));
        PCPX_HIP(hipMemcpyAsync(redo, ix.d_scalars + BUILD_REDO_WORD0, sizeof(redo), hipMemcpyDeviceToHost, s));
        PCPX_HIP(hipMemcpyAsync(plan, sh.d_plan, sizeof(plan), hipMemcpyDeviceToHost, s));
        PCPX_HIP(hipStreamSynchronize(s));
        std::memcpy(ix.bbox, &sc[2], 6 * sizeof(float));
        if (m > 0 && ix.finish_words) ix.low_pass_tiles = sc[9];
        if (m > 0 && sc[1]) {
            ix.n = 0;
            ix.nleaves = 0;
            set_error("pcpx: internal error, the radix sort's look-back gave up");
            return PCPX_ERR_DEVICE;
        }
        bool again = false;
        for (u32 w = 0; w < 8; ++w) {
            again = again || (redo[w] & ~ix.full_buckets[w]) != 0u;
            ix.full_buckets[w] |= redo[w];
        }
        if (again && attempt == 0 && m > 0) {  // a run the finish kernel could not order: once more, with its bucket taking every pass
            ++ix.build_redos;
            continue;
        }
        break;
    }
    if (plan[P_LCORE] != plan[P_CORE]) {
        ix.n = 0;
        ix.nleaves = 0;
        set_error("pcpx: internal error, the rank-local index holds %u of the core's %u points", plan[P_LCORE], plan[P_CORE]);
        return PCPX_ERR_DEVICE;
    }
    build_tree_set_shape(ix, m);
    sh.n_glob = new_n_glob;
    sh.g_first = new_g_first;
    sh.g_count = new_g_count;
    sh.core_g0 = new_core_g0;
    sh.core_l0 = plan[P_L0];
    sh.core_count = plan[P_LCORE];
    sh.everything = new_everything;
    return PCPX_OK;
}

}  // namespace

void free_shard(Index& ix)
{
    Index::Shard& sh = ix.shard;
    (void)hipFree(sh.d_sel);
    (void)hipFree(sh.d_need);
    (void)hipFree(sh.d_grid);
    (void)hipFree(sh.d_hist12);
    (void)hipFree(sh.d_plan);
    (void)hipFree(sh.d_big);
    (void)hipFree(sh.d_tile_cnt);
    (void)hipFree(sh.d_words);
    (void)hipFree(sh.d_tau);
    (void)hipFree(sh.d_fail);
    sh = Index::Shard{};
}

int shard_unsupported(const Index& ix, const char* what)
{
    (void)ix;
    set_error("%s: not available on a rank-local index (PCPX_BUILD_SHARD): it answers self queries of its own shard only", what);
    return PCPX_ERR_UNSUPPORTED;
}

int build_shard_index(Index& ix, const float* d_xyz_src, u64 n, const pcpx_build_params* params)
{
    Index::Shard& sh = ix.shard;
    const bool explicit_range = (params->flags & PCPX_BUILD_SHARD_RANGE) != 0;
    if (params->struct_size < sizeof(pcpx_build_params) || params->shard_world == 0 || params->shard_rank >= params->shard_world) {
        set_error("pcpx: PCPX_BUILD_SHARD needs shard_rank < shard_world in a full-size pcpx_build_params");
        return PCPX_ERR_INVALID;
    }
    hipStream_t s = ix.stream;
    ProfileScope prof(ix, PCPX_K_BUILD);
    const bool borrowed = (params->flags & PCPX_BUILD_BORROW_CLOUD) != 0;
    int st;
    // a handle that changes its shard, or whose grid is not pinned by the caller, forgets the cells earlier checks asked for
    const bool use_grid = (params->flags & PCPX_BUILD_USE_GRID) != 0;
    bool same_frame = sh.on && use_grid && sh.rank == params->shard_rank && sh.world == params->shard_world && sh.explicit_range == explicit_range &&
                      (!explicit_range || (sh.range_first == params->shard_first && sh.range_count == params->shard_count));
    if (same_frame)
        for (int a = 0; a < 3; ++a) same_frame = same_frame && ix.bbox[a] == params->grid_min[a] && ix.bbox[3 + a] == params->grid_max[a];
    if (!sh.d_sel) {
        if ((st = shard_alloc(sh.d_sel, SEL_WORDS)) != PCPX_OK || (st = shard_alloc(sh.d_need, SEL_WORDS)) != PCPX_OK ||
            (st = shard_alloc(sh.d_grid, 2 * SEL_CELLS / 4)) != PCPX_OK || (st = shard_alloc(sh.d_hist12, CORE_CELLS)) != PCPX_OK ||
            (st = shard_alloc(sh.d_plan, P_WORDS)) != PCPX_OK || (st = shard_alloc(sh.d_big, 1 + static_cast<size_t>(BIG_CAP) * BIG_WORDS)) != PCPX_OK)
            return st;
        same_frame = false;
    }
    sh.on = true;
    sh.n_glob = sh.g_first = sh.g_count = sh.core_g0 = sh.core_l0 = sh.core_count = 0;  // (published by shard_local_build when the tree exists)
    sh.verified_k = 0;
    sh.rank = params->shard_rank;
    sh.world = params->shard_world;
    sh.k_hint = params->shard_k_hint ? params->shard_k_hint : 32u;
    sh.explicit_range = explicit_range;
    sh.range_first = explicit_range ? params->shard_first : 0;
    sh.range_count = explicit_range ? params->shard_count : 0;
    sh.borrowed = borrowed;
    if ((st = build_grow_cloud_arrays(ix, n, !borrowed)) != PCPX_OK) return st;
    ix.n_in = n;
    ix.n = 0;
    ix.nleaves = 0;
    const bool copy_cloud = !borrowed && n > 0 && d_xyz_src != ix.d_xyz;
    if ((st = build_box_and_codes(ix, d_xyz_src, n, params, copy_cloud, nullptr, sh.d_hist12)) != PCPX_OK) return st;
    sh.cloud = borrowed ? d_xyz_src : ix.d_xyz;
    if (!same_frame) PCPX_HIP(hipMemsetAsync(sh.d_need, 0, SEL_WORDS * sizeof(u32), s));
    PCPX_HIP(hipMemsetAsync(sh.d_sel, 0, SEL_WORDS * sizeof(u32), s));
    PCPX_HIP(hipMemsetAsync(sh.d_plan, 0, P_WORDS * sizeof(u32), s));
    k_shard_plan<<<1, 1024, 0, s>>>(sh.d_hist12, sh.rank, sh.world, explicit_range ? 1u : 0u, sh.range_first, sh.range_count, sh.d_plan);
    sh.halo_cells = halo_cells_for(n, sh.k_hint);
    unsigned char* ga = reinterpret_cast<unsigned char*>(sh.d_grid);
    unsigned char* gb = ga + SEL_CELLS;
    const u32 sblocks = SEL_CELLS / 1024;
    k_sel_mark<<<sblocks, 1024, 0, s>>>(sh.d_plan, ga);
    const int h = static_cast<int>(sh.halo_cells);
    k_sel_dilate<<<sblocks, 1024, 0, s>>>(ga, gb, 0, h);
    k_sel_dilate<<<sblocks, 1024, 0, s>>>(gb, ga, 1, h);
    k_sel_dilate<<<sblocks, 1024, 0, s>>>(ga, gb, 2, h);
    k_sel_pack<<<sblocks, 1024, 0, s>>>(gb, sh.d_need, sh.d_sel, sh.d_plan);
    PCPX_HIP(hipGetLastError());
    return shard_local_build(ix);
}

// Self queries of a rank-local index.  [sorted_first, sorted_first + sorted_count) are positions of the WHOLE cloud's order
// and must lie inside the handle's shard.
int shard_knn_self(Index& ix, u64 sorted_first, u64 sorted_count, u32 k, float eps, KnnOutputs o)
{
    Index::Shard& sh = ix.shard;
    hipStream_t s = ix.stream;
    if (sorted_first > sh.n_glob) sorted_first = sh.n_glob;
    const u64 end = sorted_count > sh.n_glob - sorted_first ? sh.n_glob : sorted_first + sorted_count;
    if (end <= sorted_first) return PCPX_OK;
    if (sorted_first < sh.g_first || end > sh.g_first + sh.g_count) {
        set_error("pcpx: positions [%llu, %llu) are not inside this rank-local index's shard [%llu, %llu)", static_cast<unsigned long long>(sorted_first),
                  static_cast<unsigned long long>(end), static_cast<unsigned long long>(sh.g_first), static_cast<unsigned long long>(sh.g_first + sh.g_count));
        return PCPX_ERR_INVALID;
    }
    const u32 bias = static_cast<u32>(sh.core_g0 - sh.core_l0);  // local + bias = global (mod 2^32)
    const u32 a = static_cast<u32>(sorted_first) - bias, b = static_cast<u32>(end) - bias;
    o.pos_lo = a;
    o.pos_hi = b;
    o.pos_bias = bias;
    const bool check = !sh.everything && !(sh.verified_k == k && sh.verified_eps == eps && sh.verified_first == sorted_first && sh.verified_count == end - sorted_first);
    o.tau = check ? sh.d_tau : nullptr;
    const u64 gf = a / GROUP, ge = (static_cast<u64>(b) + GROUP - 1) / GROUP;
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix.n)};
    int st;
    if (check) {
        PCPX_HIP(hipMemsetAsync(sh.d_fail, 0, sizeof(u32), s));
        PCPX_HIP(hipMemsetAsync(sh.d_plan + P_ALL, 0, sizeof(u32), s));
        PCPX_HIP(hipMemsetAsync(sh.d_big, 0, sizeof(u32), s));
    }
    if ((st = launch_knn(ix, qv, true, gf, ge - gf, k, eps, o)) != PCPX_OK) return st;
    if (!check) return PCPX_OK;  // (the same question on the same tree: checked before, only enqueued now)
    const float* d_box = reinterpret_cast<const float*>(ix.d_scalars + 8);
    k_shard_verify<<<(b - a + 255) / 256, 256, 0, s>>>(ix.view(), d_box, sh.d_tau, a, b, sh.d_sel, sh.d_need, sh.d_fail, sh.d_plan, sh.d_big, sh.halo_cells);
    k_shard_verify_big<<<512, 256, 0, s>>>(sh.d_big, sh.d_sel, sh.d_need, sh.d_fail);
    PCPX_HIP(hipGetLastError());
    u32 nfail = 0, all = 0;
    PCPX_HIP(hipMemcpyAsync(&nfail, sh.d_fail, sizeof(u32), hipMemcpyDeviceToHost, s));
    PCPX_HIP(hipMemcpyAsync(&all, sh.d_plan + P_ALL, sizeof(u32), hipMemcpyDeviceToHost, s));
    PCPX_HIP(hipStreamSynchronize(s));
    sh.last_failed = nfail;
    sh.total_failed += nfail;
    if (nfail == 0) {
        sh.verified_k = k;
        sh.verified_eps = eps;
        sh.verified_first = sorted_first;
        sh.verified_count = end - sorted_first;
        return PCPX_OK;
    }
    // the failed queries leave the tree as a batch of arbitrary queries (positions change with the rebuild) ...
    float* d_q = nullptr;
    u32* d_dest = nullptr;
    struct Tmp {
        float*& q;
        u32*& d;
        ~Tmp()
        {
            (void)hipFree(q);
            (void)hipFree(d);
        }
    } tmp{d_q, d_dest};
    if ((st = shard_alloc(d_q, static_cast<size_t>(nfail) * 3)) != PCPX_OK || (st = shard_alloc(d_dest, nfail)) != PCPX_OK) return st;
    k_shard_collect<<<(nfail + 255) / 256, 256, 0, s>>>(ix.view(), sh.d_fail, nfail, o.by_position, bias, d_q, d_dest);
    // ... the selection takes in what they missed and the local tree is rebuilt ...
    if (all) k_fill_u32<<<SEL_WORDS / 1024, 1024, 0, s>>>(sh.d_need, SEL_WORDS, 0xFFFFFFFFu);
    k_sel_or<<<SEL_WORDS / 1024, 1024, 0, s>>>(sh.d_sel, sh.d_need, sh.d_plan);
    PCPX_HIP(hipGetLastError());
    ++sh.enlargements;
    {
        ProfileScope prof(ix, PCPX_K_BUILD);
        if ((st = shard_local_build(ix)) != PCPX_OK) return st;
    }
    // ... and they are answered again, their rows going where the failed rows were: their balls are covered now
    QueryView fq;
    if ((st = prepare_queries(ix, d_q, nfail, fq)) != PCPX_OK) return st;
    k_shard_remap_rows<<<(nfail + 255) / 256, 256, 0, s>>>(const_cast<u32*>(fq.row), nfail, d_dest);
    PCPX_HIP(hipGetLastError());
    KnnOutputs again = o;
    again.tau = nullptr;
    again.by_position = 0;
    again.pos_lo = 0;
    again.pos_hi = 0xFFFFFFFFu;
    again.pos_bias = 0;
    if ((st = launch_knn(ix, fq, false, 0, (static_cast<u64>(nfail) + GROUP - 1) / GROUP, k, eps, again)) != PCPX_OK) return st;
    PCPX_HIP(hipStreamSynchronize(s));
    sh.verified_k = k;
    sh.verified_eps = eps;
    sh.verified_first = sorted_first;
    sh.verified_count = end - sorted_first;
    return PCPX_OK;
}

// Radius counts around the shard's own points: the halo must reach `radius` from every core cell; if it does not yet, the
// selection is dilated by what is missing first.
int shard_range_count_self(Index& ix, float radius, u64 sorted_first, u64 sorted_count, u32* d_out_cnt, bool by_position)
{
    Index::Shard& sh = ix.shard;
    hipStream_t s = ix.stream;
    if (sorted_first > sh.n_glob) sorted_first = sh.n_glob;
    const u64 end = sorted_count > sh.n_glob - sorted_first ? sh.n_glob : sorted_first + sorted_count;
    if (end <= sorted_first) return PCPX_OK;
    if (sorted_first < sh.g_first || end > sh.g_first + sh.g_count) {
        set_error("pcpx: positions [%llu, %llu) are not inside this rank-local index's shard [%llu, %llu)", static_cast<unsigned long long>(sorted_first),
                  static_cast<unsigned long long>(end), static_cast<unsigned long long>(sh.g_first), static_cast<unsigned long long>(sh.g_first + sh.g_count));
        return PCPX_ERR_INVALID;
    }
    int st;
    if (!sh.everything && radius >= 0.f) {
        // cells the ball of a core point can reach: ceil(radius / narrowest cell side) + 1 (the point may sit at its cell's far side)
        double side = std::numeric_limits<double>::infinity();
        for (int a = 0; a < 3; ++a) {
            const double ext = static_cast<double>(ix.bbox[3 + a]) - ix.bbox[a];
            if (ext > 0.0) side = std::min(side, ext / SEL_AXIS);
        }
        const double want = std::isfinite(side) ? std::ceil(static_cast<double>(radius) * 1.00001 / side) + 1.0 : 1.0;
        const u32 h = want > SEL_AXIS ? SEL_AXIS : static_cast<u32>(want);
        if (h > sh.halo_cells) {
            unsigned char* ga = reinterpret_cast<unsigned char*>(sh.d_grid);
            unsigned char* gb = ga + SEL_CELLS;
            const u32 sblocks = SEL_CELLS / 1024;
            k_sel_mark<<<sblocks, 1024, 0, s>>>(sh.d_plan, ga);
            k_sel_dilate<<<sblocks, 1024, 0, s>>>(ga, gb, 0, static_cast<int>(h));
            k_sel_dilate<<<sblocks, 1024, 0, s>>>(gb, ga, 1, static_cast<int>(h));
            k_sel_dilate<<<sblocks, 1024, 0, s>>>(ga, gb, 2, static_cast<int>(h));
            k_sel_pack<<<sblocks, 1024, 0, s>>>(gb, sh.d_need, sh.d_sel, sh.d_plan);
            PCPX_HIP(hipGetLastError());
            sh.halo_cells = h;
            ++sh.enlargements;
            ProfileScope prof(ix, PCPX_K_BUILD);
            if ((st = shard_local_build(ix)) != PCPX_OK) return st;
        }
    }
    const u32 bias = static_cast<u32>(sh.core_g0 - sh.core_l0);
    const u32 a = static_cast<u32>(sorted_first) - bias, b = static_cast<u32>(end) - bias;
    const u64 gf = a / GROUP, ge = (static_cast<u64>(b) + GROUP - 1) / GROUP;
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix.n)};
    qv.pos_lo = a;
    qv.pos_hi = b;
    qv.by_position = by_position ? 1u : 0u;
    qv.pos_bias = bias;  // local position + bias = position in the whole cloud's order
    return launch_range_count(ix, qv, true, gf, ge - gf, radius, nullptr, d_out_cnt);
}

int shard_perm(Index& ix, u32* d_out_perm, u32* d_out_pos)
{
    Index::Shard& sh = ix.shard;
    if (sh.core_count == 0) return PCPX_OK;
    // (only the shard itself: the core's cells may hold a few positions of the neighbouring shards)
    const u64 l0 = sh.core_l0 + (sh.g_first - sh.core_g0);
    k_shard_perm<<<static_cast<u32>((sh.g_count + 255) / 256), 256, 0, ix.stream>>>(ix.d_perm, static_cast<u32>(l0), static_cast<u32>(sh.g_count),
                                                                                   static_cast<u32>(sh.g_first), d_out_perm, d_out_pos);
    return check_hip(hipGetLastError(), "k_shard_perm launch", __FILE__, __LINE__);
}

}  // namespace pcpx
