// pcpx_build.hip -- index construction on the GPU.
//
// Replaces the reference's sequential octree insertion (include/pcp/octree/linked_octree_node.hpp:143-331)
// and recursive nth_element kd-tree build (include/pcp/kdtree/linked_kdtree.hpp:343-424).  Only query
// RESULTS are observable through the reference API, not the tree shape, so the structure here is a
// curve-sorted implicit AABB tree (pcpx_internal.h, pcpx_curve.h) built by:
//   bbox reduce -> one sort word per point: 39-bit Hilbert key + index (+ out-of-grid drop) -> radix sort, whose first pass
//   also moves a {x, y, z, id} record per point into its top-digit bucket (pcpx_sort.hip) -> leaf records (SoA, NaN padded),
//   leaf AABBs and three tree levels in one pass -> the remaining levels, five per launch.
#include "pcpx_curve.h"

#include <cmath>
#include <limits>

namespace pcpx {

namespace {

#ifndef PCPX_BUILD_ADAPTIVE_MARGIN
#define PCPX_BUILD_ADAPTIVE_MARGIN 6  // PCPX_BUILD_COARSE_ORDER: bits of curve resolution kept beyond log2 of a top-digit bucket's size
#endif
#ifndef PCPX_BUILD_FINISH
#define PCPX_BUILD_FINISH 1  // 1: the sort stops after two bucketed passes where that leaves short runs and k_finish orders them in LDS (round 5); 0: every pass
#endif
#ifndef PCPX_BUILD_RECORDS
#define PCPX_BUILD_RECORDS 1  // 1: the sort's first pass moves a {x, y, z, id} record per point into its top-digit bucket and the leaf
                              // fill gathers from there; 0: the leaf fill gathers the coordinates from the input-order copy
#endif
constexpr size_t SCALARS = 32;  // u32 words of Index::d_scalars: [0, 6) encoded box, [6] points outside the grid, [7] sort failure, [8, 14) box,
                                //   [16, 24) buckets whose runs the finish kernel could not order (REDO_WORD0), [24, 32) the buckets to sort fully (FORCE_WORD0)
constexpr u32 REDO_WORD0 = 16, FORCE_WORD0 = 24, LOW_TILES_WORD = 15;  // ([15]: tiles of the sort's lowest pass, see sort_for_build)

// order-preserving float <-> uint encoding for atomic min/max
__device__ __forceinline__ u32 enc_f(float f)
{
    u32 b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f(u32 e)
{
    u32 b = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
    return __uint_as_float(b);
}

// pcp::bounding_box (include/pcp/common/axis_aligned_bounding_box.hpp:214-251): min/max per axis
// from +-FLT_MAX with strict comparisons (NaN coordinates never update).
__global__ void k_bbox_init(u32* enc6, u32* counter)
{
    if (threadIdx.x < 3) enc6[threadIdx.x] = enc_f(std::numeric_limits<float>::max());
    else if (threadIdx.x < 6) enc6[threadIdx.x] = enc_f(std::numeric_limits<float>::lowest());
    else if (threadIdx.x == 6) *counter = 0u;
}

// Streaming min/max: the cloud is read as 16-byte vectors.  Four points are twelve floats = three float4, so after
// peeling `head` points (which makes the address 16-byte aligned: 3 is invertible modulo 4) thread t owns point quads
// t, t + stride, ... and every load instruction of a wave covers a contiguous run of memory (three interleaved 48-byte
// strides).  Block-level reduction in LDS, then ONE set of six atomics per block (round 1: stride-3 scalar loads and six
// atomics per wave -- 576 us for 10 M points, 2.6 % of the HBM rate).
__device__ __forceinline__ void bbox_point(float (&mn)[3], float (&mx)[3], float x, float y, float z)
{
    if (x < mn[0]) mn[0] = x;
    if (y < mn[1]) mn[1] = y;
    if (z < mn[2]) mn[2] = z;
    if (x > mx[0]) mx[0] = x;
    if (y > mx[1]) mx[1] = y;
    if (z > mx[2]) mx[2] = z;
}

__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ xyz, u64 n, u32* enc6)
{
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                   std::numeric_limits<float>::max()};
    float mx[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
                   std::numeric_limits<float>::lowest()};
    const u64 word = reinterpret_cast<uintptr_t>(xyz) >> 2;          // the array is 4-byte aligned
    u64 head = ((4u - (word & 3u)) * 3u) & 3u;                       // points before the first 16-byte aligned point
    if (head > n) head = n;
    const u64 nquad = (n - head) >> 2;
    const float4* v = reinterpret_cast<const float4*>(xyz + 3 * head);
    const u64 gtid = blockIdx.x * static_cast<u64>(blockDim.x) + threadIdx.x;
    const u64 gstride = static_cast<u64>(gridDim.x) * blockDim.x;
    // two quads per trip: six 16-byte loads in flight per lane (with three the launch was pure load latency: 90 % of the waves'
    // time waiting at 16 waves per CU, profiles/r03_pmc_rebuild.json)
    for (u64 q = gtid; q < nquad; q += 2 * gstride) {
        const u64 q2 = q + gstride < nquad ? q + gstride : q;  // (the last trip may repeat its quad: min/max do not mind)
        const float4 a = v[3 * q], b = v[3 * q + 1], c = v[3 * q + 2];
        const float4 d = v[3 * q2], e = v[3 * q2 + 1], f = v[3 * q2 + 2];
        bbox_point(mn, mx, a.x, a.y, a.z);
        bbox_point(mn, mx, a.w, b.x, b.y);
        bbox_point(mn, mx, b.z, b.w, c.x);
        bbox_point(mn, mx, c.y, c.z, c.w);
        bbox_point(mn, mx, d.x, d.y, d.z);
        bbox_point(mn, mx, d.w, e.x, e.y);
        bbox_point(mn, mx, e.z, e.w, f.x);
        bbox_point(mn, mx, f.y, f.z, f.w);
    }
    // the peeled points and the tail (fewer than 8 in all): the first threads of the grid take one each
    const u64 tail0 = head + 4 * nquad;
    const u64 loose = head + (n - tail0);
    if (gtid < loose) {
        const u64 i = gtid < head ? gtid : tail0 + (gtid - head);
        bbox_point(mn, mx, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    }
    __shared__ float red[4][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    const u32 w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[w][a] = mn[a];
            red[w][3 + a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = static_cast<int>(threadIdx.x);
        float r = red[0][a];
        for (u32 i = 1; i < (blockDim.x >> 6); ++i) r = a < 3 ? fminf(r, red[i][a]) : fmaxf(r, red[i][a]);
        if (a < 3) atomicMin(&enc6[a], enc_f(r));
        else atomicMax(&enc6[a], enc_f(r));
    }
}

__global__ void k_bbox_decode(const u32* enc6, float* out6)
{
    if (threadIdx.x < 6) out6[threadIdx.x] = dec_f(enc6[threadIdx.x]);
}

// First launch of a build: the encoded box starts at (+max, -max), the counters of Index::d_scalars at zero.
__global__ void k_build_begin(u32* __restrict__ scalars)
{
    const u32 t = threadIdx.x;
    if (t < 3) scalars[t] = enc_f(std::numeric_limits<float>::max());
    else if (t < 6) scalars[t] = enc_f(std::numeric_limits<float>::lowest());
    else if (t < 8) scalars[t] = 0u;  // [6] points outside the grid, [7] the sort's failure flag
    else if (t >= REDO_WORD0 && t < REDO_WORD0 + 8u) scalars[t] = 0u;
}

// Sort word (pcpx_curve.h) of every point: curve key of a point inside the grid, the all-ones key for a point outside it
// (or NaN) -- it sorts to the end: the "silently not inserted" rule of linked_octree_node.hpp:174-175 (inclusive
// containment, include/pcp/common/axis_aligned_bounding_box.hpp:111-125).  The same sweep over the coordinates
//   * keeps the index's own copy of the cloud (the reference's containers copy their elements too, linked_kdtree.hpp:107;
//     round 2 spent a separate 120 MB device copy on it),
//   * counts the words' top digit per tile of the sort's first pass (round 2: a separate sweep over the words), which is what lets
//     that pass run without a look-back chain (pcpx_sort.hip),
//   * counts the points outside the grid (round 2: a one-thread binary search kernel after the sort),
//   * decodes the bounding box the reduction left in its order-preserving integer form (round 2: a launch of its own).
// scalars: Index::d_scalars.
constexpr int CODES_BLOCK = 1024;
__global__ __launch_bounds__(CODES_BLOCK) void k_codes(const float* __restrict__ xyz, u64 n, u32* __restrict__ scalars, bool decode_box, int idx_bits,
                                                        u64* __restrict__ codes, float* __restrict__ xyz_copy, u32* __restrict__ tile_hist,
                                                        u32* __restrict__ hist12)
{
    __shared__ u32 hist[256];
    __shared__ u32 htab[HILBERT_TABLE_WORDS];
    __shared__ u32 cells[4096];  // hist12 (a rank-local build): points per level-4 cell of the curve, this block's share
    if (hist12) {
        for (u32 c = threadIdx.x; c < 4096u; c += CODES_BLOCK) cells[c] = 0;
    }
    float* box6 = reinterpret_cast<float*>(scalars + 8);
    float b[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) b[a] = decode_box ? dec_f(scalars[a]) : box6[a];
    if (decode_box && blockIdx.x == 0 && threadIdx.x < 6) box6[threadIdx.x] = dec_f(scalars[threadIdx.x]);
    hilbert_table_to_lds(htab);
    const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5];
    const CurveGrid grid = curve_grid(b0, b1, b2, b3, b4, b5);
    u32 outside = 0;
    // a block works through whole tiles of the sort (SORT_TILE_WORDS consecutive points), so that the counts it leaves behind are
    // the sort's per-tile counts of the top digit
    const u64 ntiles = (n + SORT_TILE_WORDS - 1) / SORT_TILE_WORDS;
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < SORT_TILE_WORDS / CODES_BLOCK; ++it) {
            const u64 i = tile * SORT_TILE_WORDS + static_cast<u64>(it) * CODES_BLOCK + threadIdx.x;
            if (i < n) {
                const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
                if (xyz_copy) {
                    xyz_copy[3 * i] = x;
                    xyz_copy[3 * i + 1] = y;
                    xyz_copy[3 * i + 2] = z;
                }
                const bool ok = (x >= b0 && y >= b1 && z >= b2) && (x <= b3 && y <= b4 && z <= b5);
                const u64 word = ok ? sort_word(curve_key_inside(x, y, z, grid, htab, idx_bits), i, idx_bits) : outside_word(i, idx_bits);
                codes[i] = word;
                outside += ok ? 0u : 1u;
                if (tile_hist) atomicAdd(&hist[static_cast<u32>(word >> 56)], 1u);
                if (hist12 && ok) atomicAdd(&cells[static_cast<u32>(word >> 52)], 1u);
            }
        }
        __syncthreads();
        if (tile_hist && threadIdx.x < 256) tile_hist[tile * 256 + threadIdx.x] = hist[threadIdx.x];
        __syncthreads();
    }
    if (hist12) {
        for (u32 c = threadIdx.x; c < 4096u; c += CODES_BLOCK) {
            const u32 v = cells[c];
            if (v) atomicAdd(&hist12[c], v);
        }
    }
    const u64 some_outside = __builtin_amdgcn_ballot_w64(outside != 0u);
    if (some_outside) {  // rare: one atomic per wave that saw any
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) outside += __shfl_xor(outside, off);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&scalars[6], outside);
    }
}

__device__ __forceinline__ NodeBox padding_node()
{
    NodeBox nb;
    nb.set(0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    nb.poison = __builtin_nanf("");
    nb.pad = 0.f;
    return nb;
}

// min / max over the 8 lanes of a leaf (aligned groups of 8 lanes): two quad permutes and a half-row mirror
__device__ __forceinline__ float leaf_min(float v)
{
    v = fminf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
    v = fminf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
    v = fminf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
    return v;
}
__device__ __forceinline__ float leaf_max(float v)
{
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));
    return v;
}

// ... and over the UNIT_POINTS lanes of a unit (two leaves: one more step, the other half of the row of sixteen)
__device__ __forceinline__ float unit_min(float v)
{
    v = leaf_min(v);
    if (UNIT_LEAVES == 2) v = fminf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x128, 0xF, 0xF, true)));  // row_ror:8
    return v;
}
__device__ __forceinline__ float unit_max(float v)
{
    v = leaf_max(v);
    if (UNIT_LEAVES == 2) v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x128, 0xF, 0xF, true)));
    return v;
}

// node i of a level = union of its 4 children (padding children are skipped; no real child: a padding node)
__device__ __forceinline__ NodeBox union_of_children(const NodeBox* __restrict__ child4)
{
    float inf = std::numeric_limits<float>::infinity();
    float b[6] = {inf, inf, inf, -inf, -inf, -inf};
    bool any = false;
#pragma unroll
    for (int c = 0; c < W; ++c) {
        NodeBox cb = child4[c];
        if (cb.poison == 0.f) {
            any = true;
            b[0] = fminf(b[0], cb.lo(0));
            b[1] = fminf(b[1], cb.lo(1));
            b[2] = fminf(b[2], cb.lo(2));
            b[3] = fmaxf(b[3], cb.hi(0));
            b[4] = fmaxf(b[4], cb.hi(1));
            b[5] = fmaxf(b[5], cb.hi(2));
        }
    }
    NodeBox nb;
    nb.set(any ? b[0] : 0.f, any ? b[1] : 0.f, any ? b[2] : 0.f, any ? b[3] : 0.f, any ? b[4] : 0.f, any ? b[5] : 0.f);
    nb.poison = any ? 0.f : __builtin_nanf("");
    nb.pad = 0.f;
    return nb;
}

// Shape of the implicit tree (heap order; only REAL nodes -- and the up to three padding siblings that complete the last
// group of four of a level, the only padding a query can read -- are ever written).  real(d) = ceil(nleaves / 4^(depth - d)).
struct TreeShape {
    u32 nleaves;
    int depth;
    __host__ __device__ u32 nreal(int d) const
    {
        const int sh = 2 * (depth - d);
        return static_cast<u32>((static_cast<u64>(nleaves) + (1ull << sh) - 1ull) >> sh);
    }
    __host__ __device__ u32 nwrite(int d) const { return d == 0 ? 1u : (nreal(d) + 3u) & ~3u; }
    __host__ __device__ static u64 level_start(int d) { return ((1ull << (2 * d)) - 1ull) / 3ull; }
};

// Leaf records (16 B per point, coalesced) from the sorted words -- a word's low bits name the point's 16-byte record
// {x, y, z, id}, which the sort's first pass left in the point's top-digit bucket: the gather of a block stays inside an
// L2-sized window (round 2 gathered 12-byte coordinates from all over the cloud: 686 MB fetched for 120 MB) -- plus, in
// the same pass, the sorted position -> input index table, the leaf's tight box (min / max over the leaf's 8 lanes) and
// the three tree levels above the leaves (a block owns 128 leaves = two level-(depth - 3) nodes).  A thread takes four
// points, 256 positions apart, so that four gathers are in flight per lane.
// Blocks are dealt so that each XCD works through one contiguous eighth of the sorted order.
constexpr int FILL_BLOCK = 256;
#ifndef PCPX_FILL_PER_THREAD
#define PCPX_FILL_PER_THREAD 4
#endif
constexpr int FILL_PER_THREAD = PCPX_FILL_PER_THREAD;
constexpr int FILL_SLOTS = FILL_BLOCK * FILL_PER_THREAD;  // 1024
constexpr int FILL_LEAVES = FILL_SLOTS / LEAF;            // 128
constexpr int FILL_UNITS = FILL_LEAVES / UNIT_LEAVES;     // bottom-level boxes a block makes
static_assert(FILL_UNITS >= 64, "a block makes the three levels above its units");

// The tree's shape from the number of inserted points, on the device: the kernels below read that number where the build left it
// (Index::d_scalars: input points minus the points outside the grid), so the host need not wait for it between the sort and them.
struct BuildCount {
    const u32* scalars;  // Index::d_scalars, or nullptr: `known` is the number (a rank-local build: the host has it)
    u32 n_in, known;
    __device__ u32 valid() const { return scalars ? n_in - scalars[6] : known; }
};
// (never 1: a tree of two to four leaves gets a level of one real node between the root and them, so that a walk meets leaves only
//  under a last-level node -- k_knn's loop has no case for a leaf popped from the pending bits, which cost every pop a dozen scalar
//  instructions for a shape only clouds of up to 32 points have)
__host__ __device__ inline int depth_of(u32 nleaves)
{
    int d = 0;
    while ((1ull << (2 * d)) < nleaves) ++d;
    return d == 1 ? 2 : d;
}
// (TreeShape::nleaves is the number of nodes of the bottom level: UNITS of UNIT_LEAVES leaves)
__host__ __device__ inline TreeShape shape_of(u32 nvalid)
{
    const u32 nunits = (nvalid + UNIT_POINTS - 1) / UNIT_POINTS;
    return TreeShape{nunits, depth_of(nunits)};
}
// blocks of k_finish a tree of this shape needs: its leaf slots, and the (padding) nodes of the three levels above that a block owns
__host__ __device__ inline u32 finish_blocks(const TreeShape& ts)
{
    if (ts.nleaves == 0) return 0;
    const u32 nslots = ts.nwrite(ts.depth) * UNIT_POINTS;
    u32 fblocks = (nslots + FILL_SLOTS - 1) / FILL_SLOTS;
    for (int j = 1; j <= 3 && j <= ts.depth; ++j) {  // a block owns FILL_UNITS >> 2j nodes of level depth - j
        const u32 per_block = static_cast<u32>(FILL_UNITS) >> (2 * j);
        const u32 need = (ts.nwrite(ts.depth - j) + per_block - 1) / per_block;
        if (need > fblocks) fblocks = need;
    }
    return fblocks;
}

// k_finish = the leaf fill, and in front of it the LAST STEP OF THE SORT (round 5).  The radix sort now stops early where it can
// (SortPayload::finish): a bucket whose words are few per 24-bit cell is sorted on bits [40, 64) only, which leaves RUNS of words
// that agree on those bits -- a handful each -- in arbitrary... no: in input order.  A block takes the FILL_SLOTS words of its tile
// and FIN_MARGIN on either side into LDS, finds the runs (neighbours that agree above their bucket's shift), and every word counts
// the words of its run that are smaller: that is its place.  Two global radix passes (16 B/point each) and their look-back chains
// become ~50 LDS reads per word.  A run longer than FIN_MARGIN -- a cell far denser than its bucket's fullest 16-bit cell
// suggested -- cannot be ordered here: the block notes the bucket in `redo` and the host repeats the sort with that bucket taking
// every pass (the handle remembers it).  Buckets that took every pass have nothing to order: their runs are ties on all sorted
// bits, which the stable passes left in input order.  The ordered words themselves are not written back: what searches the sorted
// words afterwards (prepare_queries' seeds, the rank-local build's core) needs them ordered on their top 24 bits only, which the
// prefix-sorted words are (Index::sorted_codes).
constexpr int FIN_MARGIN = 64;
constexpr int FIN_WINDOW = FILL_SLOTS + 2 * FIN_MARGIN;  // words of the window; + one word on either side to see whether its ends start runs
struct FinishArgs {
    const u64* words;              // prefix-sorted words, or nullptr: `sorted` holds fully sorted words (nothing to order)
    const u64* sorted;
    const u32* bucket_first_pass;  // per top digit: the first bucketed pass the bucket took (1: all of them)
    int first_bit;                 // lowest sorted bit of a bucket that took every pass
    u32* redo;                     // bitmap, 8 words
};

__global__ __launch_bounds__(FILL_BLOCK) void k_finish(const float4* __restrict__ rec, const float* __restrict__ xyz, FinishArgs fa, int idx_bits,
                                                       BuildCount bc, Leaf* __restrict__ leaves, u32* __restrict__ perm, NodeBox* __restrict__ nodes)
{
    __shared__ __attribute__((aligned(16))) NodeBox lvl0[FILL_UNITS];      // unit boxes of the block
    __shared__ __attribute__((aligned(16))) NodeBox lvl1[FILL_UNITS / 4];
    __shared__ __attribute__((aligned(16))) NodeBox lvl2[FILL_UNITS / 16];
    // (the window of words and its flags are dead when the leaf records are staged: one piece of LDS for both -- 21 KB per block, seven
    //  blocks per CU, as before the window was there)
    union alignas(16) Piece {
        u32 rec_stage[FILL_BLOCK / 64][FILL_PER_THREAD][64 * 4];  // per wave and trip: 8 leaf records
        struct {
            u64 cov[FIN_WINDOW + 2];                 // cov[1 + s] = word at position lo - FIN_MARGIN + s
            unsigned char starts[FIN_WINDOW + 2];    // starts[j]: a run starts at cov[j] (or cov[j] is no word at all)
            unsigned char shift_of[256];             // per bucket: the words are ordered on bits [shift_of, 64)
            unsigned short place[FIN_WINDOW + 2];    // where cov[j] belongs
        } window;
    };
    __shared__ Piece piece;
    __shared__ u32 any_partial_s;
    auto& rec_stage = piece.rec_stage;
    auto& cov = piece.window.cov;
    auto& starts = piece.window.starts;
    auto& shift_of = piece.window.shift_of;
    auto& place = piece.window.place;
    const u32 n = bc.valid();
    const TreeShape ts = shape_of(n);
    const u32 nleaves_real = (n + LEAF - 1) / LEAF;
    const u32 nblocks = finish_blocks(ts);
    const u32 per = gridDim.x >> 3;  // the grid is a multiple of 8
    const u32 vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (vb >= nblocks) return;  // (the grid is sized for the input's point count; a block may exist only for the padding nodes it owns)
    const u32 p0 = vb * FILL_SLOTS + threadIdx.x;
    const u64 low = (1ull << idx_bits) - 1ull;
    u64 at[FILL_PER_THREAD];
    if (fa.words) {
        constexpr u64 NO_WORD = ~0ull;
        shift_of[threadIdx.x] = static_cast<unsigned char>(fa.first_bit + 8 * (static_cast<int>(fa.bucket_first_pass[threadIdx.x]) - 1));
        if (threadIdx.x == 0) any_partial_s = 0;
        const long long base = static_cast<long long>(vb) * FILL_SLOTS - FIN_MARGIN - 1;  // position of cov[0]
        for (u32 j = threadIdx.x; j < FIN_WINDOW + 2; j += FILL_BLOCK) {
            const long long g = base + j;
            cov[j] = (g >= 0 && g < static_cast<long long>(n)) ? fa.words[g] : NO_WORD;
        }
        __syncthreads();
        for (u32 j = threadIdx.x; j < FIN_WINDOW + 2; j += FILL_BLOCK) {
            const long long g = base + j;
            const bool word_here = g >= 0 && g < static_cast<long long>(n);
            bool st = true;  // (no word, or the first word of the order: a run cannot reach across)
            if (word_here && j > 0 && g > 0) {
                const u64 w = cov[j], before = cov[j - 1];
                const int sh = shift_of[static_cast<u32>(w >> 56)];
                st = (w >> sh) != (before >> sh);
                if (sh > fa.first_bit && !st) any_partial_s = 1;  // (a run of two or more in a bucket that stopped early: something to order)
            }
            starts[j] = st ? 1 : 0;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(any_partial_s) != 0u) {  // (block-uniform; clouds whose buckets all took every pass, and most tiles of sparse ones, skip this)
        // every word of the window: where its run starts and ends (if both are in sight), and how many of the run are smaller -- its
        // place, noted in LDS (one copy of the two scans: unrolled over a thread's words they cost 300 saved lane masks)
        constexpr int PER = (FIN_WINDOW + FILL_BLOCK - 1) / FILL_BLOCK;
        u32 redo_bucket = ~0u;
#pragma unroll 1
        for (int u = 0; u < PER; ++u) {
            const u32 j = 1u + threadIdx.x + u * FILL_BLOCK;  // cov[1 ... FIN_WINDOW] are the window
            if (j > FIN_WINDOW) break;
            place[j] = static_cast<unsigned short>(j);
            const u64 w = cov[j];
            const long long g = base + j;
            if (!(g >= 0 && g < static_cast<long long>(n))) continue;
            const u32 bucket = static_cast<u32>(w >> 56);
            if (shift_of[bucket] <= fa.first_bit) continue;  // took every pass: in order already
            const bool in_tile = j > FIN_MARGIN && j <= FIN_MARGIN + FILL_SLOTS;
            u32 a = j, smaller = 0, steps = 0;
            bool whole = true;
#pragma nounroll  // (bounded by FIN_MARGIN, hipcc would write all 64 trips out, each with a saved lane mask of its own)
            while (!starts[a]) {  // (starts[0] is always set: reaching it means the run began out of sight)
                if (++steps > FIN_MARGIN) {
                    whole = false;
                    break;
                }
                --a;
                smaller += cov[a] < w ? 1u : 0u;
            }
            if (a == 0u) whole = false;
            u32 b = j + 1u;
#pragma nounroll
            while (whole && !starts[b]) {  // (b <= FIN_WINDOW + 1 whenever this is evaluated)
                if (b == FIN_WINDOW + 1u || ++steps > FIN_MARGIN) {
                    whole = false;
                    break;
                }
                smaller += cov[b] < w ? 1u : 0u;
                ++b;
            }
            if (whole) place[j] = static_cast<unsigned short>(a + smaller);
            else if (in_tile) redo_bucket = bucket;
        }
        if (redo_bucket != ~0u) atomicOr(&fa.redo[redo_bucket >> 5], 1u << (redo_bucket & 31u));
        __syncthreads();
        u64 mine[PER];
        u32 dest[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const u32 j = 1u + threadIdx.x + u * FILL_BLOCK;
            const u32 jj = j <= FIN_WINDOW ? j : FIN_WINDOW;
            mine[u] = cov[jj];
            dest[u] = place[jj];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (1u + threadIdx.x + u * FILL_BLOCK <= FIN_WINDOW) cov[dest[u]] = mine[u];
        __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < FILL_PER_THREAD; ++u) {
            const u32 p = p0 + u * FILL_BLOCK;
            const u64 w = cov[1 + FIN_MARGIN + threadIdx.x + u * FILL_BLOCK];
            at[u] = p < n ? (w & low) : 0ull;  // (a padding slot gathers record 0 and writes nothing)
        }
        __syncthreads();  // (the window is dead: its LDS is the leaf records' stage from here on)
    } else {
#pragma unroll
        for (int u = 0; u < FILL_PER_THREAD; ++u) {
            const u32 p = p0 + u * FILL_BLOCK;
            at[u] = p < n ? (fa.sorted[p] & low) : 0ull;
        }
    }
    float x[FILL_PER_THREAD], y[FILL_PER_THREAD], z[FILL_PER_THREAD];
    u32 id[FILL_PER_THREAD];
#pragma unroll
    for (int u = 0; u < FILL_PER_THREAD; ++u) {
        if (rec) {
            const float4 r = rec[at[u]];
            x[u] = r.x;
            y[u] = r.y;
            z[u] = r.z;
            id[u] = __float_as_uint(r.w);
        } else {
            id[u] = static_cast<u32>(at[u]);
            x[u] = xyz[3 * at[u]];
            y[u] = xyz[3 * at[u] + 1];
            z[u] = xyz[3 * at[u] + 2];
        }
    }
    const float inf = std::numeric_limits<float>::infinity(), nan = __builtin_nanf("");
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < FILL_PER_THREAD; ++u) {
        const u32 p = p0 + u * FILL_BLOCK;
        const bool live = p < n;
        if (live) perm[p] = id[u];
        // The wave's 64 points are 8 whole leaf records = 1 KB of contiguous memory.  Written field by field straight from
        // the registers every store instruction touches eight 32-byte pieces of eight different lines (that is what the
        // kernel was waiting on: profiles/r03_pmc_rebuild.json); through LDS every lane stores 16 contiguous bytes and the
        // wave one contiguous kilobyte.  (Wave-private LDS region: no barrier, a wave's LDS operations execute in order.)
        u32* stage = rec_stage[w][u];
        const u32 ls = (lane >> 3) * 32u + (lane & 7u);  // slot s of leaf (lane >> 3) of the wave: x at +0, y +8, z +16, id +24
        stage[ls] = __float_as_uint(live ? x[u] : nan);
        stage[ls + 8] = __float_as_uint(live ? y[u] : nan);
        stage[ls + 16] = __float_as_uint(live ? z[u] : nan);
        stage[ls + 24] = live ? id[u] : INVALID_ID;
        __builtin_amdgcn_wave_barrier();  // (scheduling fence only: the reads below see the other lanes' writes)
        const u32 wave_leaf0 = (p - lane) / LEAF;  // first leaf of the wave's 64 points
        const uint4 piece = *reinterpret_cast<const uint4*>(stage + 4 * lane);
        if (wave_leaf0 + (lane >> 3) < nleaves_real) reinterpret_cast<uint4*>(leaves + wave_leaf0)[lane] = piece;
        // (every lane of the group of UNIT_POINTS lanes takes part in the reduction, whatever it holds)
        const float lx = unit_min(live ? x[u] : inf), ly = unit_min(live ? y[u] : inf), lz = unit_min(live ? z[u] : inf);
        const float hx = unit_max(live ? x[u] : -inf), hy = unit_max(live ? y[u] : -inf), hz = unit_max(live ? z[u] : -inf);
        if ((p % UNIT_POINTS) == 0) {
            NodeBox nb = padding_node();
            if (p / UNIT_POINTS < ts.nleaves) {
                nb.set(lx, ly, lz, hx, hy, hz);
                nb.poison = 0.f;
            }
            lvl0[(threadIdx.x + u * FILL_BLOCK) / UNIT_POINTS] = nb;
        }
    }
    // the block's leaf boxes, then the three levels above (32, 8 and 2 nodes per 128 leaves): every level leaves LDS as
    // 16-byte pieces of one contiguous run of the heap
    const u32 l0 = vb * FILL_UNITS;  // the block's first unit
    auto store_level = [&](const NodeBox* boxes, u32 count, int level, u32 first) {
        const u32 nw = ts.nwrite(level);
        uint4* dst = reinterpret_cast<uint4*>(nodes + TreeShape::level_start(level) + first);
        for (u32 t = threadIdx.x; t < 2 * count; t += FILL_BLOCK)
            if (first + (t >> 1) < nw) dst[t] = reinterpret_cast<const uint4*>(boxes)[t];
    };
    __syncthreads();
    store_level(lvl0, FILL_UNITS, ts.depth, l0);
    if (ts.depth >= 1 && threadIdx.x < FILL_UNITS / 4) {
        const u32 i = (l0 >> 2) + threadIdx.x;
        lvl1[threadIdx.x] = i < ts.nreal(ts.depth - 1) ? union_of_children(lvl0 + 4 * threadIdx.x) : padding_node();
    }
    __syncthreads();
    if (ts.depth >= 1) store_level(lvl1, FILL_UNITS / 4, ts.depth - 1, l0 >> 2);
    if (ts.depth >= 2 && threadIdx.x < FILL_UNITS / 16) {
        const u32 i = (l0 >> 4) + threadIdx.x;
        lvl2[threadIdx.x] = i < ts.nreal(ts.depth - 2) ? union_of_children(lvl1 + 4 * threadIdx.x) : padding_node();
    }
    __syncthreads();
    if (ts.depth >= 2) store_level(lvl2, FILL_UNITS / 16, ts.depth - 2, l0 >> 4);
    if (ts.depth >= 3 && threadIdx.x < FILL_UNITS / 64) {
        const u32 i = (l0 >> 6) + threadIdx.x;
        if (i < ts.nwrite(ts.depth - 3))
            nodes[TreeShape::level_start(ts.depth - 3) + i] = i < ts.nreal(ts.depth - 3) ? union_of_children(lvl2 + 4 * threadIdx.x) : padding_node();
    }
}

// Up to five levels in one launch: a block owns the 1024 nodes of level c below one node of level c - 5 and computes
// the 256 + 64 + 16 + 4 + 1 nodes above them (as many of those levels as exist: `levels`), the lower ones through LDS.
// (round 2: one launch per level, then a single workgroup walking the top six levels: 76 us at 10 M points)
constexpr int UPPER_BLOCK = 256;
// blocks a launch over the levels below level c needs
__host__ __device__ inline u32 upper_blocks(const TreeShape& ts, int c, int levels)
{
    u32 ublocks = 1;
    for (int j = 1; j <= levels; ++j) {
        const u32 per_block = static_cast<u32>(UPPER_BLOCK) >> (2 * (j - 1));
        const u32 need = (ts.nwrite(c - j) + per_block - 1) / per_block;
        if (need > ublocks) ublocks = need;
    }
    return ublocks;
}
// `stage` 0: the five levels above the three k_finish made, 1: the next five, ... (the shape is read on the device: BuildCount)
__global__ __launch_bounds__(UPPER_BLOCK) void k_upper_levels(NodeBox* __restrict__ nodes, BuildCount bc, int stage)
{
    __shared__ NodeBox buf[2][UPPER_BLOCK];
    const TreeShape ts = shape_of(bc.valid());
    const int c = ts.depth - 3 - 5 * stage;
    if (c <= 0 || ts.nleaves == 0) return;
    const int levels = c < 5 ? c : 5;
    if (blockIdx.x >= upper_blocks(ts, c, levels)) return;
    const NodeBox* child = nodes + TreeShape::level_start(c) + static_cast<u64>(blockIdx.x) * (UPPER_BLOCK * 4);
    u32 width = UPPER_BLOCK;  // nodes of this block at the level being computed
    for (int j = 1; j <= levels; ++j, width >>= 2) {
        const int d = c - j;
        NodeBox* out = buf[j & 1];
        if (threadIdx.x < width) {
            const u32 i = blockIdx.x * width + threadIdx.x;
            NodeBox nb = padding_node();
            if (i < ts.nreal(d)) nb = union_of_children(j == 1 ? child + 4 * threadIdx.x : buf[(j - 1) & 1] + 4 * threadIdx.x);
            out[threadIdx.x] = nb;
            if (i < ts.nwrite(d)) nodes[TreeShape::level_start(d) + i] = nb;
        }
        __syncthreads();
    }
}

inline u64 pow4(int d) { return 1ull << (2 * d); }
inline u64 level_start(int d) { return (pow4(d) - 1) / 3; }
inline int depth_for(u64 nleaves)  // (= depth_of)
{
    int d = 0;
    while (pow4(d) < nleaves) ++d;
    return d == 1 ? 2 : d;
}

// (out of memory: the handle's pool of staging blocks may be sitting on gigabytes -- give them back and try once more)
template <class T>
int dev_alloc(T*& p, size_t count, DevPool* pool)
{
    p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = index_block_alloc(reinterpret_cast<void**>(&p), count * sizeof(T));  // (a block of an earlier index of this device, or hipMalloc)
    if (e == hipErrorOutOfMemory && pool && pool->cached_bytes() > 0) {
        (void)hipGetLastError();
        pool->trim();
        e = index_block_alloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
    if (e != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return PCPX_ERR_ALLOC;
    }
    return PCPX_OK;
}

}  // namespace

static void launch_bbox(const float* d_xyz, u64 n, hipStream_t s, u32* d_enc6)
{
    u64 blocks = (n / 4 + 255) / 256;  // a thread takes four points per trip
    if (blocks > 1024) blocks = 1024;   // (every block ends with six atomics on the same six words: 2048 blocks spent a third of the
                                        //  10 M-point launch there)
    if (blocks < 1) blocks = 1;
    k_bbox<<<static_cast<unsigned>(blocks), 256, 0, s>>>(d_xyz, n, d_enc6);
}

int device_bbox(const float* d_xyz, u64 n, hipStream_t s, u32* d_enc6, float* d_out6)
{
    k_bbox_init<<<1, 64, 0, s>>>(d_enc6, d_enc6 + 6);
    if (n > 0) launch_bbox(d_xyz, n, s, d_enc6);
    k_bbox_decode<<<1, 64, 0, s>>>(d_enc6, d_out6);
    return check_hip(hipGetLastError(), "bbox kernels", __FILE__, __LINE__);
}

int ensure_scratch(Index& ix, size_t bytes)
{
    if (bytes <= ix.scratch_bytes) return PCPX_OK;
    if (ix.d_scratch) {
        PCPX_HIP(hipStreamSynchronize(ix.stream));
        index_block_free(ix.d_scratch);
        ix.d_scratch = nullptr;
        ix.scratch_bytes = 0;
    }
    hipError_t e = index_block_alloc(&ix.d_scratch, bytes);
    if (e == hipErrorOutOfMemory && ix.pool.cached_bytes() > 0) {
        (void)hipGetLastError();
        ix.pool.trim();
        e = index_block_alloc(&ix.d_scratch, bytes);
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for query scratch failed: %s", bytes, hipGetErrorString(e));
        return PCPX_ERR_ALLOC;
    }
    ix.scratch_bytes = bytes;
    return PCPX_OK;
}

// Device arrays of a handle come in two groups with a capacity each: the CLOUD arrays (the index's copy of the input and one
// sort word per input point) and the TREE arrays (sorted words, permutation, records, leaves, boxes, the sort's temporary
// storage).  For the whole-cloud index both hold n points; a rank-local index (pcpx_shard.hip) keeps the cloud arrays at n and
// the tree arrays at the size of its selection.  Growing: every new buffer is allocated first and swapped in only when all
// allocations have succeeded, so a failure (e.g. out of memory on a 50 M-point rebuild) leaves the handle on its previous,
// still valid index.
static int grow_arrays(Index& ix, u64 need_cloud, u64 need_tree, bool want_xyz)
{
    const bool grow_cloud = need_cloud > ix.cap || !ix.d_codes[0] || (want_xyz && (!ix.d_xyz || need_cloud > ix.cap_xyz));
    const bool grow_tree = need_tree > ix.cap_tree || !ix.d_leaves;
    if (!grow_cloud && !grow_tree) return PCPX_OK;
    PCPX_HIP(hipStreamSynchronize(ix.stream));
    struct Fresh {
        float* xyz = nullptr;
        u64* codes[2] = {nullptr, nullptr};
        u32* perm = nullptr;
        float4* rec = nullptr;
        Leaf* leaves = nullptr;
        NodeBox* nodes = nullptr;
        void* sort_tmp = nullptr;
        bool keep = false;
        ~Fresh()
        {
            if (keep) return;
            index_block_free(xyz);
            index_block_free(codes[0]); index_block_free(codes[1]);
            index_block_free(perm);
            index_block_free(rec);
            index_block_free(leaves); index_block_free(nodes); index_block_free(sort_tmp);
        }
    } nw;
    int st;
    const u64 ccap = need_cloud < 64 ? 64 : need_cloud, tcap = need_tree < 64 ? 64 : need_tree;
    u64 nodes = 0;
    size_t tb = 0;
    if (grow_cloud) {
        if (want_xyz && (st = dev_alloc(nw.xyz, ccap * 3, &ix.pool)) != PCPX_OK) return st;
        if ((st = dev_alloc(nw.codes[0], ccap, &ix.pool)) != PCPX_OK) return st;
    }
    if (grow_tree) {
        if ((st = dev_alloc(nw.codes[1], tcap, &ix.pool)) != PCPX_OK) return st;
        if ((st = dev_alloc(nw.perm, tcap, &ix.pool)) != PCPX_OK) return st;
        if ((st = dev_alloc(nw.rec, tcap, &ix.pool)) != PCPX_OK) return st;
        u32 nl = static_cast<u32>((tcap + LEAF - 1) / LEAF);
        if ((st = dev_alloc(nw.leaves, nl, &ix.pool)) != PCPX_OK) return st;
        nodes = level_start(depth_for(nl) + 1);
        if ((st = dev_alloc(nw.nodes, nodes, &ix.pool)) != PCPX_OK) return st;
        if ((st = sort_keys_u64(nullptr, tb, nullptr, nullptr, tcap, ix.stream)) != PCPX_OK) return st;
        char* tmp = nullptr;
        if ((st = dev_alloc(tmp, tb ? tb : 16, &ix.pool)) != PCPX_OK) return st;
        nw.sort_tmp = tmp;
    }
    if (grow_cloud) {
        index_block_free(ix.d_xyz);
        ix.d_xyz = nw.xyz;
        ix.cap_xyz = want_xyz ? ccap : 0;
        index_block_free(ix.d_codes[0]);
        ix.d_codes[0] = nw.codes[0];
        ix.cap = ccap;
    }
    if (grow_tree) {
        index_block_free(ix.d_codes[1]);
        ix.d_codes[1] = nw.codes[1];
        index_block_free(ix.d_perm);
        ix.d_perm = nw.perm;
        index_block_free(ix.d_rec);
        ix.d_rec = nw.rec;
        index_block_free(ix.d_leaves);
        index_block_free(ix.d_nodes);
        index_block_free(ix.d_sort_tmp);
        ix.d_leaves = nw.leaves;
        ix.d_nodes = nw.nodes;
        ix.d_sort_tmp = nw.sort_tmp;
        ix.nodes_cap = nodes;
        ix.sort_tmp_bytes = tb;
        ix.cap_tree = tcap;
    }
    nw.keep = true;
    // the old index is gone: until this build completes the handle holds an empty one
    ix.n = ix.n_in = 0;
    ix.nleaves = 0;
    ix.depth = 0;
    ix.leaf0 = 0;
    return PCPX_OK;
}

int build_grow_cloud_arrays(Index& ix, u64 n, bool want_copy)
{
    int st = grow_arrays(ix, n, 0, want_copy);
    if (st != PCPX_OK) return st;
    if (!ix.d_scalars && (st = dev_alloc(ix.d_scalars, SCALARS, &ix.pool)) != PCPX_OK) return st;
    return PCPX_OK;
}
int build_grow_tree_arrays(Index& ix, u64 m) { return grow_arrays(ix, 0, m, false); }

// Box (unless the caller gave the grid) and one sort word per point; copy_cloud: the index's copy of the input rides on the
// sweep; d_tile_hist: per-tile counts of the words' top digit for the sort's first pass (or nullptr); d_hist12: points per
// level-4 cell of the curve (4096 counters, zeroed here; or nullptr).
int build_box_and_codes(Index& ix, const float* d_xyz_src, u64 n, const pcpx_build_params* params, bool copy_cloud, u32* d_tile_hist,
                        u32* d_hist12)
{
    hipStream_t s = ix.stream;
    const bool use_grid = params && (params->flags & PCPX_BUILD_USE_GRID);
    // d_scalars: [0, 6) the box in its order-preserving integer form, [6] points outside the grid, [7] the sort's failure
    // flag, [8, 14) the box
    float* d_box = reinterpret_cast<float*>(ix.d_scalars + 8);
    k_build_begin<<<1, 64, 0, s>>>(ix.d_scalars);
    if (d_hist12) PCPX_HIP(hipMemsetAsync(d_hist12, 0, 4096 * sizeof(u32), s));
    if (use_grid) {
        float g[6] = {params->grid_min[0], params->grid_min[1], params->grid_min[2],
                      params->grid_max[0], params->grid_max[1], params->grid_max[2]};
        PCPX_HIP(hipMemcpyAsync(d_box, g, sizeof(g), hipMemcpyHostToDevice, s));
        PCPX_HIP(hipStreamSynchronize(s));  // g is a stack temporary
    } else if (n > 0) {
        launch_bbox(d_xyz_src, n, s, ix.d_scalars);
    }
    u64 blocks = (n + SORT_TILE_WORDS - 1) / SORT_TILE_WORDS;  // a block takes whole tiles
    if (blocks > 512) blocks = 512;  // two resident blocks per CU
    if (blocks < 1) blocks = 1;       // (n = 0: the launch still decodes the box)
    ix.idx_bits = index_bits_for(n);
    k_codes<<<static_cast<unsigned>(blocks), CODES_BLOCK, 0, s>>>(d_xyz_src, n, ix.d_scalars, !use_grid, ix.idx_bits, ix.d_codes[0],
                                                                 copy_cloud ? ix.d_xyz : nullptr, d_tile_hist, d_hist12);
    return check_hip(hipGetLastError(), "k_codes launch", __FILE__, __LINE__);
}

// The implicit tree over the sorted words d_codes[1][0 .. nvalid) (their low idx_bits name the points' records in d_rec):
// leaf records, leaf boxes and the three levels above them in one pass, then up to five levels per launch: real nodes only
// (+ the padding siblings of the last group of four of a level, the only padding a query can read).
// `n_at_most` points at most (the grids are sized for it); how many there are is read on the device (count_on_device: input points
// minus the points outside the grid, Index::d_scalars[6]) or is n_at_most itself.  The words are those of the last sort of this
// handle (Index::finish_words: prefix-sorted, ordered here; or nullptr: d_codes[1] is sorted).  build_tree_set_shape gives the
// handle its tree once the host knows the count.
int build_tree_from_sorted(Index& ix, u32 n_at_most, bool count_on_device)
{
    hipStream_t s = ix.stream;
    ix.sched.state = 0;        // (what an earlier tree's launches recorded says nothing about this one's groups)
    ix.pos_of_valid = false;
    const TreeShape ts = shape_of(n_at_most);
    if (ts.depth > MAXDEPTH) {
        set_error("pcpx: tree deeper than %d levels", MAXDEPTH);
        return PCPX_ERR_UNSUPPORTED;
    }
    if (level_start(ts.depth + 1) > ix.nodes_cap) {
        set_error("pcpx: internal error, node capacity");
        return PCPX_ERR_INVALID;
    }
    if (ts.nleaves == 0) return PCPX_OK;
    const BuildCount bc{count_on_device ? ix.d_scalars : nullptr, static_cast<u32>(ix.n_in), n_at_most};
    const u32 fgrid = (finish_blocks(ts) + 7u) & ~7u;
    const bool from_rec = PCPX_BUILD_RECORDS || ix.shard.on;
    const FinishArgs fa{ix.finish_words, ix.d_codes[1], ix.finish_first_pass, SORT_FIRST_BIT, ix.d_scalars + REDO_WORD0};
    k_finish<<<fgrid, FILL_BLOCK, 0, s>>>(from_rec ? reinterpret_cast<const float4*>(ix.d_rec) : nullptr, ix.d_xyz, fa, ix.idx_bits, bc, ix.d_leaves,
                                          ix.d_perm, ix.d_nodes);
    int stage = 0;
    for (int c = ts.depth - 3; c > 0; c -= 5, ++stage) {
        const int levels = c < 5 ? c : 5;
        k_upper_levels<<<upper_blocks(ts, c, levels), UPPER_BLOCK, 0, s>>>(ix.d_nodes, bc, stage);
    }
    return check_hip(hipGetLastError(), "tree kernels", __FILE__, __LINE__);
}
void build_tree_set_shape(Index& ix, u32 nvalid)
{
    const TreeShape ts = shape_of(nvalid);
    ix.n = nvalid;
    ix.nleaves = (nvalid + LEAF - 1) / LEAF;
    ix.depth = ts.depth;
    ix.leaf0 = static_cast<u32>(level_start(ts.depth));
}

// d_xyz_src: device pointer to n x 3 floats (copied into the index: the reference containers copy
// their elements too, linked_kdtree.hpp:107).
int build_index(Index& ix, const float* d_xyz_src, u64 n, const pcpx_build_params* params)
{
    if (n >= 0xFFFFFFFEull) {
        set_error("pcpx: n = %llu does not fit 32-bit point indices", static_cast<unsigned long long>(n));
        return PCPX_ERR_UNSUPPORTED;
    }
    bool use_grid = params && (params->flags & PCPX_BUILD_USE_GRID);
    if (use_grid) {
        for (int a = 0; a < 3; ++a)
            if (!(params->grid_min[a] <= params->grid_max[a])) {
                set_error("pcpx: voxel grid min must not exceed max on every axis");
                return PCPX_ERR_INVALID;
            }
    }
    if (params && (params->flags & PCPX_BUILD_SHARD)) return build_shard_index(ix, d_xyz_src, n, params);
    hipStream_t s = ix.stream;
    ProfileScope prof(ix, PCPX_K_BUILD);
    ix.shard.on = false;
    int st;
    if ((st = grow_arrays(ix, n, n, true)) != PCPX_OK) return st;
    if (!ix.d_scalars && (st = dev_alloc(ix.d_scalars, SCALARS, &ix.pool)) != PCPX_OK) return st;
    ix.n_in = n;
    const bool copy_cloud = n > 0 && d_xyz_src != ix.d_xyz;  // (the copy rides on k_codes' sweep over the coordinates)
    if ((st = build_box_and_codes(ix, d_xyz_src, n, params, copy_cloud, sort_tile_hist_buffer(ix.d_sort_tmp, n), nullptr)) != PCPX_OK) return st;
    // Sort, leaves, boxes: everything is enqueued before the host looks at anything -- how many points are inside the grid is read on
    // the device (BuildCount) -- and ONE read-back ends the build: points outside, the sort's failure flag, the box, and the buckets
    // whose runs the finish kernel could not order.  Those (a cell far denser than its surroundings) take every radix pass from now on:
    // the build is repeated once with them forced, and the handle remembers them for its later rebuilds.
    const bool coarse = params && (params->flags & PCPX_BUILD_COARSE_ORDER);
    (void)coarse;  // (round 5: every build sorts only as finely as its buckets ask for, and exactly; the flag is accepted and changes nothing)
    for (int attempt = 0;; ++attempt) {
        if (n > 0) {
            if ((st = sort_for_build(ix, ix.d_codes[0], n, PCPX_BUILD_RECORDS ? d_xyz_src : nullptr, nullptr, sort_tile_hist_buffer(ix.d_sort_tmp, n))) != PCPX_OK) return st;
            if ((st = build_tree_from_sorted(ix, static_cast<u32>(n), true)) != PCPX_OK) return st;
        }
        u32 hb[SCALARS - 6];
        PCPX_HIP(hipMemcpyAsync(hb, ix.d_scalars + 6, sizeof(hb), hipMemcpyDeviceToHost, s));
        PCPX_HIP(hipStreamSynchronize(s));
        const u32 outside = hb[0], sort_failed = hb[1];
        std::memcpy(ix.bbox, &hb[2], 6 * sizeof(float));
        if (n > 0 && sort_failed) {
            ix.n = ix.n_in = 0;
            ix.nleaves = 0;
            set_error("pcpx: internal error, the radix sort's look-back gave up");
            return PCPX_ERR_DEVICE;
        }
        if (n > 0 && ix.finish_words) ix.low_pass_tiles = hb[LOW_TILES_WORD - 6];
        bool redo = false;
        for (u32 w = 0; w < 8; ++w) {
            const u32 r = hb[REDO_WORD0 - 6 + w];
            redo = redo || (r & ~ix.full_buckets[w]) != 0u;
            ix.full_buckets[w] |= r;
        }
        if (redo && attempt == 0 && n > 0) {
            ++ix.build_redos;
            // (the tile counts of the top digit were scanned in place by the sort: the keys are made again, which also clears the flags)
            if ((st = build_box_and_codes(ix, d_xyz_src, n, params, false, sort_tile_hist_buffer(ix.d_sort_tmp, n), nullptr)) != PCPX_OK) return st;
            continue;
        }
        build_tree_set_shape(ix, static_cast<u32>(n) - outside);
        if (ix.tuning.lpt && n >= 64 * 512) (void)sched_reserve(ix, (n + GROUP - 1) / GROUP);  // (here, not in the first query; failure: no schedule)
        (void)ensure_queue(ix);  // (likewise the persistent kernels' queue counters)
        return PCPX_OK;
    }
}

// The build's sort: d_words (n words: key bits | element index) -> prefix-sorted words + the records at the positions the words' low
// bits name, in finish mode (SortPayload::finish; Index::finish_words then says where the words are for build_tree_from_sorted).
int sort_for_build(Index& ix, const u64* d_words, u64 n, const float* d_xyz_src, const float4* d_rec_in, u32* tile_hist_ready)
{
    hipStream_t s = ix.stream;
    bool any_forced = false;
    for (u32 w = 0; w < 8; ++w) any_forced = any_forced || ix.full_buckets[w] != 0u;
    if (any_forced) PCPX_HIP(hipMemcpyAsync(ix.d_scalars + FORCE_WORD0, ix.full_buckets, sizeof(ix.full_buckets), hipMemcpyHostToDevice, s));
    size_t tb = ix.sort_tmp_bytes;
    SortPayload pl;
    SortPayload::FinishOut fo;
    pl.xyz = d_xyz_src;
    pl.rec_in = d_rec_in;
    pl.rec = ix.d_rec;
    pl.idx_bits = ix.idx_bits;
    pl.tile_hist_ready = tile_hist_ready;
    pl.failed_flag = ix.d_scalars + 7;
    pl.finish = PCPX_BUILD_FINISH != 0;
    pl.force_full = any_forced ? ix.d_scalars + FORCE_WORD0 : nullptr;
    pl.finish_out = &fo;
    pl.low_pass_tiles_hint = ix.low_pass_tiles;
    pl.low_pass_tiles_out = ix.d_scalars + LOW_TILES_WORD;
    const int st = sort_keys_u64(ix.d_sort_tmp, tb, d_words, ix.d_codes[1], n, s, SORT_FIRST_BIT, &pl);
    ix.finish_words = fo.words;
    ix.finish_first_pass = fo.bucket_first_pass;
    ix.sorted_from_bit = fo.words ? 40 : SORT_FIRST_BIT;  // (Index::sorted_codes: the prefix-sorted words, ordered on bits [40, 64) everywhere)
    return st;
}

}  // namespace pcpx
