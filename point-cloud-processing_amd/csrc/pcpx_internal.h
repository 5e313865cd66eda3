// pcpx_internal.h -- shared between the build, query and API translation units of libpcpx.so.
//
// Index layout in HBM (see DESIGN.md "Data layout"):
//   * points are sorted along a Hilbert curve over the voxel grid (pcpx_curve.h; the grid is the reference's,
//     include/pcp/octree/linked_octree_node.hpp:258-265);
//   * the sorted order is cut into LEAVES of LEAF consecutive points, stored SoA per leaf
//     {x[LEAF], y[LEAF], z[LEAF], id[LEAF]} so that one leaf is one contiguous 16*LEAF-byte record
//     that a wavefront fetches with scalar (SMEM) loads and broadcasts to its 64 lanes;
//   * above the leaves sits an IMPLICIT W-ary tree of tight AABBs: node i of level l covers
//     children [i*W, i*W+W) of level l-1.  No pointers, no parent links: built by a bottom-up sweep.
#ifndef PCPX_INTERNAL_H
#define PCPX_INTERNAL_H

#include <cstring>
#include <hip/hip_runtime.h>

#include <mutex>

#include <cstdint>
#include <string>
#include <vector>

#include "pcpx.h"

namespace pcpx {

using u32 = std::uint32_t;
using u64 = std::uint64_t;

constexpr int LEAF = 8;    // points per leaf
constexpr int LOGW = 2;    // log2 of the tree arity
constexpr int W = 1 << LOGW;
constexpr int GROUP = 64;  // queries per wavefront
constexpr int LEAVES_PER_GROUP = GROUP / LEAF;
// The boxes of the tree's bottom level are over UNITS of UNIT_LEAVES consecutive leaves: 1 (a box per leaf record), or 2 -- sixteen
// points under one box, the records still eight points each; a last-level node then covers a whole query group's 64 points, the tree
// is half a level shallower and a walk expands 28 % fewer nodes for 50 % more records looked at (tools/sim_unit16.py: -14 % scalar,
// -11 % vector instructions in the walk by the model).  Measured, round 5 (every kernel and the build written for either; all parity
// tests pass with 2): k_knn 1.5 % SLOWER (3.68-3.73 -> 3.75-3.78 ms uniform, 4.20 -> 4.24 clustered), k_range 2 % faster, the rebuild
// 3 % slower.  1 stays.
#ifndef PCPX_UNIT_LEAVES
#define PCPX_UNIT_LEAVES 1
#endif
constexpr int UNIT_LEAVES = PCPX_UNIT_LEAVES;
static_assert(UNIT_LEAVES == 1 || UNIT_LEAVES == 2, "units of one or two leaves");
constexpr int UNIT_POINTS = UNIT_LEAVES * LEAF;
constexpr int MAXDEPTH = 15;  // 4^15 leaves of 8 points: far beyond 2^32 points
constexpr u32 INVALID_ID = 0xFFFFFFFFu;
// The curve order only has to make leaves spatially compact: any order gives a correct tree (boxes come from the
// points).  A sort word is {bits [25, 64): 39-bit curve key (13 bits per axis: cells of 1/8192 of the grid extent), low
// bits: the element's index; all ones above the index: outside the grid}; the radix sort looks at bits [24, 64) only -- 5
// passes -- and the index overwrites as many of the key's low bits as it needs (more than 25 only beyond 33 M elements:
// 12 key bits per axis up to 268 M); see pcpx_curve.h.
constexpr int CURVE_FIRST_BIT = 25;
#ifndef PCPX_SORT_FIRST_BIT
#define PCPX_SORT_FIRST_BIT 24
#endif
constexpr int SORT_FIRST_BIT = PCPX_SORT_FIRST_BIT;  // experiment: 32 = four passes on the top 31 key bits
inline int index_bits_for(u64 n)
{
    int b = 1;
    while (b < 32 && (1ull << b) < n) ++b;
    return b;
}

struct Leaf {
    float x[LEAF];
    float y[LEAF];
    float z[LEAF];
    u32 id[LEAF];
};
static_assert(sizeof(Leaf) == 16 * LEAF, "leaf record must be dense");

// One node of the implicit 4-ary AABB tree, 32 B so that the 4 children of a node are one 128-B
// scalar load.  `poison` is +0 for a real node and NaN for a padding node (no points below it): it is
// added to the box distance, so a padding node can never pass a `distance <= bound` test.
struct NodeBox {
    float lo3[3];
    float hi3[3];
    float poison;
    float pad;
    __host__ __device__ float lo(int a) const { return lo3[a]; }
    __host__ __device__ float hi(int a) const { return hi3[a]; }
    __host__ __device__ void set(float lx, float ly, float lz, float hx, float hy, float hz)
    {
        lo3[0] = lx, lo3[1] = ly, lo3[2] = lz, hi3[0] = hx, hi3[1] = hy, hi3[2] = hz;
    }
};
static_assert(sizeof(NodeBox) == 32, "node box must be 32 bytes");

// Heap layout: root = node 0, children of node h = 4h+1 .. 4h+4, so depth d starts at (4^d-1)/3.
// Unit u (leaves UNIT_LEAVES u ...) is node leaf0 + u with leaf0 = (4^depth - 1)/3; depth is the smallest with 4^depth >= nunits.
struct TreeView {
    const Leaf* leaves;    // nleaves records
    const NodeBox* nodes;  // (4^(depth+1)-1)/3 nodes
    u32 nleaves;
    u32 n;                 // indexed points
    int depth;
    u32 leaf0;             // heap id of unit 0
    __host__ __device__ u32 nunits() const { return (nleaves + UNIT_LEAVES - 1) / UNIT_LEAVES; }
};

// Device outputs of the fused kNN kernel, one row per query (row index = input index of the query).
struct KnnOutputs {
    u32* idx = nullptr;         // rows x k neighbour indices, ascending (d2, index), INVALID_ID padded
    u32* cnt = nullptr;         // rows: valid entries per row
    float* d2 = nullptr;        // rows x k squared distances
    float* normals = nullptr;   // rows x 3: pcp::estimate_normal of the row
    float* centroids = nullptr; // rows x 3: pcp::common::center_of_geometry of the row (tangent plane point)
    float* meandist = nullptr;  // rows: mean Euclidean distance to the row's neighbours
    u32 by_position = 0;        // self queries only: 1 = row index = the query's SORTED position instead of its input index (rows
                                // of a slice of the curve order are then one contiguous piece of every output array)
    // self queries: only sorted positions [pos_lo, pos_hi) are answered (the other lanes of their groups idle); 0, ~0 = all
    u32 pos_lo = 0, pos_hi = 0xFFFFFFFFu;
    u32 pos_bias = 0;           // by_position: row = position + pos_bias (a rank-local index: local -> global curve position)
    float* tau = nullptr;       // self queries: d2 of the k-th neighbour (+inf: fewer than k found) of sorted position p at [p]
                                // -- what the coverage check of a rank-local index needs (pcpx_shard.hip)
    u32 row_stride = 0;         // entries between the rows of idx / d2 (0: k).  A stride of KCAP (8 / 16 / 32) with k = KCAP or KCAP - 1
                                // makes a row one aligned piece written with 16-byte stores: a 60-byte row scattered by input index
                                // costs its partial 32-byte sectors twice (read for ownership + write back)
    float4* nc4 = nullptr;      // self queries: {normal, count as bits} of sorted position p at [p] INSTEAD of normals / cnt by row -- the
                                // first half of the gather-form permute to input order (k_gather_nc4: coalesced writes, cached reads)
};

// What steers a self-kNN launch besides its outputs (all optional).
struct KnnSchedule {
    const u32* order = nullptr;  // the launch answers groups order[0 .. group_count) (absolute group ids) instead of group_first + i; queue q
                                 // hands out its eighth of the array in array order (long groups first: pcpx_query.hip, k_make_order)
    u32* gtime = nullptr;        // [g - group_first]: shader-clock ticks / 64 the group took (what the next launch's order is made from)
    u32* events = nullptr;       // diagnostic-cost kernel only: 4 words per slot of the launch {expansions, dense leaves, packed leaves,
                                 // folds << 16 | packed steps}: deterministic, what pcpx_knn_group_costs_dev reports
};

// Queries of a batch, in curve-sorted order.  For self queries qx == nullptr and the query of
// sorted position p is point p of the leaves.
struct QueryView {
    const float* qx;
    const float* qy;
    const float* qz;
    const u32* row;   // output row of sorted query p
    const u32* seed;  // first leaf of the seed range of group g
    u32 nq;
    u32 pos_lo = 0, pos_hi = 0xFFFFFFFFu;  // k_range, self ranges: only sorted positions [pos_lo, pos_hi) are answered
    u32 by_position = 0, pos_bias = 0;     // k_range, self counts: 1 = the count of sorted position p goes to [p + pos_bias] (one
                                           // contiguous 256-B store per query group) instead of to the point's input index
};

// Device allocations that outlive a call: the host-pointer entry points stage through device buffers, and a
// hipMalloc / hipFree pair per buffer per call (hipFree synchronises the device) used to cost more than the kernels.
// A pool belongs to one handle (or, for the entry points without a handle, to one device) and is only touched under
// its owner's mutex; blocks are handed out best-fit and returned on scope exit, after the stream has been synchronised.
struct DevPool {
    struct Block {
        void* p;
        size_t bytes;
        bool used;
    };
    std::vector<Block> blocks;
    void* acquire(size_t bytes);  // nullptr (and the thread's error text set) when the device is out of memory
    void release(void* p);
    void trim();                  // frees every block that is not handed out
    size_t cached_bytes() const;
    ~DevPool();
};

// Device blocks of INDEXES (cloud copy, sort words, leaves, nodes, sort workspace ...), cached per device across handles:
// hipMalloc / hipFree cost 0.3 ... 1 ms each and hipFree synchronises the device, and a drop-in caller builds containers in
// a loop (benchmark/spatial_data_structures_benchmark.cpp:108-148 constructs one per iteration) -- at 2^20 points a dozen
// allocations and frees were most of the 25 ms of a construction.  index_block_free keeps a block for the next
// index_block_alloc of the CURRENT device that it fits (at most a quarter larger than asked), up to PCPX_DEVICE_CACHE_MB
// (environment, default 2048; 0: no caching) of idle blocks per device; what does not fit goes back to the driver, and an
// allocation that fails empties the cache and tries again.  The caller has finished with the block (its stream is drained).
hipError_t index_block_alloc(void** p, size_t bytes);
void index_block_free(void* p);
void index_blocks_trim();  // every idle block of the current device back to the driver

// Streams of handles, kept per device across handles: hipStreamCreate costs 8 ms on this stack and hipStreamDestroy 2 (measured:
// tools/h2d_probe.hip) -- more than a 2^20-point index build with its upload.  A handle's private streams come from here and go
// back drained (at most 8 idle per device; the rest are destroyed).
hipError_t pooled_stream_get(hipStream_t* s);
void pooled_stream_put(hipStream_t s);  // the caller has synchronised it

// pinned host memory of a handle: staging for small transfers (a query point in, one row out) that the device reads and
// writes in place, so a single-query call issues no copy at all
struct PinnedStage {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);
    ~PinnedStage();
};

struct Index {
    // Entry points that take a handle hold this for their duration: the reference's containers are queried
    // concurrently from PSTL worker threads (estimate_normals.hpp:92), and the handle's scratch buffers, work
    // queue and stream are shared state.  Recursive: some entry points are built from others.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    u64 n_in = 0;     // input points
    u64 n = 0;        // inserted points
    u64 cap = 0;      // capacity (points) of the cloud arrays: d_codes[0] (and d_xyz: cap_xyz)
    u64 cap_xyz = 0;
    u64 cap_tree = 0; // capacity (points) of the tree arrays: d_codes[1], d_perm, d_rec, d_leaves, d_nodes, d_sort_tmp
    float bbox[6] = {0, 0, 0, 0, 0, 0};

    float* d_xyz = nullptr;      // n_in x 3, input order
    u64* d_codes[2] = {nullptr, nullptr};  // sort words: [0] input order, [1] sorted
    u32* d_perm = nullptr;                 // sorted position -> input index
    float4* d_rec = nullptr;               // {x, y, z, input index} per point, grouped by the sort word's top digit (the leaf fill's source)
    int idx_bits = 1;                      // low bits of a sort word that hold the input index
    int sorted_from_bit = 24;              // the sorted words are ordered on their bits [sorted_from_bit, 64) everywhere
    const u64* finish_words = nullptr;     // the last sort's prefix-sorted words (in d_sort_tmp) for build_tree_from_sorted, or nullptr: d_codes[1] is sorted
    const u32* finish_first_pass = nullptr;  // ... and its per-bucket table
    u32 full_buckets[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // top-digit buckets that take every radix pass (a build found a run it could not order in LDS)
    u32 build_redos = 0;                   // builds that were repeated because of that (diagnostic)
    u32 low_pass_tiles = ~0u;              // tiles the sort's two lowest passes had in the last build (~0: no build yet): sizes their grids next time
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    Leaf* d_leaves = nullptr;
    NodeBox* d_nodes = nullptr;
    u64 nodes_cap = 0;           // nodes
    u32* d_scalars = nullptr;    // [0..6) encoded bbox, [6] points outside the grid, [7] sort failure, 6 floats decoded bbox at [8..14)
    u32 nleaves = 0;
    int depth = 0;
    u32 leaf0 = 0;

    // profiling (pcpx_profile_begin/end): one event pair per kernel-family interval
    struct Interval {
        int family;
        hipEvent_t a, b;
    };
    bool profiling = false;
    std::vector<Interval> intervals;

    u32* d_queue = nullptr;  // work-queue counters of the persistent query kernels: two sets of 8 (64 bytes apart each); a launch uses one and
                             // zeroes the other for the launch after it (no memset dispatch between launches: prepare_queue)
    u32 queue_set = 0;       // the set the next launch uses
    u32* queue_now = nullptr;    // prepare_queue: this launch's set ...
    u32* queue_clear = nullptr;  // ... and the one it leaves zeroed
    u64* d_multi = nullptr;  // k > 32: per-query pass keys + pass bounds
    size_t multi_bytes = 0;

    // scratch for batch queries (grown on demand)
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;

    hipStream_t copy_stream = nullptr;  // second stream of the sliced host-pointer form (copies of finished slices beside the kernels)
    DevPool pool;        // staging buffers of the host-pointer entry points
    PinnedStage pinned;  // small-transfer staging
    u32 few_epoch = 0;   // launch counter of the latency path (its completion flag carries the epoch)
    // ---- long groups first (pcpx_query.hip: k_make_order).  A self-kNN launch records what every group took; when the same question
    // (slice, k class) comes again on the same tree, the launch hands the groups out by those times -- the longest third first, curve
    // order inside a class -- so that a short launch (one rank's eighth of a cloud: 2.7 groups per resident wave) does not end with a
    // 3x-the-mean group started last.  Results do not depend on the order.
    struct Sched {
        u32* d_gtime = nullptr;   // per group of the recorded launch
        u32* d_order = nullptr;
        u64 cap = 0;              // groups both arrays hold
        u64 gf = 0, gc = 0;       // the recorded launch's groups
        int kcap = 0;
        u32 k = 0;
        int state = 0;            // 0: nothing, 1: times recorded, 2: order made
    } sched;
    float4* d_nc4 = nullptr;      // {normal, count} per curve position (gather-form outputs), n entries
    u32* d_pos_of = nullptr;      // input index -> curve position (0xFFFFFFFF: not indexed), n_in entries; valid while pos_of_valid
    u64 nc4_cap = 0, pos_of_cap = 0;
    bool pos_of_valid = false;
    struct Tuning {
        int lpt = 1;              // long groups first on repeated self-kNN launches
        int gather_counts = 0;    // input-order radius counts through the gather-form permute: measured SLOWER too (10 M counts: 1.92-2.08 ms straight,
                                  // 2.13 through the permute -- the kernel at curve positions takes 1.89 and the gather 0.24: ten million random 4-byte
                                  // reads are ten million cache lines); a switch, like `gather`
        int gather = 0;           // input-order normals + counts through the gather-form permute (measured round 5: 1-3 % SLOWER than the
                                  // direct scattered stores on the 10 M clouds -- the search kernel is not bound by them; kept as a switch)
    } tuning;
    int eps_test_mode = 0;  // k_knn's eps-box test: 0 = where it is cheaper (query.hip: eps_box_threshold), 1 = in the compaction, 2 = per candidate

    // ---- rank-local index (PCPX_BUILD_SHARD, pcpx_shard.hip): the tree holds only the points a rank's shard of the curve-sorted
    // queries can reach -- the CORE (the cells of the curve that hold the shard) and a HALO of cells around it; n, nleaves, the
    // leaves and the boxes are those of the local tree, n_glob is the size of the whole cloud's index.
    struct Shard {
        bool on = false;
        u32 rank = 0, world = 1;
        u64 n_glob = 0;              // inserted points of the whole cloud
        u64 g_first = 0, g_count = 0;  // the shard: global curve positions [g_first, g_first + g_count)
        u64 core_g0 = 0;             // global curve position of the core's first point ...
        u64 core_l0 = 0;             // ... and its position in the local tree; the core is contiguous in both orders
        u64 core_count = 0;
        u32 halo_cells = 0;          // cells (of the selection grid) the core was dilated by
        u32 k_hint = 0;
        bool explicit_range = false;  // PCPX_BUILD_SHARD_RANGE: the shard is [range_first, range_first + range_count), not pcpx_shard_range(rank, world)
        u64 range_first = 0, range_count = 0;
        bool everything = false;     // the selection covers the whole grid: nothing to verify
        bool borrowed = false;       // the cloud is read in the caller's array (PCPX_BUILD_BORROW_CLOUD)
        const float* cloud = nullptr;  // n_in x 3, input order: the index's copy or the caller's array
        u32* d_sel = nullptr;        // selection bitmap over the cells of the curve at level SHARD_SEL_LEVEL, indexed by curve prefix
        u32* d_need = nullptr;       // cells a failed coverage check asks for
        u32* d_grid = nullptr;       // scratch: the selection on the xyz grid (one byte per cell, two buffers)
        u32* d_hist12 = nullptr;     // points per level-4 cell of the curve (4096), whole cloud
        u32* d_plan = nullptr;       // device scalars of the plan (see pcpx_shard.hip)
        u32* d_big = nullptr;        // coverage check: the search boxes too large for one thread
        u32* d_tile_cnt = nullptr;   // selected words per tile of the codes
        u64 tile_cap = 0;
        u64* d_words = nullptr;      // the selected words, compacted in input order (the local sort's input)
        float* d_tau = nullptr;      // k-th squared distance per local position (coverage check)
        u32* d_fail = nullptr;       // [0] failed queries of the last check, then their local positions
        u64 cap_loc = 0;             // capacity (points) of the local arrays
        // the coverage check of a static index need not be repeated for the same question
        u32 verified_k = 0;
        float verified_eps = -1.f;
        u64 verified_first = 0, verified_count = 0;
        u64 last_failed = 0, total_failed = 0, enlargements = 0;  // diagnostics
    } shard;

    // the words of the indexed points in curve order, ordered on their bits [sorted_from_bit, 64): the last sort's prefix-sorted words
    // (they live in d_sort_tmp until the next build of this handle) or, when the sort ran every pass, d_codes[1]
    const u64* sorted_codes() const { return finish_words ? finish_words : d_codes[1]; }
    u32* perm() const { return d_perm; }
    TreeView view() const { return TreeView{d_leaves, d_nodes, nleaves, static_cast<u32>(n), depth, leaf0}; }
};

// RAII: records an event pair around a kernel family on the index's stream while profiling is on
struct ProfileScope {
    Index& ix;
    int family;
    hipEvent_t a = nullptr, b = nullptr;
    ProfileScope(Index& i, int fam) : ix(i), family(fam)
    {
        if (!ix.profiling) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            a = b = nullptr;
            return;
        }
        (void)hipEventRecord(a, ix.stream);
    }
    ~ProfileScope()
    {
        if (!a) return;
        (void)hipEventRecord(b, ix.stream);
        ix.intervals.push_back(Index::Interval{family, a, b});
    }
};

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what, const char* file, int line);
#define PCPX_HIP(expr)                                                       \
    do {                                                                     \
        int _s = ::pcpx::check_hip((expr), #expr, __FILE__, __LINE__);       \
        if (_s != PCPX_OK) return _s;                                        \
    } while (0)

// build.hip
int build_index(Index& ix, const float* d_xyz_src, u64 n, const pcpx_build_params* params);
// the pieces of build_index that the rank-local build (pcpx_shard.hip) shares with it
int build_grow_cloud_arrays(Index& ix, u64 n, bool want_copy);           // d_xyz (if want_copy), d_codes[0], d_scalars
int build_grow_tree_arrays(Index& ix, u64 m);                            // d_codes[1], d_perm, d_rec, d_leaves, d_nodes, d_sort_tmp for m points
int build_box_and_codes(Index& ix, const float* d_xyz_src, u64 n, const pcpx_build_params* params, bool copy_cloud, u32* d_tile_hist,
                        u32* d_hist12);
int build_tree_from_sorted(Index& ix, u32 n_at_most, bool count_on_device);  // leaf records, boxes, levels from the sorted words / d_rec
void build_tree_set_shape(Index& ix, u32 nvalid);
int sort_for_build(Index& ix, const u64* d_words, u64 n, const float* d_xyz_src, const float4* d_rec_in, u32* tile_hist_ready);
constexpr u32 BUILD_REDO_WORD0 = 16;  // Index::d_scalars[16 .. 24): buckets whose runs the finish kernel could not order
void free_shard(Index& ix);
// shard.hip
int build_shard_index(Index& ix, const float* d_xyz_src, u64 n, const pcpx_build_params* params);
// self queries of a rank-local index: translate the global slice, run, check coverage, enlarge and re-run what failed
int shard_knn_self(Index& ix, u64 sorted_first, u64 sorted_count, u32 k, float eps, KnnOutputs o);
int shard_range_count_self(Index& ix, float radius, u64 sorted_first, u64 sorted_count, u32* d_out_cnt, bool by_position = false);
int shard_unsupported(const Index& ix, const char* what);
int shard_perm(Index& ix, u32* d_out_perm, u32* d_out_pos);
int device_bbox(const float* d_xyz, u64 n, hipStream_t s, u32* d_enc6, float* d_out6);
// stable radix sort of 64-bit words by their bits [first_bit, 64) (first_bit a multiple of 8); words that agree on
// those bits keep their input order.  sort_failure_flag: device word of the temporary storage that the sort sets if its
// look-back ever gave up (it never should); read it after synchronising the stream.
constexpr int SORT_TILE_WORDS = 4096;  // words per tile of the radix sort's passes
// payload (optional): what the index build hangs on the sort's first pass.
struct SortPayload {
    const float* xyz = nullptr;           // n x 3 in the words' input order: with `rec` and `idx_bits`, the first pass moves {x, y, z, index}
    float4* rec = nullptr;                //   of every element to its word's position after that pass and writes that position into the
    int idx_bits = 0;                     //   word's low idx_bits (which held the element's index)
    const float4* rec_in = nullptr;       // instead of xyz: the elements' records {x, y, z, id} in the words' input order, moved as they are
                                          //   (a rank-local index: the id is the point's index in the whole cloud, not in the sorted subset)
    u32* tile_hist_ready = nullptr;       // counts of the words' top digit (bits [56, 64)) per tile of SORT_TILE_WORDS consecutive words,
                                          //   [tile][256], if the caller has them already (scanned in place by the sort)
    int adaptive_margin_bits = 0;         // > 0: a bucket (words sharing the top digit) of c words is sorted on ceil((ceil(log2 c) + margin) / 8)
                                          //   of its lower digits only, at least two: every word is then ordered on bits [40, 64), and on as many
                                          //   bits below as separate the words of its own bucket with `margin` bits to spare.  0: the full sort.
    u32* failed_flag = nullptr;           // device word (zeroed by the caller) to set if the look-back ever gives up; default: inside tmp
    // The index build's short cut (round 5).  finish = true: a bucket is sorted on its top TWO lower digits only (bits [40, 64) of
    // the words) when that is likely to leave RUNS -- words that agree on those bits -- of a handful of words: at most SORT_RUN_TARGET
    // words per 24-bit cell on average over the bucket and over its fullest 16-bit cell (k_sort_seg_plan, from the digit counts the
    // sort has anyway); other buckets -- and those the caller names in force_full, and the last one, where the words of points outside
    // the grid go -- take all passes as before.  The caller orders the runs itself (pcpx_build.hip: k_finish sorts them in LDS, in the
    // kernel that writes the leaves) and reports the buckets whose runs turned out longer than it can handle, which the next attempt
    // names in force_full.  The words come back PREFIX-sorted in the sort's temporary buffer (`finish_out`), not in kout -- the finish
    // kernel's output is what kout's buffer is for -- with the bucket table beside them.
    bool finish = false;
    const u32* force_full = nullptr;      // finish: device bitmap, 8 words; bit b: bucket b takes every pass
    u32 low_pass_tiles_hint = ~0u;        // finish: tiles the two lowest passes had in this handle's last build (~0: unknown) ...
    u32* low_pass_tiles_out = nullptr;    // ... and where this sort leaves its own count (device word)
    struct FinishOut {
        const u64* words = nullptr;           // n words, ordered on bits [first_bit + 8 (first_pass[b] - 1), 64) inside bucket b
        const u32* bucket_first_pass = nullptr;  // 256 entries (device)
    }* finish_out = nullptr;
};
constexpr u32 SORT_RUN_TARGET = 16;     // (a run the finish kernel can order is at most 64 words: four times the average it plans for)
int sort_keys_u64(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, u64 n, hipStream_t s, int first_bit = 0,
                  const SortPayload* payload = nullptr);
const u32* sort_failure_flag(void* tmp);
u32* sort_tile_hist_buffer(void* tmp, u64 n);
int ensure_scratch(Index& ix, size_t bytes);

// orient.hip: propagate_normal_orientations on the device (rows, counts, coordinates, normals are device arrays)
int orient_normals_device(const float* d_xyz, u64 n, const u32* d_nbr, const u32* d_cnt, u32 k, float* d_normals, hipStream_t s,
                          u64* out_reached, u32* out_levels);

// query.hip
// kNN + whatever per-neighbourhood products are requested (any pointer may be nullptr); all fused in k_knn
int launch_knn(Index& ix, const QueryView& qv, bool self, u64 group_first, u64 group_count, u32 k, float eps,
               const KnnOutputs& o);
// event counts of sampled groups (every `stride`-th group of the curve order, from group stride / 2 on): d_events = 4 words per sample
int launch_knn_cost_sample(Index& ix, u32 k, float eps, u32 stride, u32* d_events, u32* out_samples);
int launch_gather_u32(Index& ix, const u32* d_at_position, const u32* d_pos_of, u64 n_rows, u32 pos_lo, u32 pos_hi, u32* d_out);
int launch_gather_nc4(Index& ix, const float4* d_nc4, const u32* d_pos_of, u64 n_rows, u32 pos_lo, u32 pos_hi, float* d_normals, u32* d_cnt);
// (q_host: the queries where the host can read them, or nullptr -- a single query is then passed in the kernel arguments)
int launch_knn_few(Index& ix, const float* q_aos, const float* q_host, u32 nq, u32 k, float eps, u32* out_idx, u32* out_cnt, float* out_d2,
                   u32* flags, u32* done_count, u32* done_flag, u32 epoch);
int launch_knn_stats(Index& ix, u32 k, float eps, unsigned long long* d_stats, const float* d_known_d2 = nullptr);
int launch_range_count(Index& ix, const QueryView& qv, bool self, u64 group_first, u64 group_count, float radius,
                       const float* d_radii, u32* d_out_cnt);
int launch_range_fill(Index& ix, const QueryView& qv, float radius, const float* d_radii, const u64* d_offsets,
                      u32* d_out_idx);
int launch_range_offsets(Index& ix, const u32* d_cnt, u64 n_rows, u64* d_tile_sum, u64* d_offsets);
int launch_range_fill_self(Index& ix, u64 group_first, u64 group_count, float radius, const u64* d_offsets, u32* d_out_idx);
int launch_range_one(Index& ix, bool aabb, const float* range, u32 cap, u32* out_idx, u32* out_cnt, u32* done_flag, u32 epoch);
int launch_aabb_count(Index& ix, const float* d_boxes6, u64 nb, u32* d_out_cnt);
int launch_aabb_fill(Index& ix, const float* d_boxes6, u64 nb, const u64* d_offsets, u32* d_out_idx);
int launch_normals(Index& ix, const u32* d_nbr, const u32* d_cnt, const u32* d_rowmap, u64 first, u64 count, u32 k,
                   float* d_out, float* d_evals, float* d_centroids = nullptr, float* d_meandist = nullptr, u32 row_bias = 0);
int launch_normal_single(const float* d_xyz, u64 m, float* d_out3, hipStream_t s);
constexpr u32 NORMAL_ARG_POINTS = 64;  // largest neighbourhood that travels in the kernel arguments (launch_normal_args)
int launch_normal_args(const float* xyz_host, u32 m, float* out3_pinned, u32* done_flag_pinned, u32 epoch, hipStream_t s);
int launch_normals_csr(const float* d_xyz, const u64* d_offsets, u64 nrows, float* d_out, hipStream_t s);
// filter.hip: the loops that consume sphere ranges (bilateral filter, WLOP), fused into the range walk
size_t leaf_record_bytes(u64 points, int components);  // components: 3 (32-byte records), 1 or 0 (16-byte records)
int launch_leaf_records(Index& ix, const float* d_attr, int components, void* d_records);  // point + attribute row, leaf order
int launch_fill_f32(float* d_p, u64 n, float v, hipStream_t s);
int launch_take_rows(const float* d_xyz, u64 n, const u64* d_sample, u64 m, float* d_out, hipStream_t s);
int launch_bilateral(Index& ix, const void* d_records, float sigmaf, float sigmag, bool normals_mode, float* d_out);
int launch_wlop_density(Index& ix, float h, const void* d_records, float* d_out);
int launch_wlop_median(Index& cloud, const QueryView& samples, float h, const void* d_records_vj, float* d_median);
int launch_wlop_repulsion(Index& samples, float h, float mu, const void* d_records_wi, const float* d_median, float* d_out);
int prepare_queries(Index& ix, const float* d_q, u64 nq, QueryView& qv);
int launch_invert_perm(const u32* d_perm, u64 n, u32* d_position_of, hipStream_t s);
int ensure_queue(Index& ix);   // the counters' allocation (zeroed)
int sched_reserve(Index& ix, u64 groups);  // the arrays of the recorded order for launches of up to `groups` query groups (Index::sched)
int prepare_queue(Index& ix);  // which set of work-queue counters the next persistent launch uses (Index::queue_now) and which it zeroes (queue_clear)

}  // namespace pcpx

#endif
