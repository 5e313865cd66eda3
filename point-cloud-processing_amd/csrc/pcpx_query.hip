// pcpx_query.hip -- the throughput kNN kernel (k nearest neighbours of query batches, fused PCA normal / tangent-plane
// centroid / mean neighbour distance) for gfx950 (wave64).  The latency form for a handful of queries is pcpx_few.hip.
//
// Execution model (DESIGN.md "Kernels"):
//   * ONE WAVEFRONT = 64 queries that are consecutive on the index's Hilbert curve (pcpx_curve.h), one query per lane.
//   * Tree traversal is WAVE-UNIFORM: the pending-children bit stack (popped with one find-first-set) and all control
//     flow live in SGPRs; the 4 child boxes of a node (one 128-B record) and whole leaf records (128 B) are fetched with
//     scalar (SMEM) loads and broadcast to the 64 lanes as SGPR operands of the per-lane VALU distance code.  A subtree is
//     entered when ANY lane still needs it (ballot), so the wave walks the union of its lanes' search regions -- small,
//     because the lanes are neighbours on the curve.
//   * kNN selection: per lane a SORTED best-list of KCAP (8 / 16 / 32) 64-bit keys (d2 bits << 32 | sorted position) in
//     VGPRs plus an UNSORTED append buffer in LDS (one column per lane, conflict free).  Accepting a candidate is
//     hand-written: two v_cmpx narrow EXEC to the lanes with d2 <= tau outside the eps-box, the LDS write and the
//     address bump run under it.  When a column is nearly full the wave sorts the new keys in registers, 8 at a time
//     (bitonic network whose compare-exchange is v_min_f64 / v_max_f64: every key is the bit pattern of a finite
//     non-negative double) and merges them with the best-list, which tightens tau = d2 of the k-th best.
//   * Seeding: the 64 points at the group's own curve position (for arbitrary queries: the 64-point chunk where the
//     group's middle query would sit) and a few leaves either side (per KCAP) are processed before the walk, so tau is
//     tight when the traversal starts.
//   * Walk rounds: the first walk searches no farther than cap_mult<KCAP>() (1.125 ... 1.375) x the wave's median
//     seeded tau; a lane whose k-th distance ends up beyond that cap goes round again with a 4x larger one, accepting
//     only the new shell.  The cap steers work, never results.
//   * Persistent waves pull query groups from 8 work queues (one per XCD-sized eighth of the curve order).
//   * Keys carry the SORTED position, so neighbour ids and coordinates are gathered from the leaf records (spatially
//     coherent, L1/L2 resident); the final rows are re-ordered by (d2, original index).  After the search the row's
//     positions wait in the (then idle) LDS column, not in registers: no kernel of the single-pass family uses scratch.
//   * PCA normals are fused into the same kernel: no neighbour list round trip through HBM.
//   * Rows go to the query's input index, or (KnnOutputs::by_position, self queries) to its curve position: slices of
//     the curve order are then contiguous in every output array (pcpx_normals_knn_self_curve_order).
// The measurements behind every tuning constant below are in profiles/experiments/README.md.
//
// Arithmetic follows the reference exactly: d = p - q, dx*dx + dy*dy + dz*dz in float32 without
// FMA contraction (include/pcp/common/norm.hpp:102-112), eps-box exclusion
// (include/pcp/common/vector3d_queries.hpp:47-64).  The box lower bound is monotone in float, so
// pruning never changes the result.  Rows are the exact k nearest in ascending (d2, index) order; if
// several points tie EXACTLY with the k-th distance, which of them is kept is unspecified -- as in the
// reference, where it depends on heap order (linked_octree_node.hpp:479-489, linked_kdtree.hpp:483-488).
#include "pcpx_device.h"
#include "pcpx_eig3.h"

namespace pcpx {

namespace {

constexpr size_t STATS_EXTRA = 16 + 5 * 65536 - 8;  // diagnostic build: counters added in round 4 sit behind the per-wave records
constexpr u64 PAD_KEY = 0x7F800000FFFFFFFFull;  // (+inf, INVALID): larger than every real key, a finite double

// PAD_KEY in a fresh register pair: for the stores that refill buffer rows.  (As a plain constant hipcc hoists the pair out of
// the persistent loop and, short of registers, parks it in scratch: two v_mov where they are needed are cheaper.)
__device__ __forceinline__ u64 pad_key_here()
{
    u32 lo = static_cast<u32>(PAD_KEY), hi = static_cast<u32>(PAD_KEY >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (static_cast<u64>(hi) << 32) | lo;
}

// ---- selection network on 64-bit keys -------------------------------------------------------------
// Keys are (float32 d2 >= 0 bits) << 32 | u32: as IEEE doubles they are finite, non-negative and
// ordered like the integers, so min/max of the doubles is the integer compare-exchange in 2 VALU ops
// (a u64 compare + 4 selects costs 8 and two s_nop on gfx950).
__device__ __forceinline__ void ce(u64& a, u64& b)
{
    double x = __longlong_as_double(static_cast<long long>(a)), y = __longlong_as_double(static_cast<long long>(b));
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    a = static_cast<u64>(__double_as_longlong(lo));
    b = static_cast<u64>(__double_as_longlong(hi));
}
__device__ __forceinline__ u64 key_min(u64 a, u64 b)
{
    double x = __longlong_as_double(static_cast<long long>(a)), y = __longlong_as_double(static_cast<long long>(b));
    double lo;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    return static_cast<u64>(__double_as_longlong(lo));
}

template <int N>
__device__ __forceinline__ void bitonic_sort(u64 (&a)[N])
{
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                int l = i ^ j;
                if (l > i) {
                    if ((i & k) == 0) ce(a[i], a[l]);
                    else ce(a[l], a[i]);
                }
            }
        }
    }
}

// Sorting networks with the fewest compare-exchanges known for their size (8: 19 against the bitonic 24; 4: 5 against 6;
// checked on all 0/1 inputs): ascending.
template <int N>
__device__ __forceinline__ void sort_network(u64 (&a)[N])
{
    static_assert(N == 2 || N == 4 || N == 8, "sizes the compaction uses");
    if constexpr (N == 2) {
        ce(a[0], a[1]);
    } else if constexpr (N == 4) {
        ce(a[0], a[1]); ce(a[2], a[3]);
        ce(a[0], a[2]); ce(a[1], a[3]);
        ce(a[1], a[2]);
    } else {
        ce(a[0], a[1]); ce(a[2], a[3]); ce(a[4], a[5]); ce(a[6], a[7]);
        ce(a[0], a[2]); ce(a[1], a[3]); ce(a[4], a[6]); ce(a[5], a[7]);
        ce(a[1], a[2]); ce(a[5], a[6]); ce(a[0], a[4]); ce(a[3], a[7]);
        ce(a[1], a[5]); ce(a[2], a[6]);
        ce(a[1], a[4]); ce(a[3], a[6]);
        ce(a[2], a[4]); ce(a[3], a[5]);
        ce(a[3], a[4]);
    }
}

// SKIP: the first SKIP elements are known to hold 0, the smallest key there is (the best-list's sentinels when k < KCAP): a
// compare-exchange with one of them changes nothing and is left out (k = 15: 4 of the 32 of every merge).
// FRESH: only the last FRESH elements differ from an ascending list (the compaction has just put new keys there): in the first
// stage a pair that lies wholly below them is in order already (KCAP = 32, 8 new keys: 8 of the stage's 16).
template <int N, int SKIP = 0, int FRESH = N>
__device__ __forceinline__ void bitonic_merge(u64 (&a)[N])
{
#pragma unroll
    for (int j = N >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int l = i ^ j;
            if (l > i && i >= SKIP && (j != (N >> 1) || l >= N - FRESH)) ce(a[i], a[l]);
        }
    }
}

// Rows of the per-lane append buffer.  A leaf may append LEAF keys, so a compaction runs whenever a
// lane holds more than BUF - LEAF keys; fewer rows = less LDS per wave = more resident waves.
#ifndef PCPX_BUF16
#define PCPX_BUF16 10  // 10 rows x 512 B = 5 KB per wave = 7 waves/SIMD
#endif
#ifndef PCPX_BUF32
#define PCPX_BUF32 14  // (+ 2 rows of LDS that only the epilogue uses: lds_rows)
#endif
#ifndef PCPX_COMPACT_BY8
#define PCPX_COMPACT_BY8 1     // compaction in chunks of 8 keys
#endif
// (the multi-pass kernels -- k > 32 -- keep the 16-key compaction and the C++ accept path with its trash row)
#ifndef PCPX_BUF8
#define PCPX_BUF8 10   // k <= 8: 10 rows x 512 B = 5 KB per wave, 8 waves/SIMD
#endif
#ifndef PCPX_ASM_ACCEPT
#define PCPX_ASM_ACCEPT 1
#endif
#ifndef PCPX_SEED_DIRECT
#define PCPX_SEED_DIRECT 8  // largest KCAP whose seed leaves skip the append buffer (k <= 16 / 32: scratch in the seed phase)
#endif
#ifndef PCPX_PACKED_LEAVES
#define PCPX_PACKED_LEAVES 24  // a walk leaf that 2 ... this many lanes need is looked at eight needing lanes x eight points at a time (0: off)
#endif
#ifndef PCPX_PACKED_NODE
#define PCPX_PACKED_NODE 0  // a LAST-LEVEL node that at most this many lanes need is looked at TWO NEEDING LANES x ITS 32 POINTS at a time, its four
                             // leaf boxes never tested (0: off; <= PCPX_PACKED_LEAVES: the publish row's slots)
#endif
#ifndef PCPX_PACKED_NODE_KCAP
#define PCPX_PACKED_NODE_KCAP 16  // largest KCAP whose kernel has the node form (the k <= 32 kernel fills its buffer without the optimistic form's check: PCPX_PACKED_FREE)
#endif
#ifndef PCPX_PACKED_FREE
#define PCPX_PACKED_FREE 3  // k <= 16 kernel: free rows every needing lane has when a packed leaf starts (0: LEAF of them, like the other forms -- no key can then find its column full).  The k <= 32 kernel always waits for LEAF rows: its fold is twice the network, and what the optimistic fill loses there it does not win back (measured)
#endif
// Wave priorities (s_setprio) by phase of a group.  A fold is ~130 vector instructions back to back that wait for nothing; the
// walk and the leaf forms are short runs of arithmetic between loads whose round trips are what a wave's time is made of.  With the
// fold at a LOWER priority than the rest, a SIMD's issue slots go first to the waves that are about to issue a load, and the folds
// fill what is left: +2 % at k = 15 (uniform and clustered), +5 % at k = 32 (four waves per SIMD: fewer to hide a round trip
// behind), +1 % at k = 8.  The other way round (fold raised) -3 ... -6 %; a lower priority for the epilogue or the dense leaves
// as well: nothing, or -1 % at k = 8; base 3 instead of 1: the same (only the order matters).  profiles/experiments/README.md.
#ifndef PCPX_PRIO_BASE
#define PCPX_PRIO_BASE 1
#endif
#ifndef PCPX_PRIO_FOLD
#define PCPX_PRIO_FOLD 0
#endif
#ifndef PCPX_PRIO_PACKED
#define PCPX_PRIO_PACKED 2  // the packed leaf (three dependent LDS round trips on top of its loads) one step above the rest: +0.8 % (five rounds)
#endif
#ifndef PCPX_PRIO_WALK
#define PCPX_PRIO_WALK PCPX_PRIO_BASE
#endif
#ifndef PCPX_PRIO_DENSE
#define PCPX_PRIO_DENSE PCPX_PRIO_BASE
#endif
#ifndef PCPX_PRIO_EPI
#define PCPX_PRIO_EPI PCPX_PRIO_BASE
#endif
#ifndef PCPX_KNN_WPB16
#define PCPX_KNN_WPB16 4  // waves per workgroup of the k <= 16 kernel: its 11 rows x 512 B per wave fill the LDS allocation granule
                          // (1280 B) only in fours -- 7 waves per SIMD need <= 5851 B per wave
#endif
#ifndef PCPX_COMPACT_TIER4
#define PCPX_COMPACT_TIER4 8  // largest KCAP whose compaction has a four-key tier (k <= 8: +2 %; k <= 16: the branch costs the kernel
                              // scratch at 7 waves per SIMD and nothing at 6; k <= 32: no difference)
#endif
#ifndef PCPX_BY8_K32
#define PCPX_BY8_K32 1  // the chunk-of-8 compaction (with its PAD invariant established) for the single-pass k <= 32 kernel too
#endif
// rows of LDS per wave: the C++ accept path (multi-pass kernels only) stores rejected keys to a trash row, row BUF;
// the exec-masked path stores nothing for a rejected candidate
// (single-pass kernels: at least kcap / 2 rows -- after the search the column holds the row's kcap sorted positions, two per row)
// The packed leaf form (knn_group: packed_leaf) publishes the needing lanes' queries in rows of their own behind the buffer: 20 B per
// needing lane.  The kernels that have it: single-pass, k <= 16 and k <= 32 (the k <= 8 kernel has no LDS to spare at 8 waves per SIMD).
__host__ __device__ constexpr int pack_rows(bool multi, int kcap)
{
    return (PCPX_PACKED_LEAVES > 0 && PCPX_ASM_ACCEPT && !multi && kcap >= 16) ? (PCPX_PACKED_LEAVES * 20 + 511) / 512 : 0;
}
__host__ __device__ constexpr int lds_rows(int buf, bool multi, int kcap = 0)
{
    return (buf + ((multi || !PCPX_ASM_ACCEPT) ? 1 : 0) + pack_rows(multi, kcap)) > kcap / 2
               ? (buf + ((multi || !PCPX_ASM_ACCEPT) ? 1 : 0) + pack_rows(multi, kcap))
               : kcap / 2;
}
// waves per workgroup (every wave works alone; the workgroup only shares an LDS allocation)
__host__ __device__ constexpr int knn_wpb(int kcap, bool multi) { return (!multi && kcap == 16 && pack_rows(multi, kcap) > 0) ? PCPX_KNN_WPB16 : WAVES_PER_BLOCK; }
__host__ __device__ constexpr int buf_rows(int kcap) { return kcap <= 8 ? PCPX_BUF8 : kcap <= 16 ? PCPX_BUF16 : PCPX_BUF32; }

// Fold this lane's buffered keys (cnt <= BUF <= 16) into its sorted best-list.  All LDS traffic is
// unconditional (stale slots are masked to PAD_KEY in registers): no exec games.  The new keys are sorted
// ascending in registers -- only 8 of them when no lane of the wave holds more than 8 (the usual case late
// in a walk, when a leaf adds one or two keys per lane: 24 compare-exchanges instead of 80) -- then
// best[KCAP-1-j] = min(best[KCAP-1-j], new[j]) leaves the KCAP smallest of both as a bitonic sequence,
// which one merge network sorts.  At most 16 new keys are live beside best[]: KCAP 32 fits 128 VGPRs.
// key if j < cnt, else PAD_KEY -- by bit arithmetic: a v_cndmask_b32 whose mask is VCC (the form hipcc picks for
// `j < cnt ? key : PAD_KEY`) issues in 23 cycles on gfx950 against 4 for the SGPR-mask form and 2.4-4.3 for
// v_or_b32 / v_bfi_b32 (measured: tools/valu_rate.hip, profiles/r01_valu_issue_rates.txt)
// (hand-written: from the C++ bit arithmetic hipcc re-derives the compare + select.)
template <int J>
__device__ __forceinline__ u64 pad_from(u64 key, int cnt)
{
    u32 lo = static_cast<u32>(key), hi = static_cast<u32>(key >> 32), stale;
    asm("v_subrev_u32_e32 %[m], %[j1], %[cnt]\n\t"  // cnt - (J + 1) < 0 iff J >= cnt
        "v_ashrrev_i32_e32 %[m], 31, %[m]\n\t"    // all ones iff stale
        "v_or_b32_e32 %[lo], %[lo], %[m]\n\t"
        "v_bfi_b32 %[hi], %[m], %[padhi], %[hi]"    // stale ? padhi : hi
        : [lo] "+v"(lo), [hi] "+v"(hi), [m] "=&v"(stale)
        : [cnt] "v"(cnt), [j1] "n"(J + 1), [padhi] "s"(static_cast<u32>(PAD_KEY >> 32)));
    return (static_cast<u64>(hi) << 32) | lo;
}
template <int J0, int J1, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (J0 < J1) {
        f(std::integral_constant<int, J0>{});
        static_for<J0 + 1, J1>(f);
    }
}

// The same in chunks of 8 keys: rows 0..7, then (only if some lane holds more than 8) rows 8..BUF-1 -- never more than
// 8 new keys live beside best[], which is what lets the k <= 16 kernel fit 80 VGPRs (6 waves/SIMD).
// Invariant of the chunked path: every buffer slot that holds no key holds PAD_KEY (k_knn fills the rows once per
// wave, a compaction writes PAD_KEY back into the rows it has read) -- so rows are read unconditionally and need no
// masking by cnt: 8 LDS writes per chunk instead of 32 VALU instructions, and the vector ALU is the saturated unit.
// The eps-box test (a candidate inside the box |d.| < eps on all three axes is not a neighbour:
// include/pcp/common/vector3d_queries.hpp:47-64) can wait for the compaction: a point inside the box has d2 < thr
// = 3 eps^2 (1 + 1e-6), so only buffered keys below thr need the exact test, and after the chunk is sorted they are its
// first ones.  Self queries meet exactly one such key per lane (the query point itself), so two or three of a group's
// ~44 compactions take the slow path below, against a v_max3 + v_cmpx for each of its ~780 candidates.  A key that fails
// never reaches best[]: tau only ever comes from neighbours.  `on` is wave-uniform (launch-uniform, in fact).
#ifndef PCPX_DEFER_EPS
#define PCPX_DEFER_EPS 1
#endif
static_assert(!PCPX_DEFER_EPS || (PCPX_COMPACT_BY8 * PCPX_BY8_K32 != 0), "the deferred eps-box test lives in the chunk-of-8 compaction: compact() has none");
struct EpsFilter {
    bool on;
    float thr, eps, qx, qy, qz;
    const Leaf* leaves;
};
__device__ __forceinline__ float key_d2(u64 key) { return __uint_as_float(static_cast<u32>(key >> 32)); }

// nw[0 .. N) ascending, read from rows r0 .. r0 + R of the column (which hold PAD_KEY again): drop the keys inside the eps-box
template <int R, int N>
__device__ __forceinline__ void drop_eps_box(u64 (&nw)[N], u64* __restrict__ col_r0, const EpsFilter& f)
{
    static_assert(R <= N, "rows");
    if (!any_lane(key_d2(nw[0]) < f.thr)) return;  // (PAD_KEY: d2 = +inf)
    // back into the rows in ascending order (the real keys of the chunk are at most R and come first): the loop below is
    // rare and runs on LDS so that it costs the compaction no registers
#pragma unroll
    for (int j = 0; j < R; ++j) col_r0[j * 64] = nw[j];
#pragma unroll 1
    for (int j = 0; j < R; ++j) {
        const u64 key = col_r0[j * 64];
        const bool near = key_d2(key) < f.thr;
        if (!any_lane(near)) break;  // ascending: none of the later keys either
        if (near) {
            const u32 ps = static_cast<u32>(key);
            const Leaf& lf = f.leaves[ps / LEAF];
            const float dx = lf.x[ps % LEAF] - f.qx, dy = lf.y[ps % LEAF] - f.qy, dz = lf.z[ps % LEAF] - f.qz;
            const float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
            if (!(m >= f.eps)) col_r0[j * 64] = pad_key_here();
        }
    }
    const u64 pad = pad_key_here();
#pragma unroll
    for (int j = 0; j < N; ++j) nw[j] = j < R ? col_r0[j * 64] : pad;
#pragma unroll
    for (int j = 0; j < R; ++j) col_r0[j * 64] = pad;
    sort_network<N>(nw);
}

template <int KCAP, int BUF, int NZ>
__device__ __forceinline__ void compact_by8(u64 (&best)[KCAP], u64* __restrict__ col, int& cnt, const EpsFilter& f)
{
    static_assert(BUF >= 8 && BUF <= 16 && KCAP >= 8, "rows");
    if (KCAP <= PCPX_COMPACT_TIER4 && !any_lane(cnt > 4)) {  // about half of the compactions of a walk: four rows, five compare-exchanges
        u64 nw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) nw[j] = col[j * 64];
        const u64 pad = pad_key_here();
#pragma unroll
        for (int j = 0; j < 4; ++j) col[j * 64] = pad;
        sort_network<4>(nw);
        if (f.on) drop_eps_box<4, 4>(nw, col, f);
#pragma unroll
        for (int j = 0; j < 4; ++j) best[KCAP - 1 - j] = key_min(best[KCAP - 1 - j], nw[j]);
    } else {
        u64 nw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) nw[j] = col[j * 64];
        const u64 pad = pad_key_here();
#pragma unroll
        for (int j = 0; j < 8; ++j) col[j * 64] = pad;
        sort_network<8>(nw);
        if (f.on) drop_eps_box<8, 8>(nw, col, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) best[KCAP - 1 - j] = key_min(best[KCAP - 1 - j], nw[j]);
    }
    bitonic_merge<KCAP, NZ, 8>(best);  // (one copy for both tiers: only best[] crosses the join)
    if (BUF > 8 && any_lane(cnt > 8)) {
        constexpr int R = BUF > 8 ? BUF - 8 : 1;               // rows of the second chunk
        constexpr int N = R <= 2 ? 2 : R <= 4 ? 4 : 8;         // its sorting network (10 rows: one compare-exchange)
        u64 nw[N];
        const u64 pad = pad_key_here();
#pragma unroll
        for (int j = 0; j < N; ++j) nw[j] = j < R ? col[(8 + j) * 64] : pad;
#pragma unroll
        for (int j = 8; j < BUF; ++j) col[j * 64] = pad;
        sort_network<N>(nw);
        if (f.on) drop_eps_box<R, N>(nw, col + 8 * 64, f);
#pragma unroll
        for (int j = 0; j < N; ++j) best[KCAP - 1 - j] = key_min(best[KCAP - 1 - j], nw[j]);
        bitonic_merge<KCAP, NZ, N>(best);
    }
    cnt = 0;
}

template <int KCAP, int BUF>
__device__ __forceinline__ void compact(u64 (&best)[KCAP], u64* __restrict__ col, int& cnt)
{
    static_assert(BUF >= 8 && BUF <= 16 && KCAP >= 8, "rows");
    constexpr int TOP = KCAP < 16 ? KCAP : 16;  // the KCAP smallest of best[] and the new keys are among the TOP smallest new ones
    u64 nw[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) nw[j] = col[j * 64];
    static_for<0, 8>([&](auto J) { nw[J] = pad_from<J>(nw[J], cnt); });
    if (!any_lane(cnt > 8)) {
        u64 lo8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lo8[j] = nw[j];
        bitonic_sort<8>(lo8);
#pragma unroll
        for (int j = 0; j < 8; ++j) best[KCAP - 1 - j] = key_min(best[KCAP - 1 - j], lo8[j]);  // the other 8 would be min(., PAD)
    } else {
#pragma unroll
        for (int j = 8; j < 16; ++j) nw[j] = j < BUF ? col[j * 64] : PAD_KEY;
        static_for<8, BUF>([&](auto J) { nw[J] = pad_from<J>(nw[J], cnt); });
        bitonic_sort<16>(nw);
#pragma unroll
        for (int j = 0; j < TOP; ++j) best[KCAP - 1 - j] = key_min(best[KCAP - 1 - j], nw[j]);
    }
    cnt = 0;
    bitonic_merge<KCAP>(best);
}

// ---- exec-masked append (hand-written: hipcc has no way to emit v_cmpx from C++) ---------------------
// Appends key (d2, pos) to the lane's LDS column at byte address `wa` iff d2 <= tau and m >= eps, then
// advances wa by one row (512 B).  v_cmpx writes EXEC directly, so the two tests cost 2 VALU and the LDS
// write and the address bump simply run under the narrowed EXEC; EXEC is restored before leaving.
// The C++ equivalent (two compares, three selects, address math, count) costs 9 VALU per candidate.
// gfx9 v_cmpx also writes VCC.  No wait states are needed between the VALU EXEC write and the DS issue.
// (tried: jumping over the eps test / LDS write / address bump with s_cbranch_execz when no lane is within tau
// of a point: the branch costs more than the skipped issue slots, 998 vs 1048 Mq/s)
__device__ __forceinline__ u64 save_exec()
{
    u64 saved;
    asm volatile("s_mov_b64 %0, exec" : "=s"(saved));
    return saved;
}
// `saved` = EXEC on entry (save_exec() once per leaf: the walk is wave-uniform, EXEC does not change inside
// a leaf); also steps pos to the next point of the leaf, outside the mask.
__device__ __forceinline__ void append_if(float d2, float tau, float dx, float dy, float dz, float eps, u32& pos, u32& wa,
                                          u64 saved)
{
    float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    asm volatile("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                 "v_cmpx_le_f32_e32 %[eps], %[m]\n\t"
                 "ds_write2_b32 %[wa], %[pos], %[d2] offset1:1\n\t"
                 "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_u32_e32 %[pos], 1, %[pos]"
                 : [wa] "+v"(wa), [pos] "+v"(pos)
                 : [d2] "v"(d2), [tau] "v"(tau), [m] "v"(m), [eps] "s"(eps), [sv] "s"(saved)
                 : "vcc", "memory");
}

// The same without the eps-box test (EpsFilter: it waits for the compaction).
__device__ __forceinline__ void append_if_within(float d2, float tau, u32& pos, u32& wa, u64 saved)
{
    asm volatile("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                 "ds_write2_b32 %[wa], %[pos], %[d2] offset1:1\n\t"
                 "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_u32_e32 %[pos], 1, %[pos]"
                 : [wa] "+v"(wa), [pos] "+v"(pos)
                 : [d2] "v"(d2), [tau] "v"(tau), [sv] "s"(saved)
                 : "vcc", "memory");
}
// Four candidates of a leaf in one statement: distances (dx*dx + dy*dy + dz*dz, three roundings as in norm.hpp) and the
// exec-masked append of each.  One statement, because between two asm statements of which the second reads a register the
// first writes hipcc puts an s_nop (seven per leaf with one statement per candidate), and because it would otherwise compute
// all eight distances first: the scalar side of this kernel is as loaded as its vector side, and the registers are the
// best-list's.  `pos` steps by one per candidate for every lane; NaN padding points fail d2 <= tau.
__device__ __forceinline__ void append4_if_within(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                  float qx, float qy, float qz, float tau, u32& pos, u32& wa, u64 saved)
{
    float a, b, c;
#define PCPX_CAND(J)                                                  \
    "v_sub_f32_e32 %[a], %[x" #J "], %[qx]\n\t"                       \
    "v_sub_f32_e32 %[b], %[y" #J "], %[qy]\n\t"                       \
    "v_sub_f32_e32 %[c], %[z" #J "], %[qz]\n\t"                       \
    "v_mul_f32_e32 %[a], %[a], %[a]\n\t"                              \
    "v_mul_f32_e32 %[b], %[b], %[b]\n\t"                              \
    "v_add_f32_e32 %[a], %[a], %[b]\n\t"                              \
    "v_mul_f32_e32 %[c], %[c], %[c]\n\t"                              \
    "v_add_f32_e32 %[a], %[a], %[c]\n\t"                              \
    "v_cmpx_le_f32_e32 %[a], %[tau]\n\t"                              \
    "ds_write2_b32 %[wa], %[pos], %[a] offset1:1\n\t"                 \
    "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"                           \
    "s_mov_b64 exec, %[sv]\n\t"                                       \
    "v_add_u32_e32 %[pos], 1, %[pos]\n\t"
    asm volatile(PCPX_CAND(0) PCPX_CAND(1) PCPX_CAND(2) PCPX_CAND(3)
                 : [wa] "+v"(wa), [pos] "+v"(pos), [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c)
                 : [x0] "s"(x[0]), [x1] "s"(x[1]), [x2] "s"(x[2]), [x3] "s"(x[3]), [y0] "s"(y[0]), [y1] "s"(y[1]), [y2] "s"(y[2]),
                   [y3] "s"(y[3]), [z0] "s"(z[0]), [z1] "s"(z[1]), [z2] "s"(z[2]), [z3] "s"(z[3]), [qx] "v"(qx), [qy] "v"(qy),
                   [qz] "v"(qz), [tau] "v"(tau), [sv] "s"(saved)
                 : "vcc", "memory");
#undef PCPX_CAND
}
__device__ __forceinline__ void append_if_within_shell(float d2, float tau, float lo, u32& pos, u32& wa, u64 saved)
{
    asm volatile("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                 "v_cmpx_lt_f32_e32 %[lo], %[d2]\n\t"
                 "ds_write2_b32 %[wa], %[pos], %[d2] offset1:1\n\t"
                 "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_u32_e32 %[pos], 1, %[pos]"
                 : [wa] "+v"(wa), [pos] "+v"(pos)
                 : [d2] "v"(d2), [tau] "v"(tau), [lo] "s"(lo), [sv] "s"(saved)
                 : "vcc", "memory");
}

// The same for the later walk rounds: additionally lo < d2 (only the new shell (lo, tau] is accepted).
__device__ __forceinline__ void append_if_shell(float d2, float tau, float lo, float dx, float dy, float dz, float eps, u32& pos,
                                                u32& wa, u64 saved)
{
    float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    asm volatile("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                 "v_cmpx_lt_f32_e32 %[lo], %[d2]\n\t"
                 "v_cmpx_le_f32_e32 %[eps], %[m]\n\t"
                 "ds_write2_b32 %[wa], %[pos], %[d2] offset1:1\n\t"
                 "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_u32_e32 %[pos], 1, %[pos]"
                 : [wa] "+v"(wa), [pos] "+v"(pos)
                 : [d2] "v"(d2), [tau] "v"(tau), [lo] "s"(lo), [m] "v"(m), [eps] "s"(eps), [sv] "s"(saved)
                 : "vcc", "memory");
}

// ------------------------------------------------------------------------------------------------
// kNN (+ fused PCA normals)
// ------------------------------------------------------------------------------------------------
#ifndef PCPX_MINW8
#define PCPX_MINW8 8   // k <= 8 kernel: 63 VGPRs = 8 waves/SIMD
#endif
#ifndef PCPX_MINW32
#define PCPX_MINW32 4  // k <= 32 kernel: <= 128 VGPRs = 4 waves/SIMD
#endif
#ifndef PCPX_MINW
#define PCPX_MINW 7  // k <= 16 kernel: <= 72 VGPRs = 7 waves/SIMD
#endif
// PCPX_CAP_MULT x the median of the finite seeded taus of a sample of the wave's valid lanes (every fourth lane:
// 16 readlanes; inf if no lane has a finite tau): rank every sampled value by counting, pick the middle one.
// The cap only steers the work, never the result (a lane that fails the cap goes round again).
#ifdef PCPX_CAP_MULT
template <int KCAP>
constexpr float cap_mult() { return PCPX_CAP_MULT; }
#else
template <int KCAP>
constexpr float cap_mult() { return KCAP <= 8 ? 1.375f : KCAP <= 16 ? 1.25f : 1.125f; }
#endif
#ifndef PCPX_CAP_GROW
#define PCPX_CAP_GROW 4.f  // radius^2 growth per further round
#endif
template <int KCAP>
__device__ __forceinline__ float wave_radius_cap(float tau, bool valid, u32 lane)
{
    const float inf = std::numeric_limits<float>::infinity();
    // `lane` is made opaque here: otherwise the sixteen `l < lane` masks below are loop invariants of the persistent
    // kernel and hipcc keeps them in 32 SGPRs for its whole lifetime
    asm volatile("" : "+v"(lane));
    const bool sample = (lane & 3u) == 1u;
    const float x = (valid && tau < inf) ? tau : inf;
    const u64 finite = __builtin_amdgcn_ballot_w64(sample && x < inf);
    const u32 nfinite = static_cast<u32>(__builtin_popcountll(finite));
    if (nfinite == 0) {  // no finite sample: any finite lane, or none
        const u64 any = __builtin_amdgcn_ballot_w64(x < inf);
        if (any == 0) return inf;
        return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), static_cast<u32>(__builtin_ctzll(any)))) * cap_mult<KCAP>();
    }
    u32 rank = 0;
#pragma unroll
    for (u32 l = 1; l < 64; l += 4) {
        const float other = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), l));
        rank += (other < x || (other == x && l < lane)) ? 1u : 0u;
    }
    const u64 is_med = __builtin_amdgcn_ballot_w64(sample && x < inf && rank == nfinite / 2);
    const float med = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), static_cast<u32>(__builtin_ctzll(is_med))));
    return med * cap_mult<KCAP>();
}

// k > 32: pass p of a multi-pass search returns the (at most 32) smallest keys strictly greater than the
// last key of pass p-1, so ceil(k/32) passes enumerate the k nearest in order.  Keys (d2, sorted position)
// are kept raw per query slot; k_assemble turns them into rows.
struct MultiPass {
    const u64* lo = nullptr;  // per query slot: exclusive lower bound (nullptr on the first pass)
    u64* lo_out = nullptr;    // per query slot: last key of this pass
    u64* keys = nullptr;      // per query slot: `stride` keys; this pass writes [offset, offset + k)
    u32 stride = 0;
    u32 offset = 0;
    u32 slot0 = 0;            // query slot of the first group of the launch
};

// The arguments of k_knn, one struct: the kernel's only parameter, so the struct IS the kernarg segment.  The tree, the group
// range, eps and the queue are used all through a group and live in scalar registers; everything else -- the query arrays, k, where
// the rows go, the multi-pass bookkeeping: ~50 dwords -- is needed for a few instructions at a group's start and in its epilogue.
// Taken as ordinary parameters hipcc loads all of it at kernel entry, finds no registers for it across the persistent loop and parks
// it in lanes of a VGPR (42-58 "SGPR spills" per kernel, ~100 v_readlane per group).  knn_group reads those fields through the
// kernarg pointer instead (constant address space: scalar loads), made opaque where a group starts and where its epilogue starts so
// that nothing read through it lives across the search.
struct KnnArgs {
    TreeView t;
    QueryView qv;
    u32 group_first, group_end, k;
    float eps, eps_thr;
    KnnOutputs o;
    MultiPass mp;
    u32* queue;
    unsigned long long* stats;
    KnnSchedule sch;
    u32* queue_clear;  // the other set of queue counters: zeroed by this launch for the next one
};
typedef const __attribute__((address_space(4))) KnnArgs* knn_args_ptr;
__device__ __forceinline__ knn_args_ptr knn_args_here()
{
    u64 a = reinterpret_cast<u64>(__builtin_amdgcn_kernarg_segment_ptr());
    u32 lo = static_cast<u32>(a), hi = static_cast<u32>(a >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));  // (opaque: what is loaded through it from here on is loaded here, not at kernel entry)
    return reinterpret_cast<knn_args_ptr>((static_cast<u64>(hi) << 32) | lo);
}
template <class T>
__device__ __forceinline__ T cold(const __attribute__((address_space(4))) T* field)
{
    static_assert(sizeof(T) % 4 == 0, "dwords");
    T out;
    u32* o = reinterpret_cast<u32*>(&out);
    const __attribute__((address_space(4))) u32* c = reinterpret_cast<const __attribute__((address_space(4))) u32*>(field);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = c[i];
    return out;
}

// One query group (64 curve-consecutive queries, one per lane) from start to finish.
// NZ: the caller guarantees k <= KCAP - NZ (NZ sentinel slots at the bottom of the best-list hold 0 throughout)
// DIAG: 0 = the product; 1 = the diagnostic build (pcpx_debug_knn_stats: event counts and clocks per phase, summed over the launch);
// 2 = the product's walk with its events counted per group (launch_knn_cost_sample: what a work-balanced shard cut is made from --
// integers that depend on the tree and the question alone, so every rank of a job computes the same ones); `slot` = where its counts go
template <int KCAP, bool SELF, int DIAG, bool MULTI, bool EPS_EACH, int NZ>
__device__ __forceinline__ void knn_group(const TreeView& tree, const u32 g, const float eps, const float eps_thr,
                                          unsigned long long* __restrict__ stats, u64* __restrict__ col, float* __restrict__ pub, const u32 lane,
                                          const u32 slot, const u32 part)
{
    constexpr bool STATS = DIAG == 1, COST = DIAG == 2;
    constexpr int BUF = buf_rows(KCAP);  // usable rows (the multi-pass kernels have one more: the trash row BUF)
    // (the depth is made opaque per group, like k below: depth - 1, its multiples and the level masks are otherwise computed once
    //  per kernel and parked in spilled scalar registers)
    TreeView t = tree;
    asm volatile("" : "+s"(t.depth), "+s"(t.nodes));
    if (PCPX_PRIO_BASE != PCPX_PRIO_EPI || PCPX_PRIO_BASE != PCPX_PRIO_FOLD || PCPX_PRIO_BASE != PCPX_PRIO_DENSE) __builtin_amdgcn_s_setprio(PCPX_PRIO_BASE);
    // the cold arguments (KnnArgs), as this group's start sees them: dead before the search begins
    const knn_args_ptr ka = knn_args_here();
    const u32 k_arg = ka->k;
    const u32 pos_lo = ka->o.pos_lo, pos_hi = ka->o.pos_hi;
    MultiPass mp;
    if (MULTI) mp = cold(&ka->mp);
    const float* const known_d2 = STATS ? cold(&ka->o.d2) : nullptr;
    // k is made opaque per group: everything derived from it alone -- the initial best-list (slots below KCAP - k hold 0, the
    // others PAD_KEY) and the epilogue's per-slot predicates -- is otherwise a loop invariant of the persistent kernel and is
    // held in 4 x KCAP SGPRs from the first group to the last; the walk's own scalars then spill into VGPR lanes, and those
    // VGPRs are what pushed the k <= 16 kernel into scratch
    u32 k_here = k_arg;
    asm volatile("" : "+s"(k_here));
    const u32 k = k_here;
    // STATS build only (pcpx_debug_knn_stats): [0] leaves visited, [1] node expansions, [2] compactions,
    // [3] keys appended, [4] waves, [5] seed leaves
    //                                           [6] groups that needed the second (uncapped) walk round
    u32 st_leaves = 0, st_expand = 0, st_compact = 0, st_app = 0, st_round2 = 0, st_seed_compact = 0, st_seed_app = 0, st_sparse = 0, st_owners = 0;
    u32 st_steps = 0;  // COST: steps (eight needing lanes each) of the packed leaves
    // [7] shader cycles in the walker (pop + node expansions), [8] in compactions, [9] in leaf candidates,
    // [10] in the whole search loop, [11] whole group incl. the epilogue (id gather, tie repair, stores, fused normal)
    unsigned long long tc_walk = 0, tc_compact = 0, tc_leaf = 0, tc0 = 0, tc_mark = 0, tc_later_mark = 0;
    u32 st_later_lanes = 0;
    if (STATS) tc0 = __builtin_amdgcn_s_memtime();
    u32 tick0 = 0;  // (the low word: a group takes far less than 2^32 cycles)
    if (!MULTI && DIAG == 0) tick0 = static_cast<u32>(__builtin_amdgcn_s_memtime());

    // ---- my query ----
    u32 p_here = g * GROUP + lane;
    asm volatile("" : "+v"(p_here));  // (opaque: or the lane-only pieces of every address formed from p stay pinned in VGPR pairs across groups)
    const u32 p = p_here;
    const u32 nq = SELF ? t.n : ka->qv.nq;
    // (self queries: only the positions [pos_lo, pos_hi) the caller asked for -- a slice that starts or ends inside a group)
    // (`part` = 0: the whole group; 1 ... 8: only its lanes 8 (part - 1) ... 8 part - 1 -- an eighth of a LONG group, whose other eighths
    //  other waves answer at the same time; 9 ... 12: a quarter, 13 / 14: a half -- the groups of a launch's last, partly filled round:
    //  order_entries.  From p, which is opaque per group: from `lane` it is one more value kept from group to group.)
    const bool mine_of_part = part <= 8u ? ((p >> 3) & 7u) + 1u == part : part <= 12u ? ((p >> 4) & 3u) + 9u == part : ((p >> 5) & 1u) + 13u == part;
    const bool valid = p < nq && (!SELF || p - pos_lo < pos_hi - pos_lo) && (part == 0u || mine_of_part);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
        if (SELF) {
            const Leaf& lf = t.leaves[p / LEAF];
            qx = lf.x[p % LEAF];
            qy = lf.y[p % LEAF];
            qz = lf.z[p % LEAF];
        } else {
            qx = cold(&ka->qv.qx)[p];
            qy = cold(&ka->qv.qy)[p];
            qz = cold(&ka->qv.qz)[p];
        }
    }
    // (the query's coordinates are waited for here, once: left pending, hipcc puts three s_waitcnt vmcnt in front of their first
    //  use in every block of the search loop -- six scalar-pipe instructions per leaf + expansion)
    asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));
    u64 lo_key = 0;
    bool has_lo = false;
    if (MULTI && mp.lo && valid) {
        lo_key = mp.lo[p - mp.slot0];
        has_lo = true;
    }
    // best-list: KCAP-k leading zero keys act as -inf sentinels so that tau is always best[KCAP-1]
    u64 best[KCAP];
#pragma unroll
    for (int j = 0; j < KCAP; ++j) best[j] = (j < KCAP - static_cast<int>(k)) ? 0ull : PAD_KEY;
    const float inf = std::numeric_limits<float>::infinity();
    float tau = valid ? inf : -1.f;  // -1: an idle lane never accepts a candidate nor needs a node
    // diagnostic build only: o.d2 holds the FINAL rows of an earlier run; starting from the true k-th distance
    // measures how much of the walk is spent before tau has tightened (the floor any visiting order can reach)
    float tau_known = inf;
    if (STATS && known_d2 && valid && SELF) tau_known = known_d2[static_cast<u64>(t.leaves[p / LEAF].id[p % LEAF]) * k + (k - 1)];
    bool active = valid;             // lanes still searching (the second walk round keeps only the failed ones)
    int cnt = 0;
    const u32 col_addr = lds_address(col);  // byte address of row 0 of this lane's column
    u32 wa = col_addr;                      // byte address of the next free row (PCPX_ASM_ACCEPT)
    // a lane's column starts at lds_row0 + 8 * lane (< lds_row0 + 512), so "more than c keys buffered" is a comparison of
    // wa with a wave-uniform bound: no per-lane threshold register
    const u32 lds_row0 = __builtin_amdgcn_readfirstlane(col_addr) - 8u * __builtin_amdgcn_readfirstlane(lane);
    const u32 wa_full = lds_row0 + (static_cast<u32>(BUF - LEAF + 1) << 9);  // wa >= this: a leaf might not fit any more
    const u32 wa_end = lds_row0 + (static_cast<u32>(BUF) << 9);               // a key address >= this: beyond the column's last row
    constexpr int packed_free = KCAP <= 16 ? PCPX_PACKED_FREE : 0;  // (see PCPX_PACKED_FREE)
    const u32 wa_packed_full = packed_free > 0 ? lds_row0 + (static_cast<u32>(BUF - packed_free + 1) << 9) : wa_full;

    auto need = [&](const NodeBox& b) { return box_d2(b, qx, qy, qz) <= tau; };
    // single-pass kernels: the eps-box test waits for the compaction, unless the launcher picked the EPS_EACH form (launch_knn_t)
    const EpsFilter eps_filter{PCPX_ASM_ACCEPT && !MULTI && !EPS_EACH, eps_thr, eps, qx, qy, qz, t.leaves};

    // ---- seed range: the 64-point chunk at the group's own curve position ----
    u32 s0, s1;
    if (SELF) s0 = g * LEAVES_PER_GROUP;
    else s0 = load_const(cold(&ka->qv.seed) + g);  // (a scalar load: through a plain pointer hipcc makes this wave-uniform read a vector load, and the seed loop with it)
    s1 = s0 + LEAVES_PER_GROUP < t.nleaves ? s0 + LEAVES_PER_GROUP : t.nleaves;
    if (s0 > s1) s0 = s1;
    // leaves before and after the group's own chunk that are also processed before the walk: they are curve neighbours the walk
    // would visit anyway, and seeing them first tightens tau sooner.  How many pays depends on k.
#ifdef PCPX_SEED_EXTRA
    constexpr u32 seed_extra = PCPX_SEED_EXTRA;
#else
    constexpr u32 seed_extra = KCAP <= 8 ? 0u : KCAP <= 16 ? 2u : 4u;
#endif
    static_assert(seed_extra % UNIT_LEAVES == 0 && LEAVES_PER_GROUP % UNIT_LEAVES == 0, "the seed range is whole units of the tree's bottom level");
    if (seed_extra > 0) {
        s0 = s0 > seed_extra ? s0 - seed_extra : 0u;
        s1 = s1 + seed_extra < t.nleaves ? s1 + seed_extra : t.nleaves;
    }

    // Phases: the seed leaves, then walk rounds with a growing radius.  Each phase has its own loop (the one loop with mode
    // flags that served them all until round 3 spent more scalar instructions steering itself than the walk needs: the scalar
    // side of this kernel is as loaded as its vector side, profiles/experiments/README.md); a trip of a loop fetches the next
    // leaf, folds the append buffer into the best-list if the leaf might not fit (or, at the end of the phase, if any key is
    // buffered), then runs the leaf's candidates.
    // In the first walk round no lane searches farther than `cap` = cap_mult (1.125 ... 1.375) x the wave's median seeded tau: a
    // lane whose 64-point seed chunk lies across a jump of the curve (rare on the Hilbert curve, the rule on a Z-order) starts
    // with a tau hundreds of times too large and would drag the whole wave through thousands of leaves (measured: 7 ms groups
    // against a 0.37 ms mean).  After a round a lane is exact iff its k-th distance <= cap (then all its k nearest are within
    // cap, and everything within cap has been visited).  Lanes that fail -- points in genuinely sparse places -- go round again
    // with cap x 4, accepting only the new shell lo_d2 < d2 <= cap so that nothing is seen twice, until they verify or the cap
    // covers the whole cloud.
    float cap = inf;       // wave-uniform; inf = no cap
    float lo_d2 = -1.f;    // wave-uniform; later rounds accept only d2 > lo_d2
    constexpr bool fast = PCPX_ASM_ACCEPT && !MULTI;  // keeps only the write address `wa`; the other accept paths only `cnt`

    // (diagnostic build: the dense forms count a key where it is accepted, the packed form where it is folded -- fold() then counts EVERY
    //  key of the column, so with packed leaves in the kernel the dense forms' own counts are left out: `count_at_accept`)
    constexpr bool packed_keys_counted_at_folds = STATS && pack_rows(MULTI, KCAP) > 0 && PCPX_ASM_ACCEPT && !MULTI && !EPS_EACH;
    constexpr bool count_at_accept = STATS && !packed_keys_counted_at_folds;
    // fold the buffered keys into the best-list (one copy of the selection network per call site)
    auto fold = [&](bool in_seed_phase) {
        if (STATS) tc_mark = __builtin_amdgcn_s_memtime();
        if (fast) cnt = static_cast<int>((wa - col_addr) >> 9);
        if (STATS && packed_keys_counted_at_folds) st_app += static_cast<u32>(cnt);  // (the packed leaves' keys: see the leaf loop)
        if (PCPX_PRIO_FOLD != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_FOLD);
        if (PCPX_COMPACT_BY8 && (KCAP <= 16 || (PCPX_BY8_K32 && !MULTI))) compact_by8<KCAP, BUF, NZ>(best, col, cnt, eps_filter);
        else compact<KCAP, BUF>(best, col, cnt);
        if (PCPX_PRIO_FOLD != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_BASE);
        float nt = __uint_as_float(static_cast<u32>(best[KCAP - 1] >> 32));
        tau = active ? fminf(nt, cap) : -1.f;
        if (STATS) tau = fminf(tau, tau_known);
        wa = col_addr + (static_cast<u32>(cnt) << 9);
        if (COST) ++st_compact;
        if (STATS) {
            ++st_compact;
            if (in_seed_phase) ++st_seed_compact;
            // the clock read must not be scheduled ahead of the merge network: make it depend on tau
            asm volatile("" ::"v"(tau));
            tc_compact += __builtin_amdgcn_s_memtime() - tc_mark;
        }
    };
    // before a leaf: fold if some lane could not take LEAF more keys; at the end of a phase: if any lane holds a key
    auto fold_if_needed = [&](bool before_leaf, bool in_seed_phase) {
        if (!fast) wa = col_addr + (static_cast<u32>(cnt) << 9);
        if (any_lane(wa >= (before_leaf ? wa_full : lds_row0 + 512u))) fold(in_seed_phase);
    };

    // candidates of one leaf: SMEM broadcast, branch-free accept; `shell`: a later walk round
    auto candidates = [&](const u32 leaf, const bool shell) {
        if (COST) ++st_leaves;
        if (STATS) {
            ++st_leaves;
            tc_mark = __builtin_amdgcn_s_memtime();
        }
        const Leaf lf = load_const(t.leaves + leaf);  // (a vector-memory fetch of the record measured the same)
        const u32 posbase = leaf * LEAF;
        if (PCPX_PRIO_DENSE != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_DENSE);
        // copies of the candidate loop, switched per leaf (hipcc otherwise re-tests the mode per point)
        if (fast && eps_filter.on && !shell && !STATS) {
            u32 posv = posbase;
            const u64 saved = save_exec();
            static_assert(LEAF == 8, "two statements of four candidates");
            // (two candidates at a time with packed-float arithmetic -- 18 instead of 26 instructions per pair -- measured in round 4: the
            //  same rate; profiles/experiments/README.md)
            append4_if_within(lf.x, lf.y, lf.z, qx, qy, qz, tau, posv, wa, saved);
            append4_if_within(lf.x + 4, lf.y + 4, lf.z + 4, qx, qy, qz, tau, posv, wa, saved);
        } else if (fast && eps_filter.on && !shell) {
            u32 posv = posbase;
            const u64 saved = save_exec();
#pragma unroll
            for (int j = 0; j < LEAF; ++j) {
                float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
                float d2 = sq3(dx, dy, dz);
                if (count_at_accept) st_app += (d2 <= tau) ? 1u : 0u;
                append_if_within(d2, tau, posv, wa, saved);  // NaN padding points fail d2 <= tau
            }
        } else if (fast && eps_filter.on) {
            u32 posv = posbase;
            const u64 saved = save_exec();
#pragma unroll
            for (int j = 0; j < LEAF; ++j) {
                float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
                float d2 = sq3(dx, dy, dz);
                if (count_at_accept) st_app += (d2 <= tau && d2 > lo_d2) ? 1u : 0u;
                append_if_within_shell(d2, tau, lo_d2, posv, wa, saved);
            }
        } else if (fast && !shell) {
            u32 posv = posbase;
            const u64 saved = save_exec();
#pragma unroll
            for (int j = 0; j < LEAF; ++j) {
                float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
                float d2 = sq3(dx, dy, dz);
                if (count_at_accept) st_app += (d2 <= tau && fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) >= eps) ? 1u : 0u;
                append_if(d2, tau, dx, dy, dz, eps, posv, wa, saved);  // NaN padding points fail d2 <= tau
            }
        } else if (fast) {  // later rounds: only the shell (lo_d2, tau]
            u32 posv = posbase;
            const u64 saved = save_exec();
#pragma unroll
            for (int j = 0; j < LEAF; ++j) {
                float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
                float d2 = sq3(dx, dy, dz);
                if (count_at_accept) st_app += (d2 <= tau && d2 > lo_d2 && fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) >= eps) ? 1u : 0u;
                append_if_shell(d2, tau, lo_d2, dx, dy, dz, eps, posv, wa, saved);
            }
        } else {
#pragma unroll
            for (int j = 0; j < LEAF; ++j) {
                float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
                float d2 = sq3(dx, dy, dz);
                float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
                const u64 key = (static_cast<u64>(__float_as_uint(d2)) << 32) | (posbase + j);
                float m2 = d2 <= tau ? m : -1.f;  // NaN padding points fail here
                bool acc = m2 >= eps && d2 > lo_d2;  // outside the eps-box (eps >= 0); not seen in round one
                if (MULTI) acc = acc && (!has_lo || key > lo_key);
                int slot = acc ? cnt : BUF;
                col[slot * 64] = key;
                cnt += acc ? 1 : 0;
                if (count_at_accept) st_app += acc ? 1u : 0u;
            }
        }
        if (PCPX_PRIO_DENSE != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_BASE);
        if (STATS) {
            asm volatile("" ::"v"(wa), "v"(cnt));
            tc_leaf += __builtin_amdgcn_s_memtime() - tc_mark;
        }
    };

    // ---- the seed leaves ----
    // Early in the seed phase nearly every lane accepts nearly every point (tau starts at +inf), so a seed leaf's eight keys
    // skip the append buffer: built in registers, sorted, filtered for the eps-box and merged like a chunk of the buffer
    // (the same best-list as accepting them one by one: a key beyond tau falls off its end).  Not for the cloud's last leaf
    // (its padding slots are NaN, which the compare-exchange network cannot carry) and not in the per-candidate eps form.
    constexpr bool direct_seeds = KCAP <= PCPX_SEED_DIRECT && fast && !EPS_EACH && !STATS && PCPX_COMPACT_BY8 && (KCAP <= 16 || PCPX_BY8_K32);
    auto seed_direct = [&](const u32 leaf) {
        const Leaf lf = load_const(t.leaves + leaf);
        const u32 posbase = leaf * LEAF;
        u64 nw[LEAF];
        float qx_here = qx;
#pragma unroll
        for (int j = 0; j < LEAF; ++j) {
            float dx = lf.x[j] - qx_here, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
            nw[j] = (static_cast<u64>(__float_as_uint(sq3(dx, dy, dz))) << 32) | (posbase + j);
            asm volatile("" : "+v"(qx_here));  // one key's differences at a time (all eight at once cost registers the best-list needs)
        }
        sort_network<LEAF>(nw);
        if (eps_filter.on) drop_eps_box<LEAF, LEAF>(nw, col, eps_filter);  // (the buffer is empty: its rows are the filter's scratch)
#pragma unroll
        for (int j = 0; j < LEAF; ++j) best[KCAP - 1 - j] = key_min(best[KCAP - 1 - j], nw[j]);
        bitonic_merge<KCAP, NZ, LEAF>(best);
        tau = active ? fminf(__uint_as_float(static_cast<u32>(best[KCAP - 1] >> 32)), cap) : -1.f;
    };
    // (the seed leaves from the middle of the range outwards instead of in curve order -- the middle of the group's own chunk is near
    //  most of its lanes -- was measured in round 4: no gain; profiles/experiments/README.md)
    for (u32 leaf = s0;; ++leaf) {
        const bool more = leaf < s1;
        if (direct_seeds && more && leaf + 1u != t.nleaves) {
            seed_direct(leaf);
            continue;
        }
        fold_if_needed(more, true);
        if (!more) break;
        candidates(leaf, false);
    }
    if (STATS) st_seed_app = st_app;
    cap = wave_radius_cap<KCAP>(tau, valid, lane);
    tau = active ? fminf(tau, cap) : -1.f;

    // ---- walk rounds ----
    constexpr bool packed_leaves = pack_rows(MULTI, KCAP) > 0 && fast && !EPS_EACH;  // (wants the lanes that need each leaf: WalkerT's KEEP)
    WalkerT<(KCAP > 8), packed_leaves> wk;
    // A leaf of the walk that at most PCPX_PACKED_LEAVES lanes need (three quarters of the walk's leaves: the lane-per-query form
    // computes 512 distances there of which 8 ... 192 matter) is looked at EIGHT NEEDING LANES x EIGHT POINTS at a time: the
    // needing lanes publish {query, tau} and their write address in LDS in the order of their rank (v_mbcnt of the need mask);
    // lane 8 i + j then forms the distance from the i-th published query to point j of the leaf, and a lane whose point is
    // within that query's tau takes the next row of the query's column with a returning LDS add on the published address and
    // writes the key there; at the end each needing lane reads its address back.  ~10 vector instructions per eight needing
    // lanes (+ ~8 per leaf) against 91 per leaf.  Same keys, same arithmetic (d = p - q, three roundings), same tau as the
    // other forms; the order of a column's new keys is whatever order the adds were served in, which no result depends on
    // (keys are distinct and the selection network sorts them).
    // Slots that hold no query hold tau = -1 (k_knn sets them so, a needing lane sets its slot back when the leaf is done): the
    // lanes of a step beyond the leaf's needing lanes compare against that and take nothing -- no lane mask per step, and the
    // scalar unit is as loaded as the vector units here (profiles/experiments/README.md, round 4).
    // PCPX_PACKED_FREE > 0: the buffer is filled OPTIMISTICALLY -- a leaf is started as soon as every needing lane has that many
    // free rows (the other forms want LEAF = 8: any lane may take every point; in the walk a lane takes one key of a leaf it
    // needs, rarely three, and waiting for eight free rows of ten meant a fold -- the whole selection network, for all 64
    // lanes -- per 19 keys of the WAVE).  A key whose add comes back with an address beyond the column goes to a spare word of
    // the publish row instead; the needing lane sees from the address it reads back that keys were lost, takes its column back
    // to where it was before the leaf (the rows written since hold PAD_KEY again), and the leaf is looked at once more for
    // those lanes after a fold.  Returns the lanes that want that (0: done).
    // The same for a whole LAST-LEVEL NODE that few lanes need (`sh` = 5: lane 32 i + j forms the distance from the i-th published query
    // to point j of the node's 32, two needing lanes a step; `sh` = 3: one leaf, as above): the node's four leaf boxes are then never
    // tested -- d2 <= tau is the test that counts -- and its leaves cost one publish and read-back instead of one each.  A leaf of the
    // node that is in the seed range (seen already), or beyond the cloud's last, shows NaN to every query.
    auto packed_leaf = [&](const u32 leaf, const u64 who, const u32 how_many, const u32 sh) -> u64 {
        float4* const pub_q = reinterpret_cast<float4*>(pub);                     // [PCPX_PACKED_LEAVES] {qx, qy, qz, tau}
        u32* const pub_wa = reinterpret_cast<u32*>(pub) + 4 * PCPX_PACKED_LEAVES;  // [PCPX_PACKED_LEAVES] next free row of the column
        if (PCPX_PRIO_PACKED != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_PACKED);
        u32 lane_here = lane;
        asm volatile("" : "+v"(lane_here));  // (or everything below that depends on the lane alone sits in registers from group to group)
        const u32 j = PCPX_PACKED_NODE > 0 ? lane_here & ((1u << sh) - 1u) : lane_here & 7u, i = PCPX_PACKED_NODE > 0 ? lane_here >> sh : lane_here >> 3;
        float cx, cy, cz;
        if (PCPX_PACKED_NODE > 0) {
            const u32 lf = leaf + (j >> 3);
            const bool shown = lf < t.nleaves && lf - s0 >= s1 - s0;
            cx = std::numeric_limits<float>::quiet_NaN(), cy = 0.f, cz = 0.f;
            if (shown) {
                const float* rec = reinterpret_cast<const float*>(t.leaves + lf);
                cx = rec[j & 7u], cy = rec[LEAF + (j & 7u)], cz = rec[2 * LEAF + (j & 7u)];
            }
        } else {
            const float* rec = reinterpret_cast<const float*>(t.leaves + leaf);
            cx = rec[j], cy = rec[LEAF + j], cz = rec[2 * LEAF + j];
        }
        const u32 posj = leaf * LEAF + j;
        const u32 r = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(who >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(who), 0u));
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(who);
        if (mine) {
            pub_q[r] = make_float4(qx, qy, qz, tau);
            pub_wa[r] = wa;
        }
        __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations complete in order; this only pins the compiler's order)
        // A step takes its keys with the lanes that have none switched off (v_cmpx, like the lane-per-query leaves) instead of a branch
        // round them: five scalar-pipe instructions less per step, and nearly every step has a lane that takes one.  A row address
        // beyond the column (PCPX_PACKED_FREE > 0) becomes the spare word's by bit arithmetic: a compare and select through VCC costs a
        // register for the spare address and two wait states.
        const u64 saved = save_exec();
        const u32 per_step = PCPX_PACKED_NODE > 0 ? 64u >> sh : 8u;
        u32 s = 0;  // (how_many >= 1: some lane needs the leaf)
        do {
            const float4 q = pub_q[s + i];
            // (the step's query is on its way from LDS while the point comes from memory, which is waited for HERE, as a whole: left to
            //  itself hipcc puts three s_waitcnt vmcnt in every step, one in front of each subtraction)
            asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz));
            const float dx = cx - q.x, dy = cy - q.y, dz = cz - q.z;
            const float d2 = sq3(dx, dy, dz);
            const u32 slot = lds_address(pub_wa + s + i);
            u32 at, inside;
            // (NaN padding points fail d2 <= tau; so does every point against an empty slot's tau = -1)
            if (packed_free > 0)
                asm volatile("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                             "v_mov_b32_e32 %[at], 0x200\n\t"
                             "ds_add_rtn_u32 %[at], %[slot], %[at]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "v_subrev_u32_e32 %[m], %[end], %[at]\n\t"      // address - end of the buffer: negative while inside it
                             "v_ashrrev_i32_e32 %[m], 31, %[m]\n\t"
                             "v_bfi_b32 %[at], %[m], %[at], %[spare]\n\t"    // inside ? address : the publish row's last 32 bytes, which are nobody's
                             "ds_write2_b32 %[at], %[pos], %[d2] offset1:1\n\t"
                             "s_mov_b64 exec, %[saved]"
                             : [at] "=&v"(at), [m] "=&v"(inside)
                             : [d2] "v"(d2), [tau] "v"(q.w), [slot] "v"(slot), [end] "s"(wa_end), [spare] "s"(wa_end + 480u), [pos] "v"(posj), [saved] "s"(saved)
                             : "vcc", "memory");
            else
                asm volatile("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                             "v_mov_b32_e32 %[at], 0x200\n\t"
                             "ds_add_rtn_u32 %[at], %[slot], %[at]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "ds_write2_b32 %[at], %[pos], %[d2] offset1:1\n\t"
                             "s_mov_b64 exec, %[saved]"
                             : [at] "=&v"(at)
                             : [d2] "v"(d2), [tau] "v"(q.w), [slot] "v"(slot), [pos] "v"(posj), [saved] "s"(saved)
                             : "vcc", "memory");
            s += per_step;
        } while (s < how_many);
        __builtin_amdgcn_wave_barrier();
        u32 now = wa;
        if (mine) {
            now = pub_wa[r];
            reinterpret_cast<float*>(pub_q + r)[3] = -1.f;
        }
        __builtin_amdgcn_wave_barrier();
        if (PCPX_PRIO_PACKED != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_BASE);
        if (packed_free == 0) {
            wa = now;
            return 0ull;
        }
        // (a lane whose column ran full keeps its address of before the leaf: packed_take_back, the caller's rare branch)
        const bool lost_keys = now > col_addr + (static_cast<u32>(BUF) << 9);  // (a lane that does not need the leaf: now = wa <= the column's end)
        wa = lost_keys ? wa : now;
        return __builtin_amdgcn_ballot_w64(lost_keys);
    };
    // the rows a lane wrote in a leaf that then overran its column hold PAD_KEY again
    auto packed_take_back = [&](const u64 lost) {
        if (__builtin_amdgcn_inverse_ballot_w64(lost)) {
            const u32 column_end = col_addr + (static_cast<u32>(BUF) << 9);
            const u64 pad = pad_key_here();
            for (u32 a = wa; a < column_end; a += 512u) asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(pad) : "memory");
        }
    };
    const u32 seed_count = s1 - s0;
    u32 packed_limit = packed_leaves ? PCPX_PACKED_LEAVES : 0;
    constexpr bool packed_nodes = packed_leaves && PCPX_PACKED_NODE > 0 && KCAP <= PCPX_PACKED_NODE_KCAP;
    static_assert(!packed_nodes || (UNIT_LEAVES == 1 && PCPX_PACKED_NODE <= PCPX_PACKED_LEAVES), "the node form: one leaf per unit, one publish slot per needing lane");
    for (u32 rounds = 0;;) {  // (rounds != 0: a shell round -- asked of the counter, a bool carried round the loop becomes a lane mask)
        // (wave-uniform as it is, but carried round a loop whose exit hipcc takes for lane-dependent -- `cap` comes out of lane exchanges --
        //  it counts as a vector value, and a scalar flag made from it is an "illegal VGPR to SGPR copy")
        const u32 packed_limit_now = __builtin_amdgcn_readfirstlane(packed_limit);
        bool root_leaf = wk.start(t, need, st_expand);
        (void)root_leaf;  // depth 0: the only leaf is the seed chunk, already done
        // A trip of the outer loop pops one node: a node above the last level is expanded; a LAST-LEVEL node hands its needed
        // leaves to the inner loop directly (WalkerT::leaves_of: no trip through the pending bits for them -- a dozen scalar
        // instructions per leaf, and the scalar side of this kernel is what it is short of); `direct` = its children still to look
        // at, as leaves direct_first + c.  A leaf inside the seed range was seen under a larger tau than any later one: skipped.
        if (STATS) tc_mark = __builtin_amdgcn_s_memtime();
        u32 direct = 0, direct_first = 0;
        u32 sh = 3u;  // (packed_nodes: 5 while `direct` stands for a last-level node as a whole)
        for (;;) {
            if (direct == 0) {
                if (wk.done()) break;
                if (PCPX_PRIO_WALK != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_WALK);
                u32 node;
                const int h = wk.pop(node);
                ++st_expand;
                if (h > 1) {
                    wk.expand(t, h, node, need);
                } else {
                    direct_first = node << LOGW;
                    if (packed_nodes) {
                        // (the node's own box -- a child box of its parent's record -- says how many lanes need it NOW; none: tau has shrunk since the parent was expanded)
                        sh = 3u;
                        u32 lanes_now = ~0u;
                        u64 lanes = 0;
                        if (packed_limit_now != 0) {
                            const NodeBox own = load_const(t.nodes + (wk.level_base(t.depth - 1) + node));
                            lanes = __builtin_amdgcn_ballot_w64(need(own));
                            asm("s_bcnt1_i32_b64 %0, %1" : "=s"(lanes_now) : "s"(lanes) : "scc");  // (as __builtin_popcountll hipcc counts on the vector side)
                        }
                        if (lanes_now <= static_cast<u32>(PCPX_PACKED_NODE)) {
                            wk.leaf_need[0] = lanes;
                            wk.l = 0;
                            wk.ploc = node;
                            direct = __builtin_amdgcn_readfirstlane(lanes_now < 1u ? lanes_now : 1u);  // (wave-uniform as it is; hipcc 7.2 forms it on the vector side and then fails to bring it back: "illegal VGPR to SGPR copy")
                            sh = 5u;
                        } else {
                            direct = wk.leaves_of(t, node, need);
                        }
                    } else {
                        direct = wk.leaves_of(t, node, need);
                    }
                }
                // (h = 0 is never popped: the build gives no tree a depth of 1 -- depth_of, pcpx_build.hip -- so leaves are only ever met under a
                //  last-level node.  The case for it cost EVERY pop of the walk nine scalar instructions of dispatch.)
                if (PCPX_PRIO_WALK != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_BASE);
            }
            while (direct != 0) {
                u32 loc;
                {
                    u32 child;
                    asm("s_ff1_i32_b32 %0, %1\n\ts_bitset0_b32 %1, %0" : "=&s"(child), "+s"(direct));
                    loc = direct_first + child;
                }
                // (`loc` is a UNIT of the tree's bottom level: UNIT_LEAVES consecutive leaf records under one box, looked at one after the
                //  other for the lanes that need the unit; the seed range is whole units)
                if ((packed_nodes && sh == 5u) || loc * UNIT_LEAVES - s0 >= seed_count) {
                    if (STATS) tc_walk += __builtin_amdgcn_s_memtime() - tc_mark;
                    if (!packed_leaves) {
#pragma unroll 1
                        for (u32 leaf = loc * UNIT_LEAVES;; ++leaf) {  // (a needed unit's first leaf exists; UNIT_LEAVES = 1: no loop)
                            fold_if_needed(true, false);
                            candidates(leaf, rounds != 0u);
                            if (UNIT_LEAVES == 1 || leaf + 1u == (loc + 1u) * UNIT_LEAVES || leaf + 1u >= t.nleaves) break;
                        }
                    } else {
                        const u32 c = loc & (W - 1u);
                        u64 who;
                        u32 how_many_unit;
                        asm("s_cmp_eq_u32 %[c], 2\n\ts_cselect_b64 %[w], %[n2], %[n3]\n\t"
                            "s_cmp_eq_u32 %[c], 1\n\ts_cselect_b64 %[w], %[n1], %[w]\n\t"
                            "s_cmp_eq_u32 %[c], 0\n\ts_cselect_b64 %[w], %[n0], %[w]\n\t"
                            "s_bcnt1_i32_b64 %[m], %[w]"
                            : [w] "=&s"(who), [m] "=s"(how_many_unit)
                            : [c] "s"(c), [n0] "s"(wk.leaf_need[0]), [n1] "s"(wk.leaf_need[1]), [n2] "s"(wk.leaf_need[2]), [n3] "s"(wk.leaf_need[3])
                            : "scc");
#pragma unroll 1
                        for (u32 leaf = loc * UNIT_LEAVES;; ++leaf) {  // (a needed unit's first leaf exists; UNIT_LEAVES = 1: no loop)
                        u32 how_many = how_many_unit;
                        // Only a lane that needs the leaf can take keys from it (its box distance was within a tau that has only
                        // shrunk since, and no point of the leaf is nearer than its box): fold if one of THOSE could not take LEAF more.
                        // (The form is a NUMBER in a scalar register and "once more, after a fold" a threshold of 0: as two bools carried round
                        //  the loop hipcc kept them as 64-bit lane masks -- a dozen scalar instructions per leaf to set, copy and test them.)
                        u32 packed_form, full_from;
                        asm("s_cmp_le_u32 %[m], %[lim]\n\ts_cselect_b32 %[pf], 1, 0\n\ts_cselect_b32 %[thr], %[tp], %[td]"
                            : [pf] "=&s"(packed_form), [thr] "=&s"(full_from)
                            : [m] "s"(how_many), [lim] "s"(packed_limit_now), [tp] "s"(wa_packed_full), [td] "s"(wa_full)
                            : "scc");
                        u64 todo = who;  // (the lanes the leaf is still to be looked at for)
                        for (;;) {  // (one call site of the fold: one copy of the selection network)
                            if ((__builtin_amdgcn_ballot_w64(wa >= full_from) & todo) != 0) fold(false);
                            if (packed_form == 0u) break;
                            if (STATS) tc_mark = __builtin_amdgcn_s_memtime();
                            todo = packed_leaf(leaf, todo, how_many, sh);
                            if (COST) st_steps += (how_many + (64u >> sh) - 1u) >> (6u - sh);
                            if (STATS) {
                                asm volatile("" ::"v"(wa));
                                tc_leaf += __builtin_amdgcn_s_memtime() - tc_mark;
                            }
                            if (__builtin_expect(todo == 0, 1)) break;  // (else, PCPX_PACKED_FREE > 0 only and rare: columns ran full -- fold, and once more for their lanes)
                            packed_take_back(todo);
                            how_many = static_cast<u32>(__builtin_popcountll(todo));
                            full_from = 0u;  // (every lane of `todo` has a full column, whatever its address was taken back to: fold)
                            if (packed_nodes && sh == 5u) {  // (a node: its 32 points may be more than a column holds -- leaf by leaf for those lanes)
                                sh = 3u;
                                direct = (1u << W) - 1u;
#pragma unroll
                                for (int cc = 0; cc < W; ++cc) wk.leaf_need[cc] = todo;
                                break;
                            }
                        }
                        if (packed_form != 0u) {
                            if (COST) ++st_sparse;
                            // ([14], [15]: the packed leaves and their needing lanes.  The keys they took are counted where they are folded:
                            //  a column's address goes back to its first row at every fold, so "address now - address before" is not a count -- round 4's
                            //  `appended` figure was that difference and came out as 1.7e14)
                            if (STATS) ++st_leaves, ++st_sparse, st_owners += how_many;
                        } else {
                            candidates(leaf, rounds != 0u);
                        }
                        if (UNIT_LEAVES == 1 || leaf + 1u == (loc + 1u) * UNIT_LEAVES || leaf + 1u >= t.nleaves) break;
                        }  // the unit's next leaf
                    }
                    if (STATS) tc_mark = __builtin_amdgcn_s_memtime();
                }
            }
        }
        if (STATS) tc_walk += __builtin_amdgcn_s_memtime() - tc_mark;
        fold_if_needed(false, false);
        if (!(cap < inf)) break;
        // a capped round ended: lanes whose k-th distance is within the cap are exact; the others go round again with 4x the
        // radius^2, accepting only the new shell (lo_d2, cap]
        const float kth = __uint_as_float(static_cast<u32>(best[KCAP - 1] >> 32));
        const bool failed = active && !(kth <= cap);
        if (!any_lane(failed)) break;
        if (COST) ++st_round2, st_later_lanes += static_cast<u32>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(failed)));
        if (STATS) {
            ++st_round2;
            if (tc_later_mark == 0) tc_later_mark = __builtin_amdgcn_s_memtime();  // the group's later rounds start here
            st_later_lanes += static_cast<u32>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(failed)));
        }
        lo_d2 = cap;
        const NodeBox root = load_const(t.nodes);
        const float ex = root.hi(0) - root.lo(0), ey = root.hi(1) - root.lo(1), ez = root.hi(2) - root.lo(2);
        const float diag2 = sq3(ex, ey, ez);
        // next radius^2; the last round is uncapped: when the cap covers the whole cloud from any query inside 2x its box, when
        // it cannot grow (a cap of 0: more than half of the sampled lanes sit on >= k coincident points and eps is 0), or
        // after 12 rounds
        const float grown = cap * PCPX_CAP_GROW;
        ++rounds;
        packed_limit = 0;  // (the later rounds want lo_d2 < d2 as well: lane-per-query leaves.  Measured, round 5: the packed form there with a shell
                           //  test, and a raised priority for a group from its second round on -- neither moved the longest groups nor the rate:
                           //  what makes a group long is not a straggler's later rounds, see k_make_order)
        cap = (grown > cap && grown < diag2 * 4.f && rounds < 12u) ? grown : inf;
        active = failed;
        tau = active ? fminf(kth, cap) : -1.f;
        cnt = 0;
        wa = col_addr;
    }

    if (STATS) {
        u32 app = st_app, sapp = st_seed_app;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            app += __shfl_xor(app, off);
            sapp += __shfl_xor(sapp, off);
        }
        if (lane == 0) {
            atomicAdd(&stats[0], static_cast<unsigned long long>(st_leaves));
            atomicAdd(&stats[1], static_cast<unsigned long long>(st_expand));
            atomicAdd(&stats[2], static_cast<unsigned long long>(st_compact));
            atomicAdd(&stats[12], static_cast<unsigned long long>(st_seed_compact));
            atomicAdd(&stats[13], static_cast<unsigned long long>(sapp));
            atomicAdd(&stats[14], static_cast<unsigned long long>(st_sparse));  // leaves looked at point-per-lane ...
            atomicAdd(&stats[15], static_cast<unsigned long long>(st_owners));  // ... and the lanes they were looked at for
            atomicAdd(&stats[3], static_cast<unsigned long long>(app));
            atomicAdd(&stats[4], 1ull);
            atomicAdd(&stats[5], static_cast<unsigned long long>(s1 - s0));
            atomicAdd(&stats[6], static_cast<unsigned long long>(st_round2));
            atomicAdd(&stats[7], tc_walk);
            atomicAdd(&stats[8], tc_compact);
            atomicAdd(&stats[9], tc_leaf);
            atomicAdd(&stats[10], static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - tc0);  // search loop
            // (behind the per-wave records: [STATS_EXTRA] cycles of the groups' walk rounds after the first, [+1] the lanes those rounds ran for)
            if (tc_later_mark != 0) atomicAdd(&stats[STATS_EXTRA], static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - tc_later_mark);
            atomicAdd(&stats[STATS_EXTRA + 1], static_cast<unsigned long long>(st_later_lanes));
        }
    }

    if (!MULTI && DIAG == 0) {  // what the search took (shader clock / 64), for the next launch's order: one store per ~32 000 instructions
        const knn_args_ptr kt = knn_args_here();
        u32* const gtime = cold(&kt->sch.gtime);
        if (gtime) {
            const u32 ticks = (static_cast<u32>(__builtin_amdgcn_s_memtime()) - tick0) >> 6;
            u32 lane_again = lane;
            asm volatile("" : "+v"(lane_again));
            if (lane_again == 0) gtime[g - kt->group_first] = ticks;
        }
    }
    if (COST) {
        u32* const events = cold(&knn_args_here()->sch.events);
        if (lane == 0) {
            uint4 e;
            // (high bits, for diagnosis -- pcpx_shard_cuts_by_cost masks them off: walk rounds after the first, and the lanes they ran for)
            e.x = st_expand, e.y = (st_leaves & 0xFFFFFFu) | (st_round2 << 24), e.z = (st_sparse & 0xFFFFFu) | ((st_later_lanes < 0xFFFu ? st_later_lanes : 0xFFFu) << 20),
            e.w = (st_compact << 16) | (st_steps < 0xFFFFu ? st_steps : 0xFFFFu);
            reinterpret_cast<uint4*>(events)[slot] = e;
        }
    }
    const int first_slot = KCAP - static_cast<int>(k);
    if (PCPX_PRIO_EPI != PCPX_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_PRIO_EPI);
    // (the cold arguments are read again where the epilogue needs them: nothing read through `ka` above is alive any more)
    if (MULTI) {  // raw (d2, sorted position) keys of this pass; rows are built by k_assemble
        const MultiPass mp = cold(&knn_args_here()->mp);
        if (valid) {
            u64* dst = mp.keys + static_cast<u64>(p - mp.slot0) * mp.stride + mp.offset;
#pragma unroll
            for (int s = 0; s < KCAP; ++s) {
                int j = s - first_slot;
                if (j >= 0) dst[j] = best[s];
            }
            mp.lo_out[p - mp.slot0] = best[KCAP - 1];
        }
        return;
    }
    // ---- sorted position -> original index; order rows by (d2, index) ----
    // The append buffer is idle from here on: the row's sorted positions (which the fused normal gathers coordinates by) wait
    // in the lane's LDS column, two per row, instead of in KCAP registers beside the KCAP keys -- that pair of arrays was what
    // spilled (8 B/lane of scratch at k <= 16, 32 B at k <= 32).
    u32* const pos_col = reinterpret_cast<u32*>(col);
    auto pos_slot = [&](int s) -> u32& { return pos_col[(s >> 1) * 128 + (s & 1)]; };
#pragma unroll
    for (int s = 0; s < KCAP; ++s) {
        u64 key = best[s];
        bool real = s >= first_slot && key != PAD_KEY;
        u32 ps = real ? static_cast<u32>(key) : 0u;
        pos_slot(s) = ps;
        u32 id = t.leaves[ps / LEAF].id[ps % LEAF];
        asm volatile("" : "+v"(id));  // keep the load unconditional: sunk under `real` it becomes a branch per key, and the
                                      // values live across those branches were what spilled to scratch
        best[s] = real ? ((key & 0xFFFFFFFF00000000ull) | id) : key;
    }
    // Keys were ascending in (d2, position); with the index in the low word only runs of EXACTLY equal d2 can be out of order.
    // Rare: repaired by adjacent exchanges (odd-even transposition, at most KCAP rounds; a run of t ties needs t), the positions
    // exchanged in LDS alongside.
    bool unordered = false;
#pragma unroll
    for (int s = 0; s + 1 < KCAP; ++s) unordered |= best[s] > best[s + 1];
    for (int round = 0; round < KCAP && any_lane(unordered); ++round) {
#pragma unroll
        for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
            for (int s = parity; s + 1 < KCAP; s += 2) {
                const bool sw = best[s] > best[s + 1];
                const u64 a = sw ? best[s + 1] : best[s], b = sw ? best[s] : best[s + 1];
                best[s] = a;
                best[s + 1] = b;
                if (sw) {
                    const u32 pa = pos_slot(s), pb = pos_slot(s + 1);
                    pos_slot(s) = pb;
                    pos_slot(s + 1) = pa;
                }
            }
        }
        unordered = false;
#pragma unroll
        for (int s = 0; s + 1 < KCAP; ++s) unordered |= best[s] > best[s + 1];
    }
    // (the column goes back to the chunked compaction's invariant -- empty slots hold PAD_KEY -- at the end of this function)
    auto restore_column = [&]() {
        if (PCPX_COMPACT_BY8 && (KCAP <= 16 || PCPX_BY8_K32)) {
            const u64 pad = pad_key_here();
#pragma unroll
            for (int r = 0; r < (KCAP + 1) / 2; ++r) col[r * 64] = pad;
        }
    };

    if (!valid) {
        restore_column();
        return;
    }
    // output row = original index of the query (read here, not at the start: one VGPR less through the search loop)
    // (the query's position is formed again rather than kept: as a 64-bit record offset it sat in a VGPR pair all through the search)
    u32 g_again = g;
    asm volatile("" : "+s"(g_again));  // (or hipcc keeps the search's `p` for this: in scratch, in the k <= 16 kernel)
    u32 p_row = g_again * GROUP + lane;
    asm volatile("" : "+v"(p_row));
    const knn_args_ptr kb = knn_args_here();  // (here, not above the reordering loop: what is loaded through it is loaded where it is made)
    const KnnOutputs o = cold(&kb->o);
    u32 row = SELF ? t.leaves[p_row / LEAF].id[p_row % LEAF] : cold(&kb->qv.row)[p_row];
    if (SELF && o.by_position) row = p_row + o.pos_bias;
    if (SELF && o.tau) o.tau[p_row] = key_d2(best[KCAP - 1]);  // (+inf when the row holds fewer than k)
    u32 found = 0;
    u32 okmask = 0;
    const u32 stride = o.row_stride ? o.row_stride : k;
    const u64 ob = static_cast<u64>(row) * stride;
    // A row of KCAP entries (row_stride = KCAP; k = KCAP or KCAP - 1: the slots' shift is then a constant) leaves as 16-byte stores: one
    // aligned piece per row.  Scattered by input index, a 60-byte row written word by word costs every 32-byte sector it touches twice
    // (read for ownership, write back), and a store instruction of 64 lanes touches 64 lines whatever its width.
    // (NZ = 1: the kernel for k < KCAP, whose slot 0 holds the sentinel: k = KCAP - 1 is the shift by one; NZ = 0: k = KCAP in the
    //  deferred-eps kernels, any k in the per-candidate ones)
    const bool whole_rows = stride == static_cast<u32>(KCAP) && first_slot == NZ && !STATS;
    auto store_whole_rows = [&](auto FS) {
        constexpr int fs = decltype(FS)::value;
#pragma unroll
        for (int c = 0; c < KCAP / 4; ++c) {
            u32 id4[4];
            float d4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sl = 4 * c + e + fs;  // (the slot whose key is entry 4 c + e of the row)
                const u64 key = sl < KCAP ? best[sl < KCAP ? sl : 0] : PAD_KEY;
                const bool ok = key != PAD_KEY;
                id4[e] = ok ? static_cast<u32>(key) : INVALID_ID;
                d4[e] = __uint_as_float(static_cast<u32>(key >> 32));
            }
            if (o.idx) *reinterpret_cast<uint4*>(o.idx + ob + 4 * c) = make_uint4(id4[0], id4[1], id4[2], id4[3]);
            if (o.d2) *reinterpret_cast<float4*>(o.d2 + ob + 4 * c) = make_float4(d4[0], d4[1], d4[2], d4[3]);
        }
    };
    if (whole_rows) store_whole_rows(std::integral_constant<int, NZ>{});
#pragma unroll
    for (int s = 0; s < KCAP; ++s) {
        int j = s - first_slot;
        if (j >= 0) {
            u64 key = best[s];
            bool ok = key != PAD_KEY;
            if (!whole_rows) {
                if (o.idx) o.idx[ob + j] = ok ? static_cast<u32>(key) : INVALID_ID;
                if (o.d2 && !STATS) o.d2[ob + j] = __uint_as_float(static_cast<u32>(key >> 32));
            }
            found += ok ? 1u : 0u;
            okmask |= ok ? (1u << s) : 0u;
        }
    }
    if (o.cnt) o.cnt[row] = found;

    // ---- fused pcp::algorithm::average_distances_to_neighbors (average_distance_to_neighbors.hpp:52-70):
    //      (sum of sqrt(d2) over the row, in row order) / row size; an empty row gives 0/0 = NaN ----
    if (o.meandist) {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            bool ok = (okmask >> s) & 1u;
            float d = sqrtf(__uint_as_float(static_cast<u32>(best[s] >> 32)));
            sum += ok ? d : 0.f;
        }
        o.meandist[row] = sum / static_cast<float>(found);
    }

    // ---- fused pcp::estimate_normal over the row (normal_estimation.hpp:41-77), coordinates gathered
    //      from the leaf records in row order ----
    if (o.normals || o.centroids || (SELF && o.nc4)) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            bool ok = (okmask >> s) & 1u;
            const u32 ps = pos_slot(s);
            const Leaf& lf = t.leaves[ps / LEAF];
            float x = lf.x[ps % LEAF], y = lf.y[ps % LEAF], z = lf.z[ps % LEAF];
            sx += ok ? x : 0.f;
            sy += ok ? y : 0.f;
            sz += ok ? z : 0.f;
        }
        float fn = static_cast<float>(found);
        float mx = sx / fn, my = sy / fn, mz = sz / fn;
        if (o.centroids) {  // pcp::common::center_of_geometry (vector3d_queries.hpp:77-99): the tangent plane's point
            o.centroids[3ull * row] = mx;
            o.centroids[3ull * row + 1] = my;
            o.centroids[3ull * row + 2] = mz;
        }
        if (!o.normals && !(SELF && o.nc4)) {
            restore_column();
            return;
        }
        float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            bool ok = (okmask >> s) & 1u;
            const u32 ps = pos_slot(s);
            const Leaf& lf = t.leaves[ps / LEAF];
            float x = lf.x[ps % LEAF], y = lf.y[ps % LEAF], z = lf.z[ps % LEAF];
            float vx = ok ? x - mx : 0.f, vy = ok ? y - my : 0.f, vz = ok ? z - mz : 0.f;
            c00 += vx * vx;
            c10 += vy * vx;
            c11 += vy * vy;
            c20 += vz * vx;
            c21 += vz * vy;
            c22 += vz * vz;
        }
        float nrm[3], ev[3];
        eig3_smallest(c00, c10, c20, c11, c21, c22, nrm, ev);
        if (SELF && o.nc4) {  // {normal, count} at the query's curve position: one contiguous kilobyte per wave (k_gather_nc4 takes them to input order)
            o.nc4[p_row] = make_float4(nrm[0], nrm[1], nrm[2], __uint_as_float(found));
        } else {
            o.normals[3ull * row] = nrm[0];
            o.normals[3ull * row + 1] = nrm[1];
            o.normals[3ull * row + 2] = nrm[2];
        }
    }
    restore_column();
}

// Persistent launch: the grid is exactly the number of waves the GPU can hold, and every wave pulls query
// groups from work queues until they are empty (group run times differ by 2-3x, so a static assignment
// would leave a long tail, and 156 k single-wave workgroups per 10 M queries need not go through the
// dispatcher).  There is one queue per XCD-sized eighth of the curve order (blocks b and b+8 share an XCD,
// so its L2 keeps serving one region); a wave drains its home queue first, then helps the others.  Counters
// sit on separate 64-B lines and are zeroed before each launch.  Measured equal to one group per workgroup
// on MI355X (19.0 vs 19.1 ms); note rocprofv3's MeanOccupancyPerCU reads 11.5 of 20 for both, i.e. it
// under-reports on gfx950 (this grid is fully resident by construction).
constexpr u32 QUEUE_STRIDE = 16;  // u32 per queue counter (64 B)
// Format of KnnSchedule::order for a launch of `ngroups` groups (per = ceil(ngroups / 8) per queue): order[q] = entries of queue q,
// q = 0 ... 7; queue q's entries start at order[ORDER_HEADER + ORDER_PARTS * q * per].  An entry is a group id (absolute), with
// bits 28 ... 31 = 0 for the whole group or p + 1 for its lanes 8 p ... 8 p + 7 only.
constexpr u32 ORDER_HEADER = 16, ORDER_PARTS = 8, ORDER_PART_SHIFT = 28, ORDER_GROUP_MASK = (1u << ORDER_PART_SHIFT) - 1u;
inline u64 order_entries(u64 ngroups) { return ORDER_HEADER + ORDER_PARTS * 8ull * ((ngroups + 7) / 8); }

template <int KCAP, bool SELF, int DIAG, bool MULTI = false, bool EPS_EACH = !PCPX_DEFER_EPS, int NZ = 0>
__global__ __launch_bounds__(64 * knn_wpb(KCAP, MULTI), KCAP <= 8 ? PCPX_MINW8 : KCAP <= 16 ? PCPX_MINW : PCPX_MINW32) void k_knn(
    const KnnArgs a)
{
    constexpr bool STATS = DIAG == 1;
    // (the fields used all through a group; the others are read where they are needed: KnnArgs)
    const TreeView t = a.t;
    const float eps = a.eps, eps_thr = a.eps_thr;
    unsigned long long* const __restrict__ stats = STATS ? a.stats : nullptr;
    constexpr int BUF = buf_rows(KCAP);
    extern __shared__ u64 lds[];
    const u32 lane = threadIdx.x & 63u;
    const u32 wib = wave_in_block();
    u64* const rows = lds + static_cast<size_t>(wib) * lds_rows(BUF, MULTI, MULTI ? 0 : KCAP) * 64;  // this wave's rows
    u64* col = rows + lane;
    float* pub = reinterpret_cast<float*>(rows + BUF * 64);  // packed_leaf's rows (pack_rows), behind the buffer's
    if (pack_rows(MULTI, KCAP) > 0 && lane < static_cast<u32>(PCPX_PACKED_LEAVES)) pub[4u * lane + 3u] = -1.f;  // its invariant: a slot that holds no query holds tau = -1
    if (blockIdx.x == 0 && wib == 0 && lane < 8u) a.queue_clear[lane * QUEUE_STRIDE] = 0u;  // (the next launch's counters: prepare_queue)
    if (PCPX_COMPACT_BY8 && !MULTI && (KCAP <= 16 || PCPX_BY8_K32)) {  // the chunked compaction's invariant: empty slots hold PAD_KEY
        const u64 pad = pad_key_here();
#pragma unroll
        for (int j = 0; j < BUF; ++j) col[j * 64] = pad;
    }
    unsigned long long t_start = 0, t_max = 0, g_max = 0;
    u32 n_done = 0;
    if (STATS) t_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz constant clock
    // The loop's only state is `s`, the number of queues this wave has seen run dry: the group range, the queue's address and the
    // queue's share of the groups are read again through the kernarg pointer for every group (three scalar loads and a dozen scalar
    // instructions per ~32 000), so that none of them sits in a spilled scalar register across the search.
    for (u32 s = 0; s < 8u;) {
        const knn_args_ptr ka = knn_args_here();
        const u32 group_first = ka->group_first, group_end = ka->group_end;
        u32* const queue = cold(&ka->queue);
        const u32 ngroups = group_end - group_first;
        const u32 per = (ngroups + 7u) >> 3;
        const u32 q = (blockIdx.x + s) & 7u;
        const u32 qbeg = q * per;
        const u32 qend = qbeg + per < ngroups ? qbeg + per : ngroups;
        u32 gi = 0, lane_here = lane;
        asm volatile("" : "+v"(lane_here));  // (or the mask of "lane == 0" is one more pair of scalar registers held across the search)
        if (lane_here == 0) gi = atomicAdd(&queue[q * QUEUE_STRIDE], 1u);
        gi = __builtin_amdgcn_readfirstlane(gi);
        // (the queue's slots in curve order, or -- KnnSchedule::order -- the list a recorded launch suggests: order_entries)
        const u32* const order = cold(&ka->sch.order);
        if (order ? gi >= load_const(order + q) : qbeg + gi >= qend) {
            ++s;
            continue;
        }
        unsigned long long tg = 0, tcg = 0;
        if (STATS) {
            tg = __builtin_amdgcn_s_memrealtime();
            tcg = __builtin_amdgcn_s_memtime();
        }
        const u32 entry = order ? load_const(order + ORDER_HEADER + ORDER_PARTS * qbeg + gi) : group_first + qbeg + gi;
        const u32 g = entry & ORDER_GROUP_MASK;
        knn_group<KCAP, SELF, DIAG, MULTI, EPS_EACH, NZ>(t, g, eps, eps_thr, stats, col, pub, lane, qbeg + gi, entry >> ORDER_PART_SHIFT);
        if (STATS) {
            if (lane == 0) atomicAdd(&stats[11], static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - tcg);
            ++n_done;
            tg = __builtin_amdgcn_s_memrealtime() - tg;
            if (tg > t_max) {
                t_max = tg;
                g_max = g;
            }
        }
    }
    if (STATS && lane == 0 && wib == 0) {  // per-wave diagnostic record: start, end, groups done, slowest group (ticks, id)
        stats[16 + 5ull * blockIdx.x] = t_start;
        stats[17 + 5ull * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        stats[18 + 5ull * blockIdx.x] = n_done;
        stats[19 + 5ull * blockIdx.x] = t_max;
        stats[20 + 5ull * blockIdx.x] = g_max;
    }
}

// number of workgroups that are resident at once (occupancy API x CUs), never more than there is work
u32 persistent_grid(Index& ix, const void* fn, int block, size_t lds, u64 groups, int wpb)
{
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds) != hipSuccess || per_cu < 1) per_cu = 8;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix.device) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    u64 want = static_cast<u64>(per_cu) * cus;
    u64 need = (groups + wpb - 1) / wpb;
    // (tried for short launches -- one rank's eighth of the queries is 2.7 groups per resident wave and ends with a tail:
    //  fewer waves so that each gets >= 3 / 4 / 6 groups: 0.776 -> 0.795 / 0.862 / 0.925 ms per 1.25 M queries; the full
    //  resident grid stays)
    u64 g = want < need ? want : need;
    g = (g + 7) / 8 * 8;
    return static_cast<u32>(g < 8 ? 8 : g);
}

}  // namespace

// Work-queue counters of the persistent launches: TWO sets.  A launch takes its groups from one and its first wave zeroes the other,
// which the next launch of the handle uses (launches of a handle are stream-ordered): no memset dispatch in front of every launch.
// (Round 4 tried one set, cleared by the last wave out: that wave is found with one more atomic per wave, and was slower.)
int ensure_queue(Index& ix)
{
    if (ix.d_queue) return PCPX_OK;
    PCPX_HIP(hipMalloc(reinterpret_cast<void**>(&ix.d_queue), 2 * 8 * QUEUE_STRIDE * sizeof(u32)));
    PCPX_HIP(hipMemsetAsync(ix.d_queue, 0, 2 * 8 * QUEUE_STRIDE * sizeof(u32), ix.stream));
    ix.queue_set = 0;
    return PCPX_OK;
}
int prepare_queue(Index& ix)
{
    const int st = ensure_queue(ix);
    if (st != PCPX_OK) return st;
    ix.queue_now = ix.d_queue + ix.queue_set * 8 * QUEUE_STRIDE;
    ix.queue_set ^= 1u;
    ix.queue_clear = ix.d_queue + ix.queue_set * 8 * QUEUE_STRIDE;
    return PCPX_OK;
}

// Squared distance below which a buffered key still needs the exact eps-box test (EpsFilter), or -1: test every candidate.
// A point inside the box has |d.| < eps on every axis, so its float d2 is at most 3 eps^2 (1 + 2^-24)^3.  Waiting pays
// while such keys are rare: eps small against the spacing of the points (estimated from the cloud's box: only a
// performance decision, both forms give the same rows).  Index::eps_test_mode (pcpx_debug_eps_test_mode) forces either.
static float eps_box_threshold(const Index& ix, float eps)
{
    if (!PCPX_DEFER_EPS || ix.eps_test_mode == 2) return -1.f;
    if (!(eps > 0.f)) return 0.f;  // nothing is inside an empty box
    const double t = 3.0 * static_cast<double>(eps) * static_cast<double>(eps) * (1.0 + 1e-6);
    if (!(t < 1e37)) return ix.eps_test_mode == 1 ? std::numeric_limits<float>::infinity() : -1.f;
    float thr = static_cast<float>(t);
    if (static_cast<double>(thr) < t) thr = std::nextafterf(thr, std::numeric_limits<float>::infinity());
    const float floor_thr = 2.f * std::numeric_limits<float>::min();  // products of differences below 2^-63 underflow
    if (thr < floor_thr) thr = floor_thr;
    if (ix.eps_test_mode == 1) return thr;
    const double ex = static_cast<double>(ix.bbox[3]) - ix.bbox[0], ey = static_cast<double>(ix.bbox[4]) - ix.bbox[1],
                 ez = static_cast<double>(ix.bbox[5]) - ix.bbox[2];
    const double vol = ex * ey * ez;
    if (!(vol > 0.0) || ix.n == 0) return -1.f;
    const double spacing = std::cbrt(vol / static_cast<double>(ix.n));
    return t <= 0.01 * spacing * spacing ? thr : -1.f;
}

// ---- long groups first ---------------------------------------------------------------------------------------------------------
// The groups of a launch take 0.4 ... 3 x their mean (a chunk of the curve that lies across one of its turns, points in a sparse
// place that go round the walk again), and a persistent launch ends when its last group does: one rank's eighth of a 10 M cloud is
// 2.7 groups per resident wave, and a 3 x group handed out in the last round ends a whole group's time after everything else.
// k_knn leaves every group's time behind (KnnSchedule::gtime); when the same question comes again on the same tree, the queues
// hand their groups out in three classes -- above 2 x the queue's mean, above 1.5 x, the rest -- in curve order inside a class
// (a few per cent of the groups leave the curve order; the rest still walk the queue's eighth of the curve front to back, which
// is what keeps the XCD's L2 warm).
// And a LONG group is taken apart.  What makes a group long is seldom a straggler that goes round the walk again: on the clustered
// cloud the longest groups have no later round at all -- 690 node expansions and 600 leaves in the first, nearly every leaf needed by
// one or two lanes: 64 queries in a thin place whose search regions hardly overlap, so the wave-uniform walk does their 64 searches
// one after the other (0.9 ms of dependent instructions; an eighth of the cloud takes 0.65 ms).  A class-0 group is handed out as
// ORDER_PARTS entries of 8 lanes each: eight waves walk an eighth of the union each, at the price of seven more seed phases and
// epilogues (a fifth of a mean group each, for less than 1 % of the groups).
// One block per queue.
constexpr u32 LPT_MIN_GROUPS = 512;
#ifndef PCPX_LPT_SPLIT
#define PCPX_LPT_SPLIT 1  // class-0 groups are handed out as ORDER_PARTS entries of 8 lanes each
#endif
#ifndef PCPX_TAIL_SPLIT
#define PCPX_TAIL_SPLIT 0  // 1: the groups of a launch's last round in pieces when that round is at most half full.  Measured (round 5,
                           // tools/ab_sizes.py): SLOWER -- 1 M queries 2 130 -> 1 880 Mq/s, 0.5 M 1 520 -> 1 380, 2 M 2 475 -> 2 370: a piece of 16
                           // lanes costs most of a group (its seed leaves, its walk's common part, its epilogue), not half of one
#endif
#ifndef PCPX_LPT_HI_PCT
#define PCPX_LPT_HI_PCT 200ull   // class 0: groups that took more than twice their queue's mean ...
#endif
#ifndef PCPX_LPT_MID_PCT
#define PCPX_LPT_MID_PCT 150ull  // ... class 1: more than 1.5 x.  (7/4 and 9/8 -- a third of a uniform cloud's groups out of curve order -- cost the
                                 //  10 M uniform launch 4 %: the queue's walk along the curve is what keeps its XCD's L2 warm.)
#endif
// tail_parts (0, 2, 4, 8) / tail_groups: the last tail_groups groups of every queue's list are handed out in tail_parts pieces each
// -- the launch's last round, when it is at most 1 / tail_parts full: a persistent launch of G groups on W resident waves ends with
// G mod W groups on as many waves while the others idle for a whole group's time (1 M queries: 2.2 rounds); in pieces they fill
// the round, and a piece of 16 lanes takes about half a group's time.  gtime == nullptr: no recorded times, curve order.
__global__ __launch_bounds__(1024) void k_make_order(const u32* __restrict__ gtime, u32 ngroups, u32 group_first, u32* __restrict__ order, u32 tail_parts,
                                                     u32 tail_groups)
{
    __shared__ unsigned long long sum_s;
    __shared__ u32 cnt_s[3], base_s[3], wave_s[3][16];
    const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const u32 per = (ngroups + 7u) >> 3;
    const u32 qbeg = blockIdx.x * per, qend = qbeg + per < ngroups ? qbeg + per : ngroups;
    if (qbeg >= qend) {
        if (t == 0) order[blockIdx.x] = 0;
        return;
    }
    if (t == 0) sum_s = 0;
    if (t < 3) cnt_s[t] = 0;
    __syncthreads();
    unsigned long long mine = 0;
    if (gtime)
        for (u32 i = qbeg + t; i < qend; i += 1024u) mine += gtime[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) atomicAdd(&sum_s, mine);
    __syncthreads();
    const unsigned long long mean = sum_s / (qend - qbeg);
    const unsigned long long hi = mean * PCPX_LPT_HI_PCT / 100ull, mid = mean * PCPX_LPT_MID_PCT / 100ull;
    auto cls = [&](u32 i) -> u32 {
        if (!gtime) return 2u;
        const u32 v = gtime[i];
        return v > hi ? 0u : v > mid ? 1u : 2u;
    };
    u32 c[3] = {0, 0, 0};
    for (u32 i = qbeg + t; i < qend; i += 1024u) ++c[cls(i)];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        u32 v = c[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && v) atomicAdd(&cnt_s[b], v);
    }
    __syncthreads();
    const u32 whole2 = cnt_s[2] > tail_groups ? cnt_s[2] - tail_groups : 0u;  // class-2 groups handed out whole; the rest in tail_parts pieces
    const u32 tail_n = tail_parts ? cnt_s[2] - whole2 : 0u;
    __syncthreads();
    if (t == 0) {
        base_s[0] = 0;
        base_s[1] = cnt_s[0];
        base_s[2] = cnt_s[0] + cnt_s[1];
        order[blockIdx.x] = (PCPX_LPT_SPLIT ? ORDER_PARTS : 1u) * cnt_s[0] + cnt_s[1] + (cnt_s[2] - tail_n) + tail_parts * tail_n;
    }
    __syncthreads();
    for (u32 i0 = qbeg; i0 < qend; i0 += 1024u) {
        const u32 i = i0 + t;
        const bool in = i < qend;
        const u32 b = in ? cls(i) : 3u;
        u32 below = 0;
#pragma unroll
        for (u32 bb = 0; bb < 3; ++bb) {
            const u64 m = __builtin_amdgcn_ballot_w64(b == bb);
            if (lane == 0) wave_s[bb][w] = static_cast<u32>(__builtin_popcountll(m));
            if (b == bb) below = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(m), 0u));
        }
        __syncthreads();
        if (in) {
            u32 before = 0;
            for (u32 ww = 0; ww < w; ++ww) before += wave_s[b][ww];
            const u32 at = ORDER_HEADER + ORDER_PARTS * qbeg;
            if (b == 0 && PCPX_LPT_SPLIT) {
#pragma unroll
                for (u32 part = 0; part < ORDER_PARTS; ++part) order[at + ORDER_PARTS * (base_s[0] + before + below) + part] = (group_first + i) | ((part + 1u) << ORDER_PART_SHIFT);
            } else {
                const u32 slot = (PCPX_LPT_SPLIT ? ORDER_PARTS - 1u : 0u) * cnt_s[0] + base_s[b] + before + below;  // (were every class-1 / -2 group whole)
                const u32 i2 = base_s[b] + before + below - (cnt_s[0] + cnt_s[1]);                                 // class 2: its number in the class
                if (b == 2 && tail_n != 0u && i2 >= cnt_s[2] - tail_n) {
                    const u32 first_piece = slot - (i2 - (cnt_s[2] - tail_n)) + tail_parts * (i2 - (cnt_s[2] - tail_n));
                    const u32 code0 = tail_parts == 8u ? 1u : tail_parts == 4u ? 9u : 13u;
                    for (u32 part = 0; part < tail_parts; ++part) order[at + first_piece + part] = (group_first + i) | ((code0 + part) << ORDER_PART_SHIFT);
                } else {
                    order[at + slot] = group_first + i;
                }
            }
        }
        __syncthreads();
        if (t < 3) {
            u32 tot = 0;
            for (u32 ww = 0; ww < 16; ++ww) tot += wave_s[t][ww];
            base_s[t] += tot;
        }
        __syncthreads();
    }
}

// (blocks of the device's index cache: a handle that is created, asked once and destroyed -- the drop-in's usual life -- does not pay
//  two hipMalloc / hipFree pairs for them; build_index reserves them with the tree so that the first query does not either)
int sched_reserve(Index& ix, u64 groups)
{
    Index::Sched& sc = ix.sched;
    if (groups <= sc.cap) return PCPX_OK;
    PCPX_HIP(hipStreamSynchronize(ix.stream));
    index_block_free(sc.d_gtime);
    index_block_free(sc.d_order);
    sc = Index::Sched{};
    const u64 cap = groups + groups / 8 + 64;
    void *a = nullptr, *b = nullptr;
    if (index_block_alloc(&a, cap * sizeof(u32)) != hipSuccess || index_block_alloc(&b, order_entries(cap) * sizeof(u32)) != hipSuccess) {
        (void)hipGetLastError();
        index_block_free(a);
        index_block_free(b);
        return PCPX_ERR_ALLOC;  // (the caller goes on without a schedule)
    }
    sc.d_gtime = static_cast<u32*>(a);
    sc.d_order = static_cast<u32*>(b);
    sc.cap = cap;
    return PCPX_OK;
}

template <int KCAP, bool SELF, bool EPS_EACH, int NZ>
static int launch_knn_form(Index& ix, const QueryView& qv, u64 gfirst, u64 gcount, u32 k, float eps, float thr, const KnnOutputs& o)
{
    constexpr int BUF = buf_rows(KCAP);
    constexpr int WPB = knn_wpb(KCAP, false);
    const size_t lds = static_cast<size_t>(WPB) * lds_rows(BUF, false, KCAP) * 64 * sizeof(u64);
    const u32 gf = static_cast<u32>(gfirst), ge = static_cast<u32>(gfirst + gcount);
    if (o.row_stride != 0 && (o.row_stride < k || ((o.row_stride == static_cast<u32>(KCAP)) && ((reinterpret_cast<uintptr_t>(o.idx) | reinterpret_cast<uintptr_t>(o.d2)) & 15u) != 0))) {
        set_error("pcpx: row_stride must be >= k, and rows of %d entries must start at multiples of 16 bytes", KCAP);
        return PCPX_ERR_INVALID;
    }
    int st = prepare_queue(ix);
    if (st != PCPX_OK) return st;
    KnnSchedule sch;
    auto* fn = k_knn<KCAP, SELF, 0, false, EPS_EACH, NZ>;
    const u32 pgrid = persistent_grid(ix, reinterpret_cast<const void*>(fn), 64 * WPB, lds, gcount, WPB);
    if (SELF && ix.tuning.lpt && gcount >= LPT_MIN_GROUPS) {
        Index::Sched& sc = ix.sched;
        // the launch's last round: G mod W groups, in 2 / 4 / 8 pieces each if that still fits the resident waves (k_make_order)
        const u64 waves = static_cast<u64>(pgrid) * WPB, rest = gcount % waves;
        u32 tail_parts = 0;
        if (PCPX_TAIL_SPLIT && gcount > waves && rest != 0) tail_parts = 8 * rest <= waves ? 8u : 4 * rest <= waves ? 4u : 2 * rest <= waves ? 2u : 0u;
        const u32 tail_groups = tail_parts ? static_cast<u32>((rest + 7) / 8) : 0u;  // (per queue)
        const bool same = sc.state != 0 && sc.gf == gfirst && sc.gc == gcount && sc.kcap == KCAP && sc.k == k;
        if (same && (sc.state == 1 || sc.state == 3)) {  // times recorded: the order by them (and the tail in pieces)
            k_make_order<<<8, 1024, 0, ix.stream>>>(sc.d_gtime, static_cast<u32>(gcount), gf, sc.d_order, tail_parts, tail_groups);
            sc.state = 2;
        }
        if (same && sc.state == 2) {
            sch.order = sc.d_order;
        } else if (sched_reserve(ix, gcount) == PCPX_OK) {  // a new question: record; curve order, the tail in pieces if it pays
            sch.gtime = sc.d_gtime;
            sc.gf = gfirst, sc.gc = gcount, sc.kcap = KCAP, sc.k = k, sc.state = 1;
            if (tail_parts) {
                k_make_order<<<8, 1024, 0, ix.stream>>>(nullptr, static_cast<u32>(gcount), gf, sc.d_order, tail_parts, tail_groups);
                sch.order = sc.d_order;
                sc.state = 3;  // (recorded with an order in use: a group handed out in pieces leaves one piece's time, which is what it is worth)
            }
        }
    }
    ProfileScope prof(ix, PCPX_K_KNN);
    fn<<<pgrid, 64 * WPB, lds, ix.stream>>>(KnnArgs{ix.view(), qv, gf, ge, k, eps, thr, o, MultiPass{}, ix.queue_now, nullptr, sch, ix.queue_clear});
    return check_hip(hipGetLastError(), "k_knn launch", __FILE__, __LINE__);
}

// ---- event counts of sampled groups (what a work-balanced shard cut is made from) -------------------------------------------------
__global__ void k_sample_order(u32 nsamples, u32 stride, u32* __restrict__ order)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 per = (nsamples + 7u) >> 3;
    if (i < 8u) {
        const u32 qbeg = i * per, qend = qbeg + per < nsamples ? qbeg + per : nsamples;
        order[i] = qend > qbeg ? qend - qbeg : 0u;
    }
    if (i < nsamples) order[ORDER_HEADER + ORDER_PARTS * (i / per * per) + i % per] = i * stride + stride / 2u;
}
template <int KCAP, int NZ>
static int launch_knn_cost_form(Index& ix, u32 nsamples, u32 k, float eps, float thr, const u32* d_order, u32* d_events)
{
    constexpr int BUF = buf_rows(KCAP);
    constexpr int WPB = knn_wpb(KCAP, false);
    const size_t lds = static_cast<size_t>(WPB) * lds_rows(BUF, false, KCAP) * 64 * sizeof(u64);
    int st = prepare_queue(ix);
    if (st != PCPX_OK) return st;
    auto* fn = k_knn<KCAP, true, 2, false, false, NZ>;
    const u32 pgrid = persistent_grid(ix, reinterpret_cast<const void*>(fn), 64 * WPB, lds, nsamples, WPB);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix.n)};
    KnnSchedule sch;
    sch.order = d_order;
    sch.events = d_events;
    fn<<<pgrid, 64 * WPB, lds, ix.stream>>>(KnnArgs{ix.view(), qv, 0u, nsamples, k, eps, thr, KnnOutputs{}, MultiPass{}, ix.queue_now, nullptr, sch, ix.queue_clear});
    return check_hip(hipGetLastError(), "k_knn cost sample launch", __FILE__, __LINE__);
}

template <int KCAP>
static int launch_knn_t(Index& ix, const QueryView& qv, bool self, u64 gfirst, u64 gcount, u32 k, float eps, const KnnOutputs& o)
{
    const float thr = eps_box_threshold(ix, eps);
    if (PCPX_DEFER_EPS && thr >= 0.f) {
        if (k < static_cast<u32>(KCAP)) {  // at least one sentinel slot: its compare-exchanges are compiled out
            return self ? launch_knn_form<KCAP, true, false, 1>(ix, qv, gfirst, gcount, k, eps, thr, o)
                        : launch_knn_form<KCAP, false, false, 1>(ix, qv, gfirst, gcount, k, eps, thr, o);
        }
        return self ? launch_knn_form<KCAP, true, false, 0>(ix, qv, gfirst, gcount, k, eps, thr, o)
                    : launch_knn_form<KCAP, false, false, 0>(ix, qv, gfirst, gcount, k, eps, thr, o);
    }
    return self ? launch_knn_form<KCAP, true, true, 0>(ix, qv, gfirst, gcount, k, eps, thr, o)
                : launch_knn_form<KCAP, false, true, 0>(ix, qv, gfirst, gcount, k, eps, thr, o);
}

// ---- k > 32: stitch the per-pass keys of every query into its output row -----------------------------
// One thread per query slot.  Keys are already ascending in (d2, sorted position); converting the low word
// to the original index can only disorder runs of equal d2, which a local insertion sort repairs in place.
template <bool SELF>
__global__ __launch_bounds__(256) void k_assemble(TreeView t, QueryView qv, u32 slot0, u32 nslots, u32 k, u32 stride,
                                                  u64* __restrict__ keys, KnnOutputs o)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const u32 p = slot0 + i;
    const u32 nq = SELF ? t.n : qv.nq;
    if (p >= nq) return;
    if (SELF && !(p - o.pos_lo < o.pos_hi - o.pos_lo)) return;
    const u32 row = (SELF && o.by_position) ? p + o.pos_bias : SELF ? t.leaves[p / LEAF].id[p % LEAF] : qv.row[p];
    u64* r = keys + static_cast<u64>(i) * stride;
    u32 found = 0;
    for (u32 j = 0; j < k; ++j) {
        u64 key = r[j];
        if (key == PAD_KEY) break;
        const u32 ps = static_cast<u32>(key);
        key = (key & 0xFFFFFFFF00000000ull) | t.leaves[ps / LEAF].id[ps % LEAF];
        u32 a = j;
        while (a > 0 && r[a - 1] > key) {
            r[a] = r[a - 1];
            --a;
        }
        r[a] = key;
        ++found;
    }
    const u64 ob = static_cast<u64>(row) * k;
    for (u32 j = 0; j < k; ++j) {
        const bool ok = j < found;
        if (o.idx) o.idx[ob + j] = ok ? static_cast<u32>(r[j]) : INVALID_ID;
        if (o.d2) o.d2[ob + j] = ok ? __uint_as_float(static_cast<u32>(r[j] >> 32)) : std::numeric_limits<float>::infinity();
    }
    if (o.cnt) o.cnt[row] = found;
    if (SELF && o.tau) o.tau[p] = found == k ? __uint_as_float(static_cast<u32>(r[k - 1] >> 32)) : std::numeric_limits<float>::infinity();
}

static int launch_knn_multipass(Index& ix, const QueryView& qv, bool self, u64 gfirst, u64 gcount, u32 k, float eps,
                                const KnnOutputs& o)
{
    constexpr int KCAP = 32, BUF = buf_rows(KCAP);
    if ((o.centroids || o.meandist) && !self) {
        set_error("pcpx: tangent planes / mean distances are defined for the indexed points only");
        return PCPX_ERR_INVALID;
    }
    const u32 npass = (k + KCAP - 1) / KCAP;
    const u32 stride = npass * KCAP;
    const u64 nslots = gcount * GROUP;
    const size_t need = static_cast<size_t>(nslots) * (stride + 2) * sizeof(u64);
    if (need > ix.multi_bytes) {
        if (ix.d_multi) {
            PCPX_HIP(hipStreamSynchronize(ix.stream));
            (void)hipFree(ix.d_multi);
            ix.d_multi = nullptr;
            ix.multi_bytes = 0;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ix.d_multi), need);
        if (e == hipErrorOutOfMemory && ix.pool.cached_bytes() > 0) {  // the handle's staging pool may hold what is missing
            (void)hipGetLastError();
            ix.pool.trim();
            e = hipMalloc(reinterpret_cast<void**>(&ix.d_multi), need);
        }
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu bytes) for the k > 32 key buffer failed: %s", need, hipGetErrorString(e));
            return PCPX_ERR_ALLOC;
        }
        ix.multi_bytes = need;
    }
    u64* keys = ix.d_multi;
    u64* lo[2] = {ix.d_multi + nslots * stride, ix.d_multi + nslots * stride + nslots};
    size_t lds = static_cast<size_t>(WAVES_PER_BLOCK) * lds_rows(BUF, true) * 64 * sizeof(u64);
    u32 gf = static_cast<u32>(gfirst), ge = static_cast<u32>(gfirst + gcount);
    const void* fn = self ? reinterpret_cast<const void*>(k_knn<KCAP, true, 0, true>)
                          : reinterpret_cast<const void*>(k_knn<KCAP, false, 0, true>);
    u32 pgrid = persistent_grid(ix, fn, 64 * WAVES_PER_BLOCK, lds, gcount, WAVES_PER_BLOCK);
    ProfileScope prof(ix, PCPX_K_KNN);
    for (u32 pass = 0; pass < npass; ++pass) {
        MultiPass mp;
        mp.lo = pass ? lo[(pass + 1) & 1] : nullptr;
        mp.lo_out = lo[pass & 1];
        mp.keys = keys;
        mp.stride = stride;
        mp.offset = pass * KCAP;
        mp.slot0 = gf * GROUP;
        const u32 kp = (pass + 1 < npass) ? KCAP : k - pass * KCAP;
        int st = prepare_queue(ix);
        if (st != PCPX_OK) return st;
        KnnOutputs range_only;  // (a pass writes keys, not rows; it still answers the asked positions only)
        range_only.pos_lo = o.pos_lo;
        range_only.pos_hi = o.pos_hi;
        if (self) k_knn<KCAP, true, 0, true><<<pgrid, 64 * WAVES_PER_BLOCK, lds, ix.stream>>>(KnnArgs{ix.view(), qv, gf, ge, kp, eps, -1.f, range_only, mp, ix.queue_now, nullptr, KnnSchedule{}, ix.queue_clear});
        else k_knn<KCAP, false, 0, true><<<pgrid, 64 * WAVES_PER_BLOCK, lds, ix.stream>>>(KnnArgs{ix.view(), qv, gf, ge, kp, eps, -1.f, range_only, mp, ix.queue_now, nullptr, KnnSchedule{}, ix.queue_clear});
    }
    const u32 n32 = static_cast<u32>(nslots);
    if (self) k_assemble<true><<<(n32 + 255) / 256, 256, 0, ix.stream>>>(ix.view(), qv, gf * GROUP, n32, k, stride, keys, o);
    else k_assemble<false><<<(n32 + 255) / 256, 256, 0, ix.stream>>>(ix.view(), qv, gf * GROUP, n32, k, stride, keys, o);
    PCPX_HIP(hipGetLastError());
    if (o.normals || o.centroids || o.meandist) {
        if (!o.idx || !o.cnt) {
            set_error("pcpx: per-neighbourhood products with k > 32 need the neighbour rows as outputs too");
            return PCPX_ERR_INVALID;
        }
        // rows are complete in HBM: PCA normal per row (row order: sorted slots -> rows)
        const u32* rowmap = (self && o.by_position) ? nullptr : self ? ix.perm() : qv.row;
        u64 first = static_cast<u64>(gf) * GROUP, end = first + nslots;
        const u64 nq = self ? ix.n : qv.nq;
        if (end > nq) end = nq;
        if (self) {
            if (first < o.pos_lo) first = o.pos_lo;
            if (end > o.pos_hi) end = o.pos_hi;
        }
        if (end <= first) return PCPX_OK;
        return launch_normals(ix, o.idx, o.cnt, rowmap, first, end - first, k, o.normals, nullptr, o.centroids, o.meandist, o.pos_bias);
    }
    return PCPX_OK;
}

int launch_knn(Index& ix, const QueryView& qv, bool self, u64 group_first, u64 group_count, u32 k, float eps,
               const KnnOutputs& o)
{
    if (group_count == 0) return PCPX_OK;
    eps = sanitize_eps(eps);
    if (k <= 8) return launch_knn_t<8>(ix, qv, self, group_first, group_count, k, eps, o);
    if (k <= 16) return launch_knn_t<16>(ix, qv, self, group_first, group_count, k, eps, o);
    if (k <= 32) return launch_knn_t<32>(ix, qv, self, group_first, group_count, k, eps, o);
    return launch_knn_multipass(ix, qv, self, group_first, group_count, k, eps, o);
}

int launch_knn_cost_sample(Index& ix, u32 k, float eps, u32 stride, u32* d_events, u32* out_samples)
{
    const u64 groups = (ix.n + GROUP - 1) / GROUP;
    if (stride == 0) stride = 1;
    const u32 nsamples = static_cast<u32>(groups / stride);  // (sample i stands for groups [i stride, (i + 1) stride); a last partial block has none)
    *out_samples = nsamples;
    if (nsamples == 0) return PCPX_OK;
    if (k == 0 || k > 32) {
        set_error("pcpx: group costs are sampled with the single-pass kernels: 1 <= k <= 32");
        return PCPX_ERR_UNSUPPORTED;
    }
    eps = sanitize_eps(eps);
    float thr = eps_box_threshold(ix, eps);
    if (thr < 0.f) thr = std::numeric_limits<float>::infinity();  // (as launch_knn_stats: the deferred form, every buffered key tested)
    int st = sched_reserve(ix, nsamples);  // (the sample's group list borrows the schedule's order array: a cut is made before the queries it is for)
    if (st != PCPX_OK) {
        set_error("pcpx: out of device memory for the cost sample's group list");
        return st;
    }
    ix.sched.state = 0;
    k_sample_order<<<(nsamples + 255) / 256, 256, 0, ix.stream>>>(nsamples, stride, ix.sched.d_order);
    if (k <= 8) return k < 8 ? launch_knn_cost_form<8, 1>(ix, nsamples, k, eps, thr, ix.sched.d_order, d_events) : launch_knn_cost_form<8, 0>(ix, nsamples, k, eps, thr, ix.sched.d_order, d_events);
    if (k <= 16) return k < 16 ? launch_knn_cost_form<16, 1>(ix, nsamples, k, eps, thr, ix.sched.d_order, d_events) : launch_knn_cost_form<16, 0>(ix, nsamples, k, eps, thr, ix.sched.d_order, d_events);
    return k < 32 ? launch_knn_cost_form<32, 1>(ix, nsamples, k, eps, thr, ix.sched.d_order, d_events) : launch_knn_cost_form<32, 0>(ix, nsamples, k, eps, thr, ix.sched.d_order, d_events);
}

// {normal, count} at curve positions -> rows by input index: out[i] = nc4[position_of[i]] for the inputs whose position lies in
// [pos_lo, pos_hi).  Coalesced writes; the 16-byte reads hit the cache the kernel has just written through.
__global__ __launch_bounds__(256) void k_gather_nc4(const float4* __restrict__ nc4, const u32* __restrict__ pos_of, u32 n_rows, u32 pos_lo, u32 pos_hi,
                                                     float* __restrict__ normals, u32* __restrict__ cnt)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const u32 p = pos_of[i];
    if (!(p - pos_lo < pos_hi - pos_lo)) return;  // (0xFFFFFFFF: not indexed)
    const float4 v = nc4[p];
    normals[3ull * i] = v.x;
    normals[3ull * i + 1] = v.y;
    normals[3ull * i + 2] = v.z;
    if (cnt) cnt[i] = __float_as_uint(v.w);
}
__global__ __launch_bounds__(256) void k_gather_u32(const u32* __restrict__ at_position, const u32* __restrict__ pos_of, u32 n_rows, u32 pos_lo, u32 pos_hi,
                                                     u32* __restrict__ out)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const u32 p = pos_of[i];
    if (p - pos_lo < pos_hi - pos_lo) out[i] = at_position[p];  // (0xFFFFFFFF: not indexed)
}
int launch_gather_u32(Index& ix, const u32* d_at_position, const u32* d_pos_of, u64 n_rows, u32 pos_lo, u32 pos_hi, u32* d_out)
{
    if (n_rows == 0) return PCPX_OK;
    k_gather_u32<<<static_cast<u32>((n_rows + 255) / 256), 256, 0, ix.stream>>>(d_at_position, d_pos_of, static_cast<u32>(n_rows), pos_lo, pos_hi, d_out);
    return check_hip(hipGetLastError(), "k_gather_u32 launch", __FILE__, __LINE__);
}
int launch_gather_nc4(Index& ix, const float4* d_nc4, const u32* d_pos_of, u64 n_rows, u32 pos_lo, u32 pos_hi, float* d_normals, u32* d_cnt)
{
    if (n_rows == 0) return PCPX_OK;
    k_gather_nc4<<<static_cast<u32>((n_rows + 255) / 256), 256, 0, ix.stream>>>(d_nc4, d_pos_of, static_cast<u32>(n_rows), pos_lo, pos_hi, d_normals, d_cnt);
    return check_hip(hipGetLastError(), "k_gather_nc4 launch", __FILE__, __LINE__);
}

// instrumented self-kNN (k <= 16): traversal statistics summed over all waves into d_stats[8]
int launch_knn_stats(Index& ix, u32 k, float eps, unsigned long long* d_stats, const float* d_known_d2)
{
    constexpr int KCAP = 16, BUF = buf_rows(KCAP);
    u64 groups = (ix.n + GROUP - 1) / GROUP;
    if (groups == 0) return PCPX_OK;
    constexpr int WPB = knn_wpb(KCAP, false);
    size_t lds = static_cast<size_t>(WPB) * lds_rows(BUF, false, KCAP) * 64 * sizeof(u64);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix.n)};
    int st = prepare_queue(ix);
    if (st != PCPX_OK) return st;
    u32 pgrid = persistent_grid(ix, reinterpret_cast<const void*>(k_knn<KCAP, true, 1>), 64 * WPB, lds, groups, WPB);
    // (this instantiation is the deferred-eps form whatever the threshold says: a negative one -- "test every candidate" -- becomes
    //  +inf here, so that every buffered key takes the exact test in the compaction)
    float thr = eps_box_threshold(ix, sanitize_eps(eps));
    if (thr < 0.f) thr = std::numeric_limits<float>::infinity();
    k_knn<KCAP, true, 1><<<pgrid, 64 * WPB, lds, ix.stream>>>(
        KnnArgs{ix.view(), qv, 0u, static_cast<u32>(groups), k, sanitize_eps(eps), thr,
                KnnOutputs{nullptr, nullptr, const_cast<float*>(d_known_d2), nullptr, nullptr, nullptr}, MultiPass{}, ix.queue_now, d_stats, KnnSchedule{}, ix.queue_clear});
    return check_hip(hipGetLastError(), "k_knn stats launch", __FILE__, __LINE__);
}

}  // namespace pcpx
