// pcpx_range.hip -- radius search kernels (sphere and axis-aligned box ranges) for gfx950: the same wave-uniform
// walk as the kNN kernel (pcpx_device.h), one lane = one range, count or CSR fill.
#include "pcpx_device.h"

#ifndef PCPX_RANGE_DIRECT_LEAVES
#define PCPX_RANGE_DIRECT_LEAVES 1
#endif
#ifndef PCPX_RANGE_PACKED_LEAVES
#define PCPX_RANGE_PACKED_LEAVES 16  // count form: a leaf that at most this many lanes need is counted eight needing lanes x eight points at a time (0: off; <= 32: one 512-B row of LDS per wave)
#endif

// Wave priority by phase, as in k_knn (pcpx_query.hip): the dense leaf's 88 vector instructions run at a lower priority than the walk
// and the packed leaves (2.238 -> 2.212 ms per 10 M counts; the dense leaf raised instead: 2.25)
#ifndef PCPX_RANGE_PACKED_FILL
#define PCPX_RANGE_PACKED_FILL 1  // the list form takes the packed leaf too (0: lane-per-range leaves, rounds 1-4)
#endif
#ifndef PCPX_RANGE_PRIO_BASE
#define PCPX_RANGE_PRIO_BASE 1
#endif
#ifndef PCPX_RANGE_PRIO_DENSE
#define PCPX_RANGE_PRIO_DENSE 0
#endif
namespace pcpx {

namespace {

// ------------------------------------------------------------------------------------------------
// sphere range: count / fill (include/pcp/octree/linked_octree_node.hpp:581-614 semantics:
// every point with d2 <= r*r, query included)
// ------------------------------------------------------------------------------------------------
// One group of 64 curve-consecutive queries, one per lane: the wave-uniform walk of pcpx_device.h, a leaf's 8 points
// broadcast from SGPRs, the count kept per lane (compare + add-with-carry: 2 VALU per candidate on top of the 8 of the
// distance -- an exec-masked form would not be shorter).  The kernel sits on both issue pipes, so what pays is what takes
// instructions off both: a last-level node looks at its own leaves (no push and pop), and a leaf that few lanes need is
// counted eight needing lanes x eight points at a time (count form: packed_leaf below; the fill form keeps the lane-per-range
// leaf).  10 M counts at r = 0.01: 2.56 ms (round 3 start) -> 2.40 (direct leaves + round 3's point-per-lane form) -> 2.21
// (round 4: the packed form instead; profiles/experiments/README.md).
template <bool SELF, bool FILL>
__device__ __forceinline__ void range_group(const TreeView& t, const QueryView& qv, const u32 g, const float radius,
                                            const float* __restrict__ radii, u32* __restrict__ out_cnt,
                                            const u64* __restrict__ offsets, u32* __restrict__ out_idx, float4* __restrict__ pub,
                                            const u32 lane)
{
    const u32 p = g * GROUP + lane;
    const u32 nq = SELF ? t.n : qv.nq;
    const bool valid = p < nq && (!SELF || p - qv.pos_lo < qv.pos_hi - qv.pos_lo);  // (self ranges: only the asked positions)
    float qx = 0.f, qy = 0.f, qz = 0.f;
    u32 row = 0;
    if (valid) {
        if (SELF) {
            const Leaf& lf = t.leaves[p / LEAF];
            qx = lf.x[p % LEAF];
            qy = lf.y[p % LEAF];
            qz = lf.z[p % LEAF];
            row = lf.id[p % LEAF];
        } else {
            qx = qv.qx[p];
            qy = qv.qy[p];
            qz = qv.qz[p];
            row = qv.row[p];
        }
    }
    float r = radius;
    if (radii && valid) r = radii[row];
    const float r2 = valid ? r * r : -1.f;  // sphere.hpp:34 radius * radius in float; -1: idle lane
    if (PCPX_RANGE_PRIO_BASE != 0) __builtin_amdgcn_s_setprio(PCPX_RANGE_PRIO_BASE);
    u32 cnt = 0;
    u64 wpos = (FILL && valid) ? offsets[row] : 0;
    auto need = [&](const NodeBox& b) { return box_d2(b, qx, qy, qz) <= r2; };

    auto leaf_record_points = [&](const Leaf& lf) {
        if (!FILL) {
            if (PCPX_RANGE_PRIO_DENSE != PCPX_RANGE_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_RANGE_PRIO_DENSE);
            // count: the eight "inside" masks first (a scalar register pair each), then eight add-with-carry -- from
            // `cnt += in` hipcc pairs the points up as compare, select 0/1 under VCC, compare, add-with-carry, and the select
            // under a VCC that a compare has just written issues in 23 cycles on gfx950 (profiles/r03_valu_issue_rates.txt)
            u64 inside[LEAF];
#pragma unroll
            for (int j = 0; j < LEAF; ++j) {
                float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
                inside[j] = __builtin_amdgcn_ballot_w64(sq3(dx, dy, dz) <= r2);
            }
            static_assert(LEAF == 8, "eight masks");
            u64 carry_out;  // (one statement: between two that pass a register on hipcc puts an s_nop)
            asm("v_addc_co_u32_e64 %0, %1, %0, 0, %2\n\tv_addc_co_u32_e64 %0, %1, %0, 0, %3\n\t"
                "v_addc_co_u32_e64 %0, %1, %0, 0, %4\n\tv_addc_co_u32_e64 %0, %1, %0, 0, %5\n\t"
                "v_addc_co_u32_e64 %0, %1, %0, 0, %6\n\tv_addc_co_u32_e64 %0, %1, %0, 0, %7\n\t"
                "v_addc_co_u32_e64 %0, %1, %0, 0, %8\n\tv_addc_co_u32_e64 %0, %1, %0, 0, %9"
                : "+v"(cnt), "=&s"(carry_out)
                : "s"(inside[0]), "s"(inside[1]), "s"(inside[2]), "s"(inside[3]), "s"(inside[4]), "s"(inside[5]), "s"(inside[6]),
                  "s"(inside[7]));
            if (PCPX_RANGE_PRIO_DENSE != PCPX_RANGE_PRIO_BASE) __builtin_amdgcn_s_setprio(PCPX_RANGE_PRIO_BASE);
            return;
        }
#pragma unroll
        for (int j = 0; j < LEAF; ++j) {
            float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
            bool in = sq3(dx, dy, dz) <= r2;
            if (in) out_idx[wpos + cnt] = lf.id[j];
            cnt += in ? 1u : 0u;
        }
    };
    auto leaf_points = [&](const u32 leaf) { leaf_record_points(load_const(t.leaves + leaf)); };
    // A leaf that at most PCPX_RANGE_PACKED_LEAVES lanes need is counted EIGHT NEEDING LANES x EIGHT POINTS at a time (k_knn's
    // packed_leaf, pcpx_query.hip): the needing lanes publish {centre, r^2} in LDS in the order of their rank among the needing
    // lanes, lane 8 i + j forms the distance from the i-th published centre to point j, and a needing lane adds the number of set
    // bits of its own byte of the step's ballot -- ~12 vector instructions per eight needing lanes (+ 8 per leaf) against 88 per
    // leaf in the lane-per-range form.  Same arithmetic (d = p - c, three roundings; sphere.hpp:27-35).  A slot that holds no
    // centre holds r^2 = -1 (k_range sets the row so, a needing lane sets its slot back): the lanes of a step beyond the leaf's
    // needing lanes count nothing, without a lane mask per step (the scalar unit is as loaded as the vector units here).
    // The list form (round 5) does the same: the needing lanes also publish where their list stands (offset + what they have so far),
    // and a lane whose point is inside writes the point's index at that place plus the number of set bits below its own in its
    // group's byte of the step's ballot -- the order of a list is the order of the lane-per-range form: walk order, then point
    // order inside a leaf.
    constexpr bool packed_leaves = PCPX_RANGE_PACKED_LEAVES > 0 && (!FILL || PCPX_RANGE_PACKED_FILL);
    static_assert(PCPX_RANGE_PACKED_LEAVES <= 32, "one row of LDS per wave");
    auto packed_leaf = [&](const Leaf* record, const u64 who, const u32 how_many) {
        u32 lane_here = lane;
        asm volatile("" : "+v"(lane_here));  // (or what depends on the lane alone is kept in registers for the whole walk)
        const u32 j = lane_here & 7u, i = lane_here >> 3;
        const float* rec = reinterpret_cast<const float*>(record);
        const float cx = rec[j], cy = rec[LEAF + j], cz = rec[2 * LEAF + j];
        u32 idj = 0;
        if (FILL) idj = reinterpret_cast<const u32*>(record)[3 * LEAF + j];
        u64* const pub_at = reinterpret_cast<u64*>(pub + 32);  // (list form: the row's second part)
        const u32 rank = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(who >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(who), 0u));
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(who);
        if (mine) {
            pub[rank] = make_float4(qx, qy, qz, r2);
            if (FILL) pub_at[rank] = wpos + cnt;
        }
        __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations complete in order; this only pins the compiler's order)
        const u32 my_byte = (rank & 7u) << 3;
        for (u32 s = 0; s < how_many; s += 8u) {
            const float4 q = pub[s + i];
            const float dx = cx - q.x, dy = cy - q.y, dz = cz - q.z;
            const bool in_here = sq3(dx, dy, dz) <= q.w;  // (a NaN padding point fails; so does every point against an empty slot)
            const u64 inside = __builtin_amdgcn_ballot_w64(in_here);
            if (FILL && in_here) {
                const u32 group_byte = static_cast<u32>(inside >> (lane_here & 56u)) & 0xFFu;  // the eight points against published range s + i
                out_idx[pub_at[s + i] + static_cast<u32>(__builtin_popcount(group_byte & ((1u << j) - 1u)))] = idj;
            }
            if (mine && rank - s < 8u) cnt += static_cast<u32>(__builtin_popcount(static_cast<u32>(inside >> my_byte) & 0xFFu));
        }
        __builtin_amdgcn_wave_barrier();
        if (mine) reinterpret_cast<float*>(pub + rank)[3] = -1.f;
        __builtin_amdgcn_wave_barrier();
    };
    // the walk, with "is there another leaf" in the control flow rather than in a value (WalkerT::pop, pcpx_device.h)
    WalkerT<true, packed_leaves> wk;
    u32 nexp = 0;
    if (wk.start(t, need, nexp))  // the root is the only unit
        for (u32 leaf = 0; leaf < static_cast<u32>(UNIT_LEAVES) && leaf < t.nleaves; ++leaf) leaf_points(leaf);
    // A last-level node looks at its needed leaves itself (WalkerT::leaves_of) instead of pushing and popping them: the four
    // children written out, so that a child's record is an immediate offset from the node's first leaf and its lanes' ballot a
    // register pair known at compile time (height 0 is popped only when the root's own children are leaves).
    while (!wk.done()) {  // one pop per trip
        u32 loc;
        const int h = wk.pop(loc);
        if (h > 1 || (!PCPX_RANGE_DIRECT_LEAVES && h == 1)) {
            wk.expand(t, h, loc, need);
        } else if (PCPX_RANGE_DIRECT_LEAVES || h == 1) {  // (no tree has depth 1 -- depth_of, pcpx_build.hip --: with last-level nodes looking
                                                          //  at their leaves themselves no leaf is ever popped, and the case below is not compiled)
            // (the children of a last-level node are UNITS of UNIT_LEAVES leaf records under one box)
            const u32 needed = wk.leaves_of(t, loc, need);
            const Leaf* records = t.leaves + (loc << LOGW) * UNIT_LEAVES;
#pragma unroll
            for (int c = 0; c < W; ++c) {
                if ((needed >> c) & 1u) {
                    u32 how_many = GROUP;
                    if (packed_leaves) asm("s_bcnt1_i32_b64 %0, %1" : "=s"(how_many) : "s"(wk.leaf_need[c]) : "scc");
#pragma unroll
                    for (int r = 0; r < UNIT_LEAVES; ++r) {
                        if (UNIT_LEAVES > 1 && ((loc << LOGW) + c) * UNIT_LEAVES + r >= t.nleaves) break;  // (the cloud's last unit may hold one leaf)
                        if (packed_leaves && how_many <= static_cast<u32>(PCPX_RANGE_PACKED_LEAVES)) packed_leaf(records + c * UNIT_LEAVES + r, wk.leaf_need[c], how_many);
                        else leaf_record_points(load_const(records + c * UNIT_LEAVES + r));
                    }
                }
            }
        } else {
            wk.at_leaf(loc);
            for (u32 leaf = loc * UNIT_LEAVES; leaf < (loc + 1u) * UNIT_LEAVES && leaf < t.nleaves; ++leaf) leaf_points(leaf);
        }
    }
    if (valid && !FILL) out_cnt[(SELF && qv.by_position) ? p + qv.pos_bias : row] = cnt;
}

// One single-wave workgroup per group, XCD-aware block order (pcpx_device.h: virtual_block).  (Tried: a persistent grid
// pulling groups from the 8 work queues like k_knn -- 10 M counts at r = 0.01 went from 2.8 to 3.1 ms: these groups are
// short and even, the dispatcher is the better scheduler here.)
template <bool SELF, bool FILL>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_range(TreeView t, QueryView qv, u32 group_first, u32 group_end, float radius,
                                                               const float* __restrict__ radii, u32* __restrict__ out_cnt,
                                                               const u64* __restrict__ offsets, u32* __restrict__ out_idx)
{
    __shared__ float4 published[WAVES_PER_BLOCK][FILL ? 48 : 32];  // packed_leaf's row, one per wave (list form: + 32 list positions)
    const u32 lane = threadIdx.x & 63u;
    const u32 g = group_first + virtual_block() * WAVES_PER_BLOCK + wave_in_block();
    if (g >= group_end) return;
    if (lane < 32u) published[wave_in_block()][lane].w = -1.f;  // packed_leaf's invariant: a slot that holds no centre holds r^2 = -1
    range_group<SELF, FILL>(t, qv, g, radius, radii, out_cnt, offsets, out_idx, published[wave_in_block()], lane);
}

// AABB ranges: one wave per 64 boxes, no spatial coherence assumed (boxes are few in practice:
// test/octree/octree_range_search.cpp:80-118).  contains() is inclusive
// (axis_aligned_bounding_box.hpp:111-125); prune = box/box overlap (intersections.hpp:25-32).
template <bool FILL>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_range_aabb(TreeView t, const float* __restrict__ boxes6, u32 nb,
                                                    u32* __restrict__ out_cnt, const u64* __restrict__ offsets,
                                                    u32* __restrict__ out_idx)
{
    const u32 lane = threadIdx.x & 63u;
    const u32 g = blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    const u32 p = g * GROUP + lane;
    const bool valid = p < nb;
    // an idle lane gets an inverted box that contains and overlaps nothing
    float b0 = 1.f, b1 = 1.f, b2 = 1.f, b3 = -1.f, b4 = -1.f, b5 = -1.f;
    if (valid) {
        b0 = boxes6[6ull * p]; b1 = boxes6[6ull * p + 1]; b2 = boxes6[6ull * p + 2];
        b3 = boxes6[6ull * p + 3]; b4 = boxes6[6ull * p + 4]; b5 = boxes6[6ull * p + 5];
    }
    u32 cnt = 0;
    u64 wpos = (FILL && valid) ? offsets[p] : 0;
    auto need = [&](const NodeBox& n) {
        const float lx = n.lo(0), ly = n.lo(1), lz = n.lo(2), hx = n.hi(0), hy = n.hi(1), hz = n.hi(2);
        bool o = (hx >= b0) & (hy >= b1) & (hz >= b2) & (lx <= b3) & (ly <= b4) & (lz <= b5);
        return o & (n.poison == 0.f) & valid;
    };
    auto leaf_points = [&](const u32 leaf) {
        const Leaf lf = load_const(t.leaves + leaf);
#pragma unroll
        for (int j = 0; j < LEAF; ++j) {
            float x = lf.x[j], y = lf.y[j], z = lf.z[j];
            bool in = valid & (x >= b0) & (y >= b1) & (z >= b2) & (x <= b3) & (y <= b4) & (z <= b5);
            if (FILL) {
                if (in) out_idx[wpos + cnt] = lf.id[j];
            }
            cnt += in ? 1u : 0u;
        }
    };
    // The boxes of a wave need not lie near each other, so a leaf is needed by one or two of them as a rule: such a leaf is looked
    // at EIGHT NEEDING BOXES x EIGHT POINTS at a time, as in k_range (round 5; until then every visited leaf cost all 64 lanes its
    // eight containment tests).  The needing lanes publish their box -- and, list form, where their list stands -- by their rank
    // among the needing lanes; an unused slot holds an inverted box.
    __shared__ float4 pub_lo_s[WAVES_PER_BLOCK][PCPX_RANGE_PACKED_LEAVES > 0 ? 32 : 1], pub_hi_s[WAVES_PER_BLOCK][PCPX_RANGE_PACKED_LEAVES > 0 ? 32 : 1];
    __shared__ u64 pub_at_s[WAVES_PER_BLOCK][(PCPX_RANGE_PACKED_LEAVES > 0 && FILL) ? 32 : 1];
    constexpr bool packed_leaves = PCPX_RANGE_PACKED_LEAVES > 0 && PCPX_RANGE_PACKED_FILL;
    float4* const pub_lo = pub_lo_s[wave_in_block()];
    float4* const pub_hi = pub_hi_s[wave_in_block()];
    u64* const pub_at = pub_at_s[wave_in_block()];
    if (packed_leaves && lane < 32u) {
        pub_lo[lane] = make_float4(1.f, 1.f, 1.f, 0.f);
        pub_hi[lane] = make_float4(-1.f, -1.f, -1.f, 0.f);
    }
    auto packed_leaf = [&](const Leaf* record, const u64 who, const u32 how_many) {
        const u32 j = lane & 7u, i = lane >> 3;
        const float* rec = reinterpret_cast<const float*>(record);
        const float x = rec[j], y = rec[LEAF + j], z = rec[2 * LEAF + j];
        u32 idj = 0;
        if (FILL) idj = reinterpret_cast<const u32*>(record)[3 * LEAF + j];
        const u32 rank = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(who >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(who), 0u));
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(who);
        if (mine) {
            pub_lo[rank] = make_float4(b0, b1, b2, 0.f);
            pub_hi[rank] = make_float4(b3, b4, b5, 0.f);
            if (FILL) pub_at[rank] = wpos + cnt;
        }
        __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations complete in order; this only pins the compiler's order)
        const u32 my_byte = (rank & 7u) << 3;
#pragma nounroll
        for (u32 s = 0; s < how_many; s += 8u) {
            const float4 lo = pub_lo[s + i], hi = pub_hi[s + i];
            const bool in_here = (x >= lo.x) & (y >= lo.y) & (z >= lo.z) & (x <= hi.x) & (y <= hi.y) & (z <= hi.z);  // (NaN padding fails)
            const u64 inside = __builtin_amdgcn_ballot_w64(in_here);
            if (FILL && in_here) {
                const u32 group_byte = static_cast<u32>(inside >> (lane & 56u)) & 0xFFu;
                out_idx[pub_at[s + i] + static_cast<u32>(__builtin_popcount(group_byte & ((1u << j) - 1u)))] = idj;
            }
            if (mine && rank - s < 8u) cnt += static_cast<u32>(__builtin_popcount(static_cast<u32>(inside >> my_byte) & 0xFFu));
        }
        __builtin_amdgcn_wave_barrier();
        if (mine) {
            pub_lo[rank] = make_float4(1.f, 1.f, 1.f, 0.f);
            pub_hi[rank] = make_float4(-1.f, -1.f, -1.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
    };
    WalkerT<true, packed_leaves> wk;
    u32 nexp = 0;
    if (wk.start(t, need, nexp))
        for (u32 leaf = 0; leaf < static_cast<u32>(UNIT_LEAVES) && leaf < t.nleaves; ++leaf) leaf_points(leaf);
    while (!wk.done()) {  // one pop per trip: a node is expanded; a last-level node looks at its needed units itself
        u32 loc;
        const int h = wk.pop(loc);
        if (h > 1 || (!packed_leaves && h == 1)) {
            wk.expand(t, h, loc, need);
        } else if (packed_leaves || h == 1) {  // (as in range_group: no leaf is ever popped)
            u32 needed = wk.leaves_of(t, loc, need);
            // (one copy of the leaf forms, the needed children in a loop: written out four times two records they cost the count kernel
            //  133 saved scalar registers)
#pragma nounroll
            while (needed != 0u) {
                u32 c;
                asm("s_ff1_i32_b32 %0, %1\n\ts_bitset0_b32 %1, %0" : "=&s"(c), "+s"(needed));
                u64 who;
                u32 how_many;
                asm("s_cmp_eq_u32 %[c], 2\n\ts_cselect_b64 %[w], %[n2], %[n3]\n\t"
                    "s_cmp_eq_u32 %[c], 1\n\ts_cselect_b64 %[w], %[n1], %[w]\n\t"
                    "s_cmp_eq_u32 %[c], 0\n\ts_cselect_b64 %[w], %[n0], %[w]\n\t"
                    "s_bcnt1_i32_b64 %[m], %[w]"
                    : [w] "=&s"(who), [m] "=s"(how_many)
                    : [c] "s"(c), [n0] "s"(wk.leaf_need[0]), [n1] "s"(wk.leaf_need[1]), [n2] "s"(wk.leaf_need[2]), [n3] "s"(wk.leaf_need[3])
                    : "scc");
                const u32 first = ((loc << LOGW) + c) * UNIT_LEAVES;
#pragma unroll 1
                for (u32 leaf = first; leaf < first + UNIT_LEAVES && leaf < t.nleaves; ++leaf) {
                    if (how_many <= static_cast<u32>(PCPX_RANGE_PACKED_LEAVES)) packed_leaf(t.leaves + leaf, who, how_many);
                    else leaf_points(leaf);
                }
            }
        } else {
            wk.at_leaf(loc);
            for (u32 leaf = loc * UNIT_LEAVES; leaf < (loc + 1u) * UNIT_LEAVES && leaf < t.nleaves; ++leaf) leaf_points(leaf);
        }
    }
    if (valid && !FILL) out_cnt[p] = cnt;
}

// ONE range, latency form (the reference's per-call shape: benchmark/spatial_data_structures_benchmark.cpp:169-213): a single
// wavefront sweeps the tree breadth first for the one range -- the (up to 64) nodes of level 3 by one lane each, then two
// levels per step (a node's 16 grandchildren are contiguous in the heap layout), the surviving nodes compacted by ballot +
// prefix count into a frontier in LDS, in ascending order: the leaves come out in curve order like the batch form's walk --,
// then one lane per point of the surviving leaves.  depth / 2 + 2 dependent round trips (the depth-first walk of the batch
// kernels: one per node and leaf on the way).  The range travels in the kernel arguments; the count and up to `cap` indices
// live in the handle's pinned stage (host memory the device writes in place), and the host polls the completion word: one
// launch, no copy, no stream synchronisation -- the count / scan / fill sequence of the batch form is two launches and two
// synchronisations (41 us per call for a range that holds a handful of points).  A frontier of more than RANGE_FRONTIER nodes
// reports cap + 1 matches: the caller takes the batch form then, as for more than cap matches.
constexpr u32 RANGE_FRONTIER = 2048;
template <bool AABB>
__global__ __launch_bounds__(64) void k_range_one(TreeView t, float a0, float a1, float a2, float a3, float a4, float a5, u32 cap,
                                                  u32* __restrict__ out_idx, u32* __restrict__ out_cnt, u32* __restrict__ done_flag,
                                                  u32 epoch)
{
    __shared__ u32 front[2][RANGE_FRONTIER];
    const u32 lane = threadIdx.x;
    const float r2 = a3 * a3;  // sphere.hpp:34 radius * radius in float
    auto need = [&](const NodeBox& b) -> bool {
        const float lx = b.lo(0), ly = b.lo(1), lz = b.lo(2), hx = b.hi(0), hy = b.hi(1), hz = b.hi(2);
        if (AABB) return (hx >= a0) & (hy >= a1) & (hz >= a2) & (lx <= a3) & (ly <= a4) & (lz <= a5) & (b.poison == 0.f);
        return box_d2(b, a0, a1, a2) <= r2;  // (NaN for a padding node)
    };
    auto level_base = [](int d) { return d == 0 ? 0u : (0x55555555u >> (32 - 2 * d)); };
    // real nodes of tree level l (the ones the build writes: the children of a real node need not be real)
    auto nreal = [&](int l) { return (t.nunits() + (1u << (2 * (t.depth - l))) - 1u) >> (2 * (t.depth - l)); };
    u32 cnt = 0, m = 0;
    bool overflow = false;
    int cur = 0;
    if (t.nleaves > 0) {
        const int l0 = t.depth < 3 ? t.depth : 3;
        bool keep = false;
        if (lane < (1u << (2 * l0)) && lane < nreal(l0)) keep = need(t.nodes[level_base(l0) + lane]);
        const u64 mask = __builtin_amdgcn_ballot_w64(keep);
        const u32 below = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mask), 0u));
        if (keep) front[0][below] = lane;
        m = static_cast<u32>(__builtin_popcountll(mask));
        __syncthreads();
        for (int d = l0; d < t.depth && m > 0;) {
            const int s = t.depth - d >= 2 ? 2 : 1;
            const u32 fan = 1u << (2 * s);
            u32 next = 0;
            for (u32 c0 = 0; c0 < fan * m; c0 += 64u) {
                const u32 c = c0 + lane;
                bool k2 = false;
                u32 node = 0;
                if (c < fan * m) {
                    node = front[cur][c >> (2 * s)] * fan + (c & (fan - 1u));
                    if (node < nreal(d + s)) k2 = need(t.nodes[level_base(d + s) + node]);
                }
                const u64 mk = __builtin_amdgcn_ballot_w64(k2);
                const u32 bl = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mk >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mk), 0u));
                if (k2 && next + bl < RANGE_FRONTIER) front[cur ^ 1][next + bl] = node;
                next += static_cast<u32>(__builtin_popcountll(mk));
            }
            if (next > RANGE_FRONTIER) {
                overflow = true;
                next = RANGE_FRONTIER;
            }
            d += s;
            m = next;
            cur ^= 1;
            __syncthreads();
        }
        // the points of the surviving units, one per lane, in leaf (= curve) order
        for (u32 c0 = 0; c0 < static_cast<u32>(UNIT_POINTS) * m && !overflow; c0 += 64u) {
            const u32 c = c0 + lane;
            bool in = false;
            u32 id = 0;
            if (c < static_cast<u32>(UNIT_POINTS) * m) {
                const u32 leaf = front[cur][c / UNIT_POINTS] * UNIT_LEAVES + (c % UNIT_POINTS) / LEAF;
                if (leaf < t.nleaves) {
                    const Leaf& lf = t.leaves[leaf];
                    const u32 sl = c & 7u;
                    const float x = lf.x[sl], y = lf.y[sl], z = lf.z[sl];  // (NaN padding fails every comparison below)
                    id = lf.id[sl];
                    if (AABB) {
                        in = (x >= a0) & (y >= a1) & (z >= a2) & (x <= a3) & (y <= a4) & (z <= a5);
                    } else {
                        const float dx = x - a0, dy = y - a1, dz = z - a2;
                        in = sq3(dx, dy, dz) <= r2;
                    }
                }
            }
            const u64 mk = __builtin_amdgcn_ballot_w64(in);
            const u32 at = cnt + __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mk >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mk), 0u));
            if (in && at < cap) out_idx[at] = id;
            cnt += static_cast<u32>(__builtin_popcountll(mk));
        }
    }
    if (overflow) cnt = cap + 1u;
    if (lane == 0) *out_cnt = cnt;
    __threadfence_system();  // the row and the count are visible to the host before the completion word
    if (lane == 0) __hip_atomic_store(done_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// range: on the host -- sphere {x, y, z, r} or box {min x, y, z, max x, y, z}: it travels in the kernel arguments
int launch_range_one(Index& ix, bool aabb, const float* range, u32 cap, u32* out_idx, u32* out_cnt, u32* done_flag, u32 epoch)
{
    ProfileScope prof(ix, PCPX_K_RANGE);
    if (aabb) k_range_one<true><<<1, 64, 0, ix.stream>>>(ix.view(), range[0], range[1], range[2], range[3], range[4], range[5], cap, out_idx, out_cnt, done_flag, epoch);
    else k_range_one<false><<<1, 64, 0, ix.stream>>>(ix.view(), range[0], range[1], range[2], range[3], 0.f, 0.f, cap, out_idx, out_cnt, done_flag, epoch);
    return check_hip(hipGetLastError(), "k_range_one launch", __FILE__, __LINE__);
}

int launch_range_count(Index& ix, const QueryView& qv, bool self, u64 group_first, u64 group_count, float radius,
                       const float* d_radii, u32* d_out_cnt)
{
    if (group_count == 0) return PCPX_OK;
    u32 grid = grid_for_groups(group_count);
    u32 gf = static_cast<u32>(group_first), ge = static_cast<u32>(group_first + group_count);
    ProfileScope prof(ix, PCPX_K_RANGE);
    if (self)
        k_range<true, false><<<grid, 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), qv, gf, ge, radius, d_radii, d_out_cnt, nullptr, nullptr);
    else
        k_range<false, false><<<grid, 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), qv, gf, ge, radius, d_radii, d_out_cnt, nullptr, nullptr);
    return check_hip(hipGetLastError(), "k_range launch", __FILE__, __LINE__);
}

int launch_range_fill(Index& ix, const QueryView& qv, float radius, const float* d_radii, const u64* d_offsets,
                      u32* d_out_idx)
{
    u64 groups = (static_cast<u64>(qv.nq) + GROUP - 1) / GROUP;
    if (groups == 0) return PCPX_OK;
    ProfileScope prof(ix, PCPX_K_RANGE);
    k_range<false, true><<<grid_for_groups(groups), 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), qv, 0u, static_cast<u32>(groups), radius,
                                                                           d_radii, nullptr, d_offsets, d_out_idx);
    return check_hip(hipGetLastError(), "k_range fill launch", __FILE__, __LINE__);
}

// ---- lists of every indexed point's range, device resident: counts by input row -> offsets (exclusive scan, 64-bit) -> fill ----
namespace {
constexpr u32 SCAN_TILE = 1024;
// exclusive scan of n counts into 64-bit offsets (n + 1 of them): tile sums, their scan by one block, the tiles
__global__ __launch_bounds__(256) void k_offsets_tile_sums(const u32* __restrict__ cnt, u32 n, u64* __restrict__ tile_sum)
{
    __shared__ u64 w[4];
    const u32 base = blockIdx.x * SCAN_TILE;
    u64 v = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_TILE / 256; ++j) {
        const u32 i = base + j * 256 + threadIdx.x;
        v += i < n ? cnt[i] : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63u) == 0) w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
__global__ __launch_bounds__(1024) void k_offsets_scan_sums(u64* __restrict__ tile_sum, u32 ntiles, u64* __restrict__ total_out)
{
    __shared__ u64 wsum[16];
    __shared__ u64 carry_s;
    const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < ntiles; base += 1024) {
        const u32 i = base + t;
        const u64 v = i < ntiles ? tile_sum[i] : 0ull;
        u64 incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 up = __shfl_up(incl, off);
            if (lane >= static_cast<u32>(off)) incl += up;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        u64 before = carry_s, total = 0;
        for (u32 j = 0; j < 16; ++j) {
            before += j < w ? wsum[j] : 0ull;
            total += wsum[j];
        }
        if (i < ntiles) tile_sum[i] = before + incl - v;
        __syncthreads();
        if (t == 0) carry_s += total;
        __syncthreads();
    }
    if (t == 0) *total_out = carry_s;
}
__global__ __launch_bounds__(256) void k_offsets_tiles(const u32* __restrict__ cnt, u32 n, const u64* __restrict__ tile_base, u64* __restrict__ offsets)
{
    __shared__ u32 w[4];
    const u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * (SCAN_TILE / 256);
    u32 c[SCAN_TILE / 256], s = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_TILE / 256; ++j) {
        c[j] = base + j < n ? cnt[base + j] : 0u;
        s += c[j];
    }
    const u32 lane = threadIdx.x & 63u;
    u32 incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 up = __shfl_up(incl, off);
        if (lane >= static_cast<u32>(off)) incl += up;
    }
    if (lane == 63) w[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 before = 0;
    for (u32 j = 0; j < (threadIdx.x >> 6); ++j) before += w[j];
    u64 at = tile_base[blockIdx.x] + before + incl - s;
#pragma unroll
    for (u32 j = 0; j < SCAN_TILE / 256; ++j) {
        if (base + j < n) offsets[base + j] = at;
        at += c[j];
    }
}
}  // namespace

// d_offsets: n_rows + 1 entries (rows = input indices; a point that is not indexed has an empty list); d_total: one u64 (device).
// d_cnt: n_rows counts (scratch of the caller); d_tile_sum: ceil(n_rows / 1024) + 1 u64 (scratch).
int launch_range_offsets(Index& ix, const u32* d_cnt, u64 n_rows, u64* d_tile_sum, u64* d_offsets)
{
    if (n_rows == 0) return PCPX_OK;
    const u32 n = static_cast<u32>(n_rows), ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    k_offsets_tile_sums<<<ntiles, 256, 0, ix.stream>>>(d_cnt, n, d_tile_sum);
    k_offsets_scan_sums<<<1, 1024, 0, ix.stream>>>(d_tile_sum, ntiles, d_offsets + n_rows);
    k_offsets_tiles<<<ntiles, 256, 0, ix.stream>>>(d_cnt, n, d_tile_sum, d_offsets);
    return check_hip(hipGetLastError(), "range offset kernels", __FILE__, __LINE__);
}
int launch_range_fill_self(Index& ix, u64 group_first, u64 group_count, float radius, const u64* d_offsets, u32* d_out_idx)
{
    if (group_count == 0) return PCPX_OK;
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix.n)};
    ProfileScope prof(ix, PCPX_K_RANGE);
    k_range<true, true><<<grid_for_groups(group_count), 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), qv, static_cast<u32>(group_first),
                                                                                 static_cast<u32>(group_first + group_count), radius, nullptr, nullptr,
                                                                                 d_offsets, d_out_idx);
    return check_hip(hipGetLastError(), "k_range self fill launch", __FILE__, __LINE__);
}

int launch_aabb_count(Index& ix, const float* d_boxes6, u64 nb, u32* d_out_cnt)
{
    if (nb == 0) return PCPX_OK;
    u32 grid = static_cast<u32>((nb + 64 * WAVES_PER_BLOCK - 1) / (64 * WAVES_PER_BLOCK));
    ProfileScope prof(ix, PCPX_K_RANGE);
    k_range_aabb<false><<<grid, 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), d_boxes6, static_cast<u32>(nb), d_out_cnt, nullptr, nullptr);
    return check_hip(hipGetLastError(), "k_range_aabb launch", __FILE__, __LINE__);
}

int launch_aabb_fill(Index& ix, const float* d_boxes6, u64 nb, const u64* d_offsets, u32* d_out_idx)
{
    if (nb == 0) return PCPX_OK;
    u32 grid = static_cast<u32>((nb + 64 * WAVES_PER_BLOCK - 1) / (64 * WAVES_PER_BLOCK));
    ProfileScope prof(ix, PCPX_K_RANGE);
    k_range_aabb<true><<<grid, 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), d_boxes6, static_cast<u32>(nb), nullptr, d_offsets, d_out_idx);
    return check_hip(hipGetLastError(), "k_range_aabb fill launch", __FILE__, __LINE__);
}

}  // namespace pcpx
