// pcpx_device.h -- device-side helpers shared by the query kernels (kNN, range search, normals): arithmetic of
// the reference (squared distance, box lower bound), scalar (SMEM) record loads, and the wave-uniform walk
// over the implicit 4-ary tree.  Included by .hip translation units only.
#ifndef PCPX_DEVICE_H
#define PCPX_DEVICE_H

#include "pcpx_internal.h"

#include <cmath>
#include <limits>
#include <type_traits>

#pragma clang fp contract(off)

namespace pcpx {
namespace {

#ifndef PCPX_WPB
#define PCPX_WPB 1
#endif
constexpr int WAVES_PER_BLOCK = PCPX_WPB;  // 1: a finished wave frees its LDS at once (no intra-block tail)

__device__ __forceinline__ u32 wave_in_block() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// XCD-aware block remap: hardware deals blocks round-robin over the 8 XCDs, so give XCD x the
// contiguous range of virtual blocks [x*per, (x+1)*per): curve neighbours then share one L2.
__device__ __forceinline__ u32 virtual_block()
{
    u32 per = gridDim.x >> 3;  // grid is a multiple of 8
    return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}

__device__ __forceinline__ float sq3(float dx, float dy, float dz) { return dx * dx + dy * dy + dz * dz; }

// squared distance from q to the box (+ poison): equals d2(q, clamp(q, box)) of
// include/pcp/common/axis_aligned_bounding_box.hpp:138-148 and is a lower bound, in float arithmetic,
// of sq3(p - q) for every p inside the box; NaN for a padding node.
typedef float float_pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float box_d2(const NodeBox& b, float qx, float qy, float qz)
{
    // (v_med3_f32(q, lo, hi) takes two SGPR operands, which gfx9 VALU encodings do not allow: no cheaper.  Two axes per
    //  packed-float instruction -- v_pk_add_f32 / v_pk_mul_f32 on register pairs, 12 instead of 16 instructions per box -- was
    //  measured in round 4: k_knn the same, k_range 8 % slower; profiles/experiments/README.md)
    float dx = fmaxf(fmaxf(b.lo(0) - qx, qx - b.hi(0)), 0.f);
    float dy = fmaxf(fmaxf(b.lo(1) - qy, qy - b.hi(1)), 0.f);
    float dz = fmaxf(fmaxf(b.lo(2) - qz, qz - b.hi(2)), 0.f);
    return sq3(dx, dy, dz) + b.poison;
}

struct NodeBox4 {
    NodeBox c[W];
};

// The four child boxes of a node as two 64-byte scalar loads.  (From load_const hipcc fetches only the seven used words of
// each box -- x4 + x2 + x1: twelve SMEM instructions per expansion, and the scalar side of the query kernels is as loaded
// as their vector side.)  The wait is part of the statement: the compiler does not count loads it cannot see.
__device__ __forceinline__ NodeBox4 load_node4(const NodeBox* first_child)
{
    typedef u32 u32x16 __attribute__((ext_vector_type(16)));
    u32x16 a, b;
    // (the address is wave-uniform by construction; where hipcc cannot see that, this pins it to scalar registers)
    const u64 address = reinterpret_cast<uintptr_t>(first_child);
    const u32 address_lo = __builtin_amdgcn_readfirstlane(static_cast<u32>(address));  // (the builtin returns int: no sign extension)
    const u32 address_hi = __builtin_amdgcn_readfirstlane(static_cast<u32>(address >> 32));
    const u64 uniform = (static_cast<u64>(address_hi) << 32) | address_lo;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b)
                 : "s"(uniform));
    NodeBox4 out;
    u32* o = reinterpret_cast<u32*>(&out);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        o[i] = a[i];
        o[16 + i] = b[i];
    }
    return out;
}

__device__ __forceinline__ bool any_lane(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// Index records (leaves, node boxes) are immutable while a query kernel runs.  Reading them through
// constant-address-space pointers makes every wave-uniform read a scalar (SMEM) load unconditionally;
// through generic pointers hipcc only does that while it can prove no store (or asm with a memory
// clobber, like append_if) may alias them.
template <class T>
__device__ __forceinline__ T load_const(const T* p)
{
    static_assert(sizeof(T) % 4 == 0, "record size");
    typedef const __attribute__((address_space(4))) u32* const_u32_ptr;
    const_u32_ptr c = (const_u32_ptr)(reinterpret_cast<uintptr_t>(p));
    T out;
    u32* o = reinterpret_cast<u32*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = c[i];
    return out;
}

__device__ __forceinline__ u32 lds_address(const void* p)
{
    return static_cast<u32>(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a generic LDS pointer = LDS offset
}

// ---- wave-uniform walk over the implicit 4-ary tree ------------------------------------------------

// All members are wave-uniform (SGPRs).  next() yields, in curve (= sorted) order, every leaf whose box is still
// needed by at least one lane at the time its parent is expanded.  State: bit 4*h + c of `pend` = child c
// (a node of height h; leaves have height 0) of the current ancestor of height h+1 is still to visit.  A
// depth-first walk always continues with the LOWEST set bit of pend, so popping is one find-first-set: no
// per-level loop.  `ploc` is the level-local index of the ancestor of height l+1 (node ids are never
// stored: heap id = (4^d - 1)/3 + local index at tree level d, and a leaf's local index is its number).
// X16: the child boxes come as two 64-byte loads (load_node4) -- fewer scalar instructions, four more SGPRs at once; the
// kernel picks (k_knn for k <= 8, at 8 waves per SIMD, is faster without).
// KEEP: the lanes that needed each of the last expanded node's children are kept (leaf_need) -- k_knn looks at a leaf that few
// lanes need in a form of its own.
template <bool X16, bool KEEP = false>
struct WalkerT {
    u64 pend;
    u32 ploc;
    int l;
    u64 leaf_need[W];  // KEEP: valid for the children of the most recently expanded node (the leaves popped next, if it was a last-level node)

    // first heap id of tree level d >= 1: (4^d - 1) / 3 = 0b0101...01 (d pairs)
    static __device__ __forceinline__ u32 level_base(int d) { return 0x55555555u >> (32 - 2 * d); }

    // bit c set: child c of the node with local index `loc` at tree level d is needed by some lane
    template <class Need>
    __device__ __forceinline__ u32 child_mask(const TreeView& t, int d, u32 loc, Need&& need)
    {
        const u32 first_child = level_base(d + 1) + (loc << LOGW);  // heap id of child 0
        const NodeBox4 cb = X16 ? load_node4(t.nodes + first_child) : load_const(reinterpret_cast<const NodeBox4*>(t.nodes + first_child));
        // the mask is built on the scalar unit, two instructions per child: "some lane needs it" goes to SCC and is shifted
        // into the mask with an add-with-carry (children in reverse order, so child 0 ends up in bit 0).  (Left to itself hipcc
        // makes the mask a vector value: v_cndmask, three v_or and a v_readfirstlane per expansion.)
        u32 m = 0;
#pragma unroll
        for (int c = W - 1; c >= 0; --c) {
            const u64 lanes = __builtin_amdgcn_ballot_w64(need(cb.c[c]));
            if (KEEP) leaf_need[c] = lanes;  // (kept at every expansion, though only a last-level node's are read: keeping them only there
                                             //  measured 2 % SLOWER -- hipcc then moves the masks about at the loop's edges)
            asm("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0" : "+s"(m) : "s"(lanes) : "scc");
        }
        return m;
    }

    // returns true if the root itself is the single leaf (depth 0) and is needed
    template <class Need>
    __device__ __forceinline__ bool start(const TreeView& t, Need&& need, u32& n_expand)
    {
        pend = 0;
        ploc = 0;
        l = 0;
        if (t.nleaves == 0) return false;
        const NodeBox root = load_const(t.nodes);
        if (!any_lane(need(root))) return false;
        if (t.depth == 0) return true;
        ++n_expand;
        l = t.depth - 1;
        pend = static_cast<u64>(child_mask(t, 0, 0u, need)) << (W * l);
        return false;
    }

    // The same walk as next(), in pieces, for a caller that keeps "is there another leaf" in its control flow instead of in
    // a value (hipcc turns a wave-uniform bool that lives across blocks into a 64-bit lane mask: s_cselect_b64, s_and_b64
    // with EXEC and a VCC branch where one SCC branch would do; k_knn's scalar side is as loaded as its vector side).
    //   while (!wk.done()) { u32 loc; int h = wk.pop(loc); if (h == 0) { wk.at_leaf(loc); ...leaf loc... } else wk.expand(t, h, loc, need); }
    __device__ __forceinline__ bool done() const { return pend == 0; }
    __device__ __forceinline__ int pop(u32& loc)
    {
        int bit;  // (s_bitset0_b64: from `pend &= ~(1ull << bit)` hipcc makes a shift and an and-not)
        asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(bit), "+s"(pend));
        const int h = bit >> LOGW;
        loc = ((ploc >> (LOGW * (h - l))) << LOGW) + (static_cast<u32>(bit) & (W - 1u));  // climb h - l levels, step down
        return h;
    }
    __device__ __forceinline__ void at_leaf(u32 loc)
    {
        ploc = loc >> LOGW;
        l = 0;
    }
    template <class Need>
    __device__ __forceinline__ void expand(const TreeView& t, int h, u32 loc, Need&& need)
    {
        l = h - 1;
        ploc = loc;
        pend |= static_cast<u64>(child_mask(t, t.depth - h, loc, need)) << (W * l);
    }

    // The children of a last-level node (height 1) are leaves: their mask without the trip through the pending bits.  The
    // caller looks at leaves (loc << LOGW) + c for the set bits c and goes on popping (the walker is left as after the last of
    // them: expand() + the pops + at_leaf() of that path cost ~13 scalar instructions per leaf).
    template <class Need>
    __device__ __forceinline__ u32 leaves_of(const TreeView& t, u32 loc, Need&& need)
    {
        l = 0;
        ploc = loc;
        return child_mask(t, t.depth - 1, loc, need);
    }

    template <class Need>
    __device__ __forceinline__ bool next(const TreeView& t, Need&& need, u32& leaf, u32& n_expand)
    {
        while (pend != 0) {
            const int bit = __builtin_ctzll(pend);
            pend &= ~(1ull << bit);
            const int h = bit >> LOGW;
            const u32 loc = ((ploc >> (LOGW * (h - l))) << LOGW) + (static_cast<u32>(bit) & (W - 1u));  // climb h - l levels, step down
            if (h == 0) {
                ploc = loc >> LOGW;
                l = 0;
                leaf = loc;
                return true;
            }
            ++n_expand;
            l = h - 1;
            ploc = loc;
            pend |= static_cast<u64>(child_mask(t, t.depth - h, loc, need)) << (W * l);
        }
        return false;
    }
};
using Walker = WalkerT<true, false>;

inline u32 grid_for_groups(u64 groups)
{
    u64 blocks = (groups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    blocks = (blocks + 7) / 8 * 8;
    return static_cast<u32>(blocks);
}

inline float sanitize_eps(float eps) { return eps > 0.f ? eps : 0.f; }  // eps <= 0 or NaN: nothing is "equal"

}  // namespace
}  // namespace pcpx

#endif
