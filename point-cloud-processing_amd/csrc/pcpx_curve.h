// pcpx_curve.h -- the space-filling-curve key the index (and every arbitrary query batch) is sorted by.
//
// Round 1 sorted by the Morton (Z-order) code.  Any order gives a correct tree -- node boxes are computed from the
// points -- so the order is purely a performance choice, and Z-order is a poor one for this kernel: a run of 8 or 64
// consecutive points regularly straddles one of the curve's jumps, which makes leaf and group boxes long and thin, and
// a wavefront walks the UNION of its 64 lanes' search regions.  Consecutive points of a HILBERT curve are always
// spatial neighbours.  Measured on the CPU model of the walk (tools/sim_hilbert.py, 1 M points, k = 15): mean leaf-box
// volume 1.9e-5 -> 5.6e-6, leaves visited per 64-query group 135 -> 84, node expansions 165 -> 88 (uniform cloud;
// clustered: 131 -> 85, 156 -> 99).
//
// Sort word: 13 bits per axis of the 21-bit quantisation (the reference's octant bits, x most significant,
// include/pcp/octree/linked_octree_node.hpp:258-265, are where the grid comes from) -> 39-bit Hilbert index in bits
// [25, 64), so that the word's top byte -- the digit the radix sort partitions by first -- is the curve's top 8 bits: 256
// spatially compact buckets of about equal size for a uniform cloud.  The element's index sits in the low bits and
// overwrites as many low key bits as it needs (a 13-bit index is a refinement of the 12-bit one, so dropping low bits only
// coarsens the cells).  A point outside the voxel grid gets the all-ones key and sorts last; an inserted point's key is
// kept off that value (bit 24 is zero below 16.7 M points; beyond, the curve's very last cell is merged with its
// predecessor).  One 8-byte word per element is all the radix sort moves; it looks at bits [24, 64): 5 passes.
#ifndef PCPX_CURVE_H
#define PCPX_CURVE_H

#include "pcpx_curve_table.h"
#include "pcpx_internal.h"

namespace pcpx {

constexpr int CURVE_BITS = 13;  // per axis: cells of 1/8192 of the grid extent

__host__ __device__ __forceinline__ u64 spread21(u32 v)
{
    u64 x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// bit i of the low 10 bits of v -> bit 3i
__host__ __device__ __forceinline__ u32 spread10(u32 v)
{
    u32 x = v & 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
// the same for 3 bits
__host__ __device__ __forceinline__ u32 spread3(u32 v) { return (v | (v << 2) | (v << 4)) & 0x49u; }

// Hilbert index of the cell (x, y, z), BITS bits per axis (J. Skilling, "Programming the Hilbert curve", AIP Conf.
// Proc. 707, 2004: axes -> transposed index by undoing the excess work of the Gray code, then the bits are interleaved).
template <int BITS = CURVE_BITS>
__host__ __device__ __forceinline__ u64 hilbert_index(u32 x, u32 y, u32 z)
{
    u32 X[3] = {x, y, z};
#pragma unroll
    for (int b = BITS - 1; b > 0; --b) {
        const u32 Q = 1u << b, P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool hit = (X[i] & Q) != 0u;
            const u32 t = hit ? 0u : ((X[0] ^ X[i]) & P);  // exchange the low bits of X[0] and X[i] ...
            X[0] ^= hit ? P : t;                           // ... or invert those of X[0]
            X[i] ^= t;
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    // Skilling's tail "for every set bit b >= 1 of X[2]: t ^= 2^b - 1" makes bit j of t the parity of the bits of X[2] above j:
    // a prefix XOR from the top, shifted down by one (5 shift-xor steps instead of BITS - 1 test-and-xor steps)
    u32 t = X[2];
    t ^= t >> 1;
    t ^= t >> 2;
    t ^= t >> 4;
    t ^= t >> 8;
    if (BITS > 16) t ^= t >> 16;
    t >>= 1;
    X[0] ^= t;
    X[1] ^= t;
    X[2] ^= t;
    if (BITS <= 13) {
        // interleave in 32-bit pieces (the vector ALU has no 64-bit shifts-with-or): the low 10 bits of every axis give the low 30
        // bits of the index, the bits above them the rest
        const u32 lo = (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
        const u32 hi = (spread3(X[0] >> 10) << 2) | (spread3(X[1] >> 10) << 1) | spread3(X[2] >> 10);
        return (static_cast<u64>(hi) << 30) | lo;
    }
    return (spread21(X[0]) << 2) | (spread21(X[1]) << 1) | spread21(X[2]);
}

// The same index through the generated state machine (pcpx_curve_table.h): the Morton code of the cell, six bits at a
// time, each step one table look-up {index bits, next state}.  ~70 vector instructions and 7 LDS reads per cell instead of
// ~260 (the arithmetic form walks 12 levels x 3 axes of dependent bit-twiddling; profiles/r02_pmc_rebuild.json: 367
// instructions per point in k_codes, compute bound).  `tab`: HILBERT_TABLE_WORDS words, the layout of PCPX_HILBERT_TABLE_INIT
// (in a kernel: a copy in LDS, hilbert_table_to_lds).  tests/cpp/test_curve.hip checks it against hilbert_index<13>.
__host__ __device__ __forceinline__ u64 hilbert_index_table(u32 x, u32 y, u32 z, const u32* __restrict__ tab)
{
    static_assert(CURVE_BITS == 13, "the table walk below is laid out for 13 levels: 1 + 6 x 2");
    const u32 lo = (spread10(x) << 2) | (spread10(y) << 1) | spread10(z);                    // levels 9 .. 0
    const u32 hi = (spread3(x >> 10) << 2) | (spread3(y >> 10) << 1) | spread3(z >> 10);     // levels 12, 11, 10
    u32 e = tab[hi >> 6];                                                                    // level 12
    u32 h_hi = e & 63u;
    e = tab[8u + ((e & 0xFFFFFF00u) >> 2) + (hi & 63u)];                                     // levels 11, 10
    h_hi = (h_hi << 6) | (e & 63u);
    u32 h_lo = 0;
#pragma unroll
    for (int t = 4; t >= 0; --t) {
        e = tab[8u + ((e & 0xFFFFFF00u) >> 2) + ((lo >> (6 * t)) & 63u)];
        h_lo = (h_lo << 6) | (e & 63u);
    }
    return (static_cast<u64>(h_hi) << 30) | h_lo;
}

#ifdef __HIPCC__
// the tables of hilbert_index_table, read by every kernel that makes curve keys (each copies them into LDS once per block)
static __device__ const u32 hilbert_table_dev[HILBERT_TABLE_WORDS] = {PCPX_HILBERT_TABLE_INIT};
__device__ __forceinline__ void hilbert_table_to_lds(u32* lds_tab)
{
    for (u32 i = threadIdx.x; i < static_cast<u32>(HILBERT_TABLE_WORDS); i += blockDim.x) lds_tab[i] = hilbert_table_dev[i];
    __syncthreads();
}
#endif

// Quantisation of one axis to CURVE_BITS bits: cell = floor((v - lo) x scale), scale = 2^CURVE_BITS / extent (0 for a flat
// axis), clamped into the grid (a query may lie outside; NaN -> cell 0).
struct CurveGrid {
    float lo[3], scale[3];
};
__host__ __device__ __forceinline__ CurveGrid curve_grid(float b0, float b1, float b2, float b3, float b4, float b5)
{
    CurveGrid g;
    const float lo[3] = {b0, b1, b2}, hi[3] = {b3, b4, b5};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = hi[a] - lo[a];
        g.lo[a] = lo[a];
        g.scale[a] = ext > 0.f ? static_cast<float>(1 << CURVE_BITS) / ext : 0.f;
    }
    return g;
}
__host__ __device__ __forceinline__ u32 curve_cell(float v, float lo, float scale)
{
    const float t = fminf(fmaxf((v - lo) * scale, 0.f), static_cast<float>((1 << CURVE_BITS) - 1));
    return static_cast<u32>(t);
}

// curve key (bits [25, 64)) of a point on the grid g
__host__ __device__ __forceinline__ u64 curve_key(float x, float y, float z, const CurveGrid& g, const u32* __restrict__ lds_tab)
{
    return hilbert_index_table(curve_cell(x, g.lo[0], g.scale[0]), curve_cell(y, g.lo[1], g.scale[1]), curve_cell(z, g.lo[2], g.scale[2]), lds_tab)
           << CURVE_FIRST_BIT;
}
// the same for a point the index inserts: never the all-ones pattern above the word's index bits, which marks a point
// outside the grid
__host__ __device__ __forceinline__ u64 curve_key_inside(float x, float y, float z, const CurveGrid& g, const u32* __restrict__ lds_tab, int idx_bits)
{
    const u64 hmax = ((1ull << (3 * CURVE_BITS)) - 1ull) - (idx_bits >= CURVE_FIRST_BIT ? (1ull << (idx_bits - CURVE_FIRST_BIT)) : 0ull);
    const u64 h = hilbert_index_table(curve_cell(x, g.lo[0], g.scale[0]), curve_cell(y, g.lo[1], g.scale[1]), curve_cell(z, g.lo[2], g.scale[2]), lds_tab);
    return (h < hmax ? h : hmax) << CURVE_FIRST_BIT;
}
// the sort word of element `index`: its key with the low idx_bits replaced by the index
__host__ __device__ __forceinline__ u64 sort_word(u64 key, u64 index, int idx_bits)
{
    const u64 low = (1ull << idx_bits) - 1ull;
    return (key & ~low) | index;
}
// the word of an element outside the grid (or NaN): greater than every inserted point's word
__host__ __device__ __forceinline__ u64 outside_word(u64 index, int idx_bits) { return (~0ull << idx_bits) | index; }

}  // namespace pcpx

#endif
