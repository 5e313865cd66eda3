// pcpx_prep.hip -- arbitrary query batches: sort the queries along the index's curve and seed every group of 64.
#include "pcpx_curve.h"
#include "pcpx_device.h"

#ifndef PCPX_QUERY_SORT_FIRST_BIT
#define PCPX_QUERY_SORT_FIRST_BIT 40
#endif

#include <algorithm>

namespace pcpx {

namespace {

// ------------------------------------------------------------------------------------------------
// arbitrary query batches: sort the queries along the index's curve (same grid, same key), seed each group at the
// 64-point chunk where its first query would sit in the sorted cloud
// ------------------------------------------------------------------------------------------------
constexpr int QCODES_BLOCK = 1024;
__global__ __launch_bounds__(QCODES_BLOCK) void k_query_codes(const float* __restrict__ q, u32 nq, const float* __restrict__ box6, int idx_bits,
                                                               u64* __restrict__ codes)
{
    __shared__ u32 htab[HILBERT_TABLE_WORDS];
    hilbert_table_to_lds(htab);
    const CurveGrid grid = curve_grid(box6[0], box6[1], box6[2], box6[3], box6[4], box6[5]);
    for (u32 i = blockIdx.x * QCODES_BLOCK + threadIdx.x; i < nq; i += gridDim.x * QCODES_BLOCK) {
        const float x = q[3ull * i], y = q[3ull * i + 1], z = q[3ull * i + 2];
        // (a query outside the box is clamped onto it: the key only decides where the query sits in the batch)
        codes[i] = sort_word(curve_key(x, y, z, grid, htab), i, idx_bits);
    }
}

// sorted words -> the queries in curve order (SoA) and their output rows
__global__ __launch_bounds__(256) void k_query_gather(const float* __restrict__ q, const u64* __restrict__ sorted, int idx_bits, u32 nq,
                                                      float* __restrict__ qx, float* __restrict__ qy, float* __restrict__ qz,
                                                      u32* __restrict__ row)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    u64 o = sorted[i] & ((1ull << idx_bits) - 1ull);
    qx[i] = q[3 * o];
    qy[i] = q[3 * o + 1];
    qz[i] = q[3 * o + 2];
    row[i] = static_cast<u32>(o);
}

// seed of group g: the 64-point chunk of the cloud where the group's middle query would sit in the curve order
// (lower bound over the sorted point words, compared above the bits that hold an index in either array)
__global__ __launch_bounds__(256) void k_query_seeds(const u64* __restrict__ qcodes, u32 nq, const u64* __restrict__ pcodes,
                                                     u32 n, u32 nleaves, int cmp_shift, u32* __restrict__ seed, u32 ngroups)
{
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    u32 mid = g * GROUP + GROUP / 2;
    if (mid >= nq) mid = nq - 1;
    u64 c = qcodes[mid] >> cmp_shift;
    u32 lo = 0, hi = n;
    while (lo < hi) {
        u32 m = lo + ((hi - lo) >> 1);
        if ((pcodes[m] >> cmp_shift) < c) lo = m + 1;
        else hi = m;
    }
    u32 chunk = lo / GROUP;
    u32 s0 = chunk * LEAVES_PER_GROUP;
    if (s0 >= nleaves) s0 = nleaves > LEAVES_PER_GROUP ? ((nleaves - 1) / LEAVES_PER_GROUP) * LEAVES_PER_GROUP : 0;
    seed[g] = s0;
}

// position_of[perm[p]] = p: where every inserted point sits in the curve order
__global__ __launch_bounds__(256) void k_invert_perm(const u32* __restrict__ perm, u32 n, u32* __restrict__ position_of)
{
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) position_of[perm[p]] = p;
}

}  // namespace

int launch_invert_perm(const u32* d_perm, u64 n, u32* d_position_of, hipStream_t s)
{
    if (n == 0) return PCPX_OK;
    k_invert_perm<<<static_cast<u32>((n + 255) / 256), 256, 0, s>>>(d_perm, static_cast<u32>(n), d_position_of);
    return check_hip(hipGetLastError(), "k_invert_perm launch", __FILE__, __LINE__);
}

int prepare_queries(Index& ix, const float* d_q, u64 nq, QueryView& qv)
{
    if (nq >= 0xFFFFFFFEull) {
        set_error("pcpx: nq = %llu does not fit 32-bit rows", static_cast<unsigned long long>(nq));
        return PCPX_ERR_UNSUPPORTED;
    }
    hipStream_t s = ix.stream;
    ProfileScope prof(ix, PCPX_K_QUERY_PREP);
    u32 n32 = static_cast<u32>(nq);
    u64 ngroups = (nq + GROUP - 1) / GROUP;
    size_t tb = 0;
    int st = sort_keys_u64(nullptr, tb, nullptr, nullptr, nq, s);
    if (st != PCPX_OK) return st;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t o_codes0 = 0, o_codes1 = o_codes0 + al(nq * 8), o_row = o_codes1 + al(nq * 8), o_qx = o_row + al(nq * 4),
           o_qy = o_qx + al(nq * 4), o_qz = o_qy + al(nq * 4), o_seed = o_qz + al(nq * 4), o_tmp = o_seed + al(ngroups * 4),
           total = o_tmp + al(tb);
    if ((st = ensure_scratch(ix, total)) != PCPX_OK) return st;
    char* base = static_cast<char*>(ix.d_scratch);
    u64* codes0 = reinterpret_cast<u64*>(base + o_codes0);
    u64* codes1 = reinterpret_cast<u64*>(base + o_codes1);
    u32* row = reinterpret_cast<u32*>(base + o_row);
    float* qx = reinterpret_cast<float*>(base + o_qx);
    float* qy = reinterpret_cast<float*>(base + o_qy);
    float* qz = reinterpret_cast<float*>(base + o_qz);
    u32* seed = reinterpret_cast<u32*>(base + o_seed);
    if (nq > 0) {
        const float* d_box = reinterpret_cast<const float*>(ix.d_scalars + 8);
        const int qbits = index_bits_for(nq);
        const int cmp_shift = std::max(ix.sorted_from_bit, std::max(qbits, ix.idx_bits));  // (the binary search needs bits the index is ordered on)
        const u32 cblocks = (n32 + QCODES_BLOCK - 1) / QCODES_BLOCK;
        k_query_codes<<<cblocks < 512u ? cblocks : 512u, QCODES_BLOCK, 0, s>>>(d_q, n32, d_box, qbits, codes0);
        // (queries are only grouped by the sort -- 64 consecutive ones per wavefront --, so the top 24 bits of the curve key (256
        //  cells per axis) are enough: three passes instead of five)
        if ((st = sort_keys_u64(base + o_tmp, tb, codes0, codes1, nq, s, PCPX_QUERY_SORT_FIRST_BIT)) != PCPX_OK) return st;
        k_query_gather<<<(n32 + 255) / 256, 256, 0, s>>>(d_q, codes1, qbits, n32, qx, qy, qz, row);
        k_query_seeds<<<static_cast<u32>((ngroups + 255) / 256), 256, 0, s>>>(
            codes1, n32, ix.sorted_codes(), static_cast<u32>(ix.n), ix.nleaves, cmp_shift, seed, static_cast<u32>(ngroups));
        PCPX_HIP(hipGetLastError());
    }
    qv = QueryView{qx, qy, qz, row, seed, n32};
    return PCPX_OK;
}

}  // namespace pcpx
