// pcpx_prep.hip -- arbitrary query batches: Morton-sort the queries on the index's grid and seed every group of 64.
#include "pcpx_curve.h"
#include "pcpx_device.h"

namespace pcpx {

namespace {

// ------------------------------------------------------------------------------------------------
// arbitrary query batches: Morton-sort the queries on the index's grid, seed each group at the
// 64-point chunk where its first query would sit in the sorted cloud
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_query_codes(const float* __restrict__ q, u32 nq, const float* __restrict__ box6,
                                                     u64* __restrict__ codes, u32* __restrict__ vals)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    float x = q[3ull * i], y = q[3ull * i + 1], z = q[3ull * i + 2];
    codes[i] = curve_key(x, y, z, box6[0], box6[1], box6[2], box6[3], box6[4], box6[5]);  // (outside the box: clamped)
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_query_gather(const float* __restrict__ q, const u32* __restrict__ order, u32 nq,
                                                      float* __restrict__ qx, float* __restrict__ qy, float* __restrict__ qz)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    u64 o = order[i];
    qx[i] = q[3 * o];
    qy[i] = q[3 * o + 1];
    qz[i] = q[3 * o + 2];
}

__global__ __launch_bounds__(256) void k_query_seeds(const u64* __restrict__ qcodes, u32 nq, const u64* __restrict__ pcodes,
                                                     u32 n, u32 nleaves, u32* __restrict__ seed, u32 ngroups)
{
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    u32 mid = g * GROUP + GROUP / 2;
    if (mid >= nq) mid = nq - 1;
    u64 c = qcodes[mid] >> MORTON_SORT_FIRST_BIT;  // the codes are ordered by these bits only
    u32 lo = 0, hi = n;  // lower_bound over the sorted point codes
    while (lo < hi) {
        u32 m = lo + ((hi - lo) >> 1);
        if ((pcodes[m] >> MORTON_SORT_FIRST_BIT) < c) lo = m + 1;
        else hi = m;
    }
    u32 chunk = lo / GROUP;
    u32 s0 = chunk * LEAVES_PER_GROUP;
    if (s0 >= nleaves) s0 = nleaves > LEAVES_PER_GROUP ? ((nleaves - 1) / LEAVES_PER_GROUP) * LEAVES_PER_GROUP : 0;
    seed[g] = s0;
}

}  // namespace

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
    int st = sort_pairs_u64(nullptr, tb, nullptr, nullptr, nullptr, nullptr, nq, s);
    if (st != PCPX_OK) return st;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t o_codes0 = 0, o_codes1 = o_codes0 + al(nq * 8), o_vals0 = o_codes1 + al(nq * 8), o_vals1 = o_vals0 + al(nq * 4),
           o_qx = o_vals1 + al(nq * 4), o_qy = o_qx + al(nq * 4), o_qz = o_qy + al(nq * 4), o_seed = o_qz + al(nq * 4),
           o_tmp = o_seed + al(ngroups * 4), total = o_tmp + al(tb);
    if ((st = ensure_scratch(ix, total)) != PCPX_OK) return st;
    char* base = static_cast<char*>(ix.d_scratch);
    u64* codes0 = reinterpret_cast<u64*>(base + o_codes0);
    u64* codes1 = reinterpret_cast<u64*>(base + o_codes1);
    u32* vals0 = reinterpret_cast<u32*>(base + o_vals0);
    u32* vals1 = reinterpret_cast<u32*>(base + o_vals1);
    float* qx = reinterpret_cast<float*>(base + o_qx);
    float* qy = reinterpret_cast<float*>(base + o_qy);
    float* qz = reinterpret_cast<float*>(base + o_qz);
    u32* seed = reinterpret_cast<u32*>(base + o_seed);
    if (nq > 0) {
        const float* d_box = reinterpret_cast<const float*>(ix.d_scalars + 8);
        k_query_codes<<<(n32 + 255) / 256, 256, 0, s>>>(d_q, n32, d_box, codes0, vals0);
        if ((st = sort_pairs_u64(base + o_tmp, tb, codes0, codes1, vals0, vals1, nq, s, MORTON_SORT_FIRST_BIT)) != PCPX_OK) return st;
        k_query_gather<<<(n32 + 255) / 256, 256, 0, s>>>(d_q, vals1, n32, qx, qy, qz);
        k_query_seeds<<<static_cast<u32>((ngroups + 255) / 256), 256, 0, s>>>(
            codes1, n32, ix.sorted_codes(), static_cast<u32>(ix.n), ix.nleaves, seed, static_cast<u32>(ngroups));
        PCPX_HIP(hipGetLastError());
    }
    qv = QueryView{qx, qy, qz, vals1, seed, n32};
    return PCPX_OK;
}

}  // namespace pcpx
