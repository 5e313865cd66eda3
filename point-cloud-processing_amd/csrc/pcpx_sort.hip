// pcpx_sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs, hand-written for gfx950 (wave64).
//
// Used for the Morton order of the index build and of arbitrary query batches (the "radix sort" step of
// BASELINE.json's north_star).  8-bit digits, one pass per digit of the key bits [first_bit, 64).  Per pass:
//   k_sort_hist     each 256-thread block counts the digits of its tile          -> blockhist[block][256]
//   k_sort_scan     per digit, exclusive prefix over blocks + base of the digit  -> blockhist in place
//   k_sort_scatter  each block re-reads its tile, ranks every key among the equal digits before it in the
//                   tile (wave-level match by 8 ballots, per-wave running counts in LDS, prefix over the 4
//                   waves) and writes key and value to offset[block][digit] + rank.
// A tile is TILE = 256 x ITEMS consecutive pairs; wave w of the block owns ITEMS consecutive 64-pair chunks,
// so tile order = index order and the sort is stable.  Traffic per pass: 8 B/pair (hist) + 12 B read +
// 12 B write (scatter) = 32 B/pair, HBM bound; 10 M pairs x 8 passes = 2.6 GB.
#include "pcpx_internal.h"

namespace pcpx {

namespace {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / 64;
constexpr int SORT_ITEMS = 8;                        // 64-pair chunks per wave
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;   // pairs per block
constexpr int RADIX = 256;

__device__ __forceinline__ u32 digit_of(u64 key, int shift) { return static_cast<u32>(key >> shift) & (RADIX - 1); }

// lanes of the wave whose digit equals mine (inactive lanes pass digit = 0xFFFFFFFF and match only each other)
__device__ __forceinline__ u64 match_digit(u32 d, bool active)
{
    u64 same = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const u64 m = __builtin_amdgcn_ballot_w64(bit);
        same &= bit ? m : ~m;
    }
    const u64 act = __builtin_amdgcn_ballot_w64(active);
    return active ? (same & act) : 0ull;
}

__device__ __forceinline__ u32 lanes_below(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mask), 0u));
}

__global__ __launch_bounds__(SORT_BLOCK) void k_sort_hist(const u64* __restrict__ keys, u64 n, int shift,
                                                          u32* __restrict__ blockhist)
{
    __shared__ u32 hist[RADIX];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = static_cast<u64>(blockIdx.x) * SORT_TILE;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u64 i = base + static_cast<u64>(it) * SORT_BLOCK + threadIdx.x;  // order is irrelevant for counting
        if (i < n) atomicAdd(&hist[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    blockhist[static_cast<u64>(blockIdx.x) * RADIX + threadIdx.x] = hist[threadIdx.x];
}

// one block per digit: exclusive prefix of blockhist[*][digit] over the blocks; digit totals go to `totals`
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scan_blocks(u32* __restrict__ blockhist, u32 nblocks,
                                                                 u32* __restrict__ totals)
{
    __shared__ u32 part[SORT_BLOCK];
    const u32 d = blockIdx.x;
    const u32 per = (nblocks + SORT_BLOCK - 1) / SORT_BLOCK;
    const u32 b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    u32 sum = 0;
    for (u32 b = b0; b < b1; ++b) sum += blockhist[static_cast<u64>(b) * RADIX + d];
    part[threadIdx.x] = sum;
    __syncthreads();
    // exclusive scan of the 256 partial sums (tiny: serial in thread 0 is 256 adds)
    if (threadIdx.x == 0) {
        u32 acc = 0;
        for (int i = 0; i < SORT_BLOCK; ++i) {
            u32 v = part[i];
            part[i] = acc;
            acc += v;
        }
        totals[d] = acc;
    }
    __syncthreads();
    u32 acc = part[threadIdx.x];
    for (u32 b = b0; b < b1; ++b) {
        const u64 at = static_cast<u64>(b) * RADIX + d;
        u32 v = blockhist[at];
        blockhist[at] = acc;
        acc += v;
    }
}

// exclusive scan of the 256 digit totals -> digit bases
__global__ __launch_bounds__(RADIX) void k_sort_scan_digits(u32* __restrict__ totals)
{
    __shared__ u32 t[RADIX];
    t[threadIdx.x] = totals[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 acc = 0;
        for (int i = 0; i < RADIX; ++i) {
            u32 v = t[i];
            t[i] = acc;
            acc += v;
        }
    }
    __syncthreads();
    totals[threadIdx.x] = t[threadIdx.x];
}

__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scatter(const u64* __restrict__ kin, const u32* __restrict__ vin,
                                                             u64* __restrict__ kout, u32* __restrict__ vout, u64 n, int shift,
                                                             const u32* __restrict__ blockhist, const u32* __restrict__ digit_base)
{
    __shared__ u32 whist[SORT_WAVES][RADIX];  // per wave: keys of each digit seen so far in the tile
    __shared__ u32 wbase[SORT_WAVES][RADIX];  // per wave: destination of its first key of each digit
    const u32 lane = threadIdx.x & 63u;
    const u32 w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) whist[i][threadIdx.x] = 0;
    __syncthreads();

    const u64 base = static_cast<u64>(blockIdx.x) * SORT_TILE + static_cast<u64>(w) * SORT_ITEMS * 64;
    u64 key[SORT_ITEMS];
    u32 val[SORT_ITEMS];
    u32 rank[SORT_ITEMS];  // rank among the wave's keys with the same digit
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u64 i = base + static_cast<u64>(it) * 64 + lane;
        const bool act = i < n;
        key[it] = act ? kin[i] : ~0ull;
        val[it] = act ? vin[i] : 0u;
        const u32 d = digit_of(key[it], shift);
        const u64 same = match_digit(d, act);
        const u32 below = lanes_below(same);
        u32 prev = 0;
        if (act && below == 0) {  // leader of its digit group in this chunk
            prev = whist[w][d];
            whist[w][d] = prev + static_cast<u32>(__builtin_popcountll(same));
        }
        // broadcast the leader's previous count to its group (leader = lowest set lane of `same`)
        const int leader = act ? __builtin_ctzll(same) : static_cast<int>(lane);
        prev = __shfl(prev, leader);
        rank[it] = prev + below;
    }
    __syncthreads();
    {   // thread d: destination of the first key of digit d for each wave of this block
        const u32 d = threadIdx.x;
        u32 acc = blockhist[static_cast<u64>(blockIdx.x) * RADIX + d] + digit_base[d];
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) {
            wbase[i][d] = acc;
            acc += whist[i][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u64 i = base + static_cast<u64>(it) * 64 + lane;
        if (i < n) {
            const u32 d = digit_of(key[it], shift);
            const u32 dst = wbase[w][d] + rank[it];
            kout[dst] = key[it];
            vout[dst] = val[it];
        }
    }
}

}  // namespace

// Temporary storage: blockhist[nblocks][256] + totals[256] + one ping-pong (key, value) buffer pair.
// Result in (kout, vout).  Call with tmp == nullptr to get tmp_bytes.
int sort_pairs_u64(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, const u32* vin, u32* vout, u64 n, hipStream_t s,
                   int first_bit)
{
    const u64 nblocks = (n + SORT_TILE - 1) / SORT_TILE;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_hist = 0, o_tot = o_hist + al((nblocks ? nblocks : 1) * RADIX * sizeof(u32)), o_k = o_tot + al(RADIX * sizeof(u32)),
                 o_v = o_k + al(n * sizeof(u64)), total = o_v + al(n * sizeof(u32));
    if (!tmp) {
        tmp_bytes = total;
        return PCPX_OK;
    }
    if (tmp_bytes < total) {
        set_error("pcpx: sort temporary storage too small (%zu < %zu)", tmp_bytes, total);
        return PCPX_ERR_INVALID;
    }
    if (n == 0) return PCPX_OK;
    char* base = static_cast<char*>(tmp);
    u32* blockhist = reinterpret_cast<u32*>(base + o_hist);
    u32* totals = reinterpret_cast<u32*>(base + o_tot);
    u64* kt = reinterpret_cast<u64*>(base + o_k);
    u32* vt = reinterpret_cast<u32*>(base + o_v);
    if (first_bit < 0 || first_bit > 56 || (first_bit & 7)) {
        set_error("pcpx: radix sort first_bit %d", first_bit);
        return PCPX_ERR_INVALID;
    }
    // ping-pong between tmp and out so that the LAST pass writes out
    const int passes = (64 - first_bit) / 8;
    const u64* ksrc = kin;
    const u32* vsrc = vin;
    for (int pass = 0; pass < passes; ++pass) {
        const bool to_out = ((passes - 1 - pass) & 1) == 0;
        u64* kdst = to_out ? kout : kt;
        u32* vdst = to_out ? vout : vt;
        const int shift = first_bit + 8 * pass;
        k_sort_hist<<<static_cast<u32>(nblocks), SORT_BLOCK, 0, s>>>(ksrc, n, shift, blockhist);
        k_sort_scan_blocks<<<RADIX, SORT_BLOCK, 0, s>>>(blockhist, static_cast<u32>(nblocks), totals);
        k_sort_scan_digits<<<1, RADIX, 0, s>>>(totals);
        k_sort_scatter<<<static_cast<u32>(nblocks), SORT_BLOCK, 0, s>>>(ksrc, vsrc, kdst, vdst, n, shift, blockhist, totals);
        ksrc = kdst;
        vsrc = vdst;
    }
    return check_hip(hipGetLastError(), "radix sort kernels", __FILE__, __LINE__);
}

}  // namespace pcpx
