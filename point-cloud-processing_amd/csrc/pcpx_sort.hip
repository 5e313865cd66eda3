// pcpx_sort.hip -- stable LSD radix sort of 64-bit keys by their bits [first_bit, 64), hand-written for gfx950 (wave64).
//
// The "radix sort" step of BASELINE.json's north_star: the index build sorts one packed word per point -- curve key in
// the high bits, point index in the low bits (pcpx_curve.h) -- and an arbitrary query batch does the same, so a pass
// moves 8 bytes per element each way instead of round 1's (u64 key, u32 value) pair: 16 B/element/pass.
//
// One kernel per 8-bit digit ("onesweep", Adinets & Merrill 2022), instead of round 1's histogram / scan-over-blocks /
// scatter triple (32 B/element/pass and three launches):
//   k_sort_histograms   ONE pass over the keys counts the digits of EVERY pass (per-block counts, no global atomics),
//   k_sort_bases        reduces them to the exclusive digit bases of every pass,
//   k_sort_onesweep     per pass: a block takes the next tile by ticket (so every earlier tile is already resident and
//                       will finish: the look-back below cannot wait on a block that has not started), ranks its 2048 keys
//                       (wave-level digit matching by 8 ballots, per-wave running counts in LDS), publishes the tile's
//                       digit counts, finds the number of equal digits in all earlier tiles by DECOUPLED LOOK-BACK over
//                       the published counts / running prefixes of its predecessors (one thread per digit), publishes its
//                       own running prefix, and scatters.
// Status words carry {pass tag, state, count}: one 64-bit relaxed atomic publishes all three, the status array is cleared
// once per sort.  A wave owns ITEMS consecutive 64-key chunks of its tile and tiles are numbered in ticket order = input
// order of the pass, so equal digits keep their order: the sort is stable.  The look-back spins a bounded number of
// times; if that bound is ever hit the sort reports failure through a device flag instead of hanging.
// Traffic: 8 B/element (histograms) + passes x (16 B + 1 B of status) -- 10 M keys x 5 passes = 0.93 GB (round 1: 1.6 GB).
#include "pcpx_internal.h"

namespace pcpx {

namespace {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / 64;
#ifndef PCPX_SORT_ITEMS
#define PCPX_SORT_ITEMS 16  // measured (10 M keys, rebuild ms): 8 keys/lane 1.29, 16 keys/lane 1.19; without the LDS staging 1.37 / 1.27; look-back 16 wide 1.53 / 1.36
#endif
#ifndef PCPX_SORT_STAGE
#define PCPX_SORT_STAGE 1  // 1: the tile goes out through LDS in digit order; 0: straight from registers
#endif
#ifndef PCPX_SORT_LOOK
#define PCPX_SORT_LOOK 4  // status words per look-back round trip (10 M keys, rebuild ms: 2: 1.068, 4: 1.051, 8: 1.067, 16: 1.36)
#endif
constexpr int SORT_ITEMS = PCPX_SORT_ITEMS;          // 64-key chunks per wave
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;   // keys per tile
constexpr int RADIX = 256;
constexpr int MAX_PASSES = 8;
constexpr u32 HIST_BLOCKS_MAX = 1024;

__device__ __forceinline__ u32 digit_of(u64 key, int shift) { return static_cast<u32>(key >> shift) & (RADIX - 1); }

// lanes of the wave whose digit equals mine (inactive lanes match only each other and are masked out)
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

// digit counts of every pass in one sweep over the keys: blockhist[block][pass][256]
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_histograms(const u64* __restrict__ keys, u64 n, int first_bit, int passes,
                                                                u32* __restrict__ blockhist)
{
    __shared__ u32 hist[MAX_PASSES][RADIX];
    for (int p = 0; p < passes; ++p) hist[p][threadIdx.x] = 0;
    __syncthreads();
    const u64 stride = static_cast<u64>(gridDim.x) * SORT_BLOCK;
    for (u64 i = blockIdx.x * static_cast<u64>(SORT_BLOCK) + threadIdx.x; i < n; i += stride) {
        const u64 key = keys[i];
        for (int p = 0; p < passes; ++p) atomicAdd(&hist[p][digit_of(key, first_bit + 8 * p)], 1u);
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) blockhist[(static_cast<u64>(blockIdx.x) * passes + p) * RADIX + threadIdx.x] = hist[p][threadIdx.x];
}

// one block per pass: sum the per-block counts (4 slices of the blocks x 256 digits, independent accumulators so the
// loads pipeline), exclusive scan over the digits -> bases[pass][256]
__global__ __launch_bounds__(4 * RADIX) void k_sort_bases(const u32* __restrict__ blockhist, u32 nblocks, int passes, u32* __restrict__ bases)
{
    __shared__ u32 part[4][RADIX];
    __shared__ u32 t[RADIX];
    const int p = blockIdx.x;
    const u32 d = threadIdx.x & (RADIX - 1), slice = threadIdx.x / RADIX;
    const u32 b0 = nblocks * slice / 4, b1 = nblocks * (slice + 1) / 4;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    u32 b = b0;
    const u64 stride = static_cast<u64>(passes) * RADIX;
    const u32* src = blockhist + static_cast<u64>(p) * RADIX + d;
    for (; b + 4 <= b1; b += 4) {
        s0 += src[(b + 0) * stride];
        s1 += src[(b + 1) * stride];
        s2 += src[(b + 2) * stride];
        s3 += src[(b + 3) * stride];
    }
    for (; b < b1; ++b) s0 += src[b * stride];
    part[slice][d] = s0 + s1 + s2 + s3;
    __syncthreads();
    if (slice == 0) t[d] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 acc = 0;
        for (int i = 0; i < RADIX; ++i) {
            const u32 v = t[i];
            t[i] = acc;
            acc += v;
        }
    }
    __syncthreads();
    if (slice == 0) bases[p * RADIX + d] = t[d];
}

// status word: bits [0,40) count, [40,42) state (1 = this tile's own count, 2 = running total up to and including this
// tile), [44,48) pass tag (pass + 1; 0 = never written since the array was cleared)
constexpr u64 ST_LOCAL = 1ull << 40, ST_PREFIX = 2ull << 40, ST_STATE = 3ull << 40, ST_COUNT = (1ull << 40) - 1ull;
__device__ __forceinline__ u64 st_tag(int pass) { return static_cast<u64>(pass + 1) << 44; }
constexpr u32 SPIN_LIMIT = 1u << 24;

__global__ __launch_bounds__(SORT_BLOCK) void k_sort_onesweep(const u64* __restrict__ kin, u64* __restrict__ kout, u64 n, int shift, int pass,
                                                              const u32* __restrict__ digit_base, u64* __restrict__ status,
                                                              u32* __restrict__ ticket, u32* __restrict__ failed)
{
    __shared__ u32 whist[SORT_WAVES][RADIX];  // per wave: keys of each digit seen so far in the tile
    __shared__ u32 wbase[SORT_WAVES][RADIX];  // per wave: destination of its first key of each digit
    __shared__ u32 tile_s;
    const u32 lane = threadIdx.x & 63u;
    const u32 w = threadIdx.x >> 6;
    if (threadIdx.x == 0) tile_s = atomicAdd(&ticket[pass], 1u);
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) whist[i][threadIdx.x] = 0;
    __syncthreads();
    const u32 tile = tile_s;

    const u64 base = static_cast<u64>(tile) * SORT_TILE + static_cast<u64>(w) * SORT_ITEMS * 64;
    u64 key[SORT_ITEMS];
    u32 rank[SORT_ITEMS];  // rank among the wave's keys with the same digit
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u64 i = base + static_cast<u64>(it) * 64 + lane;
        const bool act = i < n;
        key[it] = act ? kin[i] : ~0ull;
        const u32 d = digit_of(key[it], shift);
        const u64 same = match_digit(d, act);
        const u32 below = lanes_below(same);
        u32 prev = 0;
        if (act && below == 0) {  // leader of its digit group in this chunk
            prev = whist[w][d];
            whist[w][d] = prev + static_cast<u32>(__builtin_popcountll(same));
        }
        const int leader = act ? __builtin_ctzll(same) : static_cast<int>(lane);
        prev = __shfl(prev, leader);
        rank[it] = prev + below;
    }
    __syncthreads();
    {   // thread d: this tile's count of digit d, published; then the count in all earlier tiles by look-back
        const u32 d = threadIdx.x;
        u32 local = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) local += whist[i][d];
        const u64 tag = st_tag(pass);
        u64* mine = status + static_cast<u64>(tile) * RADIX + d;
        __hip_atomic_store(mine, tag | (tile == 0 ? ST_PREFIX : ST_LOCAL) | local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 before = 0;
        if (tile > 0) {
            // Look back over the predecessors' status words, LOOK of them per round trip (independent loads in flight
            // together): most resident tiles publish their own counts at about the same time, and a walk that inspects
            // one predecessor per memory latency makes the running prefix advance by one tile per latency.
            constexpr int LOOK = PCPX_SORT_LOOK;
            u32 t = tile;  // next to inspect: t - 1
            bool done = false;
            u32 spins = 0;
            while (!done) {
                u64 v[LOOK];
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    const u32 tj = t > static_cast<u32>(j) ? t - 1u - static_cast<u32>(j) : 0u;
                    v[j] = __hip_atomic_load(status + static_cast<u64>(tj) * RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    if (done || t == 0u) {
                        done = true;
                        break;
                    }
                    const u64 w_ = v[j];
                    if ((w_ >> 44) != static_cast<u64>(pass + 1) || (w_ & ST_STATE) == 0ull) break;  // not published yet: poll again from here
                    before += static_cast<u32>(w_ & ST_COUNT);
                    --t;
                    if ((w_ & ST_STATE) == ST_PREFIX || t == 0u) done = true;
                }
                if (!done && ++spins >= SPIN_LIMIT) {  // cannot happen with ticketed tiles; never hang: report and bail out
                    atomicExch(failed, 1u);
                    done = true;
                }
            }
            __hip_atomic_store(mine, tag | ST_PREFIX | (static_cast<u64>(before) + local), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        u32 acc = digit_base[d] + before;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) {
            wbase[i][d] = acc;
            acc += whist[i][d];
        }
    }
    __syncthreads();
#if PCPX_SORT_STAGE
    // The tile goes out through LDS in digit order, so that consecutive lanes write consecutive addresses: a digit's
    // keys of this tile are one contiguous run of the output (written straight from registers every lane of a wave hits
    // a different cache line: 64 partial-line writes per instruction).
    //   tbase[d]  = position of the tile's first key of digit d within the tile's digit-sorted order
    //   gbase[d]  = its position in the output minus tbase[d]   (so out = gbase[digit] + position in the sorted tile)
    __shared__ u64 stage[SORT_TILE];
    __shared__ u32 tbase[RADIX];
    __shared__ u32 gdelta[RADIX];
    {
        const u32 d = threadIdx.x;
        u32 local = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) local += whist[i][d];
        // exclusive scan of `local` over the 256 digits: wave-level scan + the 4 wave totals
        u32 incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 up = __shfl_up(incl, off);
            if (lane >= static_cast<u32>(off)) incl += up;
        }
        __shared__ u32 wtot[SORT_WAVES];
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        u32 before_waves = 0;
        for (u32 i = 0; i < w; ++i) before_waves += wtot[i];
        const u32 excl = before_waves + incl - local;
        tbase[d] = excl;
        gdelta[d] = wbase[0][d] - excl;  // wbase[0][d] = output position of the tile's first key of digit d
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u64 i = base + static_cast<u64>(it) * 64 + lane;
        if (i < n) {
            const u32 d = digit_of(key[it], shift);
            // position in the tile's digit-sorted order: keys of digit d in earlier waves, then the rank within this wave
            stage[tbase[d] + (wbase[w][d] - wbase[0][d]) + rank[it]] = key[it];
        }
    }
    __syncthreads();
    const u64 tile_first = static_cast<u64>(tile) * SORT_TILE;
    const u32 in_tile = static_cast<u32>(n - tile_first < SORT_TILE ? n - tile_first : SORT_TILE);
    for (u32 j = threadIdx.x; j < in_tile; j += SORT_BLOCK) {
        const u64 k = stage[j];
        kout[gdelta[digit_of(k, shift)] + j] = k;
    }
}
#else
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u64 i = base + static_cast<u64>(it) * 64 + lane;
        if (i < n) kout[wbase[w][digit_of(key[it], shift)] + rank[it]] = key[it];
    }
}
#endif

}  // namespace

// Temporary storage: per-block histograms + digit bases + tickets + failure flag + tile status + one key buffer.
// Result in kout.  Call with tmp == nullptr to get tmp_bytes.  kin is not modified.
int sort_keys_u64(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, u64 n, hipStream_t s, int first_bit)
{
    const u64 ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    const u32 hblocks = static_cast<u32>(ntiles < HIST_BLOCKS_MAX ? (ntiles ? ntiles : 1) : HIST_BLOCKS_MAX);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_hist = 0, o_base = o_hist + al(static_cast<size_t>(HIST_BLOCKS_MAX) * MAX_PASSES * RADIX * sizeof(u32)),
                 o_ctl = o_base + al(MAX_PASSES * RADIX * sizeof(u32)), o_status = o_ctl + al(64 * sizeof(u32)),
                 o_k = o_status + al((ntiles ? ntiles : 1) * RADIX * sizeof(u64)), total = o_k + al(n * sizeof(u64));
    if (!tmp) {
        tmp_bytes = total;
        return PCPX_OK;
    }
    if (tmp_bytes < total) {
        set_error("pcpx: sort temporary storage too small (%zu < %zu)", tmp_bytes, total);
        return PCPX_ERR_INVALID;
    }
    if (n == 0) return PCPX_OK;
    if (first_bit < 0 || first_bit > 56 || (first_bit & 7) || n >= (1ull << 32)) {
        set_error("pcpx: radix sort first_bit %d / n %llu", first_bit, static_cast<unsigned long long>(n));
        return PCPX_ERR_INVALID;
    }
    char* base = static_cast<char*>(tmp);
    u32* blockhist = reinterpret_cast<u32*>(base + o_hist);
    u32* bases = reinterpret_cast<u32*>(base + o_base);
    u32* ctl = reinterpret_cast<u32*>(base + o_ctl);  // [0, 8) tickets, [8] failure flag
    u64* status = reinterpret_cast<u64*>(base + o_status);
    u64* kt = reinterpret_cast<u64*>(base + o_k);
    const int passes = (64 - first_bit) / 8;
    PCPX_HIP(hipMemsetAsync(base + o_ctl, 0, (o_k - o_ctl), s));  // tickets, flag and the status array: once per sort
    k_sort_histograms<<<hblocks, SORT_BLOCK, 0, s>>>(kin, n, first_bit, passes, blockhist);
    k_sort_bases<<<passes, 4 * RADIX, 0, s>>>(blockhist, hblocks, passes, bases);
    // ping-pong between tmp and out so that the LAST pass writes out
    const u64* ksrc = kin;
    for (int pass = 0; pass < passes; ++pass) {
        const bool to_out = ((passes - 1 - pass) & 1) == 0;
        u64* kdst = to_out ? kout : kt;
        k_sort_onesweep<<<static_cast<u32>(ntiles), SORT_BLOCK, 0, s>>>(ksrc, kdst, n, first_bit + 8 * pass, pass, bases + pass * RADIX, status,
                                                                       ctl, ctl + 8);
        ksrc = kdst;
    }
    return check_hip(hipGetLastError(), "radix sort kernels", __FILE__, __LINE__);
}

// the failure flag of the last sort that used this temporary storage (device word; read it after synchronising)
const u32* sort_failure_flag(void* tmp)
{
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_ctl = al(static_cast<size_t>(HIST_BLOCKS_MAX) * MAX_PASSES * RADIX * sizeof(u32)) + al(MAX_PASSES * RADIX * sizeof(u32));
    return reinterpret_cast<const u32*>(static_cast<char*>(tmp) + o_ctl) + 8;
}

}  // namespace pcpx
