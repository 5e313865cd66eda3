// pcpx_sort.hip -- stable radix sort of 64-bit words by their bits [first_bit, 64), hand-written for gfx950 (wave64).
//
// The "radix sort" step of BASELINE.json's north_star: the index build sorts one packed word per point -- curve key in
// the high bits, point index in the low bits (pcpx_curve.h) -- and an arbitrary query batch does the same.
//
// Shape (round 3): MOST significant digit first, then least-significant-digit passes INSIDE the 256 buckets.
//   k_sort_top_hist     counts the top digit (bits [56, 64)) of every word           (the build's k_codes does it itself),
//   k_sort_seg_setup    turns the counts into the bucket table: first position and first tile of every bucket,
//   k_sort_pass<false>  the top-digit pass: a stable partition of the whole array into the 256 buckets ("onesweep":
//                       ticketed tiles of 4096 words, per-wave digit ranks, decoupled look-back over the predecessors'
//                       published counts).  For the index build it also moves a 16-byte record {x, y, z, id} per point
//                       to the word's new position and writes that position into the word's low bits, so that
//                       everything after this pass -- and the leaf gather at the end -- stays inside one bucket,
//   k_sort_seg_hist     per bucket: the digit counts of every remaining pass (one sweep over the partitioned words),
//   k_sort_seg_bases    per bucket and pass: exclusive scan of the counts = the digits' first output positions,
//   k_sort_pass<true>   one per remaining digit, least significant first: the same kernel, but a tile never straddles
//                       a bucket and the look-back ends at the bucket's first tile.
// Round 2 ran five least-significant-digit passes over the whole array; its profile (profiles/r02_pmc_rebuild.json)
// showed 63 % of a wave's time waiting in a look-back chain over all ~1 000 resident tiles.  With the buckets a chain
// is n / (256 x 4096) tiles long on average (10 at 10 M words), whatever is resident, and the bucket of a leaf is an
// L2-sized window of the coordinates instead of the whole cloud.  Skewed inputs (all words in one bucket) degrade to
// the round-2 behaviour, not below it.
//
// Ranking inside a wave: lanes with the same digit find each other through ONE 64-bit LDS atomic-or per 64 words
// (a lane mask per digit, per wave) instead of eight ballots with their select/and chains: 12 vector instructions per
// 64 words instead of ~70 -- the pass was as much VALU-bound as chain-bound.  LDS operations of one wave execute in
// program order, which is all the scheme needs; different waves use different mask rows.
//
// Status words carry {pass tag, state, count}: one 64-bit relaxed atomic publishes all three, the status array is cleared
// once per sort.  A wave owns ITEMS consecutive 64-word chunks of its tile and tiles are numbered in ticket order = input
// order of the pass, so equal digits keep their order: every pass is stable, and so is the whole sort.  The look-back
// spins a bounded number of times; if that bound is ever hit the sort reports failure through a device flag instead of
// hanging (it cannot be: a tile only ever waits for tiles with smaller tickets, which are resident or finished).
#include "pcpx_internal.h"

namespace pcpx {

namespace {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / 64;
constexpr int SORT_ITEMS = 16;                      // 64-word chunks per wave
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;  // words per tile
constexpr int RADIX = 256;
constexpr int MAX_PASSES = 8;
constexpr int TOP_SHIFT = 56;
#ifndef PCPX_SORT_LOOK
#define PCPX_SORT_LOOK 8  // status words per look-back round trip once a round has used up the 4 it starts with
#endif

// bucket table of one sort (device memory, written by k_sort_seg_setup)
struct SegTable {
    u32 start[RADIX + 1];       // first position of bucket b; start[RADIX] = n
    u32 tile_first[RADIX + 1];  // first tile of bucket b in the bucketed passes; tile_first[RADIX] = number of tiles
};

__device__ __forceinline__ u32 digit_of(u64 key, int shift) { return static_cast<u32>(key >> shift) & (RADIX - 1); }

__device__ __forceinline__ u32 lanes_below(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<u32>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(mask), 0u));
}

// LDS words read and written in program order by the ranking loop (volatile accesses through generic pointers become
// flat instructions with system-scope cache bits: the pointers must carry the LDS address space themselves)
typedef __attribute__((address_space(3))) volatile u64* lds_vu64;
typedef __attribute__((address_space(3))) volatile u32* lds_vu32;
template <class P, class T>
__device__ __forceinline__ P lds_ptr(T* generic)
{
    return (P)(static_cast<uintptr_t>(static_cast<u32>(reinterpret_cast<uintptr_t>(generic))));  // low 32 bits of a generic LDS pointer = LDS offset
}

// exclusive scan of v over the 256 threads of the block (thread order); `wtot` is 4 words of LDS
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* wtot, u32& total)
{
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 up = __shfl_up(incl, off);
        if (lane >= static_cast<u32>(off)) incl += up;
    }
    __syncthreads();  // (wtot may still be in use by an earlier scan)
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (u32 i = 0; i < SORT_WAVES; ++i) {
        const u32 t = wtot[i];
        before += i < w ? t : 0u;
        all += t;
    }
    total = all;
    return before + incl - v;
}

// counts of the top digit: hist[256] (zeroed by the caller), one LDS atomic per word, one global atomic per non-empty
// digit and block
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_top_hist(const u64* __restrict__ keys, u64 n, u32* __restrict__ hist)
{
    __shared__ u32 h[RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u64 stride = static_cast<u64>(gridDim.x) * SORT_BLOCK;
    for (u64 i = blockIdx.x * static_cast<u64>(SORT_BLOCK) + threadIdx.x; i < n; i += stride) atomicAdd(&h[digit_of(keys[i], TOP_SHIFT)], 1u);
    __syncthreads();
    const u32 c = h[threadIdx.x];
    if (c) atomicAdd(&hist[threadIdx.x], c);
}

// one block: counts -> bucket table
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_seg_setup(const u32* __restrict__ hist, SegTable* __restrict__ seg)
{
    __shared__ u32 wtot[SORT_WAVES];
    const u32 c = hist[threadIdx.x];
    u32 total = 0;
    const u32 s = block_exclusive_scan(c, wtot, total);
    seg->start[threadIdx.x] = s;
    if (threadIdx.x == 0) seg->start[RADIX] = total;
    u32 tiles = 0;
    const u32 f = block_exclusive_scan((c + SORT_TILE - 1) / SORT_TILE, wtot, tiles);
    seg->tile_first[threadIdx.x] = f;
    if (threadIdx.x == 0) seg->tile_first[RADIX] = tiles;
}

// status word: bits [0,40) count, [40,42) state (1 = this tile's own count, 2 = running total up to and including this
// tile), [44,48) pass tag (pass + 1; 0 = never written since the array was cleared)
constexpr u64 ST_LOCAL = 1ull << 40, ST_PREFIX = 2ull << 40, ST_STATE = 3ull << 40, ST_COUNT = (1ull << 40) - 1ull;
constexpr u32 SPIN_LIMIT = 1u << 24;

struct SortPayloadArgs {
    const float* xyz;  // n x 3, element order of the pass's input
    float4* rec;       // out: {x, y, z, bits of the element's index} at the word's new position
    u64 low_mask;      // the word's low bits that hold the element's index (in) / the record's position (out)
};

// One tile's place in the pass.  SEG = false: tile t covers [t * TILE, ...) of the whole array.  SEG = true: the tiles
// of bucket b are seg.tile_first[b] ..., each inside [seg.start[b], seg.start[b + 1]).
struct TilePlace {
    u32 lo, hi;     // word range [lo, hi), hi - lo <= TILE
    u32 floor;      // first tile of the chain this tile's look-back walks
    u32 bucket;
};

template <bool SEG, bool PAYLOAD>
__global__ __launch_bounds__(SORT_BLOCK, 4) void k_sort_pass(const u64* __restrict__ kin, u64* __restrict__ kout, u32 n, int shift, int tag_pass,
                                                          const u32* __restrict__ digit_base, int base_stride,
                                                          const SegTable* __restrict__ seg, u64* __restrict__ status,
                                                          u32* __restrict__ ticket, u32* __restrict__ failed, SortPayloadArgs pl)
{
    // LDS: the staging buffer of the scatter doubles, before that, as the per-wave lane masks of the ranking
    __shared__ u64 stage[SORT_TILE];
    __shared__ u32 whist[SORT_WAVES][RADIX];  // ranking: words of each digit seen so far by the wave; then: position of the wave's
                                              // first word of the digit in the tile's digit-sorted order
    __shared__ u32 gdelta[RADIX];             // output position of the tile's digit-sorted word j of digit d = gdelta[d] + j
    __shared__ u32 wtot[SORT_WAVES];
    __shared__ u32 place_s[4];
    u64(*wmask)[RADIX] = reinterpret_cast<u64(*)[RADIX]>(stage);  // [SORT_WAVES][RADIX]: lanes of the wave that hold digit d

    const u32 lane = threadIdx.x & 63u;
    const u32 w = threadIdx.x >> 6;
    if (threadIdx.x == 0) place_s[0] = atomicAdd(&ticket[tag_pass], 1u);
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) {
        whist[i][threadIdx.x] = 0;
        wmask[i][threadIdx.x] = 0ull;
    }
    __syncthreads();
    const u32 tile = place_s[0];
    TilePlace tp;
    if (SEG) {
        if (tile >= seg->tile_first[RADIX]) return;  // (the grid is an upper bound of the tile count)
        // the bucket of the tile: the one non-empty bucket with tile_first[b] <= tile < tile_first[b + 1]
        const u32 f0 = seg->tile_first[threadIdx.x], f1 = seg->tile_first[threadIdx.x + 1];
        if (f0 <= tile && tile < f1) {
            place_s[1] = threadIdx.x;
            place_s[2] = f0;
        }
        __syncthreads();
        tp.bucket = place_s[1];
        tp.floor = place_s[2];
        const u32 s0 = seg->start[tp.bucket], s1 = seg->start[tp.bucket + 1];
        tp.lo = s0 + (tile - tp.floor) * SORT_TILE;
        tp.hi = s1 - tp.lo < static_cast<u32>(SORT_TILE) ? s1 : tp.lo + SORT_TILE;
    } else {
        tp.bucket = 0;
        tp.floor = 0;
        tp.lo = tile * SORT_TILE;  // (n < 2^32 - TILE: checked on the host)
        tp.hi = n - tp.lo < static_cast<u32>(SORT_TILE) ? n : tp.lo + SORT_TILE;
    }
    const u32* dbase = digit_base + static_cast<size_t>(tp.bucket) * base_stride;

    // ---- load, rank ----
    const u32 base = tp.lo + w * (SORT_ITEMS * 64) + lane;
    u64 key[SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const u32 i = base + it * 64;
        const u64 v = kin[i < tp.hi ? i : tp.hi - 1u];  // (unconditional: a branch per load otherwise)
        key[it] = i < tp.hi ? v : ~0ull;
    }
    u32 rank2[SORT_ITEMS / 2];  // rank among the wave's words with the same digit (< 1024: two per register)
    const u64 lanebit = 1ull << lane;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const bool act = base + it * 64 < tp.hi;
        const u32 d = digit_of(key[it], shift);
        const lds_vu64 m = lds_ptr<lds_vu64>(&wmask[w][d]);
        const lds_vu32 h = lds_ptr<lds_vu32>(&whist[w][d]);
        if (act) __hip_atomic_fetch_or(&wmask[w][d], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u64 same = *m;   // every active lane of the wave with my digit (the wave's LDS operations execute in order)
        const u32 prev = *h;   // words of this digit in the wave's earlier chunks
        const u32 below = lanes_below(same);
        if (act && below == 0) {  // first lane of its digit group: clears the mask for the next chunk, adds the group
            *m = 0ull;
            *h = prev + static_cast<u32>(__builtin_popcountll(same));
        }
        if (it & 1) rank2[it >> 1] |= (prev + below) << 16;
        else rank2[it >> 1] = prev + below;
    }
    __syncthreads();

    // ---- thread d: the tile's count of digit d, published; its position in the tile's digit order; the count in all
    //      earlier tiles of the chain by look-back ----
    {
        const u32 d = threadIdx.x;
        u32 cw[SORT_WAVES];
        u32 local = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) {
            cw[i] = whist[i][d];
            local += cw[i];
        }
        const u64 tag = static_cast<u64>(tag_pass + 1) << 44;
        u64* mine = status + static_cast<u64>(tile) * RADIX + d;
        const bool first = tile == tp.floor;
        __hip_atomic_store(mine, tag | (first ? ST_PREFIX : ST_LOCAL) | local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 tile_total = 0;
        const u32 tb = block_exclusive_scan(local, wtot, tile_total);  // first word of digit d in the tile's digit-sorted order
        {
            u32 acc = tb;
#pragma unroll
            for (int i = 0; i < SORT_WAVES; ++i) {
                whist[i][d] = acc;
                acc += cw[i];
            }
        }
        u32 before = 0;
        if (!first) {
            // Look back over the predecessors' status words, several per round trip (independent loads in flight
            // together).  4 at first: in the steady state a predecessor with a running total is that close; a round that
            // used up all it had fetched goes LOOK wide, so a chain of tiles that published their own counts together
            // (the start of a pass) is walked at LOOK tiles per round trip.
            constexpr int LOOK = PCPX_SORT_LOOK;
            u32 t = tile;  // next to inspect: t - 1
            bool done = false;
            u32 spins = 0;
            int look = 4;
            while (!done) {
                u64 v[LOOK];
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    const u32 back = static_cast<u32>(j) + 1u;
                    const u32 tj = t - tp.floor >= back ? t - back : tp.floor;
                    v[j] = 0;
                    if (j < look) v[j] = __hip_atomic_load(status + static_cast<u64>(tj) * RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int used = 0;
                bool stop = false;  // a status word that is not there yet: poll again from that tile
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    const u64 w_ = v[j];
                    const bool ready = (w_ >> 44) == static_cast<u64>(tag_pass + 1) && (w_ & ST_STATE) != 0ull;
                    stop = stop || done || j >= look || !ready;
                    if (!stop) {
                        before += static_cast<u32>(w_ & ST_COUNT);
                        --t;
                        ++used;
                        done = (w_ & ST_STATE) == ST_PREFIX || t == tp.floor;
                    }
                }
                look = used == look ? LOOK : 4;
                if (!done && ++spins >= SPIN_LIMIT) {  // cannot happen with ticketed tiles; never hang: report and bail out
                    atomicExch(failed, 1u);
                    done = true;
                }
            }
            __hip_atomic_store(mine, tag | ST_PREFIX | (static_cast<u64>(before) + local), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gdelta[d] = dbase[d] + before - tb;
    }
    __syncthreads();  // (also: every wave is past its ranking, the lane masks are dead: `stage` may be written)

    // ---- the tile goes out through LDS in digit order: consecutive lanes write consecutive addresses ----
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        if (base + it * 64 < tp.hi) stage[whist[w][digit_of(key[it], shift)] + ((rank2[it >> 1] >> (16 * (it & 1))) & 0xFFFFu)] = key[it];
    }
    __syncthreads();
    const u32 in_tile = tp.hi - tp.lo;
    for (u32 j = threadIdx.x; j < in_tile; j += SORT_BLOCK) {
        const u64 k = stage[j];
        const u32 dst = gdelta[digit_of(k, shift)] + j;
        if (PAYLOAD) {
            // the element's index is in the word's low bits; its record moves with the word, and the word now names
            // the record's position
            const u64 e = k & pl.low_mask;
            const float x = pl.xyz[3 * e], y = pl.xyz[3 * e + 1], z = pl.xyz[3 * e + 2];
            pl.rec[dst] = make_float4(x, y, z, __uint_as_float(static_cast<u32>(e)));
            kout[dst] = (k & ~pl.low_mask) | dst;
        } else {
            kout[dst] = k;
        }
    }
}

// Per bucket: digit counts of the bucketed passes, hist[bucket][pass][256] (zeroed by the caller).  A block takes
// `tiles_per_block` consecutive tiles of the bucketed tile order and flushes its LDS counts whenever the bucket changes.
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_seg_hist(const u64* __restrict__ keys, const SegTable* __restrict__ seg, int first_bit,
                                                              int passes, u32 tiles_per_block, u32* __restrict__ hist)
{
    __shared__ u32 h[MAX_PASSES - 1][RADIX];
    __shared__ u32 place_s[2];
    const u32 ntiles = seg->tile_first[RADIX];
    u32 tile = blockIdx.x * tiles_per_block;
    if (tile >= ntiles) return;
    const u32 tile_end = tile + tiles_per_block < ntiles ? tile + tiles_per_block : ntiles;
    for (int p = 0; p < passes; ++p) h[p][threadIdx.x] = 0;
    u32 bucket = RADIX, bfirst = 0, bend = 0;  // current bucket, its tiles [bfirst, bend)
    auto flush = [&]() {
        __syncthreads();
        for (int p = 0; p < passes; ++p) {
            const u32 c = h[p][threadIdx.x];
            if (c) atomicAdd(&hist[(static_cast<size_t>(bucket) * (MAX_PASSES - 1) + p) * RADIX + threadIdx.x], c);
            h[p][threadIdx.x] = 0;
        }
    };
    for (; tile < tile_end; ++tile) {
        if (bucket == RADIX || tile >= bend) {
            if (bucket != RADIX) flush();
            const u32 f0 = seg->tile_first[threadIdx.x], f1 = seg->tile_first[threadIdx.x + 1];
            __syncthreads();
            if (f0 <= tile && tile < f1) {
                place_s[0] = threadIdx.x;
                place_s[1] = f0;
            }
            __syncthreads();
            bucket = place_s[0];
            bfirst = place_s[1];
            bend = seg->tile_first[bucket + 1];
        }
        const u32 s1 = seg->start[bucket + 1];
        const u32 lo = seg->start[bucket] + (tile - bfirst) * SORT_TILE;
        const u32 hi = s1 - lo < static_cast<u32>(SORT_TILE) ? s1 : lo + SORT_TILE;
        for (u32 i = lo + threadIdx.x; i < hi; i += SORT_BLOCK) {
            const u64 key = keys[i];
            for (int p = 0; p < passes; ++p) atomicAdd(&h[p][digit_of(key, first_bit + 8 * p)], 1u);
        }
    }
    flush();
}

// block = bucket: exclusive scan of every pass's counts, offset by the bucket's first position
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_seg_bases(const u32* __restrict__ hist, const SegTable* __restrict__ seg, int passes,
                                                               u32* __restrict__ bases)
{
    __shared__ u32 wtot[SORT_WAVES];
    const u32 b = blockIdx.x;
    const u32 s0 = seg->start[b];
    if (seg->start[b + 1] == s0) return;
    for (int p = 0; p < passes; ++p) {
        const size_t at = (static_cast<size_t>(b) * (MAX_PASSES - 1) + p) * RADIX + threadIdx.x;
        u32 total = 0;
        bases[at] = s0 + block_exclusive_scan(hist[at], wtot, total);
    }
}

struct TmpLayout {
    size_t o_top, o_ctl, o_seg, o_seghist, o_segbase, o_status, o_k, total;
    u64 ntiles_max;
};
TmpLayout tmp_layout(u64 n)
{
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    TmpLayout t;
    t.ntiles_max = (n + SORT_TILE - 1) / SORT_TILE + RADIX;  // bucketed passes: every bucket may end with a partial tile
    t.o_top = 0;
    t.o_ctl = t.o_top + al(RADIX * sizeof(u32));
    t.o_seg = t.o_ctl + al(64 * sizeof(u32));
    t.o_seghist = t.o_seg + al(sizeof(SegTable));
    t.o_segbase = t.o_seghist + al(static_cast<size_t>(RADIX) * (MAX_PASSES - 1) * RADIX * sizeof(u32));
    t.o_status = t.o_segbase + al(static_cast<size_t>(RADIX) * (MAX_PASSES - 1) * RADIX * sizeof(u32));
    t.o_k = t.o_status + al(t.ntiles_max * RADIX * sizeof(u64));
    t.total = t.o_k + al(n * sizeof(u64));
    return t;
}

}  // namespace

// Temporary storage: top-digit counts + tickets + failure flag + bucket table + per-bucket counts and bases + tile
// status + one key buffer.  Result in kout.  Call with tmp == nullptr to get tmp_bytes.  kin is not modified.
// payload (the index build): see SortPayload in pcpx_internal.h.
int sort_keys_u64(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, u64 n, hipStream_t s, int first_bit, const SortPayload* payload)
{
    const TmpLayout L = tmp_layout(n);
    if (!tmp) {
        tmp_bytes = L.total;
        return PCPX_OK;
    }
    if (tmp_bytes < L.total) {
        set_error("pcpx: sort temporary storage too small (%zu < %zu)", tmp_bytes, L.total);
        return PCPX_ERR_INVALID;
    }
    if (n == 0) return PCPX_OK;
    if (first_bit < 0 || first_bit > 56 || (first_bit & 7) || n >= (1ull << 32) - SORT_TILE) {
        set_error("pcpx: radix sort first_bit %d / n %llu", first_bit, static_cast<unsigned long long>(n));
        return PCPX_ERR_INVALID;
    }
    char* base = static_cast<char*>(tmp);
    u32* top_hist = reinterpret_cast<u32*>(base + L.o_top);
    u32* ctl = reinterpret_cast<u32*>(base + L.o_ctl);  // [0, 8) tickets, [8] failure flag
    SegTable* seg = reinterpret_cast<SegTable*>(base + L.o_seg);
    u32* seg_hist = reinterpret_cast<u32*>(base + L.o_seghist);
    u32* seg_base = reinterpret_cast<u32*>(base + L.o_segbase);
    u64* status = reinterpret_cast<u64*>(base + L.o_status);
    u64* kt = reinterpret_cast<u64*>(base + L.o_k);
    const int passes = (64 - first_bit) / 8;  // the top-digit pass + (passes - 1) bucketed ones
    const int low = passes - 1;
    const u32 n32 = static_cast<u32>(n);
    const u64 ntiles = (n + SORT_TILE - 1) / SORT_TILE;

    // counts, tickets, flag, tables and the status array: cleared once per sort
    const bool have_top = payload && payload->top_hist_ready;
    if (have_top) {
        PCPX_HIP(hipMemcpyAsync(top_hist, payload->top_hist_ready, RADIX * sizeof(u32), hipMemcpyDeviceToDevice, s));
        PCPX_HIP(hipMemsetAsync(base + L.o_ctl, 0, L.o_k - L.o_ctl, s));
    } else {
        PCPX_HIP(hipMemsetAsync(base, 0, L.o_k, s));
        const u32 hblocks = static_cast<u32>(ntiles < 2048 ? ntiles : 2048);
        k_sort_top_hist<<<hblocks, SORT_BLOCK, 0, s>>>(kin, n, top_hist);
    }
    k_sort_seg_setup<<<1, SORT_BLOCK, 0, s>>>(top_hist, seg);

    // ping-pong between tmp and out so that the LAST pass writes out
    auto dst_of = [&](int j) { return ((passes - 1 - j) & 1) == 0 ? kout : kt; };
    SortPayloadArgs pl{nullptr, nullptr, 0};
    u64* kdst = dst_of(0);
    if (payload && payload->xyz) {
        pl.xyz = payload->xyz;
        pl.rec = reinterpret_cast<float4*>(payload->rec);
        pl.low_mask = (1ull << payload->idx_bits) - 1ull;
        k_sort_pass<false, true><<<static_cast<u32>(ntiles), SORT_BLOCK, 0, s>>>(kin, kdst, n32, TOP_SHIFT, 0, seg->start, 0, seg, status, ctl, ctl + 8, pl);
    } else {
        k_sort_pass<false, false><<<static_cast<u32>(ntiles), SORT_BLOCK, 0, s>>>(kin, kdst, n32, TOP_SHIFT, 0, seg->start, 0, seg, status, ctl, ctl + 8, pl);
    }
    if (low > 0) {
        u32 tpb = static_cast<u32>(L.ntiles_max / 1024);
        tpb = tpb < 4 ? 4 : tpb > 32 ? 32 : tpb;
        const u32 hblocks = static_cast<u32>((L.ntiles_max + tpb - 1) / tpb);
        k_sort_seg_hist<<<hblocks, SORT_BLOCK, 0, s>>>(kdst, seg, first_bit, low, tpb, seg_hist);
        k_sort_seg_bases<<<RADIX, SORT_BLOCK, 0, s>>>(seg_hist, seg, low, seg_base);
        const u64* ksrc = kdst;
        for (int p = 0; p < low; ++p) {
            kdst = dst_of(p + 1);
            k_sort_pass<true, false><<<static_cast<u32>(L.ntiles_max), SORT_BLOCK, 0, s>>>(ksrc, kdst, n32, first_bit + 8 * p, p + 1, seg_base + p * RADIX,
                                                                                          (MAX_PASSES - 1) * RADIX, seg, status, ctl, ctl + 8, pl);
            ksrc = kdst;
        }
    }
    return check_hip(hipGetLastError(), "radix sort kernels", __FILE__, __LINE__);
}

// the failure flag of the last sort that used this temporary storage (device word; read it after synchronising)
const u32* sort_failure_flag(void* tmp)
{
    const TmpLayout L = tmp_layout(0);
    return reinterpret_cast<const u32*>(static_cast<char*>(tmp) + L.o_ctl) + 8;
}

}  // namespace pcpx
