// pcpx_sort.hip -- stable radix sort of 64-bit words by their bits [first_bit, 64), hand-written for gfx950 (wave64).
//
// The "radix sort" step of BASELINE.json's north_star: the index build sorts one packed word per point -- curve key in
// the high bits, point index in the low bits (pcpx_curve.h) -- and an arbitrary query batch does the same.
//
// Shape (round 3): MOST significant digit first, then least-significant-digit passes INSIDE the 256 buckets.
//   k_sort_top_hist     counts the top digit (bits [56, 64)) of every TILE of 4096 words (the build's k_codes does it itself),
//   k_sort_tile_sums /  per digit: exclusive scan of the tiles' counts (= where a tile's words of that digit start inside
//   k_sort_tile_scan    the digit's bucket) and the digit's total,
//   k_sort_seg_setup    turns the totals into the bucket table: first position and first tile of every bucket,
//   k_sort_pass<false>  the top-digit pass: a stable partition of the whole array into the 256 buckets.  Its input order is
//                       known before it starts, so the tiles' offsets come from the scan above and a tile depends on no
//                       other tile (as one more look-back pass its chain ran over all ~2 400 tiles of 10 M words and a tile
//                       lived 57 us, 40 of them waiting: profiles/r03_pmc_rebuild.json).  For the index build it also moves
//                       a 16-byte record {x, y, z, id} per point to the word's new position and writes that position into
//                       the word's low bits, so that everything after this pass -- and the leaf gather at the end -- stays
//                       inside one bucket,
//   k_sort_seg_hist     per bucket: the digit counts of every remaining pass (one sweep over the partitioned words); a
//                       tile scans its bucket's 256 counts itself to find where its digits start,
//   k_sort_pass<true>   one per remaining digit, least significant first ("onesweep": ticketed tiles, per-wave digit ranks,
//                       decoupled look-back over the predecessors' published counts): a tile never straddles a bucket and
//                       the look-back ends at the bucket's first tile.
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
#ifndef PCPX_SORT_ITEMS
#define PCPX_SORT_ITEMS 16
#endif
constexpr int SORT_ITEMS = PCPX_SORT_ITEMS;         // 64-word chunks per wave
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;  // words per tile
static_assert(SORT_TILE == SORT_TILE_WORDS, "pcpx_internal.h names the tile size for the callers that count the top digit themselves");
constexpr int RADIX = 256;
constexpr int MAX_PASSES = 8;
constexpr int TOP_SHIFT = 56;
#ifndef PCPX_SORT_LOOK
#define PCPX_SORT_LOOK 8  // status words per look-back round trip once a round has used up the 4 it starts with
#endif

// bucket table of one sort (device memory, written by k_sort_seg_setup)
struct SegTable {
    u32 start[RADIX + 1];       // first position of bucket b; start[RADIX] = n
    u32 tile_first[RADIX + 1];  // first tile of bucket b in the bucketed tile order; tile_first[RADIX] = number of tiles
    u32 pass_tile_first[MAX_PASSES - 1][RADIX + 1];  // the same for bucketed pass q (1-based: row q - 1), over the buckets that take
                                                     // part in it: a pass hands out tickets for its own tiles only
    u32 first_pass[RADIX];      // the first bucketed pass (1-based) bucket b takes part in: 1 unless the caller lets a bucket's
                                // depth follow its size (SortPayload::adaptive_margin_bits), then passes below it are skipped
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

// two scans at once (one pair of barriers)
__device__ __forceinline__ void block_exclusive_scan2(u32 va, u32 vb, u32* wtot2, u32& ea, u32& eb)
{
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32 ia = va, ib = vb;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 ua = __shfl_up(ia, off), ub = __shfl_up(ib, off);
        if (lane >= static_cast<u32>(off)) {
            ia += ua;
            ib += ub;
        }
    }
    __syncthreads();
    if (lane == 63) {
        wtot2[w] = ia;
        wtot2[SORT_WAVES + w] = ib;
    }
    __syncthreads();
    u32 ba = 0, bb = 0;
#pragma unroll
    for (u32 i = 0; i < SORT_WAVES; ++i) {
        ba += i < w ? wtot2[i] : 0u;
        bb += i < w ? wtot2[SORT_WAVES + i] : 0u;
    }
    ea = ba + ia - va;
    eb = bb + ib - vb;
}

// counts of the top digit per tile of SORT_TILE words: tile_hist[tile][256]
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_top_hist(const u64* __restrict__ keys, u64 n, u32 ntiles, u32* __restrict__ tile_hist)
{
    __shared__ u32 h[RADIX];
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        h[threadIdx.x] = 0;
        __syncthreads();
        const u64 lo = static_cast<u64>(tile) * SORT_TILE, hi = lo + SORT_TILE < n ? lo + SORT_TILE : n;
        u64 key[SORT_ITEMS];
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const u64 i = lo + static_cast<u64>(it) * SORT_BLOCK + threadIdx.x;
            key[it] = keys[i < hi ? i : hi - 1u];
        }
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it)
            if (lo + static_cast<u64>(it) * SORT_BLOCK + threadIdx.x < hi) atomicAdd(&h[digit_of(key[it], TOP_SHIFT)], 1u);
        __syncthreads();
        tile_hist[static_cast<size_t>(tile) * RADIX + threadIdx.x] = h[threadIdx.x];
        __syncthreads();
    }
}

// Exclusive scan of the tiles' counts per digit, in place, and the digits' totals -- two launches over chunks of consecutive
// tiles, every access a 1-KB row (thread = digit): (1) the chunks' sums, (2) a chunk's base = the sums of the chunks before
// it, then the running prefix through the chunk.  About sqrt(tiles) chunks, so that neither loop of (2) is long; eight rows
// in flight per thread.  (One block per digit walking its column was 17 us at 10 M words and 84 us at 50 M: 4-byte reads a
// kilobyte apart.)
constexpr u32 SCAN_CHUNKS_MAX = 256;
constexpr int SCAN_ROWS = 8;  // rows in flight per thread
__global__ __launch_bounds__(RADIX) void k_sort_tile_sums(const u32* __restrict__ tile_hist, u32 ntiles, u32 per_chunk, u32* __restrict__ chunk_sum)
{
    const u32 d = threadIdx.x;
    const u32 t0 = blockIdx.x * per_chunk, t1 = t0 + per_chunk < ntiles ? t0 + per_chunk : ntiles;
    u32 sum = 0;
    for (u32 t = t0; t < t1; t += SCAN_ROWS) {
        u32 v[SCAN_ROWS];
#pragma unroll
        for (int j = 0; j < SCAN_ROWS; ++j) v[j] = t + j < t1 ? tile_hist[static_cast<size_t>(t + j) * RADIX + d] : 0u;
#pragma unroll
        for (int j = 0; j < SCAN_ROWS; ++j) sum += v[j];
    }
    chunk_sum[blockIdx.x * RADIX + d] = sum;
}
__global__ __launch_bounds__(RADIX) void k_sort_tile_scan(u32* __restrict__ tile_hist, u32 ntiles, u32 per_chunk, const u32* __restrict__ chunk_sum,
                                                          u32* __restrict__ hist)
{
    const u32 d = threadIdx.x;
    u32 acc = 0;
    for (u32 c = 0; c < blockIdx.x; c += SCAN_ROWS) {
        u32 v[SCAN_ROWS];
#pragma unroll
        for (int j = 0; j < SCAN_ROWS; ++j) v[j] = c + j < blockIdx.x ? chunk_sum[(c + j) * RADIX + d] : 0u;
#pragma unroll
        for (int j = 0; j < SCAN_ROWS; ++j) acc += v[j];
    }
    const u32 t0 = blockIdx.x * per_chunk, t1 = t0 + per_chunk < ntiles ? t0 + per_chunk : ntiles;
    for (u32 t = t0; t < t1; t += SCAN_ROWS) {
        u32 v[SCAN_ROWS];
#pragma unroll
        for (int j = 0; j < SCAN_ROWS; ++j) v[j] = t + j < t1 ? tile_hist[static_cast<size_t>(t + j) * RADIX + d] : 0u;
#pragma unroll
        for (int j = 0; j < SCAN_ROWS; ++j) {
            if (t + j < t1) tile_hist[static_cast<size_t>(t + j) * RADIX + d] = acc;
            acc += v[j];
        }
    }
    if (blockIdx.x == gridDim.x - 1) hist[d] = acc;
}

// one block: counts -> bucket table
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_seg_setup(const u32* __restrict__ hist, SegTable* __restrict__ seg, int low, int margin_bits)
{
    __shared__ u32 wtot[SORT_WAVES];
    const u32 c = hist[threadIdx.x];
    {
        // Depth by size (index build only): a bucket of c words is sorted on as many of its lower digits as separate its words
        // with `margin_bits` to spare -- ceil((ceil(log2 c) + margin) / 8) of them, at least two -- and takes no part in the
        // passes below those.  The sort is then exact on the top digit and the top two lower digits everywhere, and as fine
        // as its density asks for in every bucket.
        u32 first = 1;
        if (margin_bits > 0 && low >= 3) {
            const int lg = c > 1 ? 32 - __builtin_clz(c - 1u) : 0;
            int need = (lg + margin_bits + 7) / 8;
            need = need < 2 ? 2 : need > low ? low : need;
            // (the last bucket is where the words of points outside the voxel grid go -- all ones above their index bits --, and
            //  the index counts on every inside word sorting before them: that bucket is always sorted on all of its digits)
            if (threadIdx.x == RADIX - 1) need = low;
            first = static_cast<u32>(low + 1 - need);
        }
        seg->first_pass[threadIdx.x] = first;
        for (int q = 1; q <= low; ++q) {
            u32 tiles = 0;
            const u32 f = block_exclusive_scan(static_cast<u32>(q) >= first ? (c + SORT_TILE - 1) / SORT_TILE : 0u, wtot, tiles);
            seg->pass_tile_first[q - 1][threadIdx.x] = f;
            if (threadIdx.x == 0) seg->pass_tile_first[q - 1][RADIX] = tiles;
        }
    }
    u32 total = 0;
    const u32 s = block_exclusive_scan(c, wtot, total);
    seg->start[threadIdx.x] = s;
    if (threadIdx.x == 0) seg->start[RADIX] = total;
    u32 tiles = 0;
    const u32 f = block_exclusive_scan((c + SORT_TILE - 1) / SORT_TILE, wtot, tiles);
    seg->tile_first[threadIdx.x] = f;
    if (threadIdx.x == 0) seg->tile_first[RADIX] = tiles;
}

// finish mode (SortPayload::finish), after the top-digit pass and the per-bucket digit counts: which buckets may stop after their top
// two lower digits.  A bucket of c words whose fullest 16-bit cell (= its largest count of the digit at bits [48, 56)) holds m words
// has c / 65536 words per 24-bit cell on average and m / 256 in that cell's: both at most SORT_RUN_TARGET, or the bucket takes all
// its passes.  first_pass is 1 or low - 1: the same parity (low = 4), which is why the top-digit pass could put every bucket into
// the same buffer before this was known.  One block; thread = bucket.
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_seg_plan(const u32* __restrict__ fullest_of, SegTable* __restrict__ seg, int low,
                                                              const u32* __restrict__ force_full, u32* __restrict__ low_pass_tiles)
{
    __shared__ u32 wtot[SORT_WAVES];
    const u32 b = threadIdx.x;
    const u32 c = seg->start[b + 1] - seg->start[b];
    const u32 fullest = fullest_of[b];  // (k_sort_seg_hist)
    const bool forced = force_full && ((force_full[b >> 5] >> (b & 31u)) & 1u) != 0u;
    const bool short_runs = !forced && b != RADIX - 1 && (c >> 16) <= SORT_RUN_TARGET && (fullest >> 8) <= SORT_RUN_TARGET;
    const u32 first = short_runs ? static_cast<u32>(low - 1) : 1u;
    seg->first_pass[b] = first;
    for (int q = 1; q <= low; ++q) {
        u32 tiles = 0;
        const u32 f = block_exclusive_scan(static_cast<u32>(q) >= first ? (c + SORT_TILE - 1) / SORT_TILE : 0u, wtot, tiles);
        seg->pass_tile_first[q - 1][b] = f;
        if (b == 0) seg->pass_tile_first[q - 1][RADIX] = tiles;
        if (b == 0 && q == 1 && low_pass_tiles) *low_pass_tiles = tiles;  // (what the lowest passes have to do: the next build of this handle sizes their grids by it)
    }
}

// status word: bits [0,40) count, [40,42) state (1 = this tile's own count, 2 = running total up to and including this
// tile), [44,48) pass tag (pass + 1; 0 = never written since the array was cleared)
constexpr u64 ST_LOCAL = 1ull << 40, ST_PREFIX = 2ull << 40, ST_STATE = 3ull << 40, ST_COUNT = (1ull << 40) - 1ull;
constexpr u32 SPIN_LIMIT = 1u << 24;

struct SortPayloadArgs {
    const float* xyz;  // n x 3, element order of the pass's input
    float4* rec;       // out: {x, y, z, bits of the element's index} at the word's new position
    u64 low_mask;      // the word's low bits that hold the element's index (in) / the record's position (out)
    const float4* rec_in;  // instead of xyz: the elements' records, moved as they are
};

// One tile's place in the pass.  SEG = false: tile t covers [t * TILE, ...) of the whole array.  SEG = true: the tiles
// of bucket b are seg.tile_first[b] ..., each inside [seg.start[b], seg.start[b + 1]).
struct TilePlace {
    u32 lo, hi;     // word range [lo, hi), hi - lo <= TILE
    u32 floor;      // first tile of the chain this tile's look-back walks
    u32 bucket;
};

template <bool SEG, bool PAYLOAD>
__global__ __launch_bounds__(SORT_BLOCK, SORT_ITEMS <= 8 ? 6 : SORT_ITEMS <= 16 ? 4 : 3) void k_sort_pass(const u64* __restrict__ kin, u64* __restrict__ kout, u64* __restrict__ kout_odd, u32 n, u32 tiles_arg, int shift, int tag_pass,
                                                          const u32* __restrict__ seg_hist, const u32* __restrict__ tile_prefix,
                                                          const SegTable* __restrict__ seg, u64* __restrict__ status,
                                                          u32* __restrict__ ticket, u32* __restrict__ failed, SortPayloadArgs pl)
{
    // LDS: the staging buffer of the scatter doubles, before that, as the per-wave lane masks of the ranking
    __shared__ u64 stage[SORT_TILE];
    __shared__ u32 whist[SORT_WAVES][RADIX];  // ranking: words of each digit seen so far by the wave; then: position of the wave's
                                              // first word of the digit in the tile's digit-sorted order
    __shared__ u32 gdelta[RADIX];             // output position of the tile's digit-sorted word j of digit d = gdelta[d] + j
    __shared__ u32 godd[SEG ? 1 : RADIX];     // top-digit pass: bucket d's words go to kout_odd, not kout (its first bucketed pass is an even one)
    __shared__ u32 wtot[2 * SORT_WAVES];
    __shared__ u32 place_s[4];
    u64(*wmask)[RADIX] = reinterpret_cast<u64(*)[RADIX]>(stage);  // [SORT_WAVES][RADIX]: lanes of the wave that hold digit d

    // Persistent blocks: a block takes tiles by ticket until none are left; the ticket of its next tile is requested while
    // it scatters the current one (the atomic's round trip, 2-3 us under load, was a sixth of a tile's life when the block
    // waited for it with nothing else to do, profiles/r03_pmc_rebuild.json).
    // (The top-digit pass, SEG = false, has no chain: one tile per block, taken by block index.)
    const u32* my_tile_first = SEG ? seg->pass_tile_first[tag_pass - 1] : seg->tile_first;  // (tag_pass = the bucketed pass's number, 1-based)
    const u32 ntiles = SEG ? my_tile_first[RADIX] : tiles_arg;
    // (a pass that few buckets take part in -- finish mode: the two lowest digits are for the buckets with long runs only -- has fewer
    //  tiles than the grid has blocks: a block beyond them leaves before it has touched LDS or the ticket; the others take every tile)
    if (SEG && blockIdx.x >= ntiles) return;
    if (threadIdx.x == 0) place_s[0] = SEG ? atomicAdd(&ticket[tag_pass], 1u) : blockIdx.x;
    for (;;) {
    // (the thread's index is made opaque per tile: otherwise every address formed from it is a loop invariant of the
    //  persistent block, kept in registers across the ranking loop -- which then spills)
    u32 tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const u32 lane = tid & 63u;
    const u32 w = tid >> 6;
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) {
        whist[i][tid] = 0;
        wmask[i][tid] = 0ull;
    }
    __syncthreads();
    const u32 tile = place_s[0];
    if (tile >= ntiles) break;
    TilePlace tp;
    if (SEG) {
        // the bucket of the tile: the one bucket of this pass with tile_first[b] <= tile < tile_first[b + 1]
        const u32 f0 = my_tile_first[tid], f1 = my_tile_first[tid + 1];
        if (f0 <= tile && tile < f1) {
            place_s[1] = tid;
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
        const u32 d = tid;
        u32 cw[SORT_WAVES];
        u32 local = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) {
            cw[i] = whist[i][d];
            local += cw[i];
        }
        const u64 tag = static_cast<u64>(tag_pass + 1) << 44;
        u64* mine = status + static_cast<u64>(tile) * RADIX + d;
        const bool first = !SEG || tile == tp.floor;  // (top-digit pass: the offset comes from the tile scan, nothing is published)
        if (SEG) __hip_atomic_store(mine, tag | (first ? ST_PREFIX : ST_LOCAL) | local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // tb: first word of digit d in the tile's digit-sorted order; dbase: first output position of digit d in this pass --
        // the bucket's first position plus the exclusive scan of the bucket's digit counts (scanned here, by every tile: 256
        // L2-resident words, instead of a launch of its own between the count and the first pass)
        u32 tb, dbase;
        if (SEG) {
            block_exclusive_scan2(local, seg_hist[static_cast<size_t>(tp.bucket) * ((MAX_PASSES - 1) * RADIX) + d], wtot, tb, dbase);
            dbase += seg->start[tp.bucket];
        } else {
            u32 unused = 0;
            tb = block_exclusive_scan(local, wtot, unused);
            dbase = seg->start[d] + tile_prefix[static_cast<size_t>(tile) * RADIX + d];
        }
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
            constexpr int LOOK = SEG ? 4 : PCPX_SORT_LOOK;  // (bucketed chains are short; the wide form costs 16 registers)
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
        gdelta[d] = dbase + before - tb;
        if (!SEG) godd[d] = (seg->first_pass[d] - 1u) & 1u;
    }
    __syncthreads();  // (also: every wave is past its ranking, the lane masks are dead: `stage` may be written)
    // The next ticket is taken HERE: this tile has published everything its successors wait for, so the tile behind the
    // ticket starts at most one scatter later -- taken any earlier (at the top, to have it ready) the ticket would be a
    // promise that is kept only after this whole tile, and every tile behind it spins in its look-back until then (measured:
    // the top-digit pass 173 -> 209 us).  Its round trip (2-3 us under load) still hides behind the scatter below.
    u32 next_ticket = ~0u;
    if (SEG && tid == 0) next_ticket = atomicAdd(&ticket[tag_pass], 1u);

    // ---- the tile goes out through LDS in digit order: consecutive lanes write consecutive addresses ----
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        if (base + it * 64 < tp.hi) stage[whist[w][digit_of(key[it], shift)] + ((rank2[it >> 1] >> (16 * (it & 1))) & 0xFFFFu)] = key[it];
    }
    __syncthreads();
    const u32 in_tile = tp.hi - tp.lo;
    if (PAYLOAD) {
        // the element's index is in the word's low bits; its record moves with the word, and the word now names the
        // record's position.  Four words per trip, so that their coordinate gathers are in flight together.
        constexpr int U = 4;
        for (u32 j0 = tid; j0 < in_tile; j0 += U * SORT_BLOCK) {
            u64 k[U], e[U];
            u32 dst[U];
            float x[U], y[U], z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 j = j0 + u * SORT_BLOCK;
                k[u] = stage[j < in_tile ? j : j0];
                e[u] = k[u] & pl.low_mask;
                dst[u] = gdelta[digit_of(k[u], shift)] + (j < in_tile ? j : j0);
            }
            float id[U];
            if (pl.rec_in) {  // (block-uniform)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 r = pl.rec_in[e[u]];
                    x[u] = r.x;
                    y[u] = r.y;
                    z[u] = r.z;
                    id[u] = r.w;
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    x[u] = pl.xyz[3 * e[u]];
                    y[u] = pl.xyz[3 * e[u] + 1];
                    z[u] = pl.xyz[3 * e[u] + 2];
                    id[u] = __uint_as_float(static_cast<u32>(e[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u * SORT_BLOCK < in_tile) {
                    pl.rec[dst[u]] = make_float4(x[u], y[u], z[u], id[u]);
                    (godd[digit_of(k[u], shift)] ? kout_odd : kout)[dst[u]] = (k[u] & ~pl.low_mask) | dst[u];
                }
            }
        }
    } else {
        for (u32 j = tid; j < in_tile; j += SORT_BLOCK) {
            const u64 k = stage[j];
            const u32 d = digit_of(k, shift);
            ((!SEG && godd[d]) ? kout_odd : kout)[gdelta[d] + j] = k;
        }
    }
    __syncthreads();  // (everyone is done with `stage`, `whist`, `gdelta` and place_s of this tile)
    if (tid == 0) place_s[0] = next_ticket;
    }  // next tile
}

// Per bucket: digit counts of the bucketed passes, hist[bucket][pass][256] (zeroed by the caller).  A block takes
// `tiles_per_block` consecutive tiles of the bucketed tile order and flushes its LDS counts whenever the bucket changes.
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_seg_hist(const u64* __restrict__ keys_even, const u64* __restrict__ keys_odd,
                                                              const SegTable* __restrict__ seg, int first_bit, int passes, u32 tiles_per_block,
                                                              u32* __restrict__ hist, u32* __restrict__ fullest)
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
            if (c) {
                const u32 before = atomicAdd(&hist[(static_cast<size_t>(bucket) * (MAX_PASSES - 1) + p) * RADIX + threadIdx.x], c);
                // (finish mode: the bucket's largest count of its top lower digit -- the add that completes a count sees the total)
                if (fullest && p == passes - 1) atomicMax(&fullest[bucket], before + c);
            }
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
        const u64* keys = ((seg->first_pass[bucket] - 1u) & 1u) ? keys_odd : keys_even;  // where the top-digit pass put this bucket
        u64 key[SORT_ITEMS];  // (all loads of the tile in flight before the first count)
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const u32 i = lo + it * SORT_BLOCK + threadIdx.x;
            key[it] = keys[i < hi ? i : hi - 1u];
        }
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            if (lo + it * SORT_BLOCK + threadIdx.x < hi)
                for (int p = 0; p < passes; ++p) atomicAdd(&h[p][digit_of(key[it], first_bit + 8 * p)], 1u);
        }
    }
    flush();
}

struct TmpLayout {
    size_t o_top, o_ctl, o_seg, o_seghist, o_fullest, o_status, o_tilehist, o_chunks, o_k, total;
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
    t.o_fullest = t.o_seghist + al(static_cast<size_t>(RADIX) * (MAX_PASSES - 1) * RADIX * sizeof(u32));
    t.o_status = t.o_fullest + al(RADIX * sizeof(u32));
    t.o_tilehist = t.o_status + al(t.ntiles_max * RADIX * sizeof(u64));
    t.o_chunks = t.o_tilehist + al(t.ntiles_max * RADIX * sizeof(u32));
    t.o_k = t.o_chunks + al(static_cast<size_t>(SCAN_CHUNKS_MAX) * RADIX * sizeof(u32));
    t.total = t.o_k + al(n * sizeof(u64));
    return t;
}

}  // namespace

// Temporary storage: top-digit counts + tickets + failure flag + bucket table + per-bucket counts + tile status + one key
// buffer.  Result in kout.  Call with tmp == nullptr to get tmp_bytes.  kin is not modified.
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
    u64* status = reinterpret_cast<u64*>(base + L.o_status);
    u64* kt = reinterpret_cast<u64*>(base + L.o_k);
    const int passes = (64 - first_bit) / 8;  // the top-digit pass + (passes - 1) bucketed ones
    const int low = passes - 1;
    const u32 n32 = static_cast<u32>(n);
    const u64 ntiles = (n + SORT_TILE - 1) / SORT_TILE;

    // tickets, flag, tables, the per-bucket counts and the status array: cleared once per sort
    PCPX_HIP(hipMemsetAsync(base + L.o_ctl, 0, L.o_tilehist - L.o_ctl, s));
    u32* tile_hist = reinterpret_cast<u32*>(base + L.o_tilehist);
    if (payload && payload->tile_hist_ready) {
        tile_hist = payload->tile_hist_ready;
    } else {
        const u32 hblocks = static_cast<u32>(ntiles < 2048 ? ntiles : 2048);
        k_sort_top_hist<<<hblocks, SORT_BLOCK, 0, s>>>(kin, n, static_cast<u32>(ntiles), tile_hist);
    }
    {
        const u32 nt = static_cast<u32>(ntiles);
        u32 want = 16;  // about sqrt(tiles) chunks
        while (want < SCAN_CHUNKS_MAX && want * want < nt) want += 8;
        const u32 per_chunk = (nt + want - 1) / want;
        const u32 chunks = (nt + per_chunk - 1) / per_chunk;
        u32* chunk_sum = reinterpret_cast<u32*>(base + L.o_chunks);
        k_sort_tile_sums<<<chunks, RADIX, 0, s>>>(tile_hist, nt, per_chunk, chunk_sum);
        k_sort_tile_scan<<<chunks, RADIX, 0, s>>>(tile_hist, nt, per_chunk, chunk_sum, top_hist);
    }
    const u32* top = top_hist;
    u32* failed = payload && payload->failed_flag ? payload->failed_flag : ctl + 8;
    const bool finish = payload && payload->finish && low == 4;  // (the parity argument of k_sort_seg_plan)
    const int margin = (payload && low >= 3 && !finish) ? payload->adaptive_margin_bits : 0;
    k_sort_seg_setup<<<1, SORT_BLOCK, 0, s>>>(top, seg, low, margin);

    // persistent grids: as many blocks as are resident at once (4 per CU: LDS and registers), never more than tiles
    static int cus_of_device[64] = {};
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64) cus = cus_of_device[dev];
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        (void)hipGetLastError();
        if (dev >= 0 && dev < 64) cus_of_device[dev] = cus;
    }
#ifndef PCPX_SORT_PERSISTENT
#define PCPX_SORT_PERSISTENT 1  // 0: one block per tile (the dispatcher hands out blocks; every block still takes its tile by ticket)
#endif
    const u64 resident = PCPX_SORT_PERSISTENT ? static_cast<u64>(cus) * (SORT_ITEMS <= 8 ? 6 : SORT_ITEMS <= 16 ? 4 : 3) : ~0ull;
    const u32 grid_top = static_cast<u32>(ntiles);  // (one tile per block: no chain, nothing to keep resident)
    const u32 grid_seg = static_cast<u32>(L.ntiles_max < resident ? L.ntiles_max : resident);
    // ping-pong between tmp and out so that the LAST pass writes out
    // (finish mode: the last pass writes the temporary buffer -- kout's buffer is for what the caller's finish kernel makes of it)
    auto dst_of = [&](int j) { return (((passes - 1 - j) & 1) == 0) != finish ? kout : kt; };
    SortPayloadArgs pl{nullptr, nullptr, 0, nullptr};
    u64* kdst = dst_of(0);
    u64* kodd = dst_of(1);  // (a bucket whose first bucketed pass is pass q is read there from dst_of(q - 1))
    if (payload && (payload->xyz || payload->rec_in)) {
        pl.xyz = payload->xyz;
        pl.rec_in = payload->rec_in;
        pl.rec = payload->rec;
        pl.low_mask = (1ull << payload->idx_bits) - 1ull;
        k_sort_pass<false, true><<<grid_top, SORT_BLOCK, 0, s>>>(kin, kdst, kodd, n32, static_cast<u32>(ntiles), TOP_SHIFT, 0, nullptr, tile_hist, seg, status, ctl, failed, pl);
    } else {
        k_sort_pass<false, false><<<grid_top, SORT_BLOCK, 0, s>>>(kin, kdst, kodd, n32, static_cast<u32>(ntiles), TOP_SHIFT, 0, nullptr, tile_hist, seg, status, ctl, failed, pl);
    }
    if (low > 0) {
        u32 tpb = static_cast<u32>(L.ntiles_max / 2048);  // tiles per block: enough blocks to fill the chip, few flushes of the counts
        tpb = tpb < 2 ? 2 : tpb > 32 ? 32 : tpb;
        const u32 hblocks = static_cast<u32>((L.ntiles_max + tpb - 1) / tpb);
        u32* fullest = reinterpret_cast<u32*>(base + L.o_fullest);  // (cleared with the rest above)
        k_sort_seg_hist<<<hblocks, SORT_BLOCK, 0, s>>>(kdst, kodd, seg, first_bit, low, tpb, seg_hist, finish ? fullest : nullptr);
        if (finish) k_sort_seg_plan<<<1, SORT_BLOCK, 0, s>>>(fullest, seg, low, payload->force_full, payload->low_pass_tiles_out);
        const u64* ksrc = kdst;
        for (int p = 0; p < low; ++p) {
            kdst = dst_of(p + 1);
            // (finish mode: the two lowest passes are for the buckets with long runs only -- none in a uniform cloud -- and a grid of a
            //  thousand blocks costs 13 us to start and end even when no block finds a tile: the caller says how many tiles the last
            //  build of this handle had there, and the grid follows with room to spare; too few blocks are slower, never wrong)
            u32 grid_here = grid_seg;
            if (finish && p < low - 2 && payload->low_pass_tiles_hint != ~0u) {
                const u64 want = 2ull * payload->low_pass_tiles_hint + 64;
                if (want < grid_here) grid_here = static_cast<u32>(want);
            }
            k_sort_pass<true, false><<<grid_here, SORT_BLOCK, 0, s>>>(ksrc, kdst, nullptr, n32, 0u, first_bit + 8 * p, p + 1, seg_hist + p * RADIX, nullptr, seg, status, ctl, failed, pl);
            ksrc = kdst;
        }
    }
    if (payload && payload->finish_out) {
        payload->finish_out->words = finish ? dst_of(passes - 1) : nullptr;
        payload->finish_out->bucket_first_pass = seg->first_pass;
    }
    return check_hip(hipGetLastError(), "radix sort kernels", __FILE__, __LINE__);
}

// where a caller that counts the top digit per tile itself (SortPayload::tile_hist_ready) may put the counts: a region of the
// sort's own temporary storage, [tile][256] for n words
u32* sort_tile_hist_buffer(void* tmp, u64 n) { return reinterpret_cast<u32*>(static_cast<char*>(tmp) + tmp_layout(n).o_tilehist); }

// the failure flag of the last sort that used this temporary storage (device word; read it after synchronising)
const u32* sort_failure_flag(void* tmp)
{
    const TmpLayout L = tmp_layout(0);
    return reinterpret_cast<const u32*>(static_cast<char*>(tmp) + L.o_ctl) + 8;
}

}  // namespace pcpx
