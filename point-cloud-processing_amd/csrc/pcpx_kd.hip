// pcpx_kd.hip -- pcp::basic_linked_kdtree_t for K > 3 coordinates (the reference's kd-tree is generic in K:
// include/pcp/kdtree/linked_kdtree.hpp:64; the curve-sorted index of this library holds three).  Every caller of the
// reference's kd-tree in its examples and tests has K = 3 and goes through the index; this file gives the other values of
// K the same two queries with the same results, by exhaustive search on the GPU:
//   nearest_neighbours (linked_kdtree.hpp:200-262, recurse_knn :436-540): the k points with the smallest squared distance
//     <p - q, p - q>, summed over the coordinates in order starting from 0 (common/norm.hpp:123-141), a point whose every
//     coordinate is within eps of the target's left out (floating_point_equals, vector3d_queries.hpp:31-35: |a - b| < eps),
//     nearest first; ties at equal distance in index order (the reference leaves them to its heap);
//   range_search with a kd box (linked_kdtree.hpp:270-311; kd_axis_aligned_bounding_box_t::contains: min <= p <= max on
//     every axis), results unordered.
// Coordinates are stored one array per axis.  A search is a sweep over all points: n x nq distance evaluations, spread over
// (query, segment of the cloud) wavefronts, each keeping its 64 best in one register per lane, then merged per query.
// Meant for the tens of thousands to few millions of points a K-dimensional feature cloud has; no tree is built.
#include "pcpx_internal.h"

#include <mutex>
#include <new>
#include <vector>

namespace pcpx {
namespace {

constexpr u32 KD_MAX_DIMS = PCPX_KD_MAX_DIMS;
constexpr u64 KD_PAD = ~0ull;
constexpr u32 KD_SEG_MIN = 4096;   // points per (query, segment) wavefront at least
constexpr u32 KD_WAVES_WANTED = 8192;

struct KdIndex {
    int device = 0;
    hipStream_t stream = nullptr;
    u64 n = 0;
    u32 dims = 0;
    float* d_pts = nullptr;  // [dims][n]
    std::mutex mu;
};

template <class Body>
int kd_no_throw(const char* what, Body&& body)
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", what);
        return PCPX_ERR_ALLOC;
    } catch (...) {
        set_error("%s: unexpected host exception", what);
        return PCPX_ERR_INVALID;
    }
}

struct KdDeviceScope {
    int before = -1;
    int use(int device)
    {
        PCPX_HIP(hipGetDevice(&before));
        if (before != device) PCPX_HIP(hipSetDevice(device));
        else before = -1;
        return PCPX_OK;
    }
    ~KdDeviceScope()
    {
        if (before >= 0) (void)hipSetDevice(before);
    }
};

struct KdBuf {
    void* p = nullptr;
    int alloc(size_t bytes)
    {
        PCPX_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return PCPX_OK;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
    ~KdBuf()
    {
        if (p) (void)hipFree(p);
    }
};

__global__ __launch_bounds__(256) void k_kd_transpose(const float* __restrict__ rows, u64 n, u32 dims, float* __restrict__ by_axis)
{
    const u64 i = static_cast<u64>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    for (u32 a = 0; a < dims; ++a) by_axis[static_cast<u64>(a) * n + i] = rows[i * dims + a];
}

__device__ __forceinline__ u64 kd_shfl(u64 v, int src)
{
    const u32 lo = __shfl(static_cast<u32>(v), src), hi = __shfl(static_cast<u32>(v >> 32), src);
    return (static_cast<u64>(hi) << 32) | lo;
}
__device__ __forceinline__ u64 kd_shfl_up1(u64 v)
{
    const u32 lo = __shfl_up(static_cast<u32>(v), 1), hi = __shfl_up(static_cast<u32>(v >> 32), 1);
    return (static_cast<u64>(hi) << 32) | lo;
}

// The wave's `want` smallest keys so far, ascending, one per lane (lanes >= want and unfilled ones hold KD_PAD); `tau` = the
// key of lane want - 1.  Takes one candidate key per lane (KD_PAD: none).
__device__ __forceinline__ void kd_take(u64 key, u64& mine, u64& tau, const u32 want, const u32 lane)
{
    u64 m = __builtin_amdgcn_ballot_w64(key < tau);
    while (m != 0) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const u64 c = kd_shfl(key, b);
        if (c >= tau) continue;  // (tau has shrunk since the ballot)
        const u64 up = kd_shfl_up1(mine);
        const bool gt = mine > c;
        const bool prev_gt = lane > 0 && up > c;
        mine = gt ? (prev_gt ? up : c) : mine;
        tau = kd_shfl(mine, static_cast<int>(want) - 1);
    }
}

// one wavefront per (query, segment): its `want` best keys of the segment (d2 bits << 32 | index), ascending, to `part`
__global__ __launch_bounds__(64) void k_kd_knn_segment(const float* __restrict__ pts, u64 n, u32 dims, const float* __restrict__ queries, u32 nseg,
                                                       u32 seg_points, float eps, const u64* __restrict__ from_key, u32 want,
                                                       u64* __restrict__ part)
{
    const u32 lane = threadIdx.x;
    const u64 q = blockIdx.x / nseg;
    const u32 s = blockIdx.x % nseg;
    const float* qrow = queries + q * dims;
    const u64 s0 = static_cast<u64>(s) * seg_points, s1 = s0 + seg_points < n ? s0 + seg_points : n;
    const u64 from = from_key[q];  // keys below were returned by earlier passes (k > 64)
    u64 mine = KD_PAD, tau = KD_PAD;
    for (u64 base = s0; base < s1; base += 64) {
        const u64 i = base + lane;
        u64 key = KD_PAD;
        if (i < s1) {
            float acc = 0.f;
            bool same = true;
            for (u32 a = 0; a < dims; ++a) {
                const float d = pts[static_cast<u64>(a) * n + i] - qrow[a];
                acc = acc + d * d;
                same = same && fabsf(d) < eps;
            }
            if (!same && acc == acc) key = (static_cast<u64>(__float_as_uint(acc)) << 32) | static_cast<u32>(i);
            if (key < from) key = KD_PAD;
        }
        kd_take(key, mine, tau, want, lane);
    }
    part[static_cast<u64>(blockIdx.x) * 64 + lane] = mine;
}

// one wavefront per query: the `want` best of its segments' keys -> row entries [done, done + want), the count, the next pass's bound
__global__ __launch_bounds__(64) void k_kd_knn_merge(const u64* __restrict__ part, u32 nseg, u32 want, u32 k, u32 done, u32* __restrict__ out_idx,
                                                     float* __restrict__ out_d2, u32* __restrict__ out_cnt, u64* __restrict__ from_key)
{
    const u32 lane = threadIdx.x;
    const u64 q = blockIdx.x;
    u64 mine = KD_PAD, tau = KD_PAD;
    for (u32 s = 0; s < nseg; ++s) kd_take(part[(q * nseg + s) * 64 + lane], mine, tau, want, lane);
    const bool real = lane < want && mine != KD_PAD;
    if (lane < want) {
        out_idx[q * k + done + lane] = real ? static_cast<u32>(mine) : INVALID_ID;
        if (out_d2) out_d2[q * k + done + lane] = real ? __uint_as_float(static_cast<u32>(mine >> 32)) : __builtin_inff();
    }
    const u32 found = static_cast<u32>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(real)));
    const u64 last = kd_shfl(mine, static_cast<int>(want) - 1);  // (every lane takes part in the exchange)
    if (lane == 0) {
        out_cnt[q] = (done ? out_cnt[q] : 0u) + found;
        from_key[q] = found == want ? last + 1ull : KD_PAD;
    }
}

// one wavefront per (box, segment): points with lo <= p <= hi on every axis; FILL: their indices behind the box's cursor
template <bool FILL>
__global__ __launch_bounds__(64) void k_kd_boxes(const float* __restrict__ pts, u64 n, u32 dims, const float* __restrict__ boxes, u32 nseg, u32 seg_points,
                                                 u32* __restrict__ count, unsigned long long* __restrict__ cursor, u32* __restrict__ out_idx)
{
    const u32 lane = threadIdx.x;
    const u64 q = blockIdx.x / nseg;
    const u32 s = blockIdx.x % nseg;
    const float* lo = boxes + q * 2 * dims;
    const float* hi = lo + dims;
    const u64 s0 = static_cast<u64>(s) * seg_points, s1 = s0 + seg_points < n ? s0 + seg_points : n;
    u32 total = 0;
    for (u64 base = s0; base < s1; base += 64) {
        const u64 i = base + lane;
        bool in = i < s1;
        if (in)
            for (u32 a = 0; a < dims; ++a) {
                const float p = pts[static_cast<u64>(a) * n + i];
                in = in && p >= lo[a] && p <= hi[a];
            }
        const u64 m = __builtin_amdgcn_ballot_w64(in);
        if (m == 0) continue;
        const u32 c = static_cast<u32>(__builtin_popcountll(m));
        if (FILL) {
            unsigned long long at = 0;
            if (lane == 0) at = atomicAdd(cursor + q, static_cast<unsigned long long>(c));
            at = kd_shfl(at, 0);
            const u32 r = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(m), 0u));
            if (in) out_idx[at + r] = static_cast<u32>(i);
        }
        total += c;
    }
    if (!FILL && lane == 0 && total) atomicAdd(count + q, total);
}

void kd_segments(u64 n, u64 nq, u32& nseg, u32& seg_points)
{
    u64 want = nq ? (KD_WAVES_WANTED + nq - 1) / nq : 1;
    const u64 most = (n + KD_SEG_MIN - 1) / KD_SEG_MIN;
    if (want > most) want = most;
    if (want < 1) want = 1;
    u64 per = (n + want - 1) / want;
    per = (per + 63) / 64 * 64;
    if (per == 0) per = 64;
    seg_points = static_cast<u32>(per);
    nseg = static_cast<u32>((n + per - 1) / per);
    if (nseg == 0) nseg = 1;
}

}  // namespace
}  // namespace pcpx

using namespace pcpx;

extern "C" {

int pcpx_kd_create(const float* points, uint64_t n, uint32_t dims, int device, pcpx_kd_index** out)
{
    return kd_no_throw("pcpx_kd_create", [&]() -> int {
        if (!out) return PCPX_ERR_INVALID;
        *out = nullptr;
        if (dims < 1 || dims > KD_MAX_DIMS || (n > 0 && !points) || n >= 0xFFFFFFFFull) {
            set_error("pcpx_kd_create: 1 <= dims <= %u, fewer than 2^32 - 1 points", KD_MAX_DIMS);
            return PCPX_ERR_INVALID;
        }
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
            (void)hipGetLastError();
            set_error("pcpx: no HIP device available; libpcpx has no CPU fallback");
            return PCPX_ERR_DEVICE;
        }
        if (device < 0 || device >= count) {
            set_error("pcpx: device %d out of range [0,%d)", device, count);
            return PCPX_ERR_INVALID;
        }
        KdDeviceScope scope;
        int st = scope.use(device);
        if (st != PCPX_OK) return st;
        KdIndex* ix = new KdIndex;
        ix->device = device;
        ix->n = n;
        ix->dims = dims;
        auto fail = [&](int code) {
            if (ix->d_pts) (void)hipFree(ix->d_pts);
            if (ix->stream) (void)hipStreamDestroy(ix->stream);
            delete ix;
            return code;
        };
        if ((st = check_hip(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking), "hipStreamCreate", __FILE__, __LINE__)) != PCPX_OK) return fail(st);
        if ((st = check_hip(hipMalloc(reinterpret_cast<void**>(&ix->d_pts), (n ? n : 1) * dims * sizeof(float)), "hipMalloc", __FILE__, __LINE__)) != PCPX_OK)
            return fail(st);
        if (n > 0) {
            KdBuf rows;
            if ((st = rows.alloc(n * dims * sizeof(float))) != PCPX_OK) return fail(st);
            if ((st = check_hip(hipMemcpyAsync(rows.p, points, n * dims * sizeof(float), hipMemcpyHostToDevice, ix->stream), "hipMemcpyAsync", __FILE__, __LINE__)) != PCPX_OK)
                return fail(st);
            k_kd_transpose<<<static_cast<u32>((n + 255) / 256), 256, 0, ix->stream>>>(rows.as<float>(), n, dims, ix->d_pts);
            if ((st = check_hip(hipGetLastError(), "k_kd_transpose", __FILE__, __LINE__)) != PCPX_OK) return fail(st);
            if ((st = check_hip(hipStreamSynchronize(ix->stream), "hipStreamSynchronize", __FILE__, __LINE__)) != PCPX_OK) return fail(st);
        }
        *out = reinterpret_cast<pcpx_kd_index*>(ix);
        return PCPX_OK;
    });
}

void pcpx_kd_destroy(pcpx_kd_index* h)
{
    KdIndex* ix = reinterpret_cast<KdIndex*>(h);
    if (!ix) return;
    KdDeviceScope scope;
    (void)scope.use(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    if (ix->d_pts) (void)hipFree(ix->d_pts);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

uint64_t pcpx_kd_size(const pcpx_kd_index* h) { return h ? reinterpret_cast<const KdIndex*>(h)->n : 0; }
uint32_t pcpx_kd_dims(const pcpx_kd_index* h) { return h ? reinterpret_cast<const KdIndex*>(h)->dims : 0; }

int pcpx_kd_knn_batch(pcpx_kd_index* h, const float* queries, uint64_t nq, uint32_t k, float eps, uint32_t* out_idx, uint32_t* out_count,
                      float* opt_out_d2)
{
    return kd_no_throw("pcpx_kd_knn_batch", [&]() -> int {
        KdIndex* ix = reinterpret_cast<KdIndex*>(h);
        if (!ix) {
            set_error("pcpx: null index handle");
            return PCPX_ERR_INVALID;
        }
        if (nq > 0 && (!queries || !out_count || (k > 0 && !out_idx))) return PCPX_ERR_INVALID;
        if (nq == 0) return PCPX_OK;
        if (k == 0 || ix->n == 0) {
            for (u64 i = 0; i < nq; ++i) out_count[i] = 0;
            for (u64 i = 0; i < nq * k; ++i) {
                out_idx[i] = INVALID_ID;
                if (opt_out_d2) opt_out_d2[i] = __builtin_inff();
            }
            return PCPX_OK;
        }
        KdDeviceScope scope;
        int st = scope.use(ix->device);
        if (st != PCPX_OK) return st;
        std::lock_guard<std::mutex> serialise(ix->mu);
        if (!(eps > 0.f)) eps = 0.f;  // eps <= 0 or NaN: nothing is "equal"
        u32 nseg = 1, seg_points = 64;
        kd_segments(ix->n, nq, nseg, seg_points);
        if (nq * nseg > 0x7FFFFFFFull) {
            set_error("pcpx_kd_knn_batch: %llu queries at once are too many", static_cast<unsigned long long>(nq));
            return PCPX_ERR_INVALID;
        }
        KdBuf dq, dpart, dfrom, didx, dcnt, dd2;
        if ((st = dq.alloc(nq * ix->dims * sizeof(float))) != PCPX_OK) return st;
        if ((st = dpart.alloc(nq * nseg * 64 * sizeof(u64))) != PCPX_OK) return st;
        if ((st = dfrom.alloc(nq * sizeof(u64))) != PCPX_OK) return st;
        if ((st = didx.alloc(nq * k * sizeof(u32))) != PCPX_OK) return st;
        if ((st = dcnt.alloc(nq * sizeof(u32))) != PCPX_OK) return st;
        if (opt_out_d2 && (st = dd2.alloc(nq * k * sizeof(float))) != PCPX_OK) return st;
        PCPX_HIP(hipMemcpyAsync(dq.p, queries, nq * ix->dims * sizeof(float), hipMemcpyHostToDevice, ix->stream));
        PCPX_HIP(hipMemsetAsync(dfrom.p, 0, nq * sizeof(u64), ix->stream));
        for (u32 done = 0; done < k; done += 64) {  // (64 neighbours a pass: a pass takes the keys above the last one of the pass before)
            const u32 want = k - done < 64u ? k - done : 64u;
            k_kd_knn_segment<<<static_cast<u32>(nq * nseg), 64, 0, ix->stream>>>(ix->d_pts, ix->n, ix->dims, dq.as<float>(), nseg, seg_points, eps, dfrom.as<u64>(), want,
                                                                                    dpart.as<u64>());
            PCPX_HIP(hipGetLastError());
            k_kd_knn_merge<<<static_cast<u32>(nq), 64, 0, ix->stream>>>(dpart.as<u64>(), nseg, want, k, done, didx.as<u32>(), dd2.as<float>(), dcnt.as<u32>(),
                                                                          dfrom.as<u64>());
            PCPX_HIP(hipGetLastError());
        }
        PCPX_HIP(hipMemcpyAsync(out_idx, didx.p, nq * k * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
        PCPX_HIP(hipMemcpyAsync(out_count, dcnt.p, nq * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
        if (opt_out_d2) PCPX_HIP(hipMemcpyAsync(opt_out_d2, dd2.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
        PCPX_HIP(hipStreamSynchronize(ix->stream));
        return PCPX_OK;
    });
}

int pcpx_kd_range_aabb_batch(pcpx_kd_index* h, const float* boxes, uint64_t nb, uint64_t* out_offsets, uint32_t* out_idx, uint64_t idx_capacity)
{
    return kd_no_throw("pcpx_kd_range_aabb_batch", [&]() -> int {
        KdIndex* ix = reinterpret_cast<KdIndex*>(h);
        if (!ix) {
            set_error("pcpx: null index handle");
            return PCPX_ERR_INVALID;
        }
        if (!out_offsets || (nb > 0 && !boxes)) return PCPX_ERR_INVALID;
        out_offsets[0] = 0;
        if (nb == 0) return PCPX_OK;
        if (ix->n == 0) {
            for (u64 i = 0; i <= nb; ++i) out_offsets[i] = 0;
            return PCPX_OK;
        }
        KdDeviceScope scope;
        int st = scope.use(ix->device);
        if (st != PCPX_OK) return st;
        std::lock_guard<std::mutex> serialise(ix->mu);
        u32 nseg = 1, seg_points = 64;
        kd_segments(ix->n, nb, nseg, seg_points);
        if (nb * nseg > 0x7FFFFFFFull) {
            set_error("pcpx_kd_range_aabb_batch: %llu boxes at once are too many", static_cast<unsigned long long>(nb));
            return PCPX_ERR_INVALID;
        }
        KdBuf db, dcnt, dcur, dout;
        if ((st = db.alloc(nb * 2 * ix->dims * sizeof(float))) != PCPX_OK) return st;
        if ((st = dcnt.alloc(nb * sizeof(u32))) != PCPX_OK) return st;
        PCPX_HIP(hipMemcpyAsync(db.p, boxes, nb * 2 * ix->dims * sizeof(float), hipMemcpyHostToDevice, ix->stream));
        PCPX_HIP(hipMemsetAsync(dcnt.p, 0, nb * sizeof(u32), ix->stream));
        k_kd_boxes<false><<<static_cast<u32>(nb * nseg), 64, 0, ix->stream>>>(ix->d_pts, ix->n, ix->dims, db.as<float>(), nseg, seg_points, dcnt.as<u32>(), nullptr, nullptr);
        PCPX_HIP(hipGetLastError());
        std::vector<u32> cnt(nb);
        PCPX_HIP(hipMemcpyAsync(cnt.data(), dcnt.p, nb * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
        PCPX_HIP(hipStreamSynchronize(ix->stream));
        for (u64 i = 0; i < nb; ++i) out_offsets[i + 1] = out_offsets[i] + cnt[i];
        const u64 total = out_offsets[nb];
        if (total == 0) return PCPX_OK;
        if (!out_idx || idx_capacity < total) {
            set_error("pcpx_kd_range_aabb_batch: need room for %llu indices", static_cast<unsigned long long>(total));
            return PCPX_ERR_CAPACITY;
        }
        if ((st = dcur.alloc(nb * sizeof(u64))) != PCPX_OK) return st;
        if ((st = dout.alloc(total * sizeof(u32))) != PCPX_OK) return st;
        PCPX_HIP(hipMemcpyAsync(dcur.p, out_offsets, nb * sizeof(u64), hipMemcpyHostToDevice, ix->stream));
        k_kd_boxes<true><<<static_cast<u32>(nb * nseg), 64, 0, ix->stream>>>(ix->d_pts, ix->n, ix->dims, db.as<float>(), nseg, seg_points, nullptr,
                                                                             dcur.as<unsigned long long>(), dout.as<u32>());
        PCPX_HIP(hipGetLastError());
        PCPX_HIP(hipMemcpyAsync(out_idx, dout.p, total * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
        PCPX_HIP(hipStreamSynchronize(ix->stream));
        return PCPX_OK;
    });
}

}  // extern "C"
