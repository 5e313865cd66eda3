// pcpx_api.hip -- the extern "C" boundary declared in include/pcpx.h.
// Host-pointer entry points stage through device buffers and are synchronous; *_dev entry points
// enqueue on the index's stream.  No CPU fallback exists: every compute call needs a HIP device.
#include "pcpx_internal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <mutex>
#include <new>
#include <vector>

namespace pcpx {

static thread_local std::string g_err;
static const bool g_few_no_poll = std::getenv("PCPX_FEW_NO_POLL") != nullptr;  // diagnostic: wait on the stream instead

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int check_hip(hipError_t e, const char* what, const char* file, int line)
{
    if (e == hipSuccess) return PCPX_OK;
    set_error("HIP error %d (%s) at %s:%d in %s", static_cast<int>(e), hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();
    return (e == hipErrorOutOfMemory) ? PCPX_ERR_ALLOC : PCPX_ERR_DEVICE;
}

// ---- DevPool / PinnedStage (pcpx_internal.h) --------------------------------------------------------
void* DevPool::acquire(size_t bytes)
{
    if (bytes == 0) bytes = 16;
    int best = -1;
    for (size_t i = 0; i < blocks.size(); ++i)  // best fit among the free blocks that are not wastefully large
        if (!blocks[i].used && blocks[i].bytes >= bytes && blocks[i].bytes <= 2 * bytes + (1u << 20) &&
            (best < 0 || blocks[i].bytes < blocks[static_cast<size_t>(best)].bytes))
            best = static_cast<int>(i);
    if (best >= 0) {
        blocks[static_cast<size_t>(best)].used = true;
        return blocks[static_cast<size_t>(best)].p;
    }
    void* p = nullptr;
    const size_t rounded = (bytes + 4095) / 4096 * 4096;
    hipError_t e = hipMalloc(&p, rounded);
    if (e != hipSuccess) {  // give back what is cached and try once more
        (void)hipGetLastError();
        trim();
        e = hipMalloc(&p, rounded);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", rounded, hipGetErrorString(e));
        return nullptr;
    }
    blocks.push_back(Block{p, rounded, true});
    return p;
}
void DevPool::release(void* p)
{
    for (auto& b : blocks)
        if (b.p == p) {
            b.used = false;
            break;
        }
    // a cap on what sits idle: calls with differing sizes each leave a block behind (a block is reused only for a request of
    // at least half its size); beyond MAX_IDLE the largest idle blocks go back to the driver
    constexpr size_t MAX_IDLE = size_t(16) << 30;  // (of 288 GB)
    for (;;) {
        size_t idle = 0;
        int largest = -1;
        for (size_t i = 0; i < blocks.size(); ++i)
            if (!blocks[i].used) {
                idle += blocks[i].bytes;
                if (largest < 0 || blocks[i].bytes > blocks[static_cast<size_t>(largest)].bytes) largest = static_cast<int>(i);
            }
        if (idle <= MAX_IDLE || largest < 0) break;
        (void)hipFree(blocks[static_cast<size_t>(largest)].p);
        blocks.erase(blocks.begin() + largest);
    }
}
void DevPool::trim()
{
    size_t keep = 0;
    for (size_t i = 0; i < blocks.size(); ++i) {
        if (blocks[i].used) blocks[keep++] = blocks[i];
        else (void)hipFree(blocks[i].p);
    }
    blocks.resize(keep);
}
size_t DevPool::cached_bytes() const
{
    size_t t = 0;
    for (auto const& b : blocks) t += b.bytes;
    return t;
}
DevPool::~DevPool()
{
    for (auto& b : blocks) (void)hipFree(b.p);
}
// ---- device blocks of indexes, cached per device (pcpx_internal.h) ----------------------------------------------------------
namespace {
struct IndexBlocks {
    std::mutex mu;
    struct Idle {
        void* p;
        size_t bytes;
        int device;
    };
    std::vector<Idle> idle;
    std::unordered_map<void*, size_t> handed_out;  // block -> its real size
    size_t cap_bytes()
    {
        static const size_t cap = [] {
            const char* e = std::getenv("PCPX_DEVICE_CACHE_MB");
            const long long mb = e ? std::atoll(e) : 2048;
            return static_cast<size_t>(mb < 0 ? 0 : mb) << 20;
        }();
        return cap;
    }
};
IndexBlocks& index_blocks()
{
    static IndexBlocks* b = new IndexBlocks();  // (never destroyed: the HIP runtime may be gone by the time statics are)
    return *b;
}
}  // namespace

hipError_t index_block_alloc(void** p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    int dev = 0;
    (void)hipGetDevice(&dev);
    IndexBlocks& c = index_blocks();
    std::lock_guard<std::mutex> lock(c.mu);
    int best = -1;
    for (size_t i = 0; i < c.idle.size(); ++i)
        if (c.idle[i].device == dev && c.idle[i].bytes >= bytes && c.idle[i].bytes <= bytes + bytes / 4 + (size_t(64) << 10) &&
            (best < 0 || c.idle[i].bytes < c.idle[static_cast<size_t>(best)].bytes))
            best = static_cast<int>(i);
    if (best >= 0) {
        *p = c.idle[static_cast<size_t>(best)].p;
        c.handed_out[*p] = c.idle[static_cast<size_t>(best)].bytes;
        c.idle.erase(c.idle.begin() + best);
        return hipSuccess;
    }
    const size_t rounded = (bytes + 4095) / 4096 * 4096;
    hipError_t e = hipMalloc(p, rounded);
    if (e == hipErrorOutOfMemory) {  // what sits idle here may be what is missing
        (void)hipGetLastError();
        size_t keep = 0;
        for (size_t i = 0; i < c.idle.size(); ++i) {
            if (c.idle[i].device == dev) (void)hipFree(c.idle[i].p);
            else c.idle[keep++] = c.idle[i];
        }
        c.idle.resize(keep);
        e = hipMalloc(p, rounded);
    }
    if (e == hipSuccess) c.handed_out[*p] = rounded;
    else *p = nullptr;
    return e;
}

void index_block_free(void* p)
{
    if (!p) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    IndexBlocks& c = index_blocks();
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.handed_out.find(p);
    if (it == c.handed_out.end()) {  // not one of ours (cannot happen; be safe)
        (void)hipFree(p);
        return;
    }
    const size_t bytes = it->second;
    c.handed_out.erase(it);
    size_t idle_here = 0;
    for (auto const& b : c.idle)
        if (b.device == dev) idle_here += b.bytes;
    bool keep = idle_here + bytes <= c.cap_bytes();
    if (keep && idle_here + bytes > (size_t(256) << 20)) {
        // beyond a quarter of a gigabyte the cache also yields to whoever else lives on the device (a co-resident framework sees idle
        // blocks as used memory): never more than a quarter of what is free now
        size_t free_now = 0, total = 0;
        if (hipMemGetInfo(&free_now, &total) == hipSuccess) keep = idle_here + bytes <= free_now / 4;
        else (void)hipGetLastError();
    }
    if (keep) c.idle.push_back(IndexBlocks::Idle{p, bytes, dev});
    else (void)hipFree(p);
}

void index_blocks_trim()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    IndexBlocks& c = index_blocks();
    std::lock_guard<std::mutex> lock(c.mu);
    size_t keep = 0;
    for (size_t i = 0; i < c.idle.size(); ++i) {
        if (c.idle[i].device == dev) (void)hipFree(c.idle[i].p);
        else c.idle[keep++] = c.idle[i];
    }
    c.idle.resize(keep);
}

namespace {
struct IdleStreams {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;  // (device, stream)
};
IdleStreams& idle_streams()
{
    static IdleStreams* s = new IdleStreams();  // (never destroyed: the HIP runtime may be gone by the time statics are)
    return *s;
}
}  // namespace

hipError_t pooled_stream_get(hipStream_t* out)
{
    static const bool no_pool = std::getenv("PCPX_NO_STREAM_POOL") != nullptr;  // (diagnostic)
    if (no_pool) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        IdleStreams& p = idle_streams();
        std::lock_guard<std::mutex> lock(p.mu);
        for (size_t i = 0; i < p.idle.size(); ++i)
            if (p.idle[i].first == dev) {
                *out = p.idle[i].second;
                p.idle.erase(p.idle.begin() + static_cast<long>(i));
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void pooled_stream_put(hipStream_t s)
{
    if (!s) return;
    if (std::getenv("PCPX_NO_STREAM_POOL") != nullptr) {
        (void)hipStreamDestroy(s);
        return;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        IdleStreams& p = idle_streams();
        std::lock_guard<std::mutex> lock(p.mu);
        size_t here = 0;
        for (auto const& e : p.idle) here += e.first == dev ? 1u : 0u;
        if (here < 8) {
            p.idle.emplace_back(dev, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

int PinnedStage::ensure(size_t need)
{
    if (need <= bytes) return PCPX_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    const size_t want = need < (1u << 16) ? (1u << 16) : (need + 4095) / 4096 * 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();
        set_error("hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return PCPX_ERR_ALLOC;
    }
    bytes = want;
    return PCPX_OK;
}
PinnedStage::~PinnedStage()
{
    if (p) (void)hipHostFree(p);
}

namespace {

// RAII device buffer for the host-pointer entry points: a block of the owner's pool (no hipMalloc / hipFree per call);
// returned to the pool on scope exit -- every such function synchronises its stream before it returns
struct DevBuf {
    void* p = nullptr;
    DevPool* pool = nullptr;
    bool one_off = false;  // true: not a block of the handle's pool (the n x 12-byte staging copy of a build from host memory
                           // would stay cached for the handle's lifetime) but one of the device's index blocks: it serves the
                           // next index built on this device, or goes back to the driver (index_block_alloc, pcpx_internal.h)
    explicit DevBuf(DevPool& owner, bool one_off_ = false) : pool(&owner), one_off(one_off_) {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { reset(); }
    void reset()
    {
        if (!p) return;
        if (one_off) index_block_free(p);
        else pool->release(p);
        p = nullptr;
    }
    int alloc(size_t bytes)
    {
        if (one_off) {
            const hipError_t e = index_block_alloc(&p, bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
                p = nullptr;
            }
        } else {
            p = pool->acquire(bytes);
        }
        return p ? PCPX_OK : PCPX_ERR_ALLOC;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// entry points without a handle (bounding box of a host array, one normal, the sort diagnostic) share one pool and one
// pinned stage per device; they hold its mutex for their duration
struct DeviceShared {
    std::mutex mu;
    DevPool pool;
    PinnedStage pinned;
    u32 normal_epoch = 0;  // launch counter of pcpx_estimate_normal's polled form
};
DeviceShared& shared_of(int device)
{
    static std::mutex table_mu;
    static std::vector<DeviceShared*> table;  // never freed: the HIP runtime may be gone at exit
    std::lock_guard<std::mutex> lock(table_mu);
    if (static_cast<size_t>(device) >= table.size()) table.resize(static_cast<size_t>(device) + 1, nullptr);
    if (!table[static_cast<size_t>(device)]) table[static_cast<size_t>(device)] = new DeviceShared();
    return *table[static_cast<size_t>(device)];
}

// Pageable host memory -> device on `stream`; the source may be reused when this returns (64 KB and more: the bytes have arrived).  hipMemcpy from a pageable buffer it has not seen before took 12-24 ms for 12 MB on this stack (0.5-1 GB/s: the 2^20-point
// cloud of a construction; 200 MB take 4-7 ms), against 0.46 ms for a host memcpy of 12 MB plus 0.29 ms for the same copy from pinned
// memory (tools/h2d_probe.hip, PCPX_TRACE_CREATE).  So copies of up to 64 MB go through a per-device ring of two pinned 4-MB blocks --
// the host fills one while the other is on its way -- and larger ones are left to the runtime.
namespace {
struct Uploader {
    std::mutex mu;
    void* pin = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    static constexpr size_t CHUNK = size_t(4) << 20;
};
Uploader& uploader_of(int device)
{
    static std::mutex table_mu;
    static std::vector<Uploader*> table;  // never freed: the HIP runtime may be gone at exit
    std::lock_guard<std::mutex> lock(table_mu);
    if (static_cast<size_t>(device) >= table.size()) table.resize(static_cast<size_t>(device) + 1, nullptr);
    if (!table[static_cast<size_t>(device)]) table[static_cast<size_t>(device)] = new Uploader();
    return *table[static_cast<size_t>(device)];
}
}  // namespace

int upload_pageable(void* d_dst, const void* src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return PCPX_OK;
    if (bytes < (size_t(64) << 10)) {  // (a query or a few hundred: the runtime has taken its copy of a pageable source when this returns)
        PCPX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream));
        return PCPX_OK;
    }
    static const bool no_ring = std::getenv("PCPX_NO_UPLOAD_RING") != nullptr;  // (diagnostic: leave every copy to the runtime)
    if (no_ring || bytes > (size_t(64) << 20)) {
        PCPX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream));
        return check_hip(hipStreamSynchronize(stream), "upload", __FILE__, __LINE__);
    }
    int dev = 0;
    PCPX_HIP(hipGetDevice(&dev));
    Uploader& u = uploader_of(dev);
    std::lock_guard<std::mutex> lock(u.mu);
    if (!u.ev[1]) {  // (first use, or an earlier attempt that got part of the way)
        if (!u.pin) PCPX_HIP(hipHostMalloc(&u.pin, 2 * Uploader::CHUNK, hipHostMallocDefault));
        if (!u.ev[0]) PCPX_HIP(hipEventCreateWithFlags(&u.ev[0], hipEventDisableTiming));
        PCPX_HIP(hipEventCreateWithFlags(&u.ev[1], hipEventDisableTiming));
    }
    const char* from = static_cast<const char*>(src);
    char* to = static_cast<char*>(d_dst);
    hipError_t e = hipSuccess;
    size_t piece = 0;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += Uploader::CHUNK, ++piece) {
        const size_t len = bytes - off < Uploader::CHUNK ? bytes - off : Uploader::CHUNK;
        const int b = static_cast<int>(piece & 1);
        char* stage = static_cast<char*>(u.pin) + static_cast<size_t>(b) * Uploader::CHUNK;
        if (piece >= 2 && (e = hipEventSynchronize(u.ev[b])) != hipSuccess) break;  // the copy that last read this block has finished
        std::memcpy(stage, from + off, len);
        if ((e = hipMemcpyAsync(to + off, stage, len, hipMemcpyHostToDevice, stream)) != hipSuccess) break;
        e = hipEventRecord(u.ev[b], stream);
    }
    // whatever happened, the stream is drained before the ring is anyone else's (copies queued before a failure still read it)
    const hipError_t drained = hipStreamSynchronize(stream);
    return check_hip(e != hipSuccess ? e : drained, "upload", __FILE__, __LINE__);
}

int select_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("pcpx: no HIP device available (%s); libpcpx has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return PCPX_ERR_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("pcpx: device %d out of range [0,%d)", device, count);
        return PCPX_ERR_INVALID;
    }
    PCPX_HIP(hipSetDevice(device));
    return PCPX_OK;
}

// Every entry point makes its device current for its own duration only: the caller's thread gets its previous
// current device back on return (a torch caller on cuda:1 must not find itself on cuda:0 after a pcpx call).
struct DeviceScope {
    int prev = -1;
    void remember()
    {
        if (hipGetDevice(&prev) != hipSuccess) {
            prev = -1;
            (void)hipGetLastError();
        }
    }
    int select(int device)
    {
        remember();
        int st = select_device(device);
        if (prev == device) prev = -1;  // nothing to restore
        return st;
    }
    int use(Index* ix)
    {
        if (!ix) {
            set_error("pcpx: null index handle");
            return PCPX_ERR_INVALID;
        }
        remember();
        if (prev == ix->device) {
            prev = -1;
            return PCPX_OK;
        }
        PCPX_HIP(hipSetDevice(ix->device));
        return PCPX_OK;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// extern "C" entry points that allocate host memory run their body through this: nothing may be thrown across the ABI
template <class Body>
int no_throw(const char* what, Body&& body)
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

// pcpx_build_params as this library knows it, from what the caller passed: ABI 3 callers pass the first 32 bytes
constexpr size_t BUILD_PARAMS_ABI3 = 32, BUILD_PARAMS_ABI4 = 48;  // (ABI 4 ends before shard_first / shard_count)
int normalise_params(const pcpx_build_params* in, bool device_form, pcpx_build_params& out, const pcpx_build_params*& use)
{
    use = nullptr;
    if (!in) return PCPX_OK;
    static_assert(sizeof(pcpx_build_params) == 64, "pcpx_build_params is part of the ABI");
    if (in->struct_size != sizeof(pcpx_build_params) && in->struct_size != BUILD_PARAMS_ABI3 && in->struct_size != BUILD_PARAMS_ABI4) {
        set_error("pcpx: params->struct_size mismatch");
        return PCPX_ERR_INVALID;
    }
    std::memset(&out, 0, sizeof(out));
    std::memcpy(&out, in, in->struct_size);
    out.struct_size = sizeof(pcpx_build_params);
    if (in->struct_size == BUILD_PARAMS_ABI3) out.flags &= (PCPX_BUILD_USE_GRID | PCPX_BUILD_COARSE_ORDER);
    if (in->struct_size == BUILD_PARAMS_ABI4) out.flags &= ~PCPX_BUILD_SHARD_RANGE;
    if ((out.flags & PCPX_BUILD_SHARD_RANGE) && (!(out.flags & PCPX_BUILD_SHARD) || out.shard_first % GROUP != 0)) {
        set_error("pcpx: PCPX_BUILD_SHARD_RANGE goes with PCPX_BUILD_SHARD, and shard_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    if ((out.flags & PCPX_BUILD_BORROW_CLOUD) && !device_form) {
        set_error("pcpx: PCPX_BUILD_BORROW_CLOUD needs a device-pointer build (the host-pointer forms stage the cloud in a temporary)");
        return PCPX_ERR_INVALID;
    }
    if ((out.flags & PCPX_BUILD_BORROW_CLOUD) && !(out.flags & PCPX_BUILD_SHARD)) {
        set_error("pcpx: PCPX_BUILD_BORROW_CLOUD is a property of rank-local builds (PCPX_BUILD_SHARD)");
        return PCPX_ERR_INVALID;
    }
    use = &out;
    return PCPX_OK;
}

#define PCPX_WHOLE_CLOUD_ONLY(ix, name)                              \
    do {                                                             \
        if ((ix)->shard.on) return shard_unsupported(*(ix), name);   \
    } while (0)

void slice_to_groups(const Index& ix, u64 sorted_first, u64 sorted_count, u64& gfirst, u64& gcount)
{
    u64 n = ix.n;
    if (sorted_first > n) sorted_first = n;
    u64 end = (sorted_count > n - sorted_first) ? n : sorted_first + sorted_count;
    gfirst = sorted_first / GROUP;
    u64 gend = (end + GROUP - 1) / GROUP;
    gcount = gend > gfirst ? gend - gfirst : 0;
}

void free_index(Index* ix)
{
    if (!ix) return;
    DeviceScope dscope;
    (void)dscope.use(ix);
    (void)hipStreamSynchronize(ix->stream);  // (nullptr = the legacy default stream)
    for (auto& iv : ix->intervals) {
        (void)hipEventDestroy(iv.a);
        (void)hipEventDestroy(iv.b);
    }
    // (the stream is drained: the blocks may serve the next index of this device -- index_block_free)
    index_block_free(ix->d_xyz);
    for (int b = 0; b < 2; ++b) index_block_free(ix->d_codes[b]);
    index_block_free(ix->d_perm);
    index_block_free(ix->d_rec);
    index_block_free(ix->d_sort_tmp);
    index_block_free(ix->d_leaves);
    index_block_free(ix->d_nodes);
    index_block_free(ix->d_scalars);
    index_block_free(ix->d_scratch);
    index_block_free(ix->d_nc4);
    index_block_free(ix->d_pos_of);
    index_block_free(ix->sched.d_gtime);
    index_block_free(ix->sched.d_order);
    (void)hipFree(ix->d_queue);
    (void)hipFree(ix->d_multi);
    free_shard(*ix);
    if (ix->copy_stream) {
        (void)hipStreamSynchronize(ix->copy_stream);
        pooled_stream_put(ix->copy_stream);
    }
    if (ix->own_stream && ix->stream) pooled_stream_put(ix->stream);  // (synchronised above)
    delete ix;  // (the pool and the pinned stage free their memory in their destructors, while the device is still current)
}

int exclusive_scan_host(const std::vector<u32>& cnt, u64* offsets)
{
    u64 acc = 0;
    for (size_t i = 0; i < cnt.size(); ++i) {
        offsets[i] = acc;
        acc += cnt[i];
    }
    offsets[cnt.size()] = acc;
    return PCPX_OK;
}

}  // namespace
}  // namespace pcpx

using namespace pcpx;

extern "C" {

int pcpx_abi_version(void) { return PCPX_ABI_VERSION; }
const char* pcpx_last_error(void) { return g_err.c_str(); }

int pcpx_device_count(int* out_count)
{
    if (!out_count) return PCPX_ERR_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *out_count = c;
    return PCPX_OK;
}

static int create_common(const float* xyz, bool on_device, u64 n, const pcpx_build_params* params, int device,
                         void* stream, pcpx_index** out)
{
    if (!out || (n > 0 && !xyz)) {
        set_error("pcpx_index_create: null argument");
        return PCPX_ERR_INVALID;
    }
    *out = nullptr;
    pcpx_build_params full;
    int st = normalise_params(params, on_device, full, params);
    if (st != PCPX_OK) return st;
    DeviceScope dscope;
    st = dscope.select(device);
    if (st != PCPX_OK) return st;
    // PCPX_TRACE_CREATE=1 in the environment: the phases of a creation from host memory on stderr (where a construction's time goes)
    static const bool trace = std::getenv("PCPX_TRACE_CREATE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    Index* ix = new (std::nothrow) Index();
    if (!ix) return PCPX_ERR_ALLOC;
    ix->device = device;
    const double t_new = since();
    if (on_device) {
        // device-pointer form: work is enqueued on the CALLER's stream; NULL is the legacy default stream, which is
        // ordered against the caller's other default-stream work (a private stream would not be)
        ix->stream = static_cast<hipStream_t>(stream);
    } else {
        // host-pointer form: synchronous calls, so a private stream (no ordering against the caller's streams needed)
        hipError_t e = pooled_stream_get(&ix->stream);
        if (e != hipSuccess) {
            delete ix;
            return check_hip(e, "hipStreamCreate", __FILE__, __LINE__);
        }
        ix->own_stream = true;
    }
    {
        // (inner scope: the staging block belongs to the handle's pool and must be gone before a failure path deletes the handle)
        DevBuf staged(ix->pool, true);
        const float* d_src = xyz;
        const double t_stream = since();
        double t_alloc = t_stream, t_copy = t_stream;
        if (!on_device && n > 0) {
            st = staged.alloc(n * 3 * sizeof(float));
            t_alloc = since();
            if (st == PCPX_OK) st = upload_pageable(staged.p, xyz, n * 3 * sizeof(float), ix->stream);
            t_copy = since();
            d_src = staged.as<float>();
        }
        if (st == PCPX_OK) st = build_index(*ix, d_src, n, params);
        const double t_build = since();
        const std::string why = g_err;
        // (also on failure: work of a partial build may still be reading the staging block)
        const int sync = check_hip(hipStreamSynchronize(ix->stream), "build sync", __FILE__, __LINE__);
        if (trace)
            std::fprintf(stderr, "pcpx_index_create n=%llu: handle %.3f ms, stream %.3f, staging block %.3f, H2D %.3f, build enqueue (+ its allocations) %.3f, sync %.3f\n",
                         static_cast<unsigned long long>(n), t_new, t_stream - t_new, t_alloc - t_stream, t_copy - t_alloc, t_build - t_copy, since() - t_build);
        if (st == PCPX_OK) st = sync;
        else g_err = why;
    }
    if (st != PCPX_OK) {
        const std::string why = g_err;  // (free_index's own HIP calls must not replace the reason)
        free_index(ix);
        g_err = why;
        return st;
    }
    *out = reinterpret_cast<pcpx_index*>(ix);
    return PCPX_OK;
}

int pcpx_index_create(const float* xyz, uint64_t n, const pcpx_build_params* params, int device, pcpx_index** out)
{
    return create_common(xyz, false, n, params, device, nullptr, out);
}
int pcpx_index_create_dev(const float* d_xyz, uint64_t n, const pcpx_build_params* params, int device, void* stream,
                          pcpx_index** out)
{
    return create_common(d_xyz, true, n, params, device, stream, out);
}

int pcpx_index_rebuild(pcpx_index* h, const float* xyz, uint64_t n, const pcpx_build_params* params)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (n > 0 && !xyz) return PCPX_ERR_INVALID;
    pcpx_build_params full;
    if ((st = normalise_params(params, false, full, params)) != PCPX_OK) return st;
    DevBuf staged(ix->pool, true);
    if (n > 0) {
        if ((st = staged.alloc(n * 3 * sizeof(float))) != PCPX_OK) return st;
        if ((st = upload_pageable(staged.p, xyz, n * 3 * sizeof(float), ix->stream)) != PCPX_OK) return st;
    }
    st = build_index(*ix, staged.as<float>(), n, params);
    const std::string why = g_err;
    const int sync = check_hip(hipStreamSynchronize(ix->stream), "build sync", __FILE__, __LINE__);  // (before the staging block goes)
    if (st != PCPX_OK) g_err = why;
    return st != PCPX_OK ? st : sync;
}
int pcpx_index_rebuild_dev(pcpx_index* h, const float* d_xyz, uint64_t n, const pcpx_build_params* params)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (n > 0 && !d_xyz) return PCPX_ERR_INVALID;
    pcpx_build_params full;
    if ((st = normalise_params(params, true, full, params)) != PCPX_OK) return st;
    return build_index(*ix, d_xyz, n, params);
}

void pcpx_index_destroy(pcpx_index* h) { free_index(reinterpret_cast<Index*>(h)); }

int pcpx_index_size(pcpx_index* h, uint64_t* out_n)
{
    if (!h || !out_n) return PCPX_ERR_INVALID;
    const Index* ix = reinterpret_cast<Index*>(h);
    *out_n = ix->shard.on ? ix->shard.n_glob : ix->n;  // (a rank-local handle: the whole cloud's inserted points)
    return PCPX_OK;
}
int pcpx_index_shard_info(pcpx_index* h, uint64_t out[8])
{
    const Index* ix = reinterpret_cast<Index*>(h);
    if (!ix || !out || !ix->shard.on) {
        set_error("pcpx_index_shard_info: not a rank-local index");
        return PCPX_ERR_INVALID;
    }
    const Index::Shard& sh = ix->shard;
    out[0] = ix->n;
    out[1] = sh.core_g0;
    out[2] = sh.core_count;
    out[3] = sh.g_first;
    out[4] = sh.g_count;
    out[5] = sh.everything ? 64 : sh.halo_cells;
    out[6] = sh.last_failed;
    out[7] = sh.enlargements;
    return PCPX_OK;
}
int pcpx_index_bbox(pcpx_index* h, float out6[6])
{
    if (!h || !out6) return PCPX_ERR_INVALID;
    std::memcpy(out6, reinterpret_cast<Index*>(h)->bbox, 6 * sizeof(float));
    return PCPX_OK;
}
int pcpx_index_trim(pcpx_index* h)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    ix->pool.trim();
    return PCPX_OK;
}
int pcpx_index_synchronize(pcpx_index* h)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_bounding_box_dev(const float* d_xyz, uint64_t n, int device, void* stream, float* d_out6)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (!d_out6 || (n > 0 && !d_xyz)) return PCPX_ERR_INVALID;
    // d_out6 must have room for the 6 floats; the encoded scratch is a temporary
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    DevBuf enc(shared.pool);
    if ((st = enc.alloc(64 * sizeof(u32))) != PCPX_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    st = device_bbox(d_xyz, n, s, enc.as<u32>(), d_out6);
    if (st != PCPX_OK) return st;
    PCPX_HIP(hipStreamSynchronize(s));  // enc goes back to the pool on return
    return PCPX_OK;
}
int pcpx_bounding_box(const float* xyz, uint64_t n, int device, float out6[6])
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (!out6 || (n > 0 && !xyz)) return PCPX_ERR_INVALID;
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    DevBuf pts(shared.pool), box(shared.pool);
    if ((st = pts.alloc(n * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = box.alloc(64 * sizeof(u32))) != PCPX_OK) return st;
    if (n > 0) PCPX_HIP(hipMemcpy(pts.p, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    float* d_out = box.as<float>() + 32;
    if ((st = device_bbox(pts.as<float>(), n, nullptr, box.as<u32>(), d_out)) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpy(out6, d_out, 6 * sizeof(float), hipMemcpyDeviceToHost));
    return PCPX_OK;
}

// ---- kNN -----------------------------------------------------------------------------------------
int pcpx_knn_self_dev(pcpx_index* h, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                      uint32_t* d_out_idx, uint32_t* d_out_count, float* d_out_d2)
{
    return pcpx_knn_self_strided_dev(h, k, eps, sorted_first, sorted_count, 0, d_out_idx, d_out_count, d_out_d2);
}

int pcpx_knn_self_strided_dev(pcpx_index* h, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count, uint32_t row_stride,
                              uint32_t* d_out_idx, uint32_t* d_out_count, float* d_out_d2)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (k == 0) return PCPX_OK;  // linked_octree_node.hpp:464: k == 0 -> {}
    if (!d_out_idx || !d_out_count) return PCPX_ERR_INVALID;
    if (sorted_first % GROUP != 0) {
        set_error("pcpx_knn_self_dev: sorted_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    KnnOutputs o;
    o.idx = d_out_idx;
    o.cnt = d_out_count;
    o.d2 = d_out_d2;
    if (row_stride != 0 && (row_stride < k || k > 32)) {
        set_error("pcpx_knn_self_strided_dev: row_stride must be 0 or >= k, and k <= 32");
        return PCPX_ERR_INVALID;
    }
    o.row_stride = row_stride;
    if (ix->shard.on) return shard_knn_self(*ix, sorted_first, sorted_count, k, eps, o);
    u64 gf, gc;
    slice_to_groups(*ix, sorted_first, sorted_count, gf, gc);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    return launch_knn(*ix, qv, true, gf, gc, k, eps, o);
}

int pcpx_knn_self_curve_order_dev(pcpx_index* h, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count, uint32_t* d_out_idx,
                                  uint32_t* d_out_count, float* d_opt_d2, float* d_opt_normals)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    if (k == 0) return PCPX_OK;
    if (!d_out_idx || !d_out_count) return PCPX_ERR_INVALID;
    if (sorted_first % GROUP != 0) {
        set_error("pcpx_knn_self_curve_order_dev: sorted_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    KnnOutputs o;
    o.idx = d_out_idx;
    o.cnt = d_out_count;
    o.d2 = d_opt_d2;
    o.normals = d_opt_normals;
    o.by_position = 1;
    if (k > 32 && o.normals) {
        set_error("pcpx_knn_self_curve_order_dev: fused normals need k <= 32");
        return PCPX_ERR_UNSUPPORTED;
    }
    if (ix->shard.on) return shard_knn_self(*ix, sorted_first, sorted_count, k, eps, o);
    u64 gf, gc;
    slice_to_groups(*ix, sorted_first, sorted_count, gf, gc);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    return launch_knn(*ix, qv, true, gf, gc, k, eps, o);
}

int pcpx_index_perm_dev(pcpx_index* h, uint32_t* d_out_perm, uint32_t* d_opt_out_position_of)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    if (!d_out_perm && !d_opt_out_position_of) return PCPX_ERR_INVALID;
    if (d_opt_out_position_of) PCPX_HIP(hipMemsetAsync(d_opt_out_position_of, 0xFF, ix->n_in * sizeof(u32), ix->stream));
    if (ix->shard.on) return shard_perm(*ix, d_out_perm, d_opt_out_position_of);
    if (ix->n == 0) return PCPX_OK;
    if (d_out_perm) PCPX_HIP(hipMemcpyAsync(d_out_perm, ix->d_perm, ix->n * sizeof(u32), hipMemcpyDeviceToDevice, ix->stream));
    if (d_opt_out_position_of) return launch_invert_perm(ix->d_perm, ix->n, d_opt_out_position_of, ix->stream);
    return PCPX_OK;
}

// Self queries with HOST outputs (pcpx_knn_self, pcpx_normals_knn_self): device staging from the handle's pool (no
// hipMalloc / hipFree per call), one fused launch, one copy per output.  Rows are n x k x 4 bytes: at 10 M points and
// k = 15 the copy to the host (760 MB, ~13.6 ms at this box's 56 GB/s) outweighs the kernel (~5 ms).  The two cannot
// overlap: the kernel walks the curve order and writes row i = input point i, i.e. it scatters over the whole output,
// so no part of an output array is final before the launch ends.  (Tried and measured, profiles/experiments/README.md:
// chunks of the INPUT order through the batch-query form, each chunk's copy overlapping the next chunk's kernels --
// a chunk's queries are ten times sparser than the cloud, a wave's 64 queries then share little of their search
// regions, and the kernels alone took 25.7 ms against 5.3 ms.)
constexpr u64 FEW_QUERIES_MAX = 512;  // up to here pcpx_knn_batch takes the latency path

// any of out_normals / out_idx / out_d2 may be null (out_cnt is required with out_idx)
static int self_queries_to_host(Index* ix, u32 k, float eps, float* out_normals, u32* out_idx, u32* out_cnt, float* out_d2)
{
    int st;
    const u64 rows = ix->n_in;
    if (rows == 0) return PCPX_OK;
    const bool want_rows = out_idx != nullptr;
    DevBuf dn(ix->pool), di(ix->pool), dc(ix->pool), dd(ix->pool);
    if (out_normals && (st = dn.alloc(rows * 3 * sizeof(float))) != PCPX_OK) return st;
    if (want_rows && (st = di.alloc(rows * k * sizeof(u32))) != PCPX_OK) return st;
    if ((want_rows || out_cnt) && (st = dc.alloc(rows * sizeof(u32))) != PCPX_OK) return st;
    if (out_d2 && (st = dd.alloc(rows * k * sizeof(float))) != PCPX_OK) return st;
    if (ix->n != ix->n_in) {  // rows of dropped (out-of-grid) points: count 0, padding
        if (dn.p) PCPX_HIP(hipMemsetAsync(dn.p, 0, rows * 3 * sizeof(float), ix->stream));
        if (di.p) PCPX_HIP(hipMemsetAsync(di.p, 0xFF, rows * k * sizeof(u32), ix->stream));
        if (dc.p) PCPX_HIP(hipMemsetAsync(dc.p, 0, rows * sizeof(u32), ix->stream));
        if (dd.p) PCPX_HIP(hipMemsetAsync(dd.p, 0x7F, rows * k * sizeof(float), ix->stream));
    }
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    KnnOutputs o;
    o.idx = di.as<u32>();
    o.cnt = dc.as<u32>();
    o.d2 = dd.as<float>();
    o.normals = dn.as<float>();
    DevBuf srows(ix->pool);
    if (k > 32 && out_normals && !want_rows) {  // the multi-pass path builds normals from materialised rows
        if ((st = srows.alloc(rows * k * sizeof(u32))) != PCPX_OK) return st;
        o.idx = srows.as<u32>();
        if (!dc.p && (st = dc.alloc(rows * sizeof(u32))) != PCPX_OK) return st;
        o.cnt = dc.as<u32>();
    }
    if ((st = launch_knn(*ix, qv, true, 0, (ix->n + GROUP - 1) / GROUP, k, eps, o)) != PCPX_OK) return st;
    if (out_normals) PCPX_HIP(hipMemcpyAsync(out_normals, dn.p, rows * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    if (want_rows) PCPX_HIP(hipMemcpyAsync(out_idx, di.p, rows * k * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    if (out_cnt) PCPX_HIP(hipMemcpyAsync(out_cnt, dc.p, rows * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    if (out_d2) PCPX_HIP(hipMemcpyAsync(out_d2, dd.p, rows * k * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_knn_self(pcpx_index* h, uint32_t k, float eps, uint32_t* out_idx, uint32_t* out_count, float* out_d2)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_knn_self");
    if (!out_count || (k > 0 && !out_idx)) return PCPX_ERR_INVALID;
    if (k == 0) {  // linked_octree_node.hpp:464: k == 0 -> {}
        std::memset(out_count, 0, ix->n_in * sizeof(u32));
        return PCPX_OK;
    }
    return self_queries_to_host(ix, k, eps, nullptr, out_idx, out_count, out_d2);
}

int pcpx_knn_batch_dev(pcpx_index* h, const float* d_q_xyz, uint64_t nq, uint32_t k, float eps, uint32_t* d_out_idx,
                       uint32_t* d_out_count, float* d_out_d2)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_knn_batch_dev");
    if (k == 0 || nq == 0) return PCPX_OK;
    if (!d_q_xyz || !d_out_idx || !d_out_count) return PCPX_ERR_INVALID;
    QueryView qv;
    if ((st = prepare_queries(*ix, d_q_xyz, nq, qv)) != PCPX_OK) return st;
    KnnOutputs o;
    o.idx = d_out_idx;
    o.cnt = d_out_count;
    o.d2 = d_out_d2;
    return launch_knn(*ix, qv, false, 0, (nq + GROUP - 1) / GROUP, k, eps, o);
}

int pcpx_knn_batch(pcpx_index* h, const float* q_xyz, uint64_t nq, uint32_t k, float eps, uint32_t* out_idx,
                   uint32_t* out_count, float* out_d2)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_knn_batch");
    if (nq == 0) return PCPX_OK;
    if (!q_xyz || !out_count || (k > 0 && !out_idx)) return PCPX_ERR_INVALID;
    if (k == 0) {
        std::memset(out_count, 0, nq * sizeof(u32));
        return PCPX_OK;
    }
    if (nq <= FEW_QUERIES_MAX && k <= 32) {
        // latency path (pcpx_few.hip): one wavefront per query, the queries and the rows live in the handle's pinned
        // stage, which the device reads and writes in place -- no allocation, no copy, one launch, one synchronisation
        const size_t o_done = 0, o_q = 64, o_idx = o_q + nq * 3 * sizeof(float), o_cnt = o_idx + nq * k * sizeof(u32),
                     o_d2 = o_cnt + nq * sizeof(u32), o_flag = o_d2 + nq * k * sizeof(float), total = o_flag + nq * sizeof(u32);
        const bool fresh = ix->pinned.bytes < total;
        if ((st = ix->pinned.ensure(total)) != PCPX_OK) return st;
        char* stage = static_cast<char*>(ix->pinned.p);
        volatile u32* done = reinterpret_cast<volatile u32*>(stage + o_done);
        if (fresh) *done = 0u;
        if ((st = ensure_queue(*ix)) != PCPX_OK) return st;  // the work-queue counters' allocation also holds the latency path's completion counter (word 15 of queue 7 of the first set)
        u32* done_count = ix->d_queue + 8 * 16 - 1;
        std::memcpy(stage + o_q, q_xyz, nq * 3 * sizeof(float));
        const u32 epoch = ++ix->few_epoch ? ix->few_epoch : ++ix->few_epoch;  // never 0
        if ((st = launch_knn_few(*ix, reinterpret_cast<const float*>(stage + o_q), q_xyz, static_cast<u32>(nq), k, eps,
                                 reinterpret_cast<u32*>(stage + o_idx), reinterpret_cast<u32*>(stage + o_cnt),
                                 out_d2 ? reinterpret_cast<float*>(stage + o_d2) : nullptr, reinterpret_cast<u32*>(stage + o_flag),
                                 done_count, const_cast<u32*>(done), epoch)) != PCPX_OK)
            return st;
        // poll the completion flag the kernel's last block stores into the pinned stage: no trip through the runtime's
        // completion signal (the stream stays in order: the next launch on it runs after this kernel has retired)
        bool seen = false;
        if (!g_few_no_poll)
            for (u32 spin = 0; spin < 400000u && !seen; ++spin) seen = *done == epoch;
        if (!seen) PCPX_HIP(hipStreamSynchronize(ix->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
        bool complete = true;
        const u32* flags = reinterpret_cast<const u32*>(stage + o_flag);
        for (u64 q = 0; q < nq; ++q) complete = complete && flags[q] == 0u;
        if (complete) {
            std::memcpy(out_idx, stage + o_idx, nq * k * sizeof(u32));
            std::memcpy(out_count, stage + o_cnt, nq * sizeof(u32));
            if (out_d2) std::memcpy(out_d2, stage + o_d2, nq * k * sizeof(float));
            return PCPX_OK;
        }
        // a frontier or candidate list overflowed (a query far outside a large cloud, hundreds of exact ties): general path
    }
    DevBuf dq(ix->pool), di(ix->pool), dc(ix->pool), dd(ix->pool);
    if ((st = dq.alloc(nq * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = di.alloc(nq * k * sizeof(u32))) != PCPX_OK) return st;
    if ((st = dc.alloc(nq * sizeof(u32))) != PCPX_OK) return st;
    if (out_d2 && (st = dd.alloc(nq * k * sizeof(float))) != PCPX_OK) return st;
    if ((st = upload_pageable(dq.p, q_xyz, nq * 3 * sizeof(float), ix->stream)) != PCPX_OK) return st;
    st = pcpx_knn_batch_dev(h, dq.as<float>(), nq, k, eps, di.as<u32>(), dc.as<u32>(), out_d2 ? dd.as<float>() : nullptr);
    if (st != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_idx, di.p, nq * k * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipMemcpyAsync(out_count, dc.p, nq * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    if (out_d2) PCPX_HIP(hipMemcpyAsync(out_d2, dd.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

namespace pcpx {
namespace {
int ensure_gather_arrays(Index& ix, size_t bytes_per_position = sizeof(float4));
// Sphere counts of the query groups [gf, gf + gc) of a whole-cloud handle, BY INPUT INDEX.  Switch "gather_counts" (off): the kernel
// leaves the count of curve position p at [p] of a scratch array (256 contiguous bytes per query group) and k_gather_u32 takes them
// to input order with coalesced writes.  Written straight to input index perm[p] they are 4 bytes per 32-byte sector, read for
// ownership and written back (320 MB at the memory side for 40 MB of counts) -- and still faster than the permute's ten million
// random reads (measured, round 5: 1.92-2.08 ms straight, 2.13 through the permute).
int range_count_self_rows(Index& ix, u64 gf, u64 gc, float radius, u32* d_out_count)
{
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix.n)};
    if (ix.tuning.gather_counts && gc > 0 && ensure_gather_arrays(ix, sizeof(u32)) == PCPX_OK) {
        u32* at_position = reinterpret_cast<u32*>(ix.d_nc4);  // (n + 64 float4: room for n counts)
        qv.by_position = 1;
        int st = launch_range_count(ix, qv, true, gf, gc, radius, nullptr, at_position);
        if (st != PCPX_OK) return st;
        const u64 lo = gf * GROUP, hi = (gf + gc) * GROUP < ix.n ? (gf + gc) * GROUP : ix.n;
        return launch_gather_u32(ix, at_position, ix.d_pos_of, ix.n_in, static_cast<u32>(lo), static_cast<u32>(hi), d_out_count);
    }
    return launch_range_count(ix, qv, true, gf, gc, radius, nullptr, d_out_count);
}
}
}  // namespace pcpx

// ---- radius search -------------------------------------------------------------------------------
int pcpx_range_count_self_dev(pcpx_index* h, float radius, uint64_t sorted_first, uint64_t sorted_count,
                              uint32_t* d_out_count)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (!d_out_count) return PCPX_ERR_INVALID;
    if (sorted_first % GROUP != 0) {
        set_error("pcpx_range_count_self_dev: sorted_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    if (ix->shard.on) return shard_range_count_self(*ix, radius, sorted_first, sorted_count, d_out_count);
    u64 gf, gc;
    slice_to_groups(*ix, sorted_first, sorted_count, gf, gc);
    return range_count_self_rows(*ix, gf, gc, radius, d_out_count);
}

// The same with the count of sorted position p at d_out_count[p]: a query group's 64 counts are one 256-byte store (the
// input-order form scatters 4-byte stores over the whole array: eight times the bytes at the memory side).
int pcpx_range_count_self_curve_order_dev(pcpx_index* h, float radius, uint64_t sorted_first, uint64_t sorted_count,
                                          uint32_t* d_out_count)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    if (!d_out_count) return PCPX_ERR_INVALID;
    if (sorted_first % GROUP != 0) {
        set_error("pcpx_range_count_self_curve_order_dev: sorted_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    if (ix->shard.on) return shard_range_count_self(*ix, radius, sorted_first, sorted_count, d_out_count, true);
    u64 gf, gc;
    slice_to_groups(*ix, sorted_first, sorted_count, gf, gc);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    qv.by_position = 1;
    return launch_range_count(*ix, qv, true, gf, gc, radius, nullptr, d_out_count);
}

int pcpx_range_lists_self_dev(pcpx_index* h, float radius, uint64_t* d_out_offsets, uint32_t* d_out_idx, uint64_t idx_capacity,
                              uint64_t* out_total)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_range_lists_self_dev");
    if (!d_out_offsets || !out_total) return PCPX_ERR_INVALID;
    *out_total = 0;
    const u64 rows = ix->n_in;
    if (rows == 0) {
        PCPX_HIP(hipMemsetAsync(d_out_offsets, 0, sizeof(u64), ix->stream));
        PCPX_HIP(hipStreamSynchronize(ix->stream));
        return PCPX_OK;
    }
    // scratch of the handle: the counts by input row and the scan's tile sums
    const size_t cnt_bytes = (rows * sizeof(u32) + 255) / 256 * 256, sums = (rows / 1024 + 2) * sizeof(u64);
    if ((st = ensure_scratch(*ix, cnt_bytes + sums)) != PCPX_OK) return st;
    u32* d_cnt = static_cast<u32*>(ix->d_scratch);
    u64* d_sums = reinterpret_cast<u64*>(static_cast<char*>(ix->d_scratch) + cnt_bytes);
    if (ix->n != ix->n_in) PCPX_HIP(hipMemsetAsync(d_cnt, 0, rows * sizeof(u32), ix->stream));  // (points outside the grid: empty lists)
    const u64 groups = (ix->n + GROUP - 1) / GROUP;
    if ((st = range_count_self_rows(*ix, 0, groups, radius, d_cnt)) != PCPX_OK) return st;
    if ((st = launch_range_offsets(*ix, d_cnt, rows, d_sums, d_out_offsets)) != PCPX_OK) return st;
    u64 total = 0;
    PCPX_HIP(hipMemcpyAsync(&total, d_out_offsets + rows, sizeof(u64), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    *out_total = total;
    if (total == 0) return PCPX_OK;
    if (!d_out_idx || idx_capacity < total) {
        set_error("pcpx_range_lists_self_dev: need room for %llu indices", static_cast<unsigned long long>(total));
        return PCPX_ERR_CAPACITY;
    }
    return launch_range_fill_self(*ix, 0, groups, radius, d_out_offsets, d_out_idx);
}

int pcpx_range_count_self(pcpx_index* h, float radius, uint32_t* out_count)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_range_count_self");
    if (!out_count) return PCPX_ERR_INVALID;
    u64 rows = ix->n_in;
    DevBuf dc(ix->pool);
    if ((st = dc.alloc(rows * sizeof(u32))) != PCPX_OK) return st;
    PCPX_HIP(hipMemsetAsync(dc.p, 0, rows * sizeof(u32), ix->stream));
    if ((st = pcpx_range_count_self_dev(h, radius, 0, UINT64_MAX, dc.as<u32>())) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_count, dc.p, rows * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_range_count_batch(pcpx_index* h, const float* q_xyz, uint64_t nq, float radius, uint32_t* out_count)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_range_count_batch");
    if (nq == 0) return PCPX_OK;
    if (!q_xyz || !out_count) return PCPX_ERR_INVALID;
    DevBuf dq(ix->pool), dc(ix->pool);
    if ((st = dq.alloc(nq * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = dc.alloc(nq * sizeof(u32))) != PCPX_OK) return st;
    if ((st = upload_pageable(dq.p, q_xyz, nq * 3 * sizeof(float), ix->stream)) != PCPX_OK) return st;
    QueryView qv;
    if ((st = prepare_queries(*ix, dq.as<float>(), nq, qv)) != PCPX_OK) return st;
    if ((st = launch_range_count(*ix, qv, false, 0, (nq + GROUP - 1) / GROUP, radius, nullptr, dc.as<u32>())) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(out_count, dc.p, nq * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

// One sphere / one box per call: the latency form (pcpx_range.hip: k_range_one) through the handle's pinned stage.  Returns
// PCPX_OK / PCPX_ERR_CAPACITY like the batch form, or +1 when the range holds more than the stage does (caller: batch form).
constexpr u32 RANGE_ONE_CAP = 4096;
static int range_one_to_host(Index* ix, bool aabb, const float* range6, uint64_t* out_offsets, uint32_t* out_idx, uint64_t idx_capacity)
{
    int st;
    const size_t o_done = 0, o_cnt = 128, o_idx = 192, total = o_idx + RANGE_ONE_CAP * sizeof(u32);
    const bool fresh = ix->pinned.bytes < total;
    if ((st = ix->pinned.ensure(total)) != PCPX_OK) return st;
    char* stage = static_cast<char*>(ix->pinned.p);
    volatile u32* done = reinterpret_cast<volatile u32*>(stage + o_done);
    if (fresh) *done = 0u;
    const u32 epoch = ++ix->few_epoch ? ix->few_epoch : ++ix->few_epoch;  // never 0 (shared with the k-NN latency path: same word)
    if ((st = launch_range_one(*ix, aabb, range6, RANGE_ONE_CAP, reinterpret_cast<u32*>(stage + o_idx),
                               reinterpret_cast<u32*>(stage + o_cnt), const_cast<u32*>(done), epoch)) != PCPX_OK)
        return st;
    bool seen = false;
    if (!g_few_no_poll)
        for (u32 spin = 0; spin < 400000u && !seen; ++spin) seen = *done == epoch;
    if (!seen) PCPX_HIP(hipStreamSynchronize(ix->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
    const u32 cnt = *reinterpret_cast<const volatile u32*>(stage + o_cnt);
    if (cnt > RANGE_ONE_CAP) return 1;
    out_offsets[0] = 0;
    out_offsets[1] = cnt;
    if (cnt == 0) return PCPX_OK;
    if (!out_idx || idx_capacity < cnt) {
        set_error("pcpx range search: need room for %u indices", cnt);
        return PCPX_ERR_CAPACITY;
    }
    std::memcpy(out_idx, stage + o_idx, cnt * sizeof(u32));
    return PCPX_OK;
}

int pcpx_range_sphere_batch(pcpx_index* h, const float* q_xyz, const float* radii, float radius, uint64_t nq,
                            uint64_t* out_offsets, uint32_t* out_idx, uint64_t idx_capacity)
{
    return no_throw("pcpx_range_sphere_batch", [&]() -> int {
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_range_sphere_batch");
    if (!out_offsets || (nq > 0 && !q_xyz)) return PCPX_ERR_INVALID;
    if (nq == 0) {
        out_offsets[0] = 0;
        return PCPX_OK;
    }
    if (nq == 1) {  // the per-call shape of the reference's API: one launch through the pinned stage
        const float sphere[4] = {q_xyz[0], q_xyz[1], q_xyz[2], radii ? radii[0] : radius};
        st = range_one_to_host(ix, false, sphere, out_offsets, out_idx, idx_capacity);
        if (st != 1) return st;
    }
    DevBuf dq(ix->pool), dr(ix->pool), dc(ix->pool), doff(ix->pool), dout(ix->pool);
    if ((st = dq.alloc(nq * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = dc.alloc(nq * sizeof(u32))) != PCPX_OK) return st;
    if ((st = upload_pageable(dq.p, q_xyz, nq * 3 * sizeof(float), ix->stream)) != PCPX_OK) return st;
    if (radii) {
        if ((st = dr.alloc(nq * sizeof(float))) != PCPX_OK) return st;
        PCPX_HIP(hipMemcpyAsync(dr.p, radii, nq * sizeof(float), hipMemcpyHostToDevice, ix->stream));
    }
    QueryView qv;
    if ((st = prepare_queries(*ix, dq.as<float>(), nq, qv)) != PCPX_OK) return st;
    if ((st = launch_range_count(*ix, qv, false, 0, (nq + GROUP - 1) / GROUP, radius, dr.as<float>(), dc.as<u32>())) != PCPX_OK)
        return st;
    std::vector<u32> cnt(nq);
    PCPX_HIP(hipMemcpyAsync(cnt.data(), dc.p, nq * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    exclusive_scan_host(cnt, out_offsets);
    u64 total = out_offsets[nq];
    if (total == 0) return PCPX_OK;
    if (!out_idx || idx_capacity < total) {
        set_error("pcpx_range_sphere_batch: need room for %llu indices", static_cast<unsigned long long>(total));
        return PCPX_ERR_CAPACITY;
    }
    if ((st = doff.alloc((nq + 1) * sizeof(u64))) != PCPX_OK) return st;
    if ((st = dout.alloc(total * sizeof(u32))) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(doff.p, out_offsets, (nq + 1) * sizeof(u64), hipMemcpyHostToDevice, ix->stream));
    if ((st = launch_range_fill(*ix, qv, radius, dr.as<float>(), doff.as<u64>(), dout.as<u32>())) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_idx, dout.p, total * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
    });
}

int pcpx_range_aabb_batch(pcpx_index* h, const float* boxes6, uint64_t nb, uint64_t* out_offsets, uint32_t* out_idx,
                          uint64_t idx_capacity)
{
    return no_throw("pcpx_range_aabb_batch", [&]() -> int {
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_range_aabb_batch");
    if (!out_offsets || (nb > 0 && !boxes6)) return PCPX_ERR_INVALID;
    if (nb == 0) {
        out_offsets[0] = 0;
        return PCPX_OK;
    }
    if (nb == 1) {
        st = range_one_to_host(ix, true, boxes6, out_offsets, out_idx, idx_capacity);
        if (st != 1) return st;
    }
    DevBuf db(ix->pool), dc(ix->pool), doff(ix->pool), dout(ix->pool);
    if ((st = db.alloc(nb * 6 * sizeof(float))) != PCPX_OK) return st;
    if ((st = dc.alloc(nb * sizeof(u32))) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(db.p, boxes6, nb * 6 * sizeof(float), hipMemcpyHostToDevice, ix->stream));
    if ((st = launch_aabb_count(*ix, db.as<float>(), nb, dc.as<u32>())) != PCPX_OK) return st;
    std::vector<u32> cnt(nb);
    PCPX_HIP(hipMemcpyAsync(cnt.data(), dc.p, nb * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    exclusive_scan_host(cnt, out_offsets);
    u64 total = out_offsets[nb];
    if (total == 0) return PCPX_OK;
    if (!out_idx || idx_capacity < total) {
        set_error("pcpx_range_aabb_batch: need room for %llu indices", static_cast<unsigned long long>(total));
        return PCPX_ERR_CAPACITY;
    }
    if ((st = doff.alloc((nb + 1) * sizeof(u64))) != PCPX_OK) return st;
    if ((st = dout.alloc(total * sizeof(u32))) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(doff.p, out_offsets, (nb + 1) * sizeof(u64), hipMemcpyHostToDevice, ix->stream));
    if ((st = launch_aabb_fill(*ix, db.as<float>(), nb, doff.as<u64>(), dout.as<u32>())) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_idx, dout.p, total * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
    });
}

// ---- normals -------------------------------------------------------------------------------------
namespace pcpx {
namespace {
// d_pos_of (input index -> curve position) and d_nc4 ({normal, count} per curve position) of a whole-cloud handle, made on demand:
// what the gather-form permute of the input-order normals needs.  PCPX_ERR_ALLOC leaves the handle as it was (the caller takes the
// direct form).
int ensure_gather_arrays(Index& ix, size_t bytes_per_position)
{
    if (ix.n_in > ix.pos_of_cap || !ix.d_pos_of) {
        PCPX_HIP(hipStreamSynchronize(ix.stream));
        index_block_free(ix.d_pos_of);
        ix.d_pos_of = nullptr;
        ix.pos_of_cap = 0;
        ix.pos_of_valid = false;
        void* p = nullptr;
        if (index_block_alloc(&p, (ix.n_in ? ix.n_in : 1) * sizeof(u32)) != hipSuccess) {
            (void)hipGetLastError();
            return PCPX_ERR_ALLOC;
        }
        ix.d_pos_of = static_cast<u32*>(p);
        ix.pos_of_cap = ix.n_in;
    }
    const u64 need_bytes = (ix.n + GROUP) * bytes_per_position;  // (nc4_cap: bytes of the per-position scratch)
    if (need_bytes > ix.nc4_cap || !ix.d_nc4) {
        PCPX_HIP(hipStreamSynchronize(ix.stream));
        index_block_free(ix.d_nc4);
        ix.d_nc4 = nullptr;
        ix.nc4_cap = 0;
        void* p = nullptr;
        if (index_block_alloc(&p, need_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return PCPX_ERR_ALLOC;
        }
        ix.d_nc4 = static_cast<float4*>(p);
        ix.nc4_cap = need_bytes;
    }
    if (!ix.pos_of_valid) {
        PCPX_HIP(hipMemsetAsync(ix.d_pos_of, 0xFF, ix.n_in * sizeof(u32), ix.stream));
        int st = ix.n ? launch_invert_perm(ix.d_perm, ix.n, ix.d_pos_of, ix.stream) : PCPX_OK;
        if (st != PCPX_OK) return st;
        ix.pos_of_valid = true;
    }
    return PCPX_OK;
}
}  // namespace
}  // namespace pcpx

int pcpx_normals_knn_self_dev(pcpx_index* h, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                              float* d_out_normals, uint32_t* d_opt_out_idx, uint32_t* d_opt_out_count)
{
    return pcpx_normals_knn_self_strided_dev(h, k, eps, sorted_first, sorted_count, 0, d_out_normals, d_opt_out_idx, d_opt_out_count);
}

int pcpx_normals_knn_self_strided_dev(pcpx_index* h, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count, uint32_t row_stride,
                                      float* d_out_normals, uint32_t* d_opt_out_idx, uint32_t* d_opt_out_count)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (!d_out_normals || k == 0) return PCPX_ERR_INVALID;
    if (sorted_first % GROUP != 0) {
        set_error("pcpx_normals_knn_self_dev: sorted_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    // fused kernel: kNN rows stay in registers, only what the caller asked for is written
    u64 gf, gc;
    slice_to_groups(*ix, sorted_first, sorted_count, gf, gc);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    KnnOutputs o;
    o.idx = d_opt_out_idx;
    o.cnt = d_opt_out_count;
    o.normals = d_out_normals;
    if (row_stride != 0 && (row_stride < k || k > 32)) {
        set_error("pcpx_normals_knn_self_strided_dev: row_stride must be 0 or >= k, and k <= 32");
        return PCPX_ERR_INVALID;
    }
    o.row_stride = row_stride;
    if (ix->shard.on) {
        if (k > 32 && (!o.idx || !o.cnt)) {
            set_error("pcpx_normals_knn_self_dev: a rank-local index with k > 32 needs the row outputs too");
            return PCPX_ERR_UNSUPPORTED;
        }
        return shard_knn_self(*ix, sorted_first, sorted_count, k, eps, o);
    }
    // Input-order normals (+ counts) by the gather-form permute: the kernel leaves {normal, count} at the query's CURVE position (one
    // contiguous kilobyte per wave) and k_gather_nc4 takes them to input order with coalesced writes.  Written straight to row
    // perm[p] they are 12 + 4 bytes scattered over the whole output: every store instruction touches 64 lines, and every partial
    // 32-byte sector is read for ownership and written back.
    if (ix->tuning.gather && k <= 32 && gc > 0 && ensure_gather_arrays(*ix) == PCPX_OK) {
        o.nc4 = ix->d_nc4;
        o.normals = nullptr;
        o.cnt = nullptr;
        if ((st = launch_knn(*ix, qv, true, gf, gc, k, eps, o)) != PCPX_OK) return st;
        // (what the launch answered: whole groups -- a slice that ends inside a group is answered to that group's end)
        const u64 lo = gf * GROUP, hi = (gf + gc) * GROUP < ix->n ? (gf + gc) * GROUP : ix->n;
        return launch_gather_nc4(*ix, ix->d_nc4, ix->d_pos_of, ix->n_in, static_cast<u32>(lo), static_cast<u32>(hi), d_out_normals, d_opt_out_count);
    }
    if (k > 32 && (!o.idx || !o.cnt)) {  // the multi-pass path materialises rows: keep them in index scratch
        size_t need_idx = (static_cast<size_t>(ix->n_in) * k * sizeof(u32) + 255) / 256 * 256;
        if ((st = ensure_scratch(*ix, need_idx + static_cast<size_t>(ix->n_in) * sizeof(u32))) != PCPX_OK) return st;
        if (!o.idx) o.idx = static_cast<u32*>(ix->d_scratch);
        if (!o.cnt) o.cnt = reinterpret_cast<u32*>(static_cast<char*>(ix->d_scratch) + need_idx);
    }
    return launch_knn(*ix, qv, true, gf, gc, k, eps, o);
}

// estimate_tangent_planes / average_distances_to_neighbors over the index's own points
int pcpx_neighbourhoods_self_dev(pcpx_index* h, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                                 float* d_opt_normals, float* d_opt_centroids, float* d_opt_mean_dist)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (k == 0 || (!d_opt_normals && !d_opt_centroids && !d_opt_mean_dist)) return PCPX_ERR_INVALID;
    if (sorted_first % GROUP != 0) {
        set_error("pcpx_neighbourhoods_self_dev: sorted_first must be a multiple of %d", GROUP);
        return PCPX_ERR_INVALID;
    }
    u64 gf, gc;
    slice_to_groups(*ix, sorted_first, sorted_count, gf, gc);
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    KnnOutputs o;
    o.normals = d_opt_normals;
    o.centroids = d_opt_centroids;
    o.meandist = d_opt_mean_dist;
    if (ix->shard.on) {
        if (k > 32) {
            set_error("pcpx_neighbourhoods_self_dev: a rank-local index answers this for k <= 32");
            return PCPX_ERR_UNSUPPORTED;
        }
        return shard_knn_self(*ix, sorted_first, sorted_count, k, eps, o);
    }
    if (k > 32) {  // the multi-pass path materialises rows: keep them in index scratch
        size_t need_idx = (static_cast<size_t>(ix->n_in) * k * sizeof(u32) + 255) / 256 * 256;
        if ((st = ensure_scratch(*ix, need_idx + static_cast<size_t>(ix->n_in) * sizeof(u32))) != PCPX_OK) return st;
        o.idx = static_cast<u32*>(ix->d_scratch);
        o.cnt = reinterpret_cast<u32*>(static_cast<char*>(ix->d_scratch) + need_idx);
    }
    return launch_knn(*ix, qv, true, gf, gc, k, eps, o);
}

int pcpx_tangent_planes_knn_self(pcpx_index* h, uint32_t k, float eps, float* out_centroids, float* out_normals)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_tangent_planes_knn_self");
    if (!out_centroids || !out_normals || k == 0) return PCPX_ERR_INVALID;
    u64 rows = ix->n_in;
    DevBuf dc(ix->pool), dn(ix->pool);
    if ((st = dc.alloc(rows * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = dn.alloc(rows * 3 * sizeof(float))) != PCPX_OK) return st;
    PCPX_HIP(hipMemsetAsync(dc.p, 0, rows * 3 * sizeof(float), ix->stream));
    PCPX_HIP(hipMemsetAsync(dn.p, 0, rows * 3 * sizeof(float), ix->stream));
    if ((st = pcpx_neighbourhoods_self_dev(h, k, eps, 0, UINT64_MAX, dn.as<float>(), dc.as<float>(), nullptr)) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_centroids, dc.p, rows * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipMemcpyAsync(out_normals, dn.p, rows * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_mean_knn_distance_self(pcpx_index* h, uint32_t k, float eps, float* out_mean_dist)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_mean_knn_distance_self");
    if (!out_mean_dist || k == 0) return PCPX_ERR_INVALID;
    u64 rows = ix->n_in;
    DevBuf dm(ix->pool);
    if ((st = dm.alloc(rows * sizeof(float))) != PCPX_OK) return st;
    PCPX_HIP(hipMemsetAsync(dm.p, 0, rows * sizeof(float), ix->stream));
    if ((st = pcpx_neighbourhoods_self_dev(h, k, eps, 0, UINT64_MAX, nullptr, nullptr, dm.as<float>())) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_mean_dist, dm.p, rows * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_normals_knn_self(pcpx_index* h, uint32_t k, float eps, float* out_normals, uint32_t* opt_out_idx,
                          uint32_t* opt_out_count)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_normals_knn_self");
    if (!out_normals || k == 0) return PCPX_ERR_INVALID;
    return self_queries_to_host(ix, k, eps, out_normals, opt_out_idx, opt_out_count, nullptr);
}

// Rows in curve order: the kernel writes the rows of a slice of the sorted order into one contiguous piece of every output
// array, so a finished slice travels to the host while the later slices are still being computed (the input-order form
// scatters rows over the whole output: its copy cannot start before the last kernel has ended).
int pcpx_normals_knn_self_curve_order(pcpx_index* h, uint32_t k, float eps, float* opt_out_normals, uint32_t* out_idx, uint32_t* out_count,
                                      uint32_t* opt_out_perm, uint32_t* opt_out_position_of)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_normals_knn_self_curve_order");
    if (k == 0 || !out_idx || !out_count) return PCPX_ERR_INVALID;
    const u64 rows = ix->n;  // inserted points only: a point outside the voxel grid has no position on the curve
    if (opt_out_position_of && ix->n != ix->n_in) std::memset(opt_out_position_of, 0xFF, ix->n_in * sizeof(u32));
    if (rows == 0) return PCPX_OK;
    if (!ix->copy_stream) PCPX_HIP(pooled_stream_get(&ix->copy_stream));
    DevBuf dn(ix->pool), di(ix->pool), dc(ix->pool), dinv(ix->pool);
    if (opt_out_normals && (st = dn.alloc(rows * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = di.alloc(rows * k * sizeof(u32))) != PCPX_OK || (st = dc.alloc(rows * sizeof(u32))) != PCPX_OK) return st;
    if (opt_out_position_of && (st = dinv.alloc(ix->n_in * sizeof(u32))) != PCPX_OK) return st;
    QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
    KnnOutputs o;
    o.idx = di.as<u32>();
    o.cnt = dc.as<u32>();
    o.normals = dn.as<float>();
    o.by_position = 1;
    const u64 groups = (rows + GROUP - 1) / GROUP;
    constexpr int MAX_SLICES = 8;
    const int slices = static_cast<int>(groups < 8 * 1024 ? 1 : MAX_SLICES);  // (a slice should still fill the chip a few times over)
    hipEvent_t done[MAX_SLICES] = {};
    struct EventGuard {
        hipEvent_t* e;
        ~EventGuard()
        {
            for (int i = 0; i < MAX_SLICES; ++i)
                if (e[i]) (void)hipEventDestroy(e[i]);
        }
    } guard{done};
    // Whatever way this function is left -- an error half way through included -- both streams are drained first: queued
    // kernels and copies write the caller's arrays and read buffers that go back to the pool (declared last: destroyed first,
    // before the events and the buffers above).
    struct Drain {
        Index* ix;
        ~Drain()
        {
            (void)hipStreamSynchronize(ix->copy_stream);
            (void)hipStreamSynchronize(ix->stream);
        }
    } drain{ix};
    // all the kernels first (enqueueing does not block) ...
    u64 g_first[MAX_SLICES + 1];
    for (int s = 0; s <= slices; ++s) g_first[s] = groups * static_cast<u64>(s) / static_cast<u64>(slices);
    for (int s = 0; s < slices; ++s) {
        if ((st = launch_knn(*ix, qv, true, g_first[s], g_first[s + 1] - g_first[s], k, eps, o)) != PCPX_OK) return st;
        PCPX_HIP(hipEventCreateWithFlags(&done[s], hipEventDisableTiming));
        PCPX_HIP(hipEventRecord(done[s], ix->stream));
    }
    if (opt_out_position_of) {
        if (ix->n != ix->n_in) PCPX_HIP(hipMemsetAsync(dinv.p, 0xFF, ix->n_in * sizeof(u32), ix->stream));
        if ((st = launch_invert_perm(ix->d_perm, rows, dinv.as<u32>(), ix->stream)) != PCPX_OK) return st;
    }
    // ... then the copies, slice by slice, on the second stream, each behind its slice's kernel
    if (opt_out_perm) PCPX_HIP(hipMemcpyAsync(opt_out_perm, ix->d_perm, rows * sizeof(u32), hipMemcpyDeviceToHost, ix->copy_stream));
    for (int s = 0; s < slices; ++s) {
        const u64 r0 = g_first[s] * GROUP, r1 = std::min<u64>(rows, g_first[s + 1] * GROUP);
        PCPX_HIP(hipStreamWaitEvent(ix->copy_stream, done[s], 0));
        PCPX_HIP(hipMemcpyAsync(out_idx + r0 * k, di.as<u32>() + r0 * k, (r1 - r0) * k * sizeof(u32), hipMemcpyDeviceToHost, ix->copy_stream));
        PCPX_HIP(hipMemcpyAsync(out_count + r0, dc.as<u32>() + r0, (r1 - r0) * sizeof(u32), hipMemcpyDeviceToHost, ix->copy_stream));
        if (opt_out_normals)
            PCPX_HIP(hipMemcpyAsync(opt_out_normals + 3 * r0, dn.as<float>() + 3 * r0, (r1 - r0) * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->copy_stream));
    }
    PCPX_HIP(hipStreamSynchronize(ix->copy_stream));
    if (opt_out_position_of) PCPX_HIP(hipMemcpyAsync(opt_out_position_of, dinv.p, ix->n_in * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_normals_from_knn(pcpx_index* h, const uint32_t* nbr_idx, const uint32_t* count, uint64_t nq, uint32_t k,
                          float* out_normals, float* opt_out_evals)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_normals_from_knn");
    if (nq == 0) return PCPX_OK;
    if (!nbr_idx || !count || !out_normals || k == 0) return PCPX_ERR_INVALID;
    for (u64 q = 0; q < nq; ++q) {
        if (count[q] > k) {
            set_error("pcpx_normals_from_knn: count[%llu] > k", static_cast<unsigned long long>(q));
            return PCPX_ERR_INVALID;
        }
        for (u32 j = 0; j < count[q]; ++j)
            if (nbr_idx[q * k + j] >= ix->n_in) {
                set_error("pcpx_normals_from_knn: neighbour index out of range in row %llu", static_cast<unsigned long long>(q));
                return PCPX_ERR_INVALID;
            }
    }
    DevBuf di(ix->pool), dc(ix->pool), dn(ix->pool), de(ix->pool);
    if ((st = di.alloc(nq * k * sizeof(u32))) != PCPX_OK) return st;
    if ((st = dc.alloc(nq * sizeof(u32))) != PCPX_OK) return st;
    if ((st = dn.alloc(nq * 3 * sizeof(float))) != PCPX_OK) return st;
    if (opt_out_evals && (st = de.alloc(nq * 3 * sizeof(float))) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(di.p, nbr_idx, nq * k * sizeof(u32), hipMemcpyHostToDevice, ix->stream));
    PCPX_HIP(hipMemcpyAsync(dc.p, count, nq * sizeof(u32), hipMemcpyHostToDevice, ix->stream));
    st = launch_normals(*ix, di.as<u32>(), dc.as<u32>(), nullptr, 0, nq, k, dn.as<float>(),
                        opt_out_evals ? de.as<float>() : nullptr);
    if (st != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_normals, dn.p, nq * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    if (opt_out_evals) PCPX_HIP(hipMemcpyAsync(opt_out_evals, de.p, nq * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_estimate_normal(const float* xyz, uint64_t m, int device, float out_normal[3])
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (!out_normal || (m > 0 && !xyz)) return PCPX_ERR_INVALID;
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    if (m >= 1 && m <= NORMAL_ARG_POINTS) {
        // the reference's per-point shape (k neighbours, then their normal): the points travel in the kernel arguments, the
        // normal and a completion word come back through the device's pinned stage, which the host polls
        if ((st = shared.pinned.ensure(64 * sizeof(float))) != PCPX_OK) return st;
        float* stage = static_cast<float*>(shared.pinned.p);
        volatile u32* done = reinterpret_cast<volatile u32*>(stage + 8);
        const u32 epoch = ++shared.normal_epoch ? shared.normal_epoch : ++shared.normal_epoch;  // never 0
        if (*done == epoch) *done = 0u;  // (a fresh stage may hold anything)
        if ((st = launch_normal_args(xyz, static_cast<u32>(m), stage, const_cast<u32*>(done), epoch, nullptr)) != PCPX_OK) return st;
        bool seen = false;
        if (!g_few_no_poll)
            for (u32 spin = 0; spin < 400000u && !seen; ++spin) seen = *done == epoch;
        if (!seen) PCPX_HIP(hipStreamSynchronize(nullptr));
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(out_normal, stage, 3 * sizeof(float));
        return PCPX_OK;
    }
    if (m <= 4096) {
        // one neighbourhood: no allocation and no copy -- the points go into the device's pinned stage, the kernel reads
        // them and writes the normal there (host memory mapped into the device's address space)
        if ((st = shared.pinned.ensure((m * 3 + 4) * sizeof(float))) != PCPX_OK) return st;
        float* stage = static_cast<float*>(shared.pinned.p);
        if (m > 0) std::memcpy(stage + 4, xyz, m * 3 * sizeof(float));
        if ((st = launch_normal_single(stage + 4, m, stage, nullptr)) != PCPX_OK) return st;
        PCPX_HIP(hipStreamSynchronize(nullptr));
        std::memcpy(out_normal, stage, 3 * sizeof(float));
        return PCPX_OK;
    }
    DevBuf dp(shared.pool), dn(shared.pool);
    if ((st = dp.alloc(m * 3 * sizeof(float))) != PCPX_OK) return st;
    if ((st = dn.alloc(3 * sizeof(float))) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpy(dp.p, xyz, m * 3 * sizeof(float), hipMemcpyHostToDevice));
    if ((st = launch_normal_single(dp.as<float>(), m, dn.as<float>(), nullptr)) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpy(out_normal, dn.p, 3 * sizeof(float), hipMemcpyDeviceToHost));
    return PCPX_OK;
}

int pcpx_estimate_normals_batch(const float* xyz, const uint64_t* offsets, uint64_t nrows, int device, float* out_normals)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (nrows == 0) return PCPX_OK;
    if (!offsets || !out_normals) return PCPX_ERR_INVALID;
    for (u64 r = 0; r < nrows; ++r)
        if (offsets[r + 1] < offsets[r]) {
            set_error("pcpx_estimate_normals_batch: offsets must not decrease (row %llu)", static_cast<unsigned long long>(r));
            return PCPX_ERR_INVALID;
        }
    const u64 total = offsets[nrows] - offsets[0];
    if (total > 0 && !xyz) return PCPX_ERR_INVALID;
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    DevBuf dp(shared.pool), doff(shared.pool), dn(shared.pool);
    if ((st = dp.alloc(total * 3 * sizeof(float))) != PCPX_OK || (st = doff.alloc((nrows + 1) * sizeof(u64))) != PCPX_OK ||
        (st = dn.alloc(nrows * 3 * sizeof(float))) != PCPX_OK)
        return st;
    if (total > 0) PCPX_HIP(hipMemcpyAsync(dp.p, xyz + 3 * offsets[0], total * 3 * sizeof(float), hipMemcpyHostToDevice, nullptr));
    PCPX_HIP(hipMemcpyAsync(doff.p, offsets, (nrows + 1) * sizeof(u64), hipMemcpyHostToDevice, nullptr));
    if ((st = launch_normals_csr(dp.as<float>(), doff.as<u64>(), nrows, dn.as<float>(), nullptr)) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(out_normals, dn.p, nrows * 3 * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    PCPX_HIP(hipStreamSynchronize(nullptr));
    return PCPX_OK;
}

int pcpx_debug_eps_test_mode(pcpx_index* h, int mode)
{
    Index* ix = reinterpret_cast<Index*>(h);
    if (!ix || mode < 0 || mode > 2) {
        set_error("pcpx_debug_eps_test_mode: mode is 0 (automatic), 1 (in the compaction) or 2 (per candidate)");
        return PCPX_ERR_INVALID;
    }
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    ix->eps_test_mode = mode;
    return PCPX_OK;
}

int pcpx_debug_set(pcpx_index* h, const char* name, int64_t value)
{
    Index* ix = reinterpret_cast<Index*>(h);
    if (!ix || !name) return PCPX_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    const std::string key(name);
    if (key == "long_groups_first") {
        ix->tuning.lpt = value != 0;
        ix->sched.state = 0;
    } else if (key == "gather_outputs") {
        ix->tuning.gather = value != 0;
    } else if (key == "gather_counts") {
        ix->tuning.gather_counts = value != 0;
    } else {
        set_error("pcpx_debug_set: no setting called '%s'", name);
        return PCPX_ERR_INVALID;
    }
    return PCPX_OK;
}

int pcpx_debug_get(pcpx_index* h, const char* name, int64_t* out_value)
{
    Index* ix = reinterpret_cast<Index*>(h);
    if (!ix || !name || !out_value) return PCPX_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    const std::string key(name);
    if (key == "build_redos") {
        *out_value = ix->build_redos;
    } else if (key == "full_buckets") {
        int64_t c = 0;
        for (u32 w = 0; w < 8; ++w) c += __builtin_popcount(ix->full_buckets[w]);
        *out_value = c;
    } else if (key == "schedule_state") {
        *out_value = ix->sched.state;
    } else {
        set_error("pcpx_debug_get: no figure called '%s'", name);
        return PCPX_ERR_INVALID;
    }
    return PCPX_OK;
}

int pcpx_debug_group_times(pcpx_index* h, uint32_t* out_ticks, uint64_t capacity, uint64_t* out_groups)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    if (!out_groups) return PCPX_ERR_INVALID;
    *out_groups = ix->sched.state ? ix->sched.gc : 0;
    if (*out_groups == 0) return PCPX_OK;
    if (!out_ticks || capacity < *out_groups) return PCPX_ERR_CAPACITY;
    PCPX_HIP(hipMemcpyAsync(out_ticks, ix->sched.d_gtime, *out_groups * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_knn_group_costs_dev(pcpx_index* h, uint32_t k, float eps, uint32_t group_stride, uint32_t* d_out_events, uint64_t capacity,
                             uint64_t* out_samples)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);
    if (!out_samples || group_stride == 0) return PCPX_ERR_INVALID;
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_knn_group_costs_dev");
    const u64 groups = (ix->n + GROUP - 1) / GROUP;
    const u64 nsamples = groups / group_stride;
    *out_samples = nsamples;
    if (!d_out_events || capacity < nsamples) return nsamples == 0 ? PCPX_OK : PCPX_ERR_CAPACITY;
    u32 ns = 0;
    return launch_knn_cost_sample(*ix, k, eps, group_stride, d_out_events, &ns);
}

// cost of a sampled group from its event counts, in instructions of the k <= 16 kernel (ISA counts, DESIGN.md "k_knn budget"): an
// expansion is 4 box tests + the walk's scalar side, a dense leaf 8 candidates x 64 lanes, a packed leaf its publish + read-back and
// ~10 per step of eight needing lanes, a fold the selection network; the constant is a group's seed phase, cap and epilogue.
static inline uint64_t group_cost_of(const uint32_t e[4])
{
    const uint64_t folds = e[3] >> 16, steps = e[3] & 0xFFFFu, dense = e[1] & 0xFFFFFFu, packed = e[2] & 0xFFFFFu;
    return 6000ull + 112ull * e[0] + 108ull * dense + 38ull * packed + 11ull * steps + 140ull * folds;
}

int pcpx_shard_cuts_by_cost(uint64_t n, uint32_t world, uint32_t group_stride, const uint32_t* events, uint64_t nsamples, uint64_t* out_first)
{
    if (!out_first || world == 0 || group_stride == 0 || (nsamples && !events)) return PCPX_ERR_INVALID;
    const u64 groups = (n + GROUP - 1) / GROUP;
    if (nsamples != groups / group_stride) {
        set_error("pcpx_shard_cuts_by_cost: %llu samples do not describe %llu groups at stride %u", static_cast<unsigned long long>(nsamples),
                  static_cast<unsigned long long>(groups), group_stride);
        return PCPX_ERR_INVALID;
    }
    out_first[0] = 0;
    out_first[world] = n;
    if (nsamples == 0) {  // too small to sample: cut by count
        for (u32 r = 1; r < world; ++r) {
            u64 f = groups * r / world * GROUP;
            out_first[r] = f > n ? n : f;
        }
        return PCPX_OK;
    }
    // cost per group, block by block (sample i stands for groups [i stride, (i + 1) stride); the tail beyond the last whole block takes
    // the last sample's), prefix sums in 64-bit integers: every rank of a job gets the same cuts from the same counts
    std::vector<u64> prefix(nsamples + 2, 0);
    for (u64 i = 0; i < nsamples; ++i) prefix[i + 1] = prefix[i] + group_cost_of(events + 4 * i) * group_stride;
    const u64 tail_groups = groups - nsamples * group_stride;
    prefix[nsamples + 1] = prefix[nsamples] + group_cost_of(events + 4 * (nsamples - 1)) * tail_groups;
    const u64 total = prefix[nsamples + 1];
    u64 i = 0;
    for (u32 r = 1; r < world; ++r) {
        const u64 target = static_cast<u64>((static_cast<unsigned __int128>(total) * r) / world);
        while (i + 1 < nsamples + 1 && prefix[i + 1] <= target) ++i;  // block i holds the target
        const u64 block_groups = i < nsamples ? group_stride : tail_groups;
        const u64 block_cost = prefix[i + 1] - prefix[i];
        u64 inside = block_cost ? static_cast<u64>((static_cast<unsigned __int128>(target - prefix[i]) * block_groups) / block_cost) : 0;
        if (inside > block_groups) inside = block_groups;
        u64 g = i * group_stride + inside;
        if (g > groups) g = groups;
        u64 f = g * GROUP;
        if (f > n) f = n;
        if (f < out_first[r - 1]) f = out_first[r - 1];
        out_first[r] = f;
    }
    return PCPX_OK;
}

int pcpx_debug_knn_stats(pcpx_index* h, uint32_t k, float eps, uint64_t* out_stats, uint64_t capacity)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_debug_knn_stats");
    if (!out_stats || capacity < 16 || k == 0 || k > 16) return PCPX_ERR_INVALID;
    DevBuf ds(ix->pool);
    const size_t cap = 16 + 5 * 65536;  // 16 counters + 5-word records of up to 65536 persistent waves
    if ((st = ds.alloc(cap * sizeof(u64))) != PCPX_OK) return st;
    PCPX_HIP(hipMemsetAsync(ds.p, 0, cap * sizeof(u64), ix->stream));
    // capacity bit 63 set: "floor" mode -- start every lane from its true k-th distance (taken from a normal run first)
    const bool floor_mode = (capacity >> 63) != 0;
    capacity &= ~(1ull << 63);
    DevBuf known(ix->pool);
    if (floor_mode) {
        if ((st = known.alloc(static_cast<size_t>(ix->n_in) * k * sizeof(float))) != PCPX_OK) return st;
        QueryView qv{nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<u32>(ix->n)};
        KnnOutputs o;
        o.d2 = known.as<float>();
        if ((st = launch_knn(*ix, qv, true, 0, (ix->n + GROUP - 1) / GROUP, k, eps, o)) != PCPX_OK) return st;
    }
    if ((st = launch_knn_stats(*ix, k, eps, ds.as<unsigned long long>(), floor_mode ? known.as<float>() : nullptr)) != PCPX_OK) return st;
    const size_t ncopy = capacity < cap ? capacity : cap;
    PCPX_HIP(hipMemcpyAsync(out_stats, ds.p, ncopy * sizeof(u64), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_propagate_normal_orientations(const float* xyz, uint64_t n, const uint32_t* knn_idx, const uint32_t* opt_knn_count,
                                       uint32_t k, float* normals, uint64_t* opt_out_reached)
{
    return no_throw("pcpx_propagate_normal_orientations", [&]() -> int {
    if (opt_out_reached) *opt_out_reached = 0;
    if (n == 0) return PCPX_OK;
    if (!xyz || !normals || (k > 0 && !knn_idx)) {
        set_error("pcpx_propagate_normal_orientations: null argument");
        return PCPX_ERR_INVALID;
    }
    if (n > 0xFFFFFFFFull) {
        set_error("pcpx_propagate_normal_orientations: more than 2^32 - 1 vertices");
        return PCPX_ERR_UNSUPPORTED;
    }
    for (u64 i = 0; i < n; ++i) {
        const u32 c = opt_knn_count ? opt_knn_count[i] : k;
        if (c > k) {
            set_error("pcpx_propagate_normal_orientations: count[%llu] = %u exceeds k = %u", static_cast<unsigned long long>(i), c, k);
            return PCPX_ERR_INVALID;
        }
        for (u32 j = 0; j < c; ++j)
            if (knn_idx[i * k + j] >= n) {
                set_error("pcpx_propagate_normal_orientations: row %llu holds index %u >= n", static_cast<unsigned long long>(i),
                          knn_idx[i * k + j]);
                return PCPX_ERR_INVALID;
            }
    }
    // root: first point of largest z
    u32 root = 0;
    for (u64 i = 1; i < n; ++i)
        if (xyz[3 * i + 2] > xyz[3 * static_cast<u64>(root) + 2]) root = static_cast<u32>(i);
    normals[3 * static_cast<u64>(root)] = 0.f;
    normals[3 * static_cast<u64>(root) + 1] = 0.f;
    normals[3 * static_cast<u64>(root) + 2] = 1.f;
    // every vertex enters the queue at most once (it is marked when first reached), so a flat array is the queue;
    // the root is only marked after its own edges, like in the reference, which cannot matter: a vertex is not
    // its own neighbour... unless a caller's rows say so, hence the explicit mark order is kept.
    std::vector<u32> order;
    order.reserve(n);
    std::vector<unsigned char> seen(n, 0);
    order.push_back(root);
    for (size_t head = 0; head < order.size(); ++head) {
        const u64 u = order[head];
        const u32 c = opt_knn_count ? opt_knn_count[u] : k;
        for (u32 j = 0; j < c; ++j) {
            const u64 v = knn_idx[u * k + j];
            if (seen[v]) continue;
            const float* a = normals + 3 * u;
            float* b = normals + 3 * v;
            const float xx = b[0] * a[0], yy = b[1] * a[1], zz = b[2] * a[2];
            const float dot = xx + yy + zz;
            if (dot < 0.f && !(std::fabs(dot - 0.f) < 1e-5f)) {
                b[0] = -b[0];
                b[1] = -b[1];
                b[2] = -b[2];
            }
            seen[v] = 1;
            order.push_back(static_cast<u32>(v));
        }
        seen[u] = 1;
    }
    if (opt_out_reached) {
        u64 reached = 0;
        for (u64 i = 0; i < n; ++i) reached += seen[i];
        *opt_out_reached = reached;
    }
    return PCPX_OK;
    });
}

int pcpx_propagate_normal_orientations_dev(const float* d_xyz, uint64_t n, const uint32_t* d_knn_idx,
                                           const uint32_t* d_opt_knn_count, uint32_t k, float* d_normals, int device, void* stream,
                                           uint64_t* opt_out_reached, uint32_t* opt_out_levels)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (n > 0 && (!d_xyz || !d_normals || (k > 0 && !d_knn_idx))) {
        set_error("pcpx_propagate_normal_orientations_dev: null argument");
        return PCPX_ERR_INVALID;
    }
    return orient_normals_device(d_xyz, n, d_knn_idx, d_opt_knn_count, k, d_normals, static_cast<hipStream_t>(stream),
                                 opt_out_reached, opt_out_levels);
}

int pcpx_oriented_normals_knn_self(pcpx_index* h, uint32_t k, float eps, float* out_normals, uint32_t* opt_out_idx,
                                   uint32_t* opt_out_count, uint64_t* opt_out_reached)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_oriented_normals_knn_self");
    if (!out_normals || k == 0) return PCPX_ERR_INVALID;
    if (ix->n != ix->n_in) {
        set_error("pcpx_oriented_normals_knn_self: %llu of %llu points lie outside the voxel grid and have no neighbourhood",
                  static_cast<unsigned long long>(ix->n_in - ix->n), static_cast<unsigned long long>(ix->n_in));
        return PCPX_ERR_UNSUPPORTED;
    }
    const u64 rows = ix->n_in;
    DevBuf dn(ix->pool), di(ix->pool), dc(ix->pool);
    if ((st = dn.alloc(rows * 3 * sizeof(float))) != PCPX_OK || (st = di.alloc(rows * k * sizeof(u32))) != PCPX_OK ||
        (st = dc.alloc(rows * sizeof(u32))) != PCPX_OK)
        return st;
    if ((st = pcpx_normals_knn_self_dev(h, k, eps, 0, UINT64_MAX, dn.as<float>(), di.as<u32>(), dc.as<u32>())) != PCPX_OK) return st;
    if ((st = orient_normals_device(ix->d_xyz, rows, di.as<u32>(), dc.as<u32>(), k, dn.as<float>(), ix->stream, opt_out_reached,
                                    nullptr)) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(out_normals, dn.p, rows * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    if (opt_out_idx) PCPX_HIP(hipMemcpyAsync(opt_out_idx, di.p, rows * k * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    if (opt_out_count) PCPX_HIP(hipMemcpyAsync(opt_out_count, dc.p, rows * sizeof(u32), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_orient_normals_knn_self(pcpx_index* h, uint32_t k, float eps, float* normals, uint64_t* opt_out_reached)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    PCPX_WHOLE_CLOUD_ONLY(ix, "pcpx_orient_normals_knn_self");
    if (!normals || k == 0) return PCPX_ERR_INVALID;
    if (ix->n != ix->n_in) {
        set_error("pcpx_orient_normals_knn_self: %llu of %llu points lie outside the voxel grid and have no neighbourhood",
                  static_cast<unsigned long long>(ix->n_in - ix->n), static_cast<unsigned long long>(ix->n_in));
        return PCPX_ERR_UNSUPPORTED;
    }
    const u64 rows = ix->n_in;
    DevBuf dn(ix->pool), di(ix->pool), dc(ix->pool);
    if ((st = dn.alloc(rows * 3 * sizeof(float))) != PCPX_OK || (st = di.alloc(rows * k * sizeof(u32))) != PCPX_OK ||
        (st = dc.alloc(rows * sizeof(u32))) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(dn.p, normals, rows * 3 * sizeof(float), hipMemcpyHostToDevice, ix->stream));
    if ((st = pcpx_knn_self_dev(h, k, eps, 0, UINT64_MAX, di.as<u32>(), dc.as<u32>(), nullptr)) != PCPX_OK) return st;
    if ((st = orient_normals_device(ix->d_xyz, rows, di.as<u32>(), dc.as<u32>(), k, dn.as<float>(), ix->stream, opt_out_reached,
                                    nullptr)) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(normals, dn.p, rows * 3 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    return PCPX_OK;
}

int pcpx_debug_sort_keys(const uint64_t* keys, uint64_t n, int first_bit, int device, uint64_t* out_keys)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (n > 0 && (!keys || !out_keys)) return PCPX_ERR_INVALID;
    size_t tb = 0;
    if ((st = sort_keys_u64(nullptr, tb, nullptr, nullptr, n, nullptr, first_bit)) != PCPX_OK) return st;
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    DevBuf ki(shared.pool), ko(shared.pool), tmp(shared.pool);
    if ((st = ki.alloc(n * 8)) != PCPX_OK || (st = ko.alloc(n * 8)) != PCPX_OK || (st = tmp.alloc(tb)) != PCPX_OK) return st;
    if (n > 0) PCPX_HIP(hipMemcpy(ki.p, keys, n * 8, hipMemcpyHostToDevice));
    if ((st = sort_keys_u64(tmp.p, tb, ki.as<u64>(), ko.as<u64>(), n, nullptr, first_bit)) != PCPX_OK) return st;
    PCPX_HIP(hipDeviceSynchronize());
    if (n > 0) {
        u32 failed = 0;
        PCPX_HIP(hipMemcpy(&failed, sort_failure_flag(tmp.p), sizeof(u32), hipMemcpyDeviceToHost));
        if (failed) {
            set_error("pcpx: internal error, the radix sort's look-back gave up");
            return PCPX_ERR_DEVICE;
        }
        PCPX_HIP(hipMemcpy(out_keys, ko.p, n * 8, hipMemcpyDeviceToHost));
    }
    return PCPX_OK;
}

int pcpx_profile_begin(pcpx_index* h)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    for (auto& iv : ix->intervals) {
        (void)hipEventDestroy(iv.a);
        (void)hipEventDestroy(iv.b);
    }
    ix->intervals.clear();
    ix->profiling = true;
    return PCPX_OK;
}

int pcpx_profile_end(pcpx_index* h, pcpx_profile* out)
{
    Index* ix = reinterpret_cast<Index*>(h);
    DeviceScope dscope;
    int st = dscope.use(ix);
    if (st != PCPX_OK) return st;
    std::lock_guard<std::recursive_mutex> serialise(ix->mu);  // one call at a time per handle (queries share its scratch)
    if (!out) return PCPX_ERR_INVALID;
    ix->profiling = false;
    std::memset(out, 0, sizeof(*out));
    PCPX_HIP(hipStreamSynchronize(ix->stream));
    for (auto& iv : ix->intervals) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, iv.a, iv.b) == hipSuccess && iv.family >= 0 && iv.family < PCPX_K_FAMILIES) {
            out->launches[iv.family] += 1;
            out->total_ms[iv.family] += ms;
        }
        (void)hipEventDestroy(iv.a);
        (void)hipEventDestroy(iv.b);
    }
    ix->intervals.clear();
    return PCPX_OK;
}

int pcpx_device_malloc(uint64_t bytes, int device, void** out_ptr)
{
    if (!out_ptr) return PCPX_ERR_INVALID;
    *out_ptr = nullptr;
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    hipError_t e = hipMalloc(out_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        *out_ptr = nullptr;
        (void)hipGetLastError();
        set_error("hipMalloc(%llu bytes) failed: %s", static_cast<unsigned long long>(bytes), hipGetErrorString(e));
        return PCPX_ERR_ALLOC;
    }
    return PCPX_OK;
}
int pcpx_device_trim(int device)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    index_blocks_trim();
    return PCPX_OK;
}

void pcpx_device_free(void* d_ptr, int device)
{
    if (!d_ptr) return;
    DeviceScope dscope;
    if (dscope.select(device) != PCPX_OK) return;
    (void)hipFree(d_ptr);
}
int pcpx_device_upload(void* d_dst, const void* src, uint64_t bytes, int device, void* stream)
{
    if (bytes == 0) return PCPX_OK;
    if (!d_dst || !src) return PCPX_ERR_INVALID;
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if ((st = upload_pageable(d_dst, src, bytes, static_cast<hipStream_t>(stream))) != PCPX_OK) return st;
    PCPX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PCPX_OK;
}
int pcpx_device_download(void* dst, const void* d_src, uint64_t bytes, int device, void* stream)
{
    if (bytes == 0) return PCPX_OK;
    if (!dst || !d_src) return PCPX_ERR_INVALID;
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    PCPX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PCPX_OK;
}

}  // extern "C"

// ---- bilateral filter and WLOP: the in-library consumers of sphere ranges ------------------------

namespace {

// an index these calls build for themselves and drop again (the reference's functions build their own kd-trees too:
// bilateral_filter.hpp:385-389, wlop.hpp:360-363 / :383-387); works on the caller's stream
struct TempIndex {
    Index* ix = nullptr;
    int make(int device, hipStream_t stream)
    {
        ix = new (std::nothrow) Index();
        if (!ix) return PCPX_ERR_ALLOC;
        ix->device = device;
        ix->stream = stream;
        return PCPX_OK;
    }
    ~TempIndex() { free_index(ix); }  // synchronises the stream first: everything enqueued has completed
};

int check_filter_sizes(const char* what, u64 n)
{
    if (n > 0xFFFFFFFFull - 64ull) {
        set_error("%s: more than 2^32 - 65 points", what);
        return PCPX_ERR_UNSUPPORTED;
    }
    return PCPX_OK;
}

// bilateral_filter_points (bilateral_filter.hpp:303-428: p(k+1) = F(p(k)), a new tree over p(k) every iteration) and
// bilateral_filter_normals (:460-574: one tree, n(k+1) from n(k)); everything device resident, stream ordered
int bilateral_device(const float* d_xyz, const float* d_normals, u64 n, double sigmaf_, double sigmag_, u64 iterations,
                     bool normals_mode, int device, hipStream_t stream, float* d_out)
{
    const char* what = normals_mode ? "pcpx_bilateral_filter_normals" : "pcpx_bilateral_filter_points";
    const float sigmaf = static_cast<float>(sigmaf_), sigmag = static_cast<float>(sigmag_);
    if (!(sigmaf > 0.f) || !(sigmag > 0.f)) {  // the reference asserts both (:329-330)
        set_error("%s: sigmaf and sigmag must be positive", what);
        return PCPX_ERR_INVALID;
    }
    if (n == 0) return PCPX_OK;
    if (!d_xyz || !d_normals || !d_out) {
        set_error("%s: null argument", what);
        return PCPX_ERR_INVALID;
    }
    int st = check_filter_sizes(what, n);
    if (st != PCPX_OK) return st;
    const float* first = normals_mode ? d_normals : d_xyz;
    if (iterations == 0) {  // (the reference asserts K > 0; zero iterations of a filter is its input)
        if (d_out != first) PCPX_HIP(hipMemcpyAsync(d_out, first, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
        return check_hip(hipStreamSynchronize(stream), "bilateral sync", __FILE__, __LINE__);
    }
    TempIndex tmp;
    if ((st = tmp.make(device, stream)) != PCPX_OK) return st;
    Index& ix = *tmp.ix;
    DevBuf attr(ix.pool), keep(ix.pool);
    if ((st = attr.alloc(leaf_record_bytes(n, 3))) != PCPX_OK) return st;
    // d_out may alias an input (in-place filtering): the aliased input is then copied first
    const float* src_xyz = d_xyz;
    const float* src_nrm = d_normals;
    if (d_out == d_xyz || d_out == d_normals) {
        if ((st = keep.alloc(n * 3 * sizeof(float))) != PCPX_OK) return st;
        PCPX_HIP(hipMemcpyAsync(keep.p, d_out, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
        if (d_out == d_xyz) src_xyz = keep.as<float>();
        if (d_out == d_normals) src_nrm = keep.as<float>();
    }
    // a centre that is not in the tree (a NaN coordinate) has an empty range: sprime / 0 = NaN in the reference
    bool first_write = true;
    for (u64 it = 0; it < iterations; ++it) {
        if (!normals_mode || it == 0) {
            if ((st = build_index(ix, (!normals_mode && it > 0) ? d_out : src_xyz, n, nullptr)) != PCPX_OK) return st;
        }
        if ((st = launch_leaf_records(ix, (normals_mode && it > 0) ? d_out : src_nrm, 3, attr.p)) != PCPX_OK) return st;
        if (first_write) {
            PCPX_HIP(hipMemsetAsync(d_out, 0xFF, n * 3 * sizeof(float), stream));
            first_write = false;
        }
        if ((st = launch_bilateral(ix, attr.p, sigmaf, sigmag, normals_mode, d_out)) != PCPX_OK) return st;
    }
    return check_hip(hipStreamSynchronize(stream), "bilateral sync", __FILE__, __LINE__);
}

// wlop::wlop (wlop.hpp:287-428) from a given initial sample: x = cloud[sample]; every iteration builds a tree over x,
// (uniform) recomputes the sample densities w_i, and moves every x to median(cloud around x) + repulsion(x around x)
int wlop_device(const float* d_xyz, u64 n, const u64* d_sample, u64 m, double mu_, double h_, u64 iterations, bool uniform, int device,
                hipStream_t stream, float* d_out)
{
    const char* what = "pcpx_wlop";
    const float mu = static_cast<float>(mu_), h = static_cast<float>(h_);
    if (m == 0) return PCPX_OK;
    if (!d_xyz || !d_sample || !d_out || n == 0) {
        set_error("%s: null argument or an empty cloud", what);
        return PCPX_ERR_INVALID;
    }
    if (m > n) {  // the reference asserts I > 0 && J >= I (:309)
        set_error("%s: %llu samples from %llu points", what, static_cast<unsigned long long>(m), static_cast<unsigned long long>(n));
        return PCPX_ERR_INVALID;
    }
    if (!(mu >= 0.f && mu <= 0.5f) || !(h > 0.f)) {  // :310 (and a radius)
        set_error("%s: mu must lie in [0, 0.5] and h must be positive", what);
        return PCPX_ERR_INVALID;
    }
    int st = check_filter_sizes(what, n);
    if (st != PCPX_OK) return st;
    TempIndex cloud, samples;
    if ((st = cloud.make(device, stream)) != PCPX_OK || (st = samples.make(device, stream)) != PCPX_OK) return st;
    Index& P = *cloud.ix;
    Index& Q = *samples.ix;
    DevBuf vj_rows(P.pool), vj_leaf(P.pool), x(Q.pool), med(Q.pool), wi_rows(Q.pool), wi_leaf(Q.pool);
    if ((st = vj_rows.alloc(n * sizeof(float))) != PCPX_OK || (st = vj_leaf.alloc(leaf_record_bytes(n, 1))) != PCPX_OK ||
        (st = x.alloc(m * 3 * sizeof(float))) != PCPX_OK || (st = med.alloc(m * 3 * sizeof(float))) != PCPX_OK ||
        (st = wi_rows.alloc(m * sizeof(float))) != PCPX_OK || (st = wi_leaf.alloc(leaf_record_bytes(m, 1))) != PCPX_OK)
        return st;
    if ((st = build_index(P, d_xyz, n, nullptr)) != PCPX_OK) return st;
    if ((st = launch_fill_f32(vj_rows.as<float>(), n, 1.f, stream)) != PCPX_OK) return st;  // :311 v_j = 1 (LOP keeps it)
    if (uniform) {  // :365-376
        if ((st = launch_leaf_records(P, nullptr, 0, vj_leaf.p)) != PCPX_OK) return st;
        if ((st = launch_wlop_density(P, h, vj_leaf.p, vj_rows.as<float>())) != PCPX_OK) return st;
    }
    if ((st = launch_leaf_records(P, vj_rows.as<float>(), 1, vj_leaf.p)) != PCPX_OK) return st;
    if ((st = launch_take_rows(d_xyz, n, d_sample, m, x.as<float>(), stream)) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(d_out, x.p, m * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));  // :343 xp = x
    for (u64 it = 0; it < iterations; ++it) {
        if ((st = build_index(Q, x.as<float>(), m, nullptr)) != PCPX_OK) return st;
        if ((st = launch_fill_f32(wi_rows.as<float>(), m, 1.f, stream)) != PCPX_OK) return st;
        if (uniform) {  // :389-399
            if ((st = launch_leaf_records(Q, nullptr, 0, wi_leaf.p)) != PCPX_OK) return st;
            if ((st = launch_wlop_density(Q, h, wi_leaf.p, wi_rows.as<float>())) != PCPX_OK) return st;
        }
        if ((st = launch_leaf_records(Q, wi_rows.as<float>(), 1, wi_leaf.p)) != PCPX_OK) return st;
        QueryView qv{};
        if ((st = prepare_queries(P, x.as<float>(), m, qv)) != PCPX_OK) return st;
        if ((st = launch_wlop_median(P, qv, h, vj_leaf.p, med.as<float>())) != PCPX_OK) return st;
        PCPX_HIP(hipMemsetAsync(d_out, 0xFF, m * 3 * sizeof(float), stream));  // a sample with a NaN coordinate stays NaN
        if ((st = launch_wlop_repulsion(Q, h, mu, wi_leaf.p, med.as<float>(), d_out)) != PCPX_OK) return st;
        PCPX_HIP(hipMemcpyAsync(x.p, d_out, m * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));  // :417 x = xp
    }
    return check_hip(hipStreamSynchronize(stream), "wlop sync", __FILE__, __LINE__);
}

// host-pointer form of the two bilateral filters: stage in, run, stage out
int bilateral_host(const float* xyz, const float* normals, u64 n, double sigmaf, double sigmag, u64 iterations, bool normals_mode, int device,
                   float* out)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (n == 0) return PCPX_OK;
    if (!xyz || !normals || !out) {
        set_error("pcpx_bilateral_filter: null argument");
        return PCPX_ERR_INVALID;
    }
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    DevBuf dp(shared.pool), dn(shared.pool), dout(shared.pool);
    const size_t bytes = n * 3 * sizeof(float);
    if ((st = dp.alloc(bytes)) != PCPX_OK || (st = dn.alloc(bytes)) != PCPX_OK || (st = dout.alloc(bytes)) != PCPX_OK) return st;
    PCPX_HIP(hipMemcpyAsync(dp.p, xyz, bytes, hipMemcpyHostToDevice, nullptr));
    PCPX_HIP(hipMemcpyAsync(dn.p, normals, bytes, hipMemcpyHostToDevice, nullptr));
    if ((st = bilateral_device(dp.as<float>(), dn.as<float>(), n, sigmaf, sigmag, iterations, normals_mode, device, nullptr, dout.as<float>())) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, nullptr));
    PCPX_HIP(hipStreamSynchronize(nullptr));
    return PCPX_OK;
}

}  // namespace

extern "C" {

int pcpx_bilateral_filter_points(const float* xyz, const float* normals, uint64_t n, double sigmaf, double sigmag, uint64_t iterations,
                                 int device, float* out_xyz)
{
    return bilateral_host(xyz, normals, n, sigmaf, sigmag, iterations, false, device, out_xyz);
}
int pcpx_bilateral_filter_normals(const float* xyz, const float* normals, uint64_t n, double sigmaf, double sigmag, uint64_t iterations,
                                  int device, float* out_normals)
{
    return bilateral_host(xyz, normals, n, sigmaf, sigmag, iterations, true, device, out_normals);
}
int pcpx_bilateral_filter_points_dev(const float* d_xyz, const float* d_normals, uint64_t n, double sigmaf, double sigmag,
                                     uint64_t iterations, int device, void* stream, float* d_out_xyz)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    return bilateral_device(d_xyz, d_normals, n, sigmaf, sigmag, iterations, false, device, static_cast<hipStream_t>(stream), d_out_xyz);
}
int pcpx_bilateral_filter_normals_dev(const float* d_xyz, const float* d_normals, uint64_t n, double sigmaf, double sigmag,
                                      uint64_t iterations, int device, void* stream, float* d_out_normals)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    return bilateral_device(d_xyz, d_normals, n, sigmaf, sigmag, iterations, true, device, static_cast<hipStream_t>(stream), d_out_normals);
}

int pcpx_wlop_dev(const float* d_xyz, uint64_t n, const uint64_t* d_sample, uint64_t n_samples, double mu, double h, uint64_t iterations,
                  int uniform, int device, void* stream, float* d_out_xyz)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    return wlop_device(d_xyz, n, d_sample, n_samples, mu, h, iterations, uniform != 0, device, static_cast<hipStream_t>(stream), d_out_xyz);
}
int pcpx_wlop(const float* xyz, uint64_t n, const uint64_t* sample, uint64_t n_samples, double mu, double h, uint64_t iterations, int uniform,
              int device, float* out_xyz)
{
    DeviceScope dscope;
    int st = dscope.select(device);
    if (st != PCPX_OK) return st;
    if (n_samples == 0) return PCPX_OK;
    if (!xyz || !sample || !out_xyz || n == 0) {
        set_error("pcpx_wlop: null argument or an empty cloud");
        return PCPX_ERR_INVALID;
    }
    for (u64 i = 0; i < n_samples; ++i)
        if (sample[i] >= n) {
            set_error("pcpx_wlop: sample[%llu] = %llu is not an index into the cloud", static_cast<unsigned long long>(i),
                      static_cast<unsigned long long>(sample[i]));
            return PCPX_ERR_INVALID;
        }
    DeviceShared& shared = shared_of(device);
    std::lock_guard<std::mutex> lock(shared.mu);
    DevBuf dp(shared.pool), ds(shared.pool), dout(shared.pool);
    if ((st = dp.alloc(n * 3 * sizeof(float))) != PCPX_OK || (st = ds.alloc(n_samples * sizeof(u64))) != PCPX_OK ||
        (st = dout.alloc(n_samples * 3 * sizeof(float))) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(dp.p, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, nullptr));
    PCPX_HIP(hipMemcpyAsync(ds.p, sample, n_samples * sizeof(u64), hipMemcpyHostToDevice, nullptr));
    if ((st = wlop_device(dp.as<float>(), n, ds.as<u64>(), n_samples, mu, h, iterations, uniform != 0, device, nullptr, dout.as<float>())) != PCPX_OK)
        return st;
    PCPX_HIP(hipMemcpyAsync(out_xyz, dout.p, n_samples * 3 * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    PCPX_HIP(hipStreamSynchronize(nullptr));
    return PCPX_OK;
}

int pcpx_shard_range(uint64_t n, uint32_t rank, uint32_t world, uint64_t* out_first, uint64_t* out_count)
{
    if (!out_first || !out_count || world == 0 || rank >= world) return PCPX_ERR_INVALID;
    u64 groups = (n + GROUP - 1) / GROUP;
    u64 g0 = groups * rank / world, g1 = groups * (static_cast<u64>(rank) + 1) / world;
    u64 first = g0 * GROUP, end = g1 * GROUP;
    if (first > n) first = n;
    if (end > n) end = n;
    *out_first = first;
    *out_count = end - first;
    return PCPX_OK;
}

}  // extern "C"
