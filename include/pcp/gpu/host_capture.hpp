// pcp/gpu/host_capture.hpp -- host-side helpers of the container constructors: the coordinate buffer that feeds the device
// index, and a chunked std::thread loop for evaluating a property map over a large element range.
//
// The reference's constructors walk the range once on one thread (include/pcp/octree/linked_octree.hpp:83-121,
// include/pcp/kdtree/linked_kdtree.hpp:100-135) because they build a pointer tree while they walk.  Here the walk only
// evaluates the property map and stores three floats per element -- the tree is built on the GPU in about a millisecond -- so
// the walk itself was the whole construction time (round 3: 233-286 ms of a 2^24-point construction).  It is split over the
// host's cores; property maps are pure functions of the element (the reference itself calls them from PSTL worker threads,
// include/pcp/algorithm/estimate_normals.hpp:92).
#ifndef PCP_GPU_HOST_CAPTURE_HPP
#define PCP_GPU_HOST_CAPTURE_HPP

#include <cstddef>
#include <cstdlib>
#include <memory>
#include <new>
#include <exception>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#if defined(__linux__)
#include <sys/mman.h>
#endif

namespace pcp {
namespace gpu {

// Memory of the containers' big arrays.  From 4 MB on: 2-MB aligned and advised for transparent huge pages (Linux, where the
// system allows it on advice) -- a 2^24-point container is 400 MB that is touched once, copied to the device and freed, and with
// 4-KB pages the page faults of filling it and the unmapping at destruction (100 000 pages: 60 ms) cost more than the work.
inline void* array_memory(std::size_t bytes)
{
#if defined(__linux__)
    if (bytes >= (std::size_t(4) << 20))
    {
        std::size_t const huge = std::size_t(2) << 20;
        std::size_t const len  = (bytes + huge - 1) / huge * huge;
        if (void* p = std::aligned_alloc(huge, len))
        {
            (void)::madvise(p, len, MADV_HUGEPAGE);
            return p;
        }
    }
#endif
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}

// std::allocator whose value-less construct() default-initialises: resize() of a vector<float, ...> does not write zeros over
// memory that the capture threads are about to fill (a 2^24-point cloud: 200 MB, first touched by the threads that fill it).
template <class T>
struct default_init_allocator : std::allocator<T>
{
    using std::allocator<T>::allocator;
    template <class U>
    struct rebind
    {
        using other = default_init_allocator<U>;
    };
    T* allocate(std::size_t n) { return static_cast<T*>(array_memory(n * sizeof(T))); }
    void deallocate(T* p, std::size_t) noexcept { std::free(p); }
    template <class U, class... Args>
    void construct(U* p, Args&&... args)
    {
        if constexpr (sizeof...(Args) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
};

// Element types whose copies the capture threads may construct in place, each in its own piece of the container's storage:
// a copy cannot throw half way through the range and an element that was never constructed needs no destructor.
template <class T>
constexpr bool constructible_in_pieces = std::is_nothrow_copy_constructible_v<T> && std::is_trivially_destructible_v<T>;

// Allocator of the containers' element storage.  For such element types the value-less construct() that resize() calls does
// NOTHING: resize(size() + n) only claims n slots, and the caller constructs every one of them (placement new, on several
// threads) before anything reads them.  A single thread copying a 2^24-element range into fresh memory spends most of its time
// in page faults; the pieces are first touched by the threads that fill them instead.  For any other element type this is
// std::allocator and the containers copy the range with vector::insert.
template <class T>
struct piecewise_allocator : std::allocator<T>
{
    using std::allocator<T>::allocator;
    template <class U>
    struct rebind
    {
        using other = piecewise_allocator<U>;
    };
    T* allocate(std::size_t n) { return static_cast<T*>(array_memory(n * sizeof(T))); }
    void deallocate(T* p, std::size_t) noexcept { std::free(p); }
    template <class U, class... Args>
    void construct(U* p, Args&&... args)
    {
        if constexpr (sizeof...(Args) == 0 && constructible_in_pieces<U>) (void)p;
        else ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using element_storage_t = std::vector<T, piecewise_allocator<T>>;

// n x 3 floats, point i at [3 i, 3 i + 3): what pcpx_index_create / pcpx_index_rebuild read
using coord_buffer_t = std::vector<float, default_init_allocator<float>>;

// ranges shorter than this are walked on the calling thread
constexpr std::size_t parallel_capture_threshold = 32768;

// THREAD SAFETY OF THE PROPERTY MAP.  The reference's constructors walk the range on the calling thread
// (include/pcp/octree/linked_octree.hpp:103-121); here a range of parallel_capture_threshold elements or more is walked by several
// threads, each on a contiguous piece, so the CoordinateMap / PointViewMap and the element's copy constructor are called
// CONCURRENTLY for different elements.  A pure map (the usual lambda returning the element's coordinates) is fine; a map that
// caches, counts or logs through unsynchronised shared state is not.  Opt out: define PCP_CAPTURE_THREADS to 1 before including
// the headers, or set the environment variable PCPX_CAPTURE_THREADS=1 (any positive number caps the thread count).
#ifndef PCP_CAPTURE_THREADS
#define PCP_CAPTURE_THREADS 0  // 0: by range size and std::thread::hardware_concurrency()
#endif
inline unsigned capture_thread_limit()
{
    static unsigned const limit = [] {
        if (PCP_CAPTURE_THREADS > 0) return static_cast<unsigned>(PCP_CAPTURE_THREADS);
        char const* e = std::getenv("PCPX_CAPTURE_THREADS");
        long const v = e ? std::strtol(e, nullptr, 10) : 0;
        return v > 0 ? static_cast<unsigned>(v) : 0u;
    }();
    return limit;
}

inline unsigned capture_threads(std::size_t n)
{
    unsigned hw = std::thread::hardware_concurrency();
    if (capture_thread_limit() != 0 && (hw == 0 || hw > capture_thread_limit())) hw = capture_thread_limit();
    if (hw == 0) hw = 1;
    if (hw > 32) hw = 32;
    std::size_t const by_size = n / (parallel_capture_threshold / 2);
    return by_size < 2 ? 1u : (by_size < hw ? static_cast<unsigned>(by_size) : hw);
}

// f(first, last, chunk) for `chunks` contiguous pieces of [0, n), chunk 0 on the calling thread after `meanwhile()` has run
// there (work that overlaps the others' pieces), the rest on threads of their own; returns when all are done.
template <class F, class G>
inline void parallel_chunks(std::size_t n, unsigned chunks, F&& f, G&& meanwhile)
{
    if (chunks <= 1)
    {
        meanwhile();
        f(std::size_t{0}, n, 0u);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(chunks - 1);
    auto piece = [n, chunks](unsigned c) { return static_cast<std::size_t>((static_cast<unsigned long long>(n) * c) / chunks); };
    // an exception thrown by a piece (the caller's property map, an element's copy) reaches the caller as it would from the
    // one-by-one loop: the first one caught is rethrown after every thread has been joined
    std::exception_ptr failed;
    std::mutex failed_mu;
    auto guarded = [&](std::size_t first, std::size_t last, unsigned c) {
        try
        {
            f(first, last, c);
        }
        catch (...)
        {
            std::lock_guard<std::mutex> lock(failed_mu);
            if (!failed) failed = std::current_exception();
        }
    };
    unsigned started = 1;  // chunks [1, started) have a thread
    try
    {
        for (; started < chunks; ++started) pool.emplace_back([&guarded, piece, c = started] { guarded(piece(c), piece(c + 1), c); });
    }
    catch (...)  // no more threads to be had: the calling thread takes the pieces that are left
    {
    }
    try
    {
        meanwhile();
    }
    catch (...)
    {
        std::lock_guard<std::mutex> lock(failed_mu);
        if (!failed) failed = std::current_exception();
    }
    guarded(piece(0), piece(1), 0u);
    for (unsigned c = started; c < chunks; ++c) guarded(piece(c), piece(c + 1), c);
    for (auto& t : pool) t.join();
    if (failed) std::rethrow_exception(failed);
}
template <class F>
inline void parallel_chunks(std::size_t n, unsigned chunks, F&& f)
{
    parallel_chunks(n, chunks, std::forward<F>(f), [] {});
}

} // namespace gpu
} // namespace pcp

#endif
