// tools/pcie_rate.hip -- what the host-pointer ABI can hope for on this box: D2H / H2D rates for pageable memory,
// hipHostRegister'ed user memory (registration cost included and excluded) and a pinned bounce buffer, at the sizes the
// host-pointer entry points move (120 MB normals, 600 MB rows).  Build: hipcc -O2 --offload-arch=gfx950 tools/pcie_rate.hip -o /tmp/pcie_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) {                                                             \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(char* dst, const char* src, size_t n, int threads)
{
    std::vector<std::thread> th;
    size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t a = t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        th.emplace_back([=] { std::memcpy(dst + a, src + a, b - a); });
    }
    for (auto& t : th) t.join();
}

int main()
{
    const size_t sizes[] = {120ull << 20, 600ull << 20};
    std::printf("{\"results\": [\n");
    bool first = true;
    for (size_t bytes : sizes) {
        void* d = nullptr;
        CK(hipMalloc(&d, bytes));
        CK(hipMemset(d, 1, bytes));
        char* h = static_cast<char*>(std::aligned_alloc(4096, bytes));
        std::memset(h, 0, bytes);  // touch
        hipStream_t s;
        CK(hipStreamCreate(&s));
        auto emit = [&](const char* what, double sec) {
            std::printf("%s  {\"bytes\": %zu, \"case\": \"%s\", \"ms\": %.3f, \"GBps\": %.2f}", first ? "" : ",\n", bytes, what, sec * 1e3,
                        bytes / sec / 1e9);
            first = false;
        };
        for (int rep = 0; rep < 2; ++rep) {  // second repetition reported (first warms page tables)
            double t0 = now();
            CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
            double dt = now() - t0;
            if (rep) emit("d2h pageable hipMemcpy", dt);
            t0 = now();
            CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
            dt = now() - t0;
            if (rep) emit("h2d pageable hipMemcpy", dt);
        }
        {
            double t0 = now();
            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
            double treg = now() - t0;
            t0 = now();
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            double tcopy = now() - t0;
            t0 = now();
            CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            double tcopy2 = now() - t0;
            t0 = now();
            CK(hipHostUnregister(h));
            double tun = now() - t0;
            emit("hipHostRegister", treg);
            emit("d2h registered", tcopy);
            emit("h2d registered", tcopy2);
            emit("hipHostUnregister", tun);
            emit("d2h register+copy+unregister", treg + tcopy + tun);
        }
        {   // pinned bounce buffers (2 x 16 MB), D2H chunk c overlapped with the host memcpy of chunk c-1
            const size_t chunk = 16ull << 20;
            char* p[2];
            CK(hipHostMalloc(reinterpret_cast<void**>(&p[0]), chunk, hipHostMallocDefault));
            CK(hipHostMalloc(reinterpret_cast<void**>(&p[1]), chunk, hipHostMallocDefault));
            hipEvent_t ev[2];
            CK(hipEventCreate(&ev[0]));
            CK(hipEventCreate(&ev[1]));
            for (int threads : {1, 4, 8}) {
                double t0 = now();
                size_t nchunks = (bytes + chunk - 1) / chunk;
                for (size_t c = 0; c <= nchunks; ++c) {
                    if (c < nchunks) {
                        size_t off = c * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
                        CK(hipMemcpyAsync(p[c & 1], static_cast<char*>(d) + off, len, hipMemcpyDeviceToHost, s));
                        CK(hipEventRecord(ev[c & 1], s));
                    }
                    if (c > 0) {
                        size_t off = (c - 1) * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
                        CK(hipEventSynchronize(ev[(c - 1) & 1]));
                        par_memcpy(h + off, p[(c - 1) & 1], len, threads);
                    }
                }
                double dt = now() - t0;
                char name[64];
                std::snprintf(name, sizeof name, "d2h pinned bounce 16MB x2, %d memcpy threads", threads);
                emit(name, dt);
            }
            CK(hipHostFree(p[0]));
            CK(hipHostFree(p[1]));
        }
        {   // all-pinned reference: the line rate
            char* hp;
            CK(hipHostMalloc(reinterpret_cast<void**>(&hp), bytes, hipHostMallocDefault));
            double t0 = now();
            CK(hipMemcpyAsync(hp, d, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            emit("d2h hipHostMalloc (line rate)", now() - t0);
            t0 = now();
            CK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            emit("h2d hipHostMalloc (line rate)", now() - t0);
            CK(hipHostFree(hp));
        }
        std::free(h);
        CK(hipFree(d));
        CK(hipStreamDestroy(s));
    }
    std::printf("\n]}\n");
    return 0;
}
