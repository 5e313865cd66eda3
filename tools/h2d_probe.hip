// What a host -> device copy of a cloud costs by size and by kind of host memory (pageable / pinned), and what hipMalloc / hipFree /
// stream creation cost: the pieces of pcpx_index_create from a host array.  build: hipcc -O2 --offload-arch=gfx950 tools/h2d_probe.hip -o /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    const size_t sizes[] = {size_t(1) << 20, size_t(12) << 20, size_t(48) << 20, size_t(200) << 20};
    std::printf("{");
    for (size_t bytes : sizes) {
        std::vector<char> host(bytes, 1);
        void* d = nullptr;
        double t0 = now();
        hipMalloc(&d, bytes);
        double t_malloc = now() - t0;
        hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice);  // warm
        t0 = now();
        for (int i = 0; i < 5; ++i) hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice);
        double t_pageable = (now() - t0) / 5;
        void* pin = nullptr;
        t0 = now();
        hipHostMalloc(&pin, bytes, hipHostMallocDefault);
        double t_hostmalloc = now() - t0;
        std::memset(pin, 1, bytes);
        hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice);
        t0 = now();
        for (int i = 0; i < 5; ++i) hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice);
        double t_pinned = (now() - t0) / 5;
        t0 = now();
        std::memcpy(pin, host.data(), bytes);
        double t_memcpy = now() - t0;
        t0 = now();
        hipHostFree(pin);
        double t_hostfree = now() - t0;
        t0 = now();
        hipFree(d);
        double t_free = now() - t0;
        std::printf("\"%zu_MB\": {\"hipMalloc_ms\": %.3f, \"hipFree_ms\": %.3f, \"h2d_pageable_ms\": %.3f, \"h2d_pinned_ms\": %.3f, \"host_memcpy_ms\": %.3f, "
                    "\"hipHostMalloc_ms\": %.3f, \"hipHostFree_ms\": %.3f}, ",
                    bytes >> 20, t_malloc * 1e3, t_free * 1e3, t_pageable * 1e3, t_pinned * 1e3, t_memcpy * 1e3, t_hostmalloc * 1e3, t_hostfree * 1e3);
    }
    hipStream_t s;
    double t0 = now();
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t_sc = now() - t0;
    t0 = now();
    hipStreamDestroy(s);
    double t_sd = now() - t0;
    std::printf("\"hipStreamCreate_ms\": %.3f, \"hipStreamDestroy_ms\": %.3f}\n", t_sc * 1e3, t_sd * 1e3);
    return 0;
}
