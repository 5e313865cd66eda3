// pcpx_comm.hip -- the multi-GPU step of the path behind the C ABI: ONE collective, an RCCL all-gather of the per-rank
// bounding boxes (6 floats = 24 B per rank over xGMI), so that every rank quantises curve keys on the same grid
// (BASELINE.json north_star: "an RCCL all-gather of per-rank bounding boxes over xGMI and nothing
// else").  One process per GPU; every rank then builds the same index on the union box and answers its shard of the
// curve-sorted queries (pcpx_shard_range).  Round 1 had this only in Python (torch.distributed in bench.py): a C++ host
// of the drop-in headers had no multi-GPU path.
//
// librccl.so.1 is bound at the first pcpx_comm_* call (dlopen), not at load time: single-GPU users of libpcpx.so neither
// need nor load RCCL, and in a process that already has an RCCL (PyTorch ships one under the same soname) that copy is
// the one the loader returns.
#include "pcpx_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>
#include <string>

namespace pcpx {

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;  // the loader's message, captured where the load failed
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            const char* e = dlerror();  // (one call: it also clears the message)
            if (e) r.why = e;
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
        if (!r.ok) r.why = "symbols missing";
    });
    return r;
}

int need_rccl()
{
    if (rccl().ok) return PCPX_OK;
    set_error("pcpx: librccl.so.1 could not be loaded (%s)", rccl().why.empty() ? "unknown reason" : rccl().why.c_str());
    return PCPX_ERR_UNSUPPORTED;
}

int check_nccl(ncclResult_t r, const char* what)
{
    if (r == ncclSuccess) return PCPX_OK;
    set_error("RCCL error %d (%s) in %s", static_cast<int>(r), rccl().GetErrorString(r), what);
    return PCPX_ERR_DEVICE;
}

struct Comm {
    ncclComm_t comm = nullptr;
    bool owned = false;
    int world = 1, rank = 0, device = 0;
    float* d_work = nullptr;  // [0, 16): bbox scratch (encoded + decoded), [16, 16 + 6 world): gathered boxes, then 6: the union
    std::mutex mu;
};

// union of `world` boxes {min xyz, max xyz}
__global__ void k_union_boxes(const float* __restrict__ all, int world, float* __restrict__ out6)
{
    const int a = threadIdx.x;
    if (a >= 6) return;
    float r = all[a];
    for (int w = 1; w < world; ++w) r = a < 3 ? fminf(r, all[6 * w + a]) : fmaxf(r, all[6 * w + a]);
    out6[a] = r;
}

int make_comm(ncclComm_t c, bool owned, int world, int rank, int device, Comm** out)
{
    Comm* cm = new (std::nothrow) Comm();
    if (!cm) return PCPX_ERR_ALLOC;
    cm->comm = c;
    cm->owned = owned;
    cm->world = world;
    cm->rank = rank;
    cm->device = device;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&cm->d_work), (16 + 6 * static_cast<size_t>(world) + 6 + 64) * sizeof(float));
    if (e != hipSuccess) {
        delete cm;
        return check_hip(e, "hipMalloc (communicator scratch)", __FILE__, __LINE__);
    }
    *out = cm;
    return PCPX_OK;
}

struct DeviceGuard {  // make `device` current for the call, restore the caller's on return
    int prev = -1;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void)hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace
}  // namespace pcpx

using namespace pcpx;

extern "C" {

int pcpx_comm_unique_id(char out_id[PCPX_COMM_ID_BYTES])
{
    static_assert(PCPX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!out_id) return PCPX_ERR_INVALID;
    int st = need_rccl();
    if (st != PCPX_OK) return st;
    ncclUniqueId id;
    if ((st = check_nccl(rccl().GetUniqueId(&id), "ncclGetUniqueId")) != PCPX_OK) return st;
    std::memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return PCPX_OK;
}

int pcpx_comm_init_rank(const char id[PCPX_COMM_ID_BYTES], int world, int rank, int device, pcpx_comm** out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return PCPX_ERR_INVALID;
    *out = nullptr;
    int st = need_rccl();
    if (st != PCPX_OK) return st;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        (void)hipGetLastError();
        set_error("pcpx_comm_init_rank: device %d is not available", device);
        return PCPX_ERR_DEVICE;
    }
    DeviceGuard guard(device);
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    if ((st = check_nccl(rccl().CommInitRank(&c, world, uid, rank), "ncclCommInitRank")) != PCPX_OK) return st;
    Comm* cm = nullptr;
    if ((st = make_comm(c, true, world, rank, device, &cm)) != PCPX_OK) {
        (void)rccl().CommDestroy(c);
        return st;
    }
    *out = reinterpret_cast<pcpx_comm*>(cm);
    return PCPX_OK;
}

int pcpx_comm_wrap(void* nccl_comm, int world, int rank, int device, pcpx_comm** out)
{
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return PCPX_ERR_INVALID;
    *out = nullptr;
    int st = need_rccl();
    if (st != PCPX_OK) return st;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        (void)hipGetLastError();
        set_error("pcpx_comm_wrap: device %d is not available", device);
        return PCPX_ERR_DEVICE;
    }
    DeviceGuard guard(device);
    Comm* cm = nullptr;
    if ((st = make_comm(static_cast<ncclComm_t>(nccl_comm), false, world, rank, device, &cm)) != PCPX_OK) return st;
    *out = reinterpret_cast<pcpx_comm*>(cm);
    return PCPX_OK;
}

void pcpx_comm_destroy(pcpx_comm* h)
{
    Comm* cm = reinterpret_cast<Comm*>(h);
    if (!cm) return;
    DeviceGuard guard(cm->device);
    (void)hipDeviceSynchronize();
    if (cm->owned && cm->comm && rccl().ok) (void)rccl().CommDestroy(cm->comm);
    (void)hipFree(cm->d_work);
    delete cm;
}

int pcpx_comm_allgather_boxes_dev(pcpx_comm* h, const float* d_local6, float* d_all, void* stream)
{
    Comm* cm = reinterpret_cast<Comm*>(h);
    if (!cm || !d_local6 || !d_all) return PCPX_ERR_INVALID;
    DeviceGuard guard(cm->device);
    std::lock_guard<std::mutex> lock(cm->mu);
    return check_nccl(rccl().AllGather(d_local6, d_all, 6, ncclFloat32, cm->comm, static_cast<hipStream_t>(stream)), "ncclAllGather");
}

int pcpx_comm_global_grid_dev(pcpx_comm* h, const float* d_xyz_slice, uint64_t n_slice, void* stream, float out_grid6[6])
{
    Comm* cm = reinterpret_cast<Comm*>(h);
    if (!cm || !out_grid6 || (n_slice > 0 && !d_xyz_slice)) return PCPX_ERR_INVALID;
    DeviceGuard guard(cm->device);
    std::lock_guard<std::mutex> lock(cm->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* d_local = cm->d_work + 8;   // decoded box of this rank's slice
    float* d_all = cm->d_work + 16;
    float* d_union = d_all + 6 * cm->world;
    // an empty slice contributes the empty box {+FLT_MAX.., -FLT_MAX..}: the identity of the union (axis_aligned_bounding_box.hpp:214-251)
    int st = device_bbox(d_xyz_slice, n_slice, s, reinterpret_cast<u32*>(cm->d_work), d_local);
    if (st != PCPX_OK) return st;
    if ((st = check_nccl(rccl().AllGather(d_local, d_all, 6, ncclFloat32, cm->comm, s), "ncclAllGather")) != PCPX_OK) return st;
    k_union_boxes<<<1, 64, 0, s>>>(d_all, cm->world, d_union);
    PCPX_HIP(hipGetLastError());
    PCPX_HIP(hipMemcpyAsync(out_grid6, d_union, 6 * sizeof(float), hipMemcpyDeviceToHost, s));
    PCPX_HIP(hipStreamSynchronize(s));
    return PCPX_OK;
}

}  // extern "C"
