// comm.cpp -- the ONE exchange the multi-GPU path has (BASELINE.json north_star: "RCCL over
// xGMI used only to gather per-rank compressed sizes/offsets"), behind the C-ABI: one ncclAllGather of 8 bytes
// per rank, in-stream behind the compaction pass that produces the count, so that the write path
// (encode -> compact -> gather) is one queue of device work with no host round trip inside it.
//
// RCCL is resolved at run time (dlopen of librccl.so.1 by SONAME: inside a torch process that is the copy
// torch.distributed's "nccl" backend already loaded, so both share one set of xGMI rings) -- a single-GPU user of
// the codec never pays for loading it, and the library has no link-time dependency on it.
#include "../../include/sprintz_mi355x.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace sprintz { int set_error(int code, const char* what); }

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

const Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // SPRINTZ_MI355X_RCCL_SONAME names the one library to try instead of the default list (a site whose RCCL lives
        // elsewhere; also how tests/test_abi.py makes the load fail on purpose)
        const char* forced = getenv("SPRINTZ_MI355X_RCCL_SONAME");
        std::string last = "?";
        if (forced && *forced) {
            r.handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
            if (!r.handle) { const char* e = dlerror(); if (e) last = e; }      // dlerror() clears what it returns: ask once
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.handle) break;
                const char* e = dlerror();
                if (e) last = e;
            }
        }
        if (!r.handle) { r.why = std::string("RCCL not loadable: ") + last; return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
            r.why = "RCCL loaded but a symbol is missing";
            r.handle = nullptr;
        }
    });
    return r;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int nccl_fail(const char* what, ncclResult_t e)
{
    static thread_local std::string msg;
    msg = what;
    const Rccl& r = rccl();
    if (r.GetErrorString) { msg += ": "; msg += r.GetErrorString(e); }
    return sprintz::set_error(SPRINTZ_E_HIP, msg.c_str());
}

}  // namespace

extern "C" {

int sprintz_mi355x_comm_unique_id(void* id_out)
{
    if (!id_out) return sprintz::set_error(SPRINTZ_E_INVALID, "comm_unique_id: null pointer");
    const Rccl& r = rccl();
    if (!r.handle) return sprintz::set_error(SPRINTZ_E_UNSUPPORTED, r.why.c_str());
    static_assert(sizeof(ncclUniqueId) == SPRINTZ_MI355X_COMM_ID_BYTES, "id size");
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int sprintz_mi355x_comm_init(const void* id_in, int rank, int world, void** comm_out)
{
    if (!id_in || !comm_out || world < 1 || rank < 0 || rank >= world) return sprintz::set_error(SPRINTZ_E_INVALID, "comm_init: bad argument");
    const Rccl& r = rccl();
    if (!r.handle) return sprintz::set_error(SPRINTZ_E_UNSUPPORTED, r.why.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "comm_init: no usable HIP device (there is no CPU fallback)");
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);     // binds to the CURRENT HIP device
    if (e != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", e); }
    *comm_out = c;
    return 0;
}

int sprintz_mi355x_gather_layout(void* comm, const uint64_t* d_local_total, uint64_t* d_all, void* hip_stream)
{
    if (!comm || !d_local_total || !d_all) return sprintz::set_error(SPRINTZ_E_INVALID, "gather_layout: null pointer");
    Comm* c = (Comm*)comm;
    const ncclResult_t e = rccl().AllGather(d_local_total, d_all, 1, ncclUint64, c->comm, (hipStream_t)hip_stream);
    if (e != ncclSuccess) return nccl_fail("ncclAllGather", e);
    return 0;
}

int sprintz_mi355x_layout_bases(const uint64_t* d_all, int world, uint64_t* bases_out, void* hip_stream)
{
    if (!d_all || !bases_out || world < 1 || world > 4096) return sprintz::set_error(SPRINTZ_E_INVALID, "layout_bases: bad argument");
    uint64_t counts[4096];
    if (hipMemcpyAsync(counts, d_all, (size_t)world * 8, hipMemcpyDeviceToHost, (hipStream_t)hip_stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess)
        return sprintz::set_error(SPRINTZ_E_HIP, "layout_bases: reading the gathered counts back failed");
    uint64_t acc = 0;
    for (int r = 0; r < world; r++) { bases_out[r] = acc; acc += counts[r]; }
    bases_out[world] = acc;
    return 0;
}

int sprintz_mi355x_comm_destroy(void* comm)
{
    if (!comm) return 0;
    Comm* c = (Comm*)comm;
    const Rccl& r = rccl();
    if (r.handle && c->comm) (void)r.CommDestroy(c->comm);
    delete c;
    return 0;
}

}  // extern "C"
