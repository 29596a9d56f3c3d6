// comm.hpp — the one exchange step of the multi-GPU path: an RCCL all-gather of decoded label ids over xGMI.
//
// The reference is single-device (no torch.distributed / DataParallel anywhere); pages shard by
// the reference's own chunks (pero_ocr/ocr_engine/line_ocr_engine.py:79-90 - independent forwards), one process per
// GPU, and the only data every rank needs from the others is the decoded text.  Every rank
// derives the same chunk plan from the same widths, so the payload geometry (lines per rank, longest label row) is
// known everywhere in advance: ONE fixed-stride ncclAllGather per page stream, no size exchange.
//
// librccl.so is opened at run time (dlopen) by pocr_comm_unique_id / pocr_comm_init: the single-GPU path neither
// links nor loads it.  The rendezvous of the 128-byte unique id is the caller's business (any out-of-band channel:
// the host side uses a TCP socket, pero_ocr_amd/sharding.py).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

namespace pocr {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    std::string err;

    // 0 on success.  POCR_RCCL_LIB overrides the library name.  The ROCm installation this library was built against comes
    // first, by path: a process that has imported PyTorch already holds PyTorch's own bundled librccl.so (same SONAME)
    // together with a second, uninitialised HSA runtime - that copy fails in ncclCommInitRank with "no ROCm-capable device".
    int load() {
        if (lib) return 0;
        const char *names[] = {getenv("POCR_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
            err = dlerror();
        }
        if (!lib) { err = "cannot open librccl: " + err; return 1; }
        bool ok = true;
        auto sym = [&](const char *nm) { void *p = dlsym(lib, nm); if (!p) { ok = false; err = std::string("librccl lacks ") + nm; } return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        GetVersion = reinterpret_cast<decltype(GetVersion)>(sym("ncclGetVersion"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        CommUserRank = reinterpret_cast<decltype(CommUserRank)>(sym("ncclCommUserRank"));
        if (!ok) { dlclose(lib); lib = nullptr; return 1; }
        return 0;
    }
};

inline RcclApi &rccl() {
    static RcclApi api;
    return api;
}

// Per-engine communicator state: one RCCL communicator, its own stream (the exchange overlaps the engine's kernels),
// device send/receive buffers and pinned host mirrors.
struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    hipStream_t stream = nullptr;
    void *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    bool active() const { return comm != nullptr; }
};

}  // namespace pocr
