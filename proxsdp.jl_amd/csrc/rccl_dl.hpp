// RCCL entry points, loaded at run time (dlopen): the library does not LINK librccl, a process that never
// hands over a communicator never loads it, and a box without RCCL can still run single-GPU solves.
// Used by the block-sharded solve: the packed scalar record of
// every PDHG iteration and the coupling rows of M x are reduced by the library itself, on its own stream,
// over the communicator the caller created (proxsdp_problem.nccl_comm) -- xGMI, no host callback.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enums only; every call goes through the table below

#include <mutex>
#include <stdexcept>
#include <string>

namespace proxsdp {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;     // optional: releases THIS rank's pending collectives
    std::string load_error;

    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, []() { r.load(); });
        return r;
    }
    bool ok() const { return handle != nullptr; }
    void require() const {
        if (!ok()) throw std::runtime_error("librccl could not be loaded: " + load_error);
    }
    void check(ncclResult_t rc, const char* what) const {
        if (rc != ncclSuccess)
            throw std::runtime_error(std::string(what) + ": " + (GetErrorString ? GetErrorString(rc) : "RCCL error"));
    }

private:
    template <typename F>
    bool sym(F& f, const char* name) {
        f = reinterpret_cast<F>(dlsym(handle, name));
        if (f == nullptr) { load_error = std::string("missing symbol ") + name; return false; }
        return true;
    }
    void load() {
        // a librccl already in the process (e.g. the one a host framework loaded) is preferred: one RCCL per process
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (handle) break;
            }
        if (!handle) { const char* e = dlerror(); load_error = e ? e : "dlopen failed"; return; }
        const bool all = sym(GetUniqueId, "ncclGetUniqueId") && sym(CommInitRank, "ncclCommInitRank") &&
                         sym(CommDestroy, "ncclCommDestroy") && sym(CommCount, "ncclCommCount") &&
                         sym(CommUserRank, "ncclCommUserRank") && sym(AllReduce, "ncclAllReduce") &&
                         sym(AllGather, "ncclAllGather") && sym(GetErrorString, "ncclGetErrorString");
        if (!all) handle = nullptr;       // (left loaded; the table is unusable)
        else CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(handle, "ncclCommAbort"));
    }
};

}  // namespace proxsdp
