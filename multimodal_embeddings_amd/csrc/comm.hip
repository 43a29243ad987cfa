// The path's ONE collective behind the C ABI: an all-gather of the [rows, d] bf16 embedding shards over RCCL
// (xGMI), for binders that do not go through torch.distributed.
//
// libmme.so carries no link-time dependency on RCCL (the single-GPU path must load on a box without it): the
// five entry points used here are resolved at first use from the RCCL that is already in the process (PyTorch
// ships and loads its own librccl.so) or else from librccl.so.1 on the loader path; MME_RCCL_LIB overrides.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>

#include "kernels.h"

struct Id128 {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char bytes[128];
};

namespace {

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* env = getenv("MME_RCCL_LIB");
    const char* names[] = {env, "librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !g_rccl.lib; ++pass)  // pass 0: only what the process already holds
        for (const char* n : names) {
            if (!n || !*n) continue;
            g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (g_rccl.lib) break;
        }
    if (!g_rccl.lib) {
        g_rccl.error = "RCCL not found (tried the loaded process image, librccl.so, librccl.so.1; set MME_RCCL_LIB)";
        return;
    }
    auto sym = [&](const char* name) {
        void* p = dlsym(g_rccl.lib, name);
        if (!p && g_rccl.error.empty()) g_rccl.error = std::string("RCCL symbol missing: ") + name;
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}

}  // namespace

// returns nullptr when RCCL is usable, else a message
const char* rccl_ready() {
    std::call_once(g_once, load_rccl);
    return g_rccl.error.empty() ? nullptr : g_rccl.error.c_str();
}
const char* rccl_error_string(int code) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "unknown RCCL error"; }
int rccl_unique_id(void* id128) { return g_rccl.GetUniqueId(id128); }
int rccl_comm_init(void** comm, int world, const void* id128, int rank) {
    Id128 id;
    __builtin_memcpy(id.bytes, id128, 128);
    return g_rccl.CommInitRank(comm, world, id, rank);
}
int rccl_comm_destroy(void* comm) { return g_rccl.CommDestroy(comm); }
int rccl_allgather_bytes(const void* send, void* recv, size_t bytes, void* comm, hipStream_t s) {
    return g_rccl.AllGather(send, recv, bytes, /* ncclUint8 */ 1, comm, s);
}
