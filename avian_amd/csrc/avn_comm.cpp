// Library transport of the level-2 halo exchange: RCCL point-to-point over xGMI (include/avian_mi355x.h, "level-2 sharding").
// librccl is opened lazily by avn_comm_unique_id / avn_comm_init: a world that never shards loads nothing, and the library itself has
// no link-time dependency on RCCL (single-GPU hosts without it still load libavian_mi355x.so).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "avn_world.hpp"

namespace avn {

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
Rccl g_rccl;
std::once_flag g_once;

template <class F> bool sym(F& f, const char* name) {
    f = (F)dlsym(g_rccl.lib, name);
    if (!f) g_rccl.why = std::string("librccl: missing symbol ") + name;
    return f != nullptr;
}
void load_once() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) { g_rccl.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    bool ok = sym(g_rccl.GetUniqueId, "ncclGetUniqueId") && sym(g_rccl.CommInitRank, "ncclCommInitRank") && sym(g_rccl.CommDestroy, "ncclCommDestroy") &&
              sym(g_rccl.GroupStart, "ncclGroupStart") && sym(g_rccl.GroupEnd, "ncclGroupEnd") && sym(g_rccl.Send, "ncclSend") && sym(g_rccl.Recv, "ncclRecv") && sym(g_rccl.AllGather, "ncclAllGather") &&
              sym(g_rccl.GetErrorString, "ncclGetErrorString");
    if (!ok) { dlclose(g_rccl.lib); g_rccl.lib = nullptr; }
}
bool loaded(std::string& err) {
    std::call_once(g_once, load_once);
    if (!g_rccl.lib) { err = g_rccl.why; return false; }
    return true;
}
avn_status fail(std::string& err, const char* what, ncclResult_t r) {
    err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
    return AVN_ERR_HIP;
}
}  // namespace

avn_status comm_unique_id(uint8_t* out, std::string& err) {
    if (!out) { err = "comm_unique_id: null output"; return AVN_ERR_BAD_ARG; }
    if (!loaded(err)) return AVN_ERR_STATE;
    static_assert(sizeof(ncclUniqueId) == AVN_COMM_ID_BYTES, "AVN_COMM_ID_BYTES must hold an ncclUniqueId");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(err, "ncclGetUniqueId", r);
    std::memcpy(out, &id, sizeof id);
    return AVN_OK;
}

Comm::~Comm() { if (handle && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)handle); }

avn_status Comm::init(const uint8_t* unique_id, int n, int r, std::string& err) {
    if (!unique_id || n <= 0 || r < 0 || r >= n) { err = "comm_init: bad rank / rank count"; return AVN_ERR_BAD_ARG; }
    if (!loaded(err)) return AVN_ERR_STATE;
    if (handle) { (void)g_rccl.CommDestroy((ncclComm_t)handle); handle = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t res = g_rccl.CommInitRank(&c, n, id, r);
    if (res != ncclSuccess) return fail(err, "ncclCommInitRank", res);
    handle = c; n_ranks = n; rank = r;
    return AVN_OK;
}

// One grouped exchange: every (send, recv) of the call is posted between ncclGroupStart / ncclGroupEnd, so that two ranks which both send
// first cannot deadlock and RCCL can run all peers' transfers as one launch on `s`.
avn_status Comm::exchange(const CommXfer* sends, size_t n_sends, const CommXfer* recvs, size_t n_recvs, hipStream_t s, std::string& err) {
    if (!handle) { err = "halo exchange without a communicator (avn_comm_init)"; return AVN_ERR_STATE; }
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) return fail(err, "ncclGroupStart", r);
    for (size_t i = 0; i < n_sends && r == ncclSuccess; ++i)
        r = g_rccl.Send(sends[i].ptr, sends[i].bytes, ncclUint8, sends[i].peer, (ncclComm_t)handle, s);
    for (size_t i = 0; i < n_recvs && r == ncclSuccess; ++i)
        r = g_rccl.Recv(recvs[i].ptr, recvs[i].bytes, ncclUint8, recvs[i].peer, (ncclComm_t)handle, s);
    ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fail(err, "ncclSend/ncclRecv", r);
    if (e != ncclSuccess) return fail(err, "ncclGroupEnd", e);
    return AVN_OK;
}

// Level-1 sharding's only per-step exchange: every rank's dynamic bounds (48 bytes) to every rank.
avn_status Comm::all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s, std::string& err) {
    if (!handle) { err = "all-gather without a communicator (avn_comm_init)"; return AVN_ERR_STATE; }
    ncclResult_t r = g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, (ncclComm_t)handle, s);
    if (r != ncclSuccess) return fail(err, "ncclAllGather", r);
    return AVN_OK;
}

}  // namespace avn
