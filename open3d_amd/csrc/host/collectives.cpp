// Multi-GPU collectives of the sharded hot path, owned by the C++ host side
// (the reference has no multi-device execution, section
// 2.2). One process (or host thread) per GPU; the exchanges are
//   * ICP: in-place sum of 32 float64 per Gauss-Newton iteration, enqueued on
//     the launch stream between the final reduction kernel and the kernel that
//     posts the sums to the host mailbox -- ncclAllReduce, no host in between;
//   * frame-sharded integration: all-gather of per-owner block counts and an
//     all-to-all of block IDs + voxel rows to the ranks that own them
//     (o3dmi_vbg_merge_frame_sharded in voxel_block_grid.cpp).
// RCCL is resolved with dlopen at first use, so the library carries no link
// dependency on it and single-GPU users never load it. When the process
// already holds an RCCL (PyTorch bundles one) that instance is used: a
// communicator is only valid inside the library instance that created it.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <set>
#include <string>

#include "../collectives.h"

using namespace o3dmi;

namespace {

// The slice of the RCCL API used here (rccl/rccl.h: ncclResult_t 0 = success;
// ncclUint8 = 1, ncclFloat64 = 8; ncclSum = 0; a 128-byte unique id passed by
// value).
struct UniqueId {
    char internal[128];
};
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*,
                     hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*,
                     hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int kNcclUint8 = 1, kNcclFloat64 = 8, kNcclSum = 0;

// Why RCCL could not be loaded: written once, inside LoadRccl's call_once
// (it used to be assigned on every failing call, a data race when
// several rank threads probe RCCL).
std::string g_rccl_why;

Rccl* LoadRccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        struct SetWhy {
            ~SetWhy() { g_rccl_why = r.why; }
        } set_why;
        const char* env = std::getenv("O3DMI_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so"};
        // an instance the process already holds first (RTLD_NOLOAD)
        for (const char* n : names)
            if (n && !r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (n && !r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            const char* e = dlerror();
            r.why = std::string("librccl.so not found: ") + (e ? e : "");
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.handle, name);
            if (!p && r.why.empty()) r.why = std::string("missing ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString =
                (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!r.why.empty()) r.handle = nullptr;
    });
    // (r.why is written once, inside call_once: immutable from here on)
    return r.handle ? &r : nullptr;
}

int RcclUnavailable() {
    SetLastError("RCCL is not available in this process (" + g_rccl_why +
                 "; O3DMI_RCCL_LIB names the librccl.so to use)");
    return O3DMI_ERR_UNSUPPORTED;
}

int Check(Rccl* r, int rc, const char* what) {
    if (rc == 0) return O3DMI_OK;
    SetLastError(std::string(what) + ": " +
                 (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
    return O3DMI_ERR_HIP;
}

thread_local o3dmi_comm* g_thread_comm = nullptr;
// the communicator o3dmi_set_rccl_comm made for the calling thread
thread_local o3dmi_comm* g_adopted = nullptr;

// Communicators alive in this process. A communicator may be installed
// (o3dmi_set_comm) on one thread and destroyed on another: the installing
// thread's slot cannot be reached from there, so ThreadComm() checks that
// what the slot names is still alive instead (a dangling slot).
std::mutex g_live_mu;
std::set<const o3dmi_comm*> g_live;
void Register(const o3dmi_comm* c) {
    std::lock_guard<std::mutex> lock(g_live_mu);
    g_live.insert(c);
}
bool Unregister(const o3dmi_comm* c) {
    std::lock_guard<std::mutex> lock(g_live_mu);
    return g_live.erase(c) != 0;
}
bool Alive(const o3dmi_comm* c) {
    std::lock_guard<std::mutex> lock(g_live_mu);
    return g_live.count(c) != 0;
}

}  // namespace

namespace o3dmi {
o3dmi_comm* ThreadComm() {
    if (g_thread_comm && !Alive(g_thread_comm)) g_thread_comm = nullptr;
    return g_thread_comm;
}
}  // namespace o3dmi

int o3dmi_comm::AllreduceSumF64(double* dev, int64_t n, hipStream_t s) {
    if (world <= 1 || n <= 0) return O3DMI_OK;
    if (custom) {
        O3DMI_REQUIRE(table.allreduce_sum_f64, "transport has no all-reduce");
        if (table.allreduce_sum_f64(user, dev, n, (o3dmi_stream_t)s) != 0) {
            SetLastError("transport all-reduce failed");
            return O3DMI_ERR_INVALID_ARG;
        }
        return O3DMI_OK;
    }
    Rccl* r = LoadRccl();
    if (!r) return RcclUnavailable();
    return Check(r, r->AllReduce(dev, dev, (size_t)n, kNcclFloat64, kNcclSum,
                                 nccl, s),
                 "ncclAllReduce");
}

int o3dmi_comm::Allgather(const void* send_dev, void* recv_dev,
                          int64_t bytes_per_rank, hipStream_t s) {
    if (bytes_per_rank <= 0) return O3DMI_OK;
    if (world <= 1) {
        if (send_dev != recv_dev)
            O3DMI_HIP_CHECK(hipMemcpyAsync(recv_dev, send_dev,
                                           (size_t)bytes_per_rank,
                                           hipMemcpyDeviceToDevice, s));
        return O3DMI_OK;
    }
    if (custom) {
        O3DMI_REQUIRE(table.allgather, "transport has no all-gather");
        if (table.allgather(user, send_dev, recv_dev, bytes_per_rank,
                            (o3dmi_stream_t)s) != 0) {
            SetLastError("transport all-gather failed");
            return O3DMI_ERR_INVALID_ARG;
        }
        return O3DMI_OK;
    }
    Rccl* r = LoadRccl();
    if (!r) return RcclUnavailable();
    return Check(r, r->AllGather(send_dev, recv_dev, (size_t)bytes_per_rank,
                                 kNcclUint8, nccl, s),
                 "ncclAllGather");
}

o3dmi_comm::~o3dmi_comm() {
    if (status_dev) (void)hipFree(status_dev);
    if (status_host) (void)hipHostFree(status_host);
}

int o3dmi_comm::AgreeStatus(int local_status, hipStream_t s) {
    if (world <= 1) return local_status;
    if (!status_dev) {
        if (hipMalloc((void**)&status_dev, sizeof(int) * (size_t)world) !=
                    hipSuccess ||
            hipHostMalloc((void**)&status_host, sizeof(int) * (size_t)world) !=
                    hipSuccess) {
            (void)hipGetLastError();
            if (status_dev) (void)hipFree(status_dev);
            status_dev = nullptr;
            // nothing to exchange the status through: the peers cannot be
            // told
            if (local_status) return local_status;
            SetLastError("AgreeStatus: no memory for " +
                         std::to_string(world) + " status words");
            return O3DMI_ERR_HIP;
        }
    }
    O3DMI_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(status_dev + rank),
                                      local_status, 1, s));
    // (SetLastError below must not hide the stage's own message)
    int st = Allgather(status_dev + rank, status_dev, sizeof(int), s);
    if (st) return local_status ? local_status : st;
    O3DMI_HIP_CHECK(hipMemcpyAsync(status_host, status_dev,
                                   sizeof(int) * (size_t)world,
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (local_status) return local_status;
    for (int r = 0; r < world; ++r)
        if (status_host[r] != 0) {
            SetLastError("rank " + std::to_string(r) +
                         " failed its part of a collective call (status " +
                         std::to_string(status_host[r]) +
                         "); nothing was exchanged");
            return O3DMI_ERR_PEER;
        }
    return O3DMI_OK;
}

int o3dmi_comm::Alltoallv(const void* send_dev, const int64_t* send_bytes,
                          const int64_t* send_offsets, void* recv_dev,
                          const int64_t* recv_bytes,
                          const int64_t* recv_offsets, hipStream_t s) {
    if (world <= 1) {
        if (send_bytes[0] > 0)
            O3DMI_HIP_CHECK(hipMemcpyAsync(
                    (char*)recv_dev + recv_offsets[0],
                    (const char*)send_dev + send_offsets[0],
                    (size_t)send_bytes[0], hipMemcpyDeviceToDevice, s));
        return O3DMI_OK;
    }
    if (custom) {
        O3DMI_REQUIRE(table.alltoallv, "transport has no all-to-all");
        if (table.alltoallv(user, send_dev, send_bytes, send_offsets, recv_dev,
                            recv_bytes, recv_offsets, (o3dmi_stream_t)s) != 0) {
            SetLastError("transport all-to-all failed");
            return O3DMI_ERR_INVALID_ARG;
        }
        return O3DMI_OK;
    }
    Rccl* r = LoadRccl();
    if (!r) return RcclUnavailable();
    // one grouped exchange: every pair's send and receive progress together
    // over its own xGMI link
    int st = Check(r, r->GroupStart(), "ncclGroupStart");
    if (st) return st;
    for (int p = 0; p < world && !st; ++p) {
        if (send_bytes[p] > 0)
            st = Check(r, r->Send((const char*)send_dev + send_offsets[p],
                                  (size_t)send_bytes[p], kNcclUint8, p, nccl,
                                  s),
                       "ncclSend");
        if (!st && recv_bytes[p] > 0)
            st = Check(r, r->Recv((char*)recv_dev + recv_offsets[p],
                                  (size_t)recv_bytes[p], kNcclUint8, p, nccl,
                                  s),
                       "ncclRecv");
    }
    const int st_end = Check(r, r->GroupEnd(), "ncclGroupEnd");
    return st ? st : st_end;
}

extern "C" {

int o3dmi_rccl_available(void) { return LoadRccl() != nullptr; }

int o3dmi_rccl_unique_id(void* id128) {
    O3DMI_REQUIRE(id128 != nullptr, "id is null");
    Rccl* r = LoadRccl();
    if (!r) return RcclUnavailable();
    return Check(r, r->GetUniqueId((UniqueId*)id128), "ncclGetUniqueId");
}

int o3dmi_comm_create_rccl(const void* id128, int rank, int world,
                           o3dmi_comm_t** out) {
    O3DMI_REQUIRE(id128 && out, "null argument");
    O3DMI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    Rccl* r = LoadRccl();
    if (!r) return RcclUnavailable();
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    void* nccl = nullptr;
    int st = Check(r, r->CommInitRank(&nccl, world, id, rank),
                   "ncclCommInitRank");
    if (st) return st;
    auto* c = new o3dmi_comm();
    c->rank = rank;
    c->world = world;
    c->nccl = nccl;
    c->owns_nccl = true;
    Register(c);
    *out = c;
    return O3DMI_OK;
}

int o3dmi_comm_adopt_rccl(void* nccl_comm, o3dmi_comm_t** out) {
    O3DMI_REQUIRE(nccl_comm && out, "null argument");
    Rccl* r = LoadRccl();
    if (!r) return RcclUnavailable();
    int rank = 0, world = 1;
    int st = Check(r, r->CommUserRank(nccl_comm, &rank), "ncclCommUserRank");
    if (!st) st = Check(r, r->CommCount(nccl_comm, &world), "ncclCommCount");
    if (st) return st;
    auto* c = new o3dmi_comm();
    c->rank = rank;
    c->world = world;
    c->nccl = nccl_comm;
    Register(c);
    *out = c;
    return O3DMI_OK;
}

int o3dmi_comm_create_custom(const o3dmi_transport_t* table, void* user,
                             int rank, int world, o3dmi_comm_t** out) {
    O3DMI_REQUIRE(table && out, "null argument");
    O3DMI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    auto* c = new o3dmi_comm();
    c->rank = rank;
    c->world = world;
    c->custom = true;
    c->table = *table;
    c->user = user;
    Register(c);
    *out = c;
    return O3DMI_OK;
}

int o3dmi_comm_destroy(o3dmi_comm_t* c) {
    if (!c) return O3DMI_OK;
    if (!Unregister(c)) {
        SetLastError("o3dmi_comm_destroy: not a live communicator");
        return O3DMI_ERR_INVALID_ARG;
    }
    if (g_thread_comm == c) g_thread_comm = nullptr;
    if (g_adopted == c) g_adopted = nullptr;
    int st = O3DMI_OK;
    if (c->owns_nccl && c->nccl) {
        Rccl* r = LoadRccl();
        if (r) st = Check(r, r->CommDestroy(c->nccl), "ncclCommDestroy");
    }
    delete c;
    return st;
}

int o3dmi_comm_rank(const o3dmi_comm_t* c) { return c ? c->rank : 0; }
int o3dmi_comm_world(const o3dmi_comm_t* c) { return c ? c->world : 1; }

int o3dmi_comm_rccl_ranks(const o3dmi_comm_t* c) {
    // asked of RCCL itself, not the number remembered at creation: what a
    // scaling line reports as the span of the communicator that carried it
    if (!c || !c->nccl) return 0;
    Rccl* r = LoadRccl();
    int world = 0;
    if (!r || Check(r, r->CommCount(c->nccl, &world), "ncclCommCount"))
        return 0;
    return world;
}

int o3dmi_set_comm(o3dmi_comm_t* c) {
    g_thread_comm = c;
    return O3DMI_OK;
}

int o3dmi_set_rccl_comm(void* nccl_comm) {
    if (g_adopted) {
        o3dmi_comm* old = g_adopted;
        g_adopted = nullptr;
        if (g_thread_comm == old) g_thread_comm = nullptr;
        (void)Unregister(old);
        delete old;  // adopted: the ncclComm_t stays the caller's
    }
    if (!nccl_comm) {
        g_thread_comm = nullptr;
        return O3DMI_OK;
    }
    o3dmi_comm_t* c = nullptr;
    int st = o3dmi_comm_adopt_rccl(nccl_comm, &c);
    if (st) return st;
    g_adopted = c;
    g_thread_comm = c;
    return O3DMI_OK;
}

int o3dmi_comm_allreduce_sum_f64(o3dmi_comm_t* c, double* dev_buf, int64_t n,
                                 o3dmi_stream_t stream) {
    O3DMI_REQUIRE(c && dev_buf && n >= 0, "bad argument");
    return c->AllreduceSumF64(dev_buf, n, (hipStream_t)stream);
}

int o3dmi_comm_allgather(o3dmi_comm_t* c, const void* send_dev, void* recv_dev,
                         int64_t bytes_per_rank, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(c && send_dev && recv_dev && bytes_per_rank >= 0,
                  "bad argument");
    return c->Allgather(send_dev, recv_dev, bytes_per_rank,
                        (hipStream_t)stream);
}

int o3dmi_comm_alltoallv(o3dmi_comm_t* c, const void* send_dev,
                         const int64_t* send_bytes,
                         const int64_t* send_offsets, void* recv_dev,
                         const int64_t* recv_bytes,
                         const int64_t* recv_offsets, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(c && send_bytes && send_offsets && recv_bytes &&
                          recv_offsets,
                  "bad argument");
    return c->Alltoallv(send_dev, send_bytes, send_offsets, recv_dev,
                        recv_bytes, recv_offsets, (hipStream_t)stream);
}

}  // extern "C"
