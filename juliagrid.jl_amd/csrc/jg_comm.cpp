// jg_comm.cpp -- the ONE collective of a sharded contingency screen behind the C ABI: an RCCL all-gather of the packed result records
// (include/jgrid.h: jg_comm_*, jg_nr_allgather_results).  The reference has no counterpart (single process: the user-level loop of
// /root/reference/src/powerSystem/branch.jl:453-459 per scenario); SURVEY.md 8(e) asks for "one ncclAllGather (RCCL, xGMI) at the end".
//
// librccl is bound at run time, on the first jg_comm_* call: it is a 0.3 - 0.6 GB library that a single-GPU user of the NR / GN path never
// needs, and a host that already carries an RCCL (Python ML frameworks ship their own copy under the same soname) must end up with ONE of them -- the
// copy that is already mapped wins (RTLD_NOLOAD first), exactly as for the HIP runtime.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

// The five entry points of RCCL this file binds, declared here (the library is bound at run time so that a host without RCCL can still
// run single-GPU analyses -- a hard #include <rccl/rccl.h> made the BUILD depend on its headers all the same).  The ABI of these has been stable
// since NCCL 2.0: an opaque communicator pointer, a 128-byte id passed by value, int-sized enums (ncclSuccess = 0; ncclFloat64 = ncclDouble = 8).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclDouble = 8;

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/jgrid.h"
#include "jg_engine.hpp"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) if ((r.so = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;      // a copy the process already has
        if (!r.so) for (const char* nm : names) if ((r.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.so) { r.error = std::string("librccl not found: ") + dlerror(); return; }
        auto sym = [&](const char* s) { void* p = dlsym(r.so, s); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

int failc(int code, const std::string& msg) { jg::set_last_error(msg); return code; }

#define JG_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r__ = (expr);                                                                                 \
        if (r__ != ncclSuccess) return failc(2, std::string(#expr) + ": " + rccl()->GetErrorString(r__));          \
    } while (0)

}  // namespace

struct jg_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;      // for gathers that are not tied to a handle
};

static_assert(sizeof(ncclUniqueId) == JG_COMM_ID_BYTES, "JG_COMM_ID_BYTES must be the size of ncclUniqueId");

extern "C" {

int jg_comm_unique_id(uint8_t* id) {
    if (!id) return failc(1, "jg_comm_unique_id: bad argument");
    Rccl* r = rccl();
    if (!r->error.empty()) return failc(2, r->error);
    ncclUniqueId u;
    JG_NCCL(r->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return 0;
}

int jg_comm_create(jg_comm** out, int64_t rank, int64_t world, const uint8_t* id, int device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return failc(1, "jg_comm_create: bad argument");
    Rccl* r = rccl();
    if (!r->error.empty()) return failc(2, r->error);
    if (hipSetDevice(device) != hipSuccess) return failc(2, "jg_comm_create: no such HIP device");
    jg_comm* c = new jg_comm();
    c->rank = (int)rank; c->world = (int)world; c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t rc = r->CommInitRank(&c->comm, (int)world, u, (int)rank);
    if (rc != ncclSuccess) { delete c; return failc(2, std::string("ncclCommInitRank: ") + r->GetErrorString(rc)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { r->CommDestroy(c->comm); delete c; return failc(2, "jg_comm_create: stream"); }
    *out = c;
    return 0;
}

void jg_comm_destroy(jg_comm* c) {
    if (!c) return;
    if (c->comm) rccl()->CommDestroy(c->comm);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int jg_comm_rank(const jg_comm* c) { return c ? c->rank : -1; }
int jg_comm_world(const jg_comm* c) { return c ? c->world : -1; }

// Precondition: the collective runs on the communicator's PRIVATE stream, which is not ordered against whatever stream produced
// send_dev -- the producer must have been synchronised (a ContingencyPipeline's record is complete, and its stream synchronised, before on_done
// hands it over; jg_nr_allgather_results needs no such care: it gathers on the handle's own stream).  The call returns after the gather has completed.
int jg_comm_allgather_device(jg_comm* c, const double* send_dev, double* recv_dev, int64_t count) {
    if (!c || !send_dev || !recv_dev || count < 1) return failc(1, "jg_comm_allgather_device: bad argument");
    if (hipSetDevice(c->device) != hipSuccess) return failc(2, "jg_comm_allgather_device: device");
    JG_NCCL(rccl()->AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, c->comm, c->stream));
    if (hipStreamSynchronize(c->stream) != hipSuccess) return failc(2, "jg_comm_allgather_device: stream");
    return 0;
}

}  // extern "C"

namespace jg {
// all-gather of `count` doubles per rank on the caller's stream (no synchronisation): recv = [world][count], send may be recv + rank * count
int comm_allgather(jg_comm* c, const double* send, double* recv, size_t count, hipStream_t st) {
    if (!c) return failc(1, "all-gather: no communicator");
    JG_NCCL(rccl()->AllGather(send, recv, count, ncclDouble, c->comm, st));
    return 0;
}
int comm_rank(const jg_comm* c) { return c->rank; }
int comm_device(const jg_comm* c) { return c->device; }
}  // namespace jg
