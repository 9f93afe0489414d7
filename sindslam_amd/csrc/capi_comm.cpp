// C ABI: the one collective of the path -- the gather of the per-frame dynamic masks over the GPUs of a node (frames of a sequence shard
// across ranks, one ncclAllGather per step over xGMI).  A C++ caller of include/DynaDetect.h gets the multi-GPU path without Python: rank 0 makes a
// 128-byte id (sind_comm_unique_id), hands it to the other ranks by whatever channel the application has, every rank calls sind_comm_create, and after
// each pipeline step sind_pipe_gather_masks.  RCCL is loaded with dlopen at first use (librccl.so.1 of the ROCm installation, RTLD_LOCAL): the library
// keeps loading on boxes without RCCL, and a process that also carries PyTorch's bundled RCCL does not mix the two.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>
#include "../../include/sind_hip.h"
#include "common.hpp"

namespace {
struct Rccl {
    void* so = nullptr; std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        static std::once_flag once;
        std::call_once(once, [this] {
            // SIND_RCCL_LIB: one explicit library name / path instead of the default search list (tests point it at a missing file)
            const char* forced = getenv("SIND_RCCL_LIB");
            if (forced) so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            else for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { so = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (so) break; }
            if (!so) { const char* e = dlerror(); err = e ? e : "librccl.so.1 not found"; return; }       // (dlerror() clears the message: one call)
            GetUniqueId = (decltype(GetUniqueId))dlsym(so, "ncclGetUniqueId"); CommInitRank = (decltype(CommInitRank))dlsym(so, "ncclCommInitRank");
            CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy"); AllGather = (decltype(AllGather))dlsym(so, "ncclAllGather");
            GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
            Send = (decltype(Send))dlsym(so, "ncclSend"); Recv = (decltype(Recv))dlsym(so, "ncclRecv"); GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
            if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString || !Send || !Recv || !GroupStart || !GroupEnd) { err = "librccl lacks an expected symbol"; (void)dlclose(so); so = nullptr; }
        });
        return so != nullptr;
    }
} g_rccl;
}  // namespace

struct sind_comm { ncclComm_t comm = nullptr; int rank = 0, world = 1, device = 0; hipStream_t stream = nullptr; DevBuf<uint8_t> send, p2p_send, p2p_recv; };

#define RCCL_TRY(expr)                                                                                          \
    do { const ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { sind_set_error("%s -> %s", #expr, g_rccl.GetErrorString(r_)); return SIND_E_HIP; } } while (0)

extern "C" {

int sind_comm_unique_id(void* id, size_t bytes) {
    if (!id || bytes < NCCL_UNIQUE_ID_BYTES) { sind_set_error("sind_comm_unique_id: need %d bytes", NCCL_UNIQUE_ID_BYTES); return SIND_E_ARG; }
    if (!g_rccl.load()) { sind_set_error("RCCL is not available: %s", g_rccl.err.c_str()); return SIND_E_STATE; }
    ncclUniqueId u; RCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, NCCL_UNIQUE_ID_BYTES); return SIND_OK;
}
int sind_comm_destroy(sind_comm* c) {
    if (!c) return SIND_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c; return SIND_OK;
}
int sind_comm_create(const void* id, int rank, int world, int device, sind_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { sind_set_error("sind_comm_create: bad arguments"); return SIND_E_ARG; }
    if (!g_rccl.load()) { sind_set_error("RCCL is not available: %s", g_rccl.err.c_str()); return SIND_E_STATE; }
    HIP_TRY(hipSetDevice(device));
    sind_comm* c = new sind_comm(); c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u; std::memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { sind_set_error("ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, device, g_rccl.GetErrorString(r)); c->comm = nullptr; sind_comm_destroy(c); return SIND_E_HIP; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { sind_set_error("sind_comm_create: hipStreamCreate failed"); sind_comm_destroy(c); return SIND_E_HIP; }
    *out = c; return SIND_OK;
}
// One all-gather of `bytes` bytes per rank: local (host memory, e.g. the dyna array a pipeline step just filled) -> all_dev (device, world * bytes, rank
// order) and, if given, all_host.  Blocks until the result is there.
int sind_comm_allgather_u8(sind_comm* c, const uint8_t* local, size_t bytes, uint8_t* all_dev, uint8_t* all_host) {
    if (!c || !local || !all_dev || bytes == 0) { sind_set_error("sind_comm_allgather_u8: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    SIND_TRY(c->send.alloc(bytes));
    HIP_TRY(hipMemcpyAsync(c->send.p, local, bytes, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(g_rccl.AllGather(c->send.p, all_dev, bytes, ncclUint8, c->comm, c->stream));
    if (all_host) HIP_TRY(hipMemcpyAsync(all_host, all_dev, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(sind_stream_wait(c->stream));
    return SIND_OK;
}
// One hand-over along the chain of ranks (the state blob of a mismatching chunk seam between two ranks, sequence driver): `bytes` bytes from host memory to rank `to`
// (-1: nothing to send) and as many from rank `from` (-1: nothing to receive) into host memory, as ONE RCCL group (a rank in the middle does both).  Blocks until done.
int sind_comm_sendrecv_u8(sind_comm* c, const uint8_t* send, int to, uint8_t* recv, int from, size_t bytes) {
    if (!c || bytes == 0 || (to >= 0 && (!send || to >= c->world || to == c->rank)) || (from >= 0 && (!recv || from >= c->world || from == c->rank))) { sind_set_error("sind_comm_sendrecv_u8: bad arguments"); return SIND_E_ARG; }
    if (to < 0 && from < 0) return SIND_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (to >= 0) { SIND_TRY(c->p2p_send.alloc(bytes)); HIP_TRY(hipMemcpyAsync(c->p2p_send.p, send, bytes, hipMemcpyHostToDevice, c->stream)); }
    if (from >= 0) SIND_TRY(c->p2p_recv.alloc(bytes));
    RCCL_TRY(g_rccl.GroupStart());
    if (to >= 0) RCCL_TRY(g_rccl.Send(c->p2p_send.p, bytes, ncclUint8, to, c->comm, c->stream));
    if (from >= 0) RCCL_TRY(g_rccl.Recv(c->p2p_recv.p, bytes, ncclUint8, from, c->comm, c->stream));
    RCCL_TRY(g_rccl.GroupEnd());
    if (from >= 0) HIP_TRY(hipMemcpyAsync(recv, c->p2p_recv.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(sind_stream_wait(c->stream));
    return SIND_OK;
}
int sind_pipe_mask_bytes(sind_pipe* p, size_t* bytes);       // pipeline_capi.cpp: S * T * H * W
// The step's dynamic masks ([S][T][H][W] u8, as sind_pipe_process / submit / flush wrote them to `dyna`) of every rank, in rank order
int sind_pipe_gather_masks(sind_pipe* p, sind_comm* c, const uint8_t* dyna, uint8_t* all_dev, uint8_t* all_host) {
    size_t n = 0; SIND_TRY(sind_pipe_mask_bytes(p, &n));
    return sind_comm_allgather_u8(c, dyna, n, all_dev, all_host);
}
int sind_comm_rank(const sind_comm* c) { return c ? c->rank : -1; }
int sind_comm_world(const sind_comm* c) { return c ? c->world : 0; }

}  // extern "C"
