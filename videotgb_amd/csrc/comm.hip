// comm.hip -- vtgb_allreduce_f32: the thin RCCL wrapper of SURVEY.md 8b / 8e.  The one real exchange of the path is the sum
// all-reduce of the trainable gradients (config C5; DDP in the reference: configs/trainer/ddp.yaml:4): one flat fp32 bucket,
// in place, on the caller's stream.  RCCL is bound at run time (dlopen of the librccl the process already has -- PyTorch's --
// else ROCm's): libvtgb.so carries no link-time dependency on it, and inference-only users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = {0};      // why RCCL could not be used (the failing dlopen's dlerror(), or the missing symbols)
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {   // an already loaded copy first (one RCCL per process), then the system one
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (g_rccl.h) break;
    }
    for (int i = 0; !g_rccl.h && i < 3; i++) {
        g_rccl.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.h) {      // dlerror() clears itself when read: capture the reason of THIS failure now
            const char* e = dlerror();
            snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen failed");
        }
    }
    if (!g_rccl.h) return;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GetErrorString;
    if (!g_rccl.ok) snprintf(g_rccl.why, sizeof(g_rccl.why), "the library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString");
}
int need_rccl() {
    std::call_once(g_once, load_rccl);
    VTGB_REQUIRE(g_rccl.ok, VTGB_EUNSUPPORTED, "RCCL is not available in this process (%s)", g_rccl.why[0] ? g_rccl.why : "librccl.so not found");
    return VTGB_OK;
}
}   // namespace

struct vtgb_comm { ncclComm_t comm; int rank, world; };
static_assert(sizeof(ncclUniqueId) == VTGB_COMM_ID_BYTES, "vtgb.h: VTGB_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

#define VTGB_NCCL(expr)                                                                                      \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) { vtgb_set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r_)); return VTGB_EHIP; } \
    } while (0)

extern "C" int vtgb_comm_unique_id(void* id_out) {
    VTGB_REQUIRE(id_out, VTGB_EINVAL, "comm_unique_id: NULL output");
    VTGB_TRY(need_rccl());
    ncclUniqueId id;
    VTGB_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return VTGB_OK;
}

extern "C" int vtgb_comm_init(vtgb_comm** comm, const void* unique_id, int32_t rank, int32_t world) {
    VTGB_REQUIRE(comm && unique_id && world >= 1 && rank >= 0 && rank < world, VTGB_EINVAL, "comm_init: bad arguments (rank %d of %d)", rank, world);
    VTGB_TRY(need_rccl());
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c;
    VTGB_NCCL(g_rccl.CommInitRank(&c, world, id, rank));      // collective: every rank calls it with the same id, on its own current device
    *comm = new vtgb_comm{c, rank, world};
    return VTGB_OK;
}

extern "C" int vtgb_comm_destroy(vtgb_comm* comm) {
    if (!comm) return VTGB_OK;
    VTGB_TRY(need_rccl());
    VTGB_NCCL(g_rccl.CommDestroy(comm->comm));
    delete comm;
    return VTGB_OK;
}

extern "C" int vtgb_allreduce_f32(vtgb_comm* comm, float* buf, size_t count, int32_t average, vtgb_stream_t stream) {
    VTGB_REQUIRE(comm && (buf || count == 0), VTGB_EINVAL, "allreduce_f32: NULL communicator or buffer");
    if (count == 0) return VTGB_OK;
    VTGB_TRY(need_rccl());
    VTGB_NCCL(g_rccl.AllReduce(buf, buf, count, ncclFloat32, average ? ncclAvg : ncclSum, comm->comm, stream));
    return VTGB_OK;
}
