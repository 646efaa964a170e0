// qg_comm.h -- native per-step exchange over RCCL (included by qg_capi.hip).
//
// The reference has no distributed layer (SB3's process-per-env SubprocVecEnv only, src/train_quadruped.py:49-50);
// the MI355X counterpart shards the env batch over the GPUs of a node and gathers the packed (obs, reward, done) rows
// to the learner rank once per env-step.  torch.distributed can do that gather, but at 4096 envs per GPU an env-step is
// an 18 us kernel and torch's per-collective host cost (~30 us) is what bounds the rate.  This file is the same exchange
// issued from C: one loop, per step { wait for the reader of the buffer, launch the step, ncclGroupStart, ncclRecv x N on
// the root / ncclSend, ncclGroupEnd } on two HIP streams chained by events.  RCCL is loaded with dlopen(): the library
// has no link-time dependency on it, and a process that already holds torch's copy (same SONAME) reuses that one.
#pragma once
#include <dlfcn.h>

typedef struct { char internal[128]; } qg_nccl_unique_id;     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *qg_nccl_comm;                                    // ncclComm_t
enum { QG_NCCL_FLOAT32 = 7 };                                  // ncclFloat32

struct qg_rccl_api {
    void *handle;
    int (*GetUniqueId)(qg_nccl_unique_id *);
    int (*CommInitRank)(qg_nccl_comm *, int, qg_nccl_unique_id, int);
    int (*CommDestroy)(qg_nccl_comm);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void *, size_t, int, int, qg_nccl_comm, hipStream_t);
    int (*Recv)(void *, size_t, int, int, qg_nccl_comm, hipStream_t);
    const char *(*GetErrorString)(int);
};

static qg_rccl_api g_rccl = {};

static int qg_rccl_load(void) {
    if (g_rccl.handle) return QG_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(QG_ERR_DEVICE, "RCCL not found (librccl.so.1): %s", dlerror());
#define QG_SYM(field, sym)                                                                 \
    *(void **)(&g_rccl.field) = dlsym(h, sym);                                             \
    if (!g_rccl.field) return fail(QG_ERR_DEVICE, "RCCL symbol %s missing", sym)
    QG_SYM(GetUniqueId, "ncclGetUniqueId");
    QG_SYM(CommInitRank, "ncclCommInitRank");
    QG_SYM(CommDestroy, "ncclCommDestroy");
    QG_SYM(GroupStart, "ncclGroupStart");
    QG_SYM(GroupEnd, "ncclGroupEnd");
    QG_SYM(Send, "ncclSend");
    QG_SYM(Recv, "ncclRecv");
    QG_SYM(GetErrorString, "ncclGetErrorString");
#undef QG_SYM
    g_rccl.handle = h;
    return QG_OK;
}

#define RCCL_TRY(expr)                                                                     \
    do {                                                                                   \
        int r_ = (expr);                                                                   \
        if (r_ != 0) return fail(QG_ERR_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

struct qg_comm {
    qg_sim *sim;
    qg_nccl_comm comm;
    int rank, world;
    hipStream_t comm_stream;
    hipEvent_t produced[2], consumed[2];
    int consumed_valid[2];
};
