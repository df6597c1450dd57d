/* fake_rccl.c -- TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so that libterrarium_hip.so opens when
 * TRM_RCCL_LIBRARY names it (terrarium_hip.hip: rccl()).  It exports the seven entry points the library resolves and implements
 * them for n communicators in ONE process and ONE thread through host memory, so that the grouped call sequences of
 * trm_comm_init_all / trm_reduce_global_all / trm_status_global_all run with n > 1 on a one-GPU box.
 *
 * What it checks (and real RCCL would answer with a hang): outside a group a call that needs other ranks fails; at ncclGroupEnd
 * every communicator group touched inside the group must have been posted by ALL its ranks, exactly once, with the same count,
 * type and operator.  What it does NOT model: transport, topology, streams running concurrently (it synchronises each stream),
 * the order in which RCCL folds the ranks (here: rank order, so sums equal the library's host fold bit for bit).
 *
 * Built by the test: gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -L/opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_COMMS 64
#define MAX_PENDING 256

struct ncclComm {
    int rank, nranks, complete;
    unsigned long long group;      /* from the unique id */
    int alive;
};
static struct ncclComm comms[MAX_COMMS];
static int group_depth = 0;
static unsigned long long next_id = 1;
/* FAKE_RCCL_FAIL_INIT_RANK = k: ncclCommInitRank of rank k fails (tests the library's clean-up) */
static int fail_rank(void) { const char* e = getenv("FAKE_RCCL_FAIL_INIT_RANK"); return e ? atoi(e) : -1; }

struct pending_reduce { const void* send; void* recv; size_t count; ncclDataType_t type; ncclRedOp_t op; struct ncclComm* comm; hipStream_t stream; };
static struct pending_reduce pending[MAX_PENDING];
static int npending = 0;
static struct ncclComm* pending_init[MAX_COMMS];
static int ninit = 0;
static char last_error[256] = "";

static ncclResult_t err(ncclResult_t code, const char* msg) { snprintf(last_error, sizeof last_error, "fake_rccl: %s", msg); return code; }

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (last_error[0] ? last_error : "fake_rccl: error"); }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    memcpy(id->internal, &next_id, sizeof next_id);
    memcpy(id->internal + 8, "fakerccl", 8);
    next_id++;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { group_depth++; return ncclSuccess; }

static double fold(ncclRedOp_t op, double a, double b) {
    switch (op) {
        case ncclSum: return a + b;
        case ncclProd: return a * b;
        case ncclMax: return a > b ? a : b;
        case ncclMin: return a < b ? a : b;
        default: return a;
    }
}

/* every rank of `group` among pending[first..]: checked, then reduced on the host in rank order */
static ncclResult_t run_allreduce(unsigned long long group, int nranks) {
    struct pending_reduce* by_rank[MAX_COMMS];
    memset(by_rank, 0, sizeof by_rank);
    for (int i = 0; i < npending; ++i) {
        struct pending_reduce* p = &pending[i];
        if (p->comm->group != group) continue;
        if (by_rank[p->comm->rank]) return err(ncclInvalidUsage, "a rank posted two all-reduces of one communicator group inside one group call");
        by_rank[p->comm->rank] = p;
    }
    for (int r = 0; r < nranks; ++r)
        if (!by_rank[r]) return err(ncclInvalidUsage, "ncclGroupEnd: not every rank of the communicator group posted its all-reduce (RCCL would wait forever)");
    for (int r = 1; r < nranks; ++r)
        if (by_rank[r]->count != by_rank[0]->count || by_rank[r]->type != by_rank[0]->type || by_rank[r]->op != by_rank[0]->op)
            return err(ncclInvalidArgument, "the ranks disagree about count / type / operator");
    if (by_rank[0]->type != ncclDouble) return err(ncclInvalidArgument, "only ncclDouble is modelled");
    const size_t count = by_rank[0]->count;
    double* acc = (double*)malloc(count * sizeof(double));
    double* x = (double*)malloc(count * sizeof(double));
    for (int r = 0; r < nranks; ++r) {
        if (hipStreamSynchronize(by_rank[r]->stream) != hipSuccess) return err(ncclUnhandledCudaError, "hipStreamSynchronize");
        if (hipMemcpy(r == 0 ? acc : x, by_rank[r]->send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return err(ncclUnhandledCudaError, "hipMemcpy (send)");
        if (r > 0) for (size_t j = 0; j < count; ++j) acc[j] = fold(by_rank[0]->op, acc[j], x[j]);
    }
    for (int r = 0; r < nranks; ++r)
        if (hipMemcpy(by_rank[r]->recv, acc, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return err(ncclUnhandledCudaError, "hipMemcpy (recv)");
    free(acc); free(x);
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (group_depth <= 0) return err(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    /* communicator creation: every rank 0 .. n-1 of a unique id exactly once */
    for (int i = 0; i < ninit && rc == ncclSuccess; ++i) {
        struct ncclComm* c = pending_init[i];
        if (c->complete) continue;
        int seen[MAX_COMMS] = {0}, total = 0;
        for (int j = 0; j < ninit; ++j)
            if (pending_init[j]->group == c->group) {
                if (pending_init[j]->nranks != c->nranks || seen[pending_init[j]->rank]++) rc = err(ncclInvalidArgument, "ncclCommInitRank: inconsistent ranks inside one group");
                total++;
            }
        if (rc == ncclSuccess && total != c->nranks) rc = err(ncclInvalidUsage, "ncclGroupEnd: not every rank of the unique id called ncclCommInitRank (RCCL would wait forever)");
        if (rc == ncclSuccess)
            for (int j = 0; j < ninit; ++j)
                if (pending_init[j]->group == c->group) pending_init[j]->complete = 1;
    }
    if (rc != ncclSuccess)
        for (int i = 0; i < ninit; ++i)
            if (!pending_init[i]->complete) pending_init[i]->alive = 0;
    ninit = 0;
    /* all-reduces */
    for (int i = 0; i < npending && rc == ncclSuccess; ++i) {
        int done_before = 0;
        for (int j = 0; j < i; ++j) done_before |= pending[j].comm->group == pending[i].comm->group;
        if (!done_before) rc = run_allreduce(pending[i].comm->group, pending[i].comm->nranks);
    }
    npending = 0;
    return rc;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_COMMS || rank < 0 || rank >= nranks) return err(ncclInvalidArgument, "ncclCommInitRank: bad argument");
    if (rank == fail_rank()) return err(ncclSystemError, "ncclCommInitRank: failure requested by FAKE_RCCL_FAIL_INIT_RANK");
    if (nranks > 1 && group_depth == 0) return err(ncclInvalidUsage, "ncclCommInitRank of several ranks from one thread outside a group (RCCL would wait forever)");
    struct ncclComm* c = NULL;
    for (int i = 0; i < MAX_COMMS; ++i)
        if (!comms[i].alive) { c = &comms[i]; break; }
    if (!c) return err(ncclInternalError, "out of communicator slots");
    c->alive = 1; c->rank = rank; c->nranks = nranks; c->complete = nranks == 1;
    memcpy(&c->group, id.internal, sizeof c->group);
    if (nranks > 1) pending_init[ninit++] = c;
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    struct ncclComm* c = (struct ncclComm*)comm;
    if (!c || !c->alive) return err(ncclInvalidArgument, "ncclCommDestroy: not a live communicator");
    c->alive = 0;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    struct ncclComm* c = (struct ncclComm*)comm;
    if (!c || !c->alive || !c->complete) return err(ncclInvalidArgument, "ncclAllReduce: not a complete communicator");
    if (c->nranks == 1) {
        if (hipMemcpyAsync(recv, send, count * (type == ncclDouble ? 8 : 4), hipMemcpyDeviceToDevice, stream) != hipSuccess) return err(ncclUnhandledCudaError, "hipMemcpyAsync");
        return ncclSuccess;
    }
    if (group_depth == 0) return err(ncclInvalidUsage, "ncclAllReduce of a multi-rank communicator from one thread outside a group (RCCL would wait forever)");
    if (npending >= MAX_PENDING) return err(ncclInternalError, "too many pending operations");
    struct pending_reduce p = {send, recv, count, type, op, c, stream};
    pending[npending++] = p;
    return ncclSuccess;
}

/* test hooks (ctypes) */
int fake_rccl_live_communicators(void) { int n = 0; for (int i = 0; i < MAX_COMMS; ++i) n += comms[i].alive; return n; }
