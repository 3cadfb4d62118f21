// libntm_rccl.so (include/ntm_rccl.h): the one collective of the sharded path -- a SUM all-reduce of a few fp64 scalars over
// RCCL -- for callers of the C ABI that do not have torch.distributed.  Thin by design: communicator life cycle + ncclAllReduce.
#include "ntm_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

static_assert(sizeof(ncclUniqueId) == NTM_RCCL_ID_BYTES, "ncclUniqueId size");

namespace {
thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int rccl_fail(ncclResult_t r, const char *where) { return fail(NTM_ERCCL, std::string(where) + ": " + ncclGetErrorString(r)); }
}  // namespace

extern "C" {

const char *ntm_rccl_last_error(void) { return g_err.c_str(); }

int ntm_rccl_unique_id(void *id_out)
{
    if (!id_out) return fail(-1, "ntm_rccl_unique_id: null pointer");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int ntm_rccl_comm_create(void **comm_out, int nranks, int rank, const void *id)
{
    if (!comm_out || !id) return fail(-1, "ntm_rccl_comm_create: null pointer");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "ntm_rccl_comm_create: rank must lie in [0, nranks)");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    ncclResult_t r = ncclCommInitRank(&c, nranks, uid, rank);       // binds the calling thread's current HIP device
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    *comm_out = (void *)c;
    return 0;
}

int ntm_rccl_allreduce_f64(double *buf, int64_t count, void *comm, void *stream)
{
    if (count < 0) return fail(-1, "ntm_rccl_allreduce_f64: negative count");
    if (count == 0) return 0;
    if (!buf || !comm) return fail(-1, "ntm_rccl_allreduce_f64: null pointer");
    ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(r, "ncclAllReduce");
}

int ntm_rccl_comm_destroy(void *comm)
{
    if (!comm) return 0;
    ncclResult_t r = ncclCommDestroy((ncclComm_t)comm);
    return r == ncclSuccess ? 0 : rccl_fail(r, "ncclCommDestroy");
}

}  // extern "C"
