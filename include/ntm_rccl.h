/*
 * ntm_rccl.h -- the ONE collective of the sharded path, for callers without torch.distributed
 * (neural-tape-modeling_amd/libntm_rccl.so; a library of its own so that libntm.so carries no RCCL dependency).
 *
 * BASELINE.json north_star: "the test-model batch shards trivially across 8 GPUs with RCCL only for the final ESR reduction".
 * Streams never interact (code/model.py:81, :314-315), so a job of one process per GPU exchanges exactly one thing: the four
 * fp64 loss scalars of ntm_loss_scalars (include/ntm.h) -- what code/test-model.py:386-398 accumulates over the dataset.
 * The Python layer does this with torch.distributed.all_reduce on the nccl backend (= RCCL; distributed.py, bench.py); a C / C++
 * caller of the C ABI does it with the four entry points below:
 *
 *     rank 0:  ntm_rccl_unique_id(id)            -> 128 bytes, shipped to the other ranks by whatever launched them (MPI, a file, a socket)
 *     every rank (its GPU current):  ntm_rccl_comm_create(&comm, nranks, rank, id)
 *     per evaluation:  ntm_gru_forward_esr(...) ; ntm_loss_scalars(esr_out, B, T - skip, 1e-5, v4, stream) ;
 *                      ntm_rccl_allreduce_f64(v4, 4, comm, stream)     // SUM, in place, asynchronous on `stream`
 *                      -> job ESR = v4[0] / v4[1] once the stream has been synchronised
 *     at the end:  ntm_rccl_comm_destroy(comm)
 *
 * Return values: 0 on success, NTM_EINVAL (-1) for bad arguments, NTM_ERCCL (-4) when RCCL reports an error
 * (ntm_rccl_last_error() gives its text; thread-local).  The reference has no counterpart (single device; replicas under SLURM,
 * scripts/sbatch-train-exp1a.sh:7).
 */
#ifndef NTM_RCCL_H
#define NTM_RCCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTM_ERCCL (-4)
#define NTM_RCCL_ID_BYTES 128 /* sizeof(ncclUniqueId) */

int ntm_rccl_unique_id(void *id_out /* host, NTM_RCCL_ID_BYTES */);
int ntm_rccl_comm_create(void **comm_out, int nranks, int rank, const void *id /* host, NTM_RCCL_ID_BYTES */);
int ntm_rccl_allreduce_f64(double *buf /* device, in place */, int64_t count, void *comm, void *stream);
int ntm_rccl_comm_destroy(void *comm);
const char *ntm_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NTM_RCCL_H */
