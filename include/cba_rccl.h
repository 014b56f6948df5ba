/*
 * cba_rccl.h -- native RCCL all-reduce callback for the image-sharded path of libcalib_ba_hip.so (include/cba.h,
 * cba_config.allreduce), for hosts that are not Python: one process per GPU, one RCCL communicator over xGMI, the one
 * packed-reduced-system all-reduce per Gauss-Newton step plus the 8-double scalar reductions.  Built as
 * libcalib_ba_rccl.so (links librccl); the engine itself does not depend on RCCL.
 *
 * No reference counterpart: the reference has no multi-GPU path.
 */
#ifndef CBA_RCCL_H_
#define CBA_RCCL_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { CBA_RCCL_ID_BYTES = 128 };           /* sizeof(ncclUniqueId) */
typedef struct cba_rccl cba_rccl;

/* rank 0 creates the id and hands it to the other ranks by whatever the host has (MPI_Bcast, a file, a socket) */
int cba_rccl_unique_id(char id[CBA_RCCL_ID_BYTES]);
/* collective over all ranks; `device` is this rank's HIP device */
int cba_rccl_create(int rank, int world, const char id[CBA_RCCL_ID_BYTES], int device, cba_rccl** out);
/* the same, with the id exchanged through a file (single-node launchers): rank 0 replaces whatever is at `path`, the others
 * wait for a FRESH file (leftovers older than two minutes are ignored), and rank 0 removes the file once every rank has joined,
 * so a later launch with the same path never reads this run's id */
int cba_rccl_create_via_file(int rank, int world, const char* path, int device, cba_rccl** out);
/* the reader half of the above (tests): 0 = id read, -1 = no fresh file within timeout_ms */
int cba_rccl_debug_read_id_file(const char* path, char id[CBA_RCCL_ID_BYTES], int timeout_ms, int max_age_s);
void cba_rccl_destroy(cba_rccl* c);
/* ncclCommCount of the communicator (-1 on error) */
int cba_rccl_comm_count(cba_rccl* c);
/* cba_allreduce_fn: in-place fp64 sum of a DEVICE buffer over all ranks; `user` is the cba_rccl*.  Enqueues
 * ncclAllReduce on the communicator's own stream and waits for THAT stream only (no device-wide synchronisation). */
int cba_rccl_allreduce(void* device_ptr, int64_t count, void* user);
/* cba_collective_fn: ncclAllReduce / ncclReduceScatter / ncclAllGather (fp64) for cba_config.collective -- the reduce-scatter
 * and all-gathers of the distributed reduced solve; same stream discipline as above. */
int cba_rccl_collective(int32_t op, void* sendbuf, void* recvbuf, int64_t count, void* user);
const char* cba_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
