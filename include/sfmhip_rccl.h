/*
 * sfmhip_rccl.h -- native RCCL binding of the sharded bundle adjustment (libsfmhip_rccl.so).
 *
 * The reference has no communication layer; this is what a C++ orchestration
 * with one process per GPU links next to libsfmhip.so: every rank holds all cameras + the focal and
 * its own block of points (sfmhip_ba_create on the rank's shard), and the per-iteration sum of the
 * reduced camera system -- the exchange step behind ceres::Solve(DENSE_SCHUR), reference
 * src/BundleAdjustment.cpp:116,123 -- is ONE ncclAllReduce(sum, ncclDouble) over xGMI on the context's
 * stream (the packed upper triangle, or only the 6 x 6 blocks of co-visible camera pairs when the camera graph is
 * sparse), plus the 136-double step evaluation.  Kept in a library of its own so that libsfmhip.so carries
 * no RCCL dependency (a process that already hosts another RCCL, e.g. torch's, keeps using the
 * callback form sfmhip_ba_set_allreduce).
 */
#ifndef SFMHIP_RCCL_H
#define SFMHIP_RCCL_H
#include "sfmhip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SFMHIP_RCCL_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* rank 0 creates the id and ships it to the other ranks by whatever the host program uses (a file, MPI, a socket) */
int sfmhip_rccl_unique_id(void* id_out /* SFMHIP_RCCL_ID_BYTES */);
/* ncclCommInitRank on the context's device; *comm is the ncclComm_t */
int sfmhip_rccl_comm_create(sfmhip_ctx* ctx, int rank, int world, const void* id, void** comm);
/* route the all-reduces of `ba` through `comm` (ncclAllReduce in place, on the context's stream);
 * rank / world as in sfmhip_ba_set_allreduce */
int sfmhip_ba_use_rccl(sfmhip_ba* ba, sfmhip_ctx* ctx, void* comm, int rank, int world);
void sfmhip_rccl_comm_destroy(void* comm);
/* the last ncclResult_t that was not ncclSuccess (0 if none) */
int sfmhip_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
