/*
 * pvface_dist.h -- C ABI of the one exchange step of the multi-GPU path (libpvface_dist.so, built from pyannote-video_amd/csrc/dist.hip).
 *
 * The reference has no distributed code at all; BASELINE.json's north_star asks for videos / frame ranges
 * partitioned over the 8 GPUs of a node "with an RCCL all-gather over xGMI of per-shard 128-D track embeddings before a single global
 * clustering".  That all-gather is this library: one process per GPU, ranks exchange their (time, track id, 128 values) rows -- and, for
 * the split distance matrix, their rows of the T x T matrix -- with ncclAllGather on a stream of their own.  It is a separate shared object
 * so that single-GPU users of libpvface.so do not need RCCL.  The 128-byte communicator id travels out of band (torch.distributed, MPI, a file).
 *
 * Conventions as in pvface.h: 0 on success, < 0 on error with pvfd_last_error(); caller owns host buffers.
 */
#ifndef PVFACE_DIST_H
#define PVFACE_DIST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t pvfd_handle;
#define PVFD_ID_BYTES 128

const char* pvfd_last_error(void);
/* rank 0: a fresh communicator id (ncclGetUniqueId) to hand to every rank */
int32_t pvfd_unique_id(uint8_t id[PVFD_ID_BYTES]);
/* every rank: join the communicator (ncclCommInitRank) on GPU `device`; collective call */
int32_t pvfd_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t id[PVFD_ID_BYTES], pvfd_handle* comm);
int32_t pvfd_comm_destroy(pvfd_handle comm);
/* All-gather of a different number of float64 rows per rank (collective): rank r contributes n_rows x row_doubles values; `counts`
 * receives every rank's row count, `out` (room for out_cap_rows rows) all rows in rank order, *total_rows their number.
 * Two ncclAllGather calls (counts, then the payload padded to the largest share) over xGMI; payloads here are <= 32 MB per rank. */
int32_t pvfd_allgather_rows(pvfd_handle comm, const double* rows, int64_t n_rows, int32_t row_doubles, int64_t* counts,
                            double* out, int64_t out_cap_rows, int64_t* total_rows);
/* the largest row count over the ranks for a contribution of n_rows (collective; sizes the buffers of the call above) */
int32_t pvfd_max_rows(pvfd_handle comm, int64_t n_rows, int64_t* counts, int64_t* total_rows);

#ifdef __cplusplus
}
#endif
#endif
