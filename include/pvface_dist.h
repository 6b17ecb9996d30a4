/*
 * pvface_dist.h -- C ABI of the one exchange step of the multi-GPU path (libpvface_dist.so, built from pyannote-video_amd/csrc/dist.hip).
 *
 * The reference has no distributed code at all; BASELINE.json's north_star asks for videos / frame ranges
 * partitioned over the 8 GPUs of a node "with an RCCL all-gather over xGMI of per-shard 128-D track embeddings before a single global
 * clustering".  That all-gather is this library: one process per GPU, ranks exchange their (128 float32 values, time, track id) rows --
 * and, for the split distance matrix, their rows of the T x T matrix -- over RCCL on a stream of their own, HBM to HBM.  It is a separate shared object
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
/* every rank's value of n (collective; one ncclAllGather of int64): the row counts that size the payload exchange */
int32_t pvfd_allgather_counts(pvfd_handle comm, int64_t n, int64_t* counts /* [world] */);
/* All-gather of a different number of BYTES per rank, device memory to device memory (collective): rank r contributes nbytes[r]
 * bytes from d_send; d_recv (sum of nbytes) receives all contributions in rank order.  One ncclGroup of ncclBroadcast calls, exact
 * sizes; the call returns when the data is in place.  The caller orders its own streams around the call (the producer of d_send
 * must have finished; consumers of d_recv may start afterwards).  What travels in a run: 528 bytes per face (float32[128], float64
 * time, int32 track id) -- 95 MB per rank at configs[2]'s 180 000 faces -- and each rank's rows of the T x T track-pair matrix. */
int32_t pvfd_allgatherv_dev(pvfd_handle comm, const void* d_send, const int64_t* nbytes /* [world] */, void* d_recv);
/* Every collective above is WATCHED: its stream is polled, the communicator's asynchronous error state beside it, and after
 * PVF_DIST_TIMEOUT_S seconds (default 120) the communicator is aborted, the stream drained, and the call fails with a message that names
 * this rank, the world size, what was being exchanged (the counts / the bytes per rank) and that the communicator was aborted; later calls
 * on the handle fail at once ("aborted by an earlier failure"), pvfd_comm_destroy still releases it.
 * pvfd_debug_stall: the watchdog's own test entry (tests/test_gpu_sharded.py) -- a kernel that spins for `milliseconds` is put on the
 * communicator's stream and waited for exactly as a collective is: a peer that never joins looks like this from the waiting rank's side. */
int32_t pvfd_debug_stall(pvfd_handle comm, int32_t milliseconds);

#ifdef __cplusplus
}
#endif
#endif
