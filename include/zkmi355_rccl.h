/* libzkmi355_rccl.so — the collective of a sharded proof over RCCL / xGMI, for a host that is not Python.
 *
 * BASELINE north_star: "the MSM base table shards across the 8 GPUs of one node with a final RCCL all-reduce over xGMI"; SURVEY 5: "RCCL over xGMI, one process
 * x 8 devices (ncclCommInitAll)".  In libzkmi355.so the exchange between the ranks of one proof is a callback of the caller (zk_allgather_fn, zkmi355.h): EC
 * addition is not an RCCL reduction, so the "all-reduce" is an all-gather of 128-byte partial points (one per commitment of a phase) that every rank sums itself
 * (zk_g1_sum_xyzz_batch), plus one bulk all-gather of the quotient's numerators.  This OPTIONAL library is that callback written on RCCL: it links librccl and
 * nothing else — not libzkmi355.so, not torch — so that a Rust or C host (the reference's: circuits/src/sgx_dcap_verifier.rs:814-822 calls create_proof from
 * Rust) runs the sharded proof with no Python in the process.  The core library stays RCCL-free.
 *
 * Two deployments:
 *   one process per GPU (torchrun-style launchers, MPI):   rank 0: zk_rccl_unique_id(id) -> hand the 128 bytes to every rank by any means -> every rank:
 *                                                          zk_rccl_comm_create(world, rank, id, device, timeout_ms, &comm)
 *   one process, N devices (ncclCommInitAll):    zk_rccl_comm_init_all(ndev, devices, timeout_ms, comms) -> thread r proves on zk_ctx_create(devices[r])
 *                                                          with comms[r]
 * and in both: zk_plonk_pk_host / zk_plonk_pk_desc .allgather = zk_rccl_allgather, .allgather_user = comm.
 *
 * Conventions are zkmi355.h's: 0 or a negative ZK_ERR_* code (the callback itself: 0 or non-zero, as zk_allgather_fn says), nothing throws or aborts, text of the last
 * failure through zk_rccl_last_error.  A communicator serves one proving thread at a time (the ranks of one proof call it in lockstep by construction).
 * Status: compiled and symbol-checked in the build image, run with ONE rank on one MI355X (tests/test_rccl_adapter.py); two and more ranks need the multi-GPU node no
 * builder session has had — unmeasured on hardware. */
#ifndef ZKMI355_RCCL_H
#define ZKMI355_RCCL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZK_RCCL_UNIQUE_ID_BYTES 128                      /* = NCCL_UNIQUE_ID_BYTES */
typedef struct zk_rccl_comm zk_rccl_comm;

/* ncclGetUniqueId: called once, by one rank; `out` receives ZK_RCCL_UNIQUE_ID_BYTES bytes that every rank passes to zk_rccl_comm_create. */
int zk_rccl_unique_id(void* out);

/* ncclCommInitRank on `device` (hipSetDevice ordinal) + a stream of its own for the collectives.  Blocks until all `world` ranks have called it.
 * timeout_ms: how long zk_rccl_allgather waits for one collective before it gives the communicator up (0 = 30 000). */
int zk_rccl_comm_create(uint32_t world, uint32_t rank, const void* unique_id, int device, uint32_t timeout_ms, zk_rccl_comm** comm);

/* ncclCommInitAll: `ndev` communicators of ONE process, comms[r] on devices[r] (devices = NULL: 0 .. ndev-1) with rank r. */
int zk_rccl_comm_init_all(uint32_t ndev, const int* devices, uint32_t timeout_ms, zk_rccl_comm** comms);

/* A zk_allgather_fn (zkmi355.h): user = the rank's zk_rccl_comm*.  ncclAllGather(send_dev, recv_dev, bytes, ncclChar) on the communicator's stream, then waits
 * for it — the library reads recv_dev right after the call.  Returns non-zero when RCCL reports an error or the collective has not finished after timeout_ms (a
 * rank that died, a link that hangs): the communicator is aborted (ncclCommAbort), every later call fails at once, and the proof ends with ZK_ERR_COMM. */
int zk_rccl_allgather(void* user, const void* send_dev, void* recv_dev, size_t bytes);

uint32_t zk_rccl_comm_world(const zk_rccl_comm* comm);
uint32_t zk_rccl_comm_rank(const zk_rccl_comm* comm);
uint64_t zk_rccl_comm_calls(const zk_rccl_comm* comm);  /* collectives completed (a proof makes 8: tests count them) */
const char* zk_rccl_last_error(const zk_rccl_comm* comm);   /* comm = NULL: of the last failed create / init_all / unique_id on the calling thread */
void zk_rccl_comm_destroy(zk_rccl_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
