/* GPU-box check of libzkmi355_rccl.so (include/zkmi355_rccl.h) with the ranks one GPU allows — ONE (RCCL refuses two ranks on a device): both ways of making a
 * communicator (zk_rccl_unique_id + zk_rccl_comm_create = ncclCommInitRank; zk_rccl_comm_init_all = ncclCommInitAll), then zk_rccl_allgather as the library calls
 * it — device buffers of a zk_ctx, bytes, blocking — with the sizes of a proof's exchanges (128-byte points, a 1 MiB block), and the data checked.  This is the
 * collective's call path from plain C, not scaling.  exit 0 ok, 3 no GPU, 1 failure.  gcc -std=c99, links libzkmi355.so + libzkmi355_rccl.so. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkmi355.h"
#include "zkmi355_rccl.h"

static int roundtrip(zk_ctx* ctx, zk_rccl_comm* comm, size_t bytes) {
    unsigned char* h = (unsigned char*)malloc(bytes);
    unsigned char* back = (unsigned char*)malloc(bytes);
    void* send = NULL; void* recv = NULL;
    for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 131 + 7);
    int bad = zk_dev_alloc(ctx, bytes, &send) || zk_dev_alloc(ctx, bytes, &recv) || zk_dev_upload(ctx, send, h, bytes) || zk_dev_zero(ctx, recv, bytes) || zk_dev_sync(ctx);
    if (!bad) bad = zk_rccl_allgather(comm, send, recv, bytes);          /* exactly a zk_allgather_fn call: user, send_dev, recv_dev, bytes */
    if (bad) fprintf(stderr, "all-gather of %zu bytes: %s | %s\n", bytes, zk_rccl_last_error(comm), zk_last_error(ctx));
    if (!bad) bad = zk_dev_download(ctx, back, recv, bytes) || memcmp(h, back, bytes);
    if (send) zk_dev_free(ctx, send);
    if (recv) zk_dev_free(ctx, recv);
    free(h); free(back);
    return bad;
}

int main(void) {
    zk_ctx* ctx = NULL;
    int rc = zk_ctx_create(0, &ctx);
    if (rc == ZK_ERR_NODEV) { printf("no usable GPU\n"); return 3; }
    if (rc) return 1;
    unsigned char id[ZK_RCCL_UNIQUE_ID_BYTES];
    zk_rccl_comm* comm = NULL;
    if (zk_rccl_unique_id(id) || zk_rccl_comm_create(1, 0, id, 0, 10000, &comm)) { fprintf(stderr, "ncclCommInitRank path: %s\n", zk_rccl_last_error(NULL)); return 1; }
    if (zk_rccl_comm_world(comm) != 1 || zk_rccl_comm_rank(comm) != 0) return 1;
    if (roundtrip(ctx, comm, 128) || roundtrip(ctx, comm, 25 * 128) || roundtrip(ctx, comm, 1 << 20)) return 1;
    if (zk_rccl_comm_calls(comm) != 3) { fprintf(stderr, "calls = %llu\n", (unsigned long long)zk_rccl_comm_calls(comm)); return 1; }
    zk_rccl_comm_destroy(comm);
    printf("ncclCommInitRank communicator: 3 all-gathers (128 B, 3200 B, 1 MiB) through zk_rccl_allgather, data intact\n");
    zk_rccl_comm* all[1] = { NULL };
    if (zk_rccl_comm_init_all(1, NULL, 10000, all)) { fprintf(stderr, "ncclCommInitAll path: %s\n", zk_rccl_last_error(NULL)); return 1; }
    if (roundtrip(ctx, all[0], 4096)) return 1;
    zk_rccl_comm_destroy(all[0]);
    printf("ncclCommInitAll communicator: ok\n");
    /* argument errors come back as codes */
    if (zk_rccl_comm_create(2, 5, id, 0, 0, &comm) != ZK_ERR_ARG || zk_rccl_allgather(NULL, NULL, NULL, 0) == 0) return 1;
    zk_ctx_destroy(ctx);
    printf("capi_rccl OK\n");
    return 0;
}
