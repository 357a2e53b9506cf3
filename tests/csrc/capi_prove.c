/* Plain-C prover over include/zkmi355.h: reads a ZKPK1 file (tools/dump_pk_blob.py — the HOST data a halo2 ProvingKey + ParamsKZG + witness hold), registers
 * the SRS, builds the proving key with zk_plonk_pk_build, proves with zk_plonk_prove and compares the bytes with the expected proof of the file (the golden
 * of the independent CPU prover for the toy / sgx-shaped circuits).  Then a SECOND context borrows the SRS tables and the key (zk_bases_share,
 * zk_plonk_pk_share) and must emit the same proof.  This is the call sequence of the Rust binding (shim/halo2_proofs_mi355x/src/pk_desc.rs), with gcc -std=c99:
 * no Python, no C++, no HIP on this side of the ABI.
 * With a second argument W > 1 the same proof is then made by W "ranks" — W threads of this process, one context each, every one holding 1/W of both SRS
 * tables and a sharded key (zk_plonk_pk_host.shard_world) — whose zk_allgather_fn is a barrier + device-to-device copies: the multi-GPU call sequence of a
 * Rust / C host (with RCCL's ncclAllGather in the callback's place), and every rank must emit the same expected bytes.
 * With ZK_TAMPER_LAST_RANK=1 the ranks first run a proof in which the LAST rank's witness leaves its lookup table (advice column 0 := column 2 — the toy circuit):
 * that rank must fail with its own error after telling the others through the library-owned exchange buffers, every other rank must return ZK_ERR_COMM from the
 * same exchange, and the untampered proof that follows on the same contexts must still come out byte for byte.
 * With ZK_RANK_DEVICES=1 rank r proves on GPU r % (number of GPUs) instead of GPU 0, and — when there is a GPU per rank and libzkmi355_rccl.so loads — the ranks'
 * collective is zk_rccl_allgather on the communicators of zk_rccl_comm_init_all (ncclCommInitAll: one process, W devices; include/zkmi355_rccl.h): the deployment
 * SURVEY 5 names, with no Python and no torch in the process.  On a one-GPU box that setting changes nothing (barrier + copies on GPU 0).
 * usage: capi_prove FILE.zkpk [W]  env ZK_TUNE="key=value,..." applies zk_tune_set pairs (the emulator build wants small launch shapes)
 * exit: 0 proof == expected on both contexts, 3 no usable GPU, 1 anything else */
#include <dlfcn.h>
#include <pthread.h>
#include "zkpk_reader.h"
#include "zkmi355_rccl.h"

#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zk_last_error(ctx)); return 1; } } while (0)

/* ---- W ranks in one process: the collective is a barrier and W x W device copies --------------------------------------------------------------------- */
typedef struct {
    pthread_barrier_t bar;
    int world;
    zk_ctx* ctx[8];
    const void* send[8];
    int failed;
} fabric;
typedef struct {
    fabric* fab; int rank;
    const zk_plonk_pk_host* host; const void* g; const void* g_lagrange; size_t n;
    const void** advice; const void** inst; const uint32_t* inst_len;
    stream st; const unsigned char* want; size_t want_len; int ok;
    int tamper;                                                      /* 1: one failing proof first (see the header comment) */
    zk_allgather_fn rccl_gather; zk_rccl_comm* comm;                 /* ZK_RANK_DEVICES=1 with a GPU per rank: RCCL is the collective — zk_rccl_allgather IS a zk_allgather_fn, user = the communicator */
    uint64_t (*comm_calls)(const zk_rccl_comm*);
    int gathers;                                                     /* calls of the collective this rank made (the same on every rank: 7 commitment phases + the numerators) */
} rank_job;
static int gather(void* user, const void* send_dev, void* recv_dev, size_t bytes) {
    rank_job* j = (rank_job*)user;
    fabric* f = j->fab;
    j->gathers++;
    f->send[j->rank] = send_dev;
    pthread_barrier_wait(&f->bar);                                   /* every rank's send buffer is complete and published */
    int rc = 0;
    for (int r = 0; r < f->world; r++) rc |= zk_dev_copy(f->ctx[j->rank], (char*)recv_dev + (size_t)r * bytes, f->send[r], bytes);
    if (rc) f->failed = 1;
    pthread_barrier_wait(&f->bar);                                   /* nobody overwrites a send buffer a peer is still copying */
    return f->failed;
}
static void* rank_main(void* arg) {
    rank_job* j = (rank_job*)arg;
    zk_ctx* ctx = j->fab->ctx[j->rank];
    const int W = j->fab->world;
    const size_t n_loc = j->n / (size_t)W, lo = (size_t)j->rank * n_loc;
    zk_plonk_pk_host host = *j->host;
    host.shard_world = (uint32_t)W; host.shard_rank = (uint32_t)j->rank; host.allgather = gather; host.allgather_user = j;
    if (j->rccl_gather) { host.allgather = j->rccl_gather; host.allgather_user = j->comm; }
    uint64_t h_g = 0, h_gl = 0, pk = 0;
    unsigned char* proof = (unsigned char*)malloc(j->want_len + 4096);
    size_t len = 0;
    int rc = zk_bases_register(ctx, (const char*)j->g + lo * 64, n_loc, &h_g);
    if (!rc) rc = zk_bases_register(ctx, (const char*)j->g_lagrange + lo * 64, n_loc, &h_gl);
    if (!rc) rc = zk_bases_enable_runs(ctx, h_gl);
    if (!rc) rc = zk_plonk_pk_build(ctx, &host, h_g, h_gl, &pk);
    if (rc) { fprintf(stderr, "rank %d setup -> %d: %s\n", j->rank, rc, zk_last_error(ctx)); j->fab->failed = 1; }
    pthread_barrier_wait(&j->fab->bar);                              /* all ranks ready (or all see `failed`): the collectives below stay matched */
    if (!j->fab->failed && j->tamper) {
        const int last = j->rank == W - 1;
        const void** adv = j->advice;
        if (last) {
            adv = (const void**)calloc(host.n_advice + 1, sizeof(void*));
            memcpy(adv, j->advice, host.n_advice * sizeof(void*));
            if (host.n_advice > 2) adv[0] = j->advice[2];
        }
        rc = zk_plonk_prove(ctx, pk, adv, 0, j->inst, j->inst_len, serve, &j->st, proof, j->want_len + 4096, &len);
        const int as_expected = last ? (rc != ZK_OK && rc != ZK_ERR_COMM && strstr(zk_last_error(ctx), "failure signalled to the other ranks") != NULL) : rc == ZK_ERR_COMM;
        if (!as_expected) { fprintf(stderr, "rank %d, tampered round: rc %d (%s)\n", j->rank, rc, zk_last_error(ctx)); j->fab->failed = 1; }
        if (last) free(adv);
        j->st.at = 0; j->gathers = 0;
        pthread_barrier_wait(&j->fab->bar);
    }
    if (!j->fab->failed) {
        const uint64_t calls0 = j->rccl_gather ? j->comm_calls(j->comm) : 0;
        rc = zk_plonk_prove(ctx, pk, j->advice, 0, j->inst, j->inst_len, serve, &j->st, proof, j->want_len + 4096, &len);
        if (j->rccl_gather) j->gathers = (int)(j->comm_calls(j->comm) - calls0);
        if (rc) fprintf(stderr, "rank %d zk_plonk_prove -> %d: %s\n", j->rank, rc, zk_last_error(ctx));
        j->ok = !rc && len == j->want_len && !memcmp(proof, j->want, len) && j->st.at == j->st.n;
    }
    if (pk) zk_plonk_pk_release(ctx, pk);
    if (h_g) zk_bases_release(ctx, h_g);
    if (h_gl) zk_bases_release(ctx, h_gl);
    free(proof);
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE.zkpk [W]\n", argv[0]); return 1; }
    static zkpk z;
    if (zkpk_read(argv[1], &z)) return 1;
    zk_plonk_pk_host host = z.host;
    const uint32_t k = z.k, n_advice = z.n_advice, n_fixed = z.n_fixed, n_instance = z.n_instance, L = z.L, P = z.P, draw_schedule = z.draw_schedule;
    const size_t n = z.n, want_len = z.want_len;
    const void* g = z.g; const void* g_lagrange = z.g_lagrange;
    const void** advice = z.advice; const void** inst = z.inst; const uint32_t* inst_len = z.inst_len;
    stream st = z.st;
    const unsigned char* want = z.want;

    zk_ctx* ctx = NULL;
    int rc = zk_ctx_create(0, &ctx);
    if (rc == ZK_ERR_NODEV) { printf("no usable GPU (ZK_ERR_NODEV) - there is no CPU fallback\n"); return 3; }
    if (rc) { fprintf(stderr, "zk_ctx_create -> %d\n", rc); return 1; }
    apply_tune(ctx);
    printf("%s: k = %u, %u advice / %u fixed / %u instance columns, %u lookups, %u equality columns, draw schedule %u\n", zk_version(), k, n_advice, n_fixed, n_instance, L, P, draw_schedule);

    /* gen_srs / ParamsKZG::read -> two resident tables; keygen_pk's device half -> one key handle */
    uint64_t h_g, h_gl, pk;
    CK(zk_bases_register(ctx, g, n, &h_g));
    CK(zk_bases_register(ctx, g_lagrange, n, &h_gl));
    CK(zk_bases_enable_runs(ctx, h_gl));
    CK(zk_plonk_pk_build(ctx, &host, h_g, h_gl, &pk));
    unsigned char* proof = (unsigned char*)malloc(want_len + 4096);
    size_t len = 0;
    CK(zk_plonk_prove(ctx, pk, advice, 0, inst, inst_len, serve, &st, proof, want_len + 4096, &len));
    if (st.overrun || st.at != st.n) { fprintf(stderr, "the prover asked for %s draws than the plan holds (%llu of %llu)\n", st.overrun ? "more" : "fewer", (unsigned long long)st.at, (unsigned long long)st.n); return 1; }
    if (len != want_len || memcmp(proof, want, len)) { fprintf(stderr, "proof differs from the expected bytes (%zu vs %zu bytes)\n", len, want_len); return 1; }
    printf("context 1: %zu proof bytes == expected\n", len);

    /* a second context on the same GPU (a second proving thread): borrows the tables and the key, must emit the same bytes */
    {
        zk_ctx* owner = ctx;
        zk_ctx* ctx2 = NULL;
        if (zk_ctx_create(0, &ctx2)) { fprintf(stderr, "second context failed\n"); return 1; }
        ctx = ctx2;
        apply_tune(ctx);
        uint64_t h_g2, h_gl2, pk2;
        CK(zk_bases_share(ctx, owner, h_g, &h_g2));
        CK(zk_bases_share(ctx, owner, h_gl, &h_gl2));
        CK(zk_plonk_pk_share(ctx, owner, pk, h_g2, h_gl2, &pk2));
        st.at = 0;
        memset(proof, 0, want_len);
        CK(zk_plonk_prove(ctx, pk2, advice, 0, inst, inst_len, serve, &st, proof, want_len + 4096, &len));
        if (len != want_len || memcmp(proof, want, len)) { fprintf(stderr, "second context: proof differs\n"); return 1; }
        /* the owner lets go first: the borrower keeps the key alive */
        ctx = owner;
        CK(zk_plonk_pk_release(ctx, pk));
        ctx = ctx2;
        st.at = 0;
        CK(zk_plonk_prove(ctx, pk2, advice, 0, inst, inst_len, serve, &st, proof, want_len + 4096, &len));
        if (len != want_len || memcmp(proof, want, len)) { fprintf(stderr, "second context after the owner released the key: proof differs\n"); return 1; }
        printf("context 2 (shared tables + shared key): same bytes, also after the owner released its handle\n");
        CK(zk_plonk_pk_release(ctx, pk2));
        if (zk_plonk_prove(ctx, pk2, advice, 0, inst, inst_len, serve, &st, proof, want_len + 4096, &len) != ZK_ERR_ARG) { fprintf(stderr, "released key still usable\n"); return 1; }
        zk_ctx_destroy(ctx2);
        ctx = owner;
    }
    CK(zk_bases_release(ctx, h_g)); CK(zk_bases_release(ctx, h_gl));
    zk_ctx_destroy(ctx);
    const int W = argc > 2 ? atoi(argv[2]) : 1;
    if (W > 1) {
        if (W > 8 || n % (size_t)W) { fprintf(stderr, "W must divide n and be <= 8\n"); return 1; }
        static fabric fab;
        static rank_job jobs[8];
        pthread_t th[8];
        fab.world = W;
        pthread_barrier_init(&fab.bar, NULL, (unsigned)W);
        /* ZK_RANK_DEVICES=1: a GPU per rank when the box has them (counted through the ABI: no HIP on this side), RCCL between them when its adapter loads */
        int ndev = 1;
        zk_allgather_fn rccl_gather = NULL;
        zk_rccl_comm* comms[8] = { NULL };
        uint64_t (*comm_calls)(const zk_rccl_comm*) = NULL;
        void (*comm_destroy)(zk_rccl_comm*) = NULL;
        if (getenv("ZK_RANK_DEVICES")) {
            for (ndev = 0; ndev < 64; ndev++) { zk_ctx* probe = NULL; if (zk_ctx_create(ndev, &probe)) break; zk_ctx_destroy(probe); }
            if (ndev < 1) ndev = 1;
            void* so = ndev >= W ? dlopen(getenv("ZK_RCCL_LIB") ? getenv("ZK_RCCL_LIB") : "libzkmi355_rccl.so", RTLD_NOW) : NULL;
            if (so) {
                int (*init_all)(uint32_t, const int*, uint32_t, zk_rccl_comm**) = (int (*)(uint32_t, const int*, uint32_t, zk_rccl_comm**))dlsym(so, "zk_rccl_comm_init_all");
                const char* (*last_error)(const zk_rccl_comm*) = (const char* (*)(const zk_rccl_comm*))dlsym(so, "zk_rccl_last_error");
                rccl_gather = (zk_allgather_fn)dlsym(so, "zk_rccl_allgather");
                comm_calls = (uint64_t (*)(const zk_rccl_comm*))dlsym(so, "zk_rccl_comm_calls");
                comm_destroy = (void (*)(zk_rccl_comm*))dlsym(so, "zk_rccl_comm_destroy");
                if (!init_all || !rccl_gather || !comm_calls || !comm_destroy) { fprintf(stderr, "libzkmi355_rccl.so lacks a symbol of zkmi355_rccl.h\n"); return 1; }
                const int rc_all = init_all((uint32_t)W, NULL, 20000, comms);
                if (rc_all) { fprintf(stderr, "zk_rccl_comm_init_all(%d) -> %d: %s\n", W, rc_all, last_error ? last_error(NULL) : ""); return 1; }
            }
            printf("%d GPUs for %d ranks: %s\n", ndev, W, rccl_gather ? "one GPU per rank, RCCL all-gather (ncclCommInitAll)" : "barrier + device copies");
        }
        for (int r = 0; r < W; r++) {
            if (zk_ctx_create(r % ndev, &fab.ctx[r])) { fprintf(stderr, "context of rank %d failed\n", r); return 1; }
            ctx = fab.ctx[r];
            apply_tune(ctx);
            rank_job* j = &jobs[r];
            j->fab = &fab; j->rank = r; j->host = &host; j->g = g; j->g_lagrange = g_lagrange; j->n = n;
            j->advice = advice; j->inst = inst; j->inst_len = inst_len; j->st = st; j->st.at = 0; j->want = want; j->want_len = want_len; j->ok = 0; j->gathers = 0;
            j->tamper = getenv("ZK_TAMPER_LAST_RANK") != NULL;
            j->rccl_gather = rccl_gather; j->comm = comms[r]; j->comm_calls = comm_calls;
        }
        for (int r = 0; r < W; r++) pthread_create(&th[r], NULL, rank_main, &jobs[r]);
        for (int r = 0; r < W; r++) pthread_join(th[r], NULL);
        for (int r = 0; r < W; r++) { if (!jobs[r].ok) { fprintf(stderr, "rank %d of %d: proof differs or failed\n", r, W); return 1; } zk_ctx_destroy(fab.ctx[r]); }
        if (rccl_gather) for (int r = 0; r < W; r++) comm_destroy(comms[r]);
        for (int r = 1; r < W; r++) if (jobs[r].gathers != jobs[0].gathers) { fprintf(stderr, "rank %d made %d all-gathers, rank 0 made %d\n", r, jobs[r].gathers, jobs[0].gathers); return 1; }
        if (jobs[0].tamper) printf("tampered round: the last rank failed on its own and signalled it through the library's exchange buffers, the others returned ZK_ERR_COMM\n");
        printf("%d ranks (sharded tables, sharded keys, all-gather callback): every rank emitted the expected bytes, %d all-gathers per proof\n", W, jobs[0].gathers);
    }
    printf("capi_prove OK\n");
    return 0;
}
