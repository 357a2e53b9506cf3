/* Plain-C prover over include/zkmi355.h: reads a ZKPK1 file (tools/dump_pk_blob.py — the HOST data a halo2 ProvingKey + ParamsKZG + witness hold), registers
 * the SRS, builds the proving key with zk_plonk_pk_build, proves with zk_plonk_prove and compares the bytes with the expected proof of the file (the golden
 * of the independent CPU prover for the toy / sgx-shaped circuits).  Then a SECOND context borrows the SRS tables and the key (zk_bases_share,
 * zk_plonk_pk_share) and must emit the same proof.  This is the call sequence of the Rust binding (shim/halo2_proofs_mi355x/src/pk_desc.rs), with gcc -std=c99:
 * no Python, no C++, no HIP on this side of the ABI.
 * With a second argument W > 1 the same proof is then made by W "ranks" — W threads of this process, one context each, every one holding 1/W of both SRS
 * tables and a sharded key (zk_plonk_pk_host.shard_world) — whose zk_allgather_fn is a barrier + device-to-device copies: the multi-GPU call sequence of a
 * Rust / C host (with RCCL's ncclAllGather in the callback's place), and every rank must emit the same expected bytes.
 * usage: capi_prove FILE.zkpk [W]  env ZK_TUNE="key=value,..." applies zk_tune_set pairs (the emulator build wants small launch shapes)
 * exit: 0 proof == expected on both contexts, 3 no usable GPU, 1 anything else */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkmi355.h"

#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zk_last_error(ctx)); return 1; } } while (0)

typedef struct { const unsigned char* p; size_t left; } rd;
static const void* take(rd* r, size_t bytes) {
    bytes = (bytes + 7) & ~(size_t)7;
    if (bytes > r->left) { fprintf(stderr, "ZKPK1 file truncated\n"); exit(1); }
    const void* q = r->p; r->p += bytes; r->left -= bytes; return q;
}
static uint64_t take_u64(rd* r) { uint64_t v; memcpy(&v, take(r, 8), 8); return v; }

/* the caller's rng: serves the recorded Fr::random stream in the order the library asks for it */
typedef struct { const unsigned char* draws; uint64_t n, at; int overrun; } stream;
static void serve(void* user, size_t count, void* out) {
    stream* s = (stream*)user;
    if (s->at + count > s->n) { s->overrun = 1; memset(out, 0, count * 32); return; }
    memcpy(out, s->draws + 32 * s->at, count * 32);
    s->at += count;
}

static void apply_tune(zk_ctx* ctx) {
    const char* t = getenv("ZK_TUNE");
    if (!t) return;
    char* copy = strdup(t);
    for (char* tok = strtok(copy, ","); tok; tok = strtok(NULL, ",")) {
        char* eq = strchr(tok, '=');
        if (!eq) continue;
        *eq = 0;
        if (zk_tune_set(ctx, tok, atoi(eq + 1))) fprintf(stderr, "zk_tune_set(%s) refused\n", tok);
    }
    free(copy);
}

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
    uint64_t h_g = 0, h_gl = 0, pk = 0;
    unsigned char* proof = (unsigned char*)malloc(j->want_len + 4096);
    size_t len = 0;
    int rc = zk_bases_register(ctx, (const char*)j->g + lo * 64, n_loc, &h_g);
    if (!rc) rc = zk_bases_register(ctx, (const char*)j->g_lagrange + lo * 64, n_loc, &h_gl);
    if (!rc) rc = zk_bases_enable_runs(ctx, h_gl);
    if (!rc) rc = zk_plonk_pk_build(ctx, &host, h_g, h_gl, &pk);
    if (rc) { fprintf(stderr, "rank %d setup -> %d: %s\n", j->rank, rc, zk_last_error(ctx)); j->fab->failed = 1; }
    pthread_barrier_wait(&j->fab->bar);                              /* all ranks ready (or all see `failed`): the collectives below stay matched */
    if (!j->fab->failed) {
        rc = zk_plonk_prove(ctx, pk, j->advice, 0, j->inst, j->inst_len, serve, &j->st, proof, j->want_len + 4096, &len);
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
    if (argc < 2) { fprintf(stderr, "usage: %s FILE.zkpk\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* file = (unsigned char*)malloc((size_t)size + 8);
    if (!file || fread(file, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "read failed\n"); return 1; }
    fclose(f);
    rd r = { file, (size_t)size };
    const uint32_t* head = (const uint32_t*)take(&r, 8 + 12 * 4);
    if (memcmp(head, "ZKPK", 4) || head[1] != 1) { fprintf(stderr, "not a ZKPK1 file\n"); return 1; }
    const uint32_t k = head[2], cs_degree = head[3], bf = head[4], n_fixed = head[5], n_advice = head[6], n_instance = head[7], L = head[8], P = head[9],
                   n_aq = head[10], n_fq = head[11], transcript = head[12], draw_schedule = head[13];
    const size_t n = (size_t)1 << k;
    const uint32_t* lists = (const uint32_t*)take(&r, 4 * (2 * (size_t)P + 2 * n_aq + 2 * n_fq + L));
    const void* transcript_repr = take(&r, 32);

    zk_plonk_pk_host host;
    ZK_STRUCT_INIT(host);
    host.k = k; host.cs_degree = cs_degree; host.blinding_factors = bf;
    host.n_fixed = n_fixed; host.n_advice = n_advice; host.n_instance = n_instance; host.n_lookups = L; host.n_perm_columns = P;
    host.perm_columns = lists; host.advice_queries = lists + 2 * P; host.n_advice_queries = n_aq;
    host.fixed_queries = lists + 2 * P + 2 * n_aq; host.n_fixed_queries = n_fq;
    host.lookup_table_key = lists + 2 * P + 2 * n_aq + 2 * n_fq;
    host.transcript_repr = transcript_repr; host.transcript = transcript; host.draw_schedule = draw_schedule;
    host.evaluator_zkq1_len = (size_t)take_u64(&r); host.evaluator_zkq1 = take(&r, host.evaluator_zkq1_len);
    const void** in_blob = (const void**)calloc(L + 1, sizeof(void*)); const void** tab_blob = (const void**)calloc(L + 1, sizeof(void*));
    size_t* in_len = (size_t*)calloc(L + 1, sizeof(size_t)); size_t* tab_len = (size_t*)calloc(L + 1, sizeof(size_t));
    for (uint32_t l = 0; l < L; l++) {
        in_len[l] = (size_t)take_u64(&r); in_blob[l] = take(&r, in_len[l]);
        tab_len[l] = (size_t)take_u64(&r); tab_blob[l] = take(&r, tab_len[l]);
    }
    host.lookup_input_zkq1 = in_blob; host.lookup_input_zkq1_len = in_len; host.lookup_table_zkq1 = tab_blob; host.lookup_table_zkq1_len = tab_len;
    const void* g = take(&r, n * 64);
    const void* g_lagrange = take(&r, n * 64);
    const void** fixed = (const void**)calloc(n_fixed + 1, sizeof(void*)); const void** sigma = (const void**)calloc(P + 1, sizeof(void*));
    for (uint32_t i = 0; i < n_fixed; i++) fixed[i] = take(&r, n * 32);
    for (uint32_t i = 0; i < P; i++) sigma[i] = take(&r, n * 32);
    host.fixed_values = fixed; host.sigma_values = sigma;
    const void** advice = (const void**)calloc(n_advice + 1, sizeof(void*));
    for (uint32_t i = 0; i < n_advice; i++) advice[i] = take(&r, n * 32);
    const void** inst = (const void**)calloc(n_instance + 1, sizeof(void*)); uint32_t* inst_len = (uint32_t*)calloc(n_instance + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n_instance; i++) { inst_len[i] = (uint32_t)take_u64(&r); inst[i] = take(&r, (size_t)inst_len[i] * 32); }
    stream st = { NULL, 0, 0, 0 };
    st.n = take_u64(&r); st.draws = (const unsigned char*)take(&r, (size_t)st.n * 32);
    const size_t want_len = (size_t)take_u64(&r);
    const unsigned char* want = (const unsigned char*)take(&r, want_len);

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
        for (int r = 0; r < W; r++) {
            if (zk_ctx_create(0, &fab.ctx[r])) { fprintf(stderr, "context of rank %d failed\n", r); return 1; }
            ctx = fab.ctx[r];
            apply_tune(ctx);
            rank_job* j = &jobs[r];
            j->fab = &fab; j->rank = r; j->host = &host; j->g = g; j->g_lagrange = g_lagrange; j->n = n;
            j->advice = advice; j->inst = inst; j->inst_len = inst_len; j->st = st; j->st.at = 0; j->want = want; j->want_len = want_len; j->ok = 0; j->gathers = 0;
        }
        for (int r = 0; r < W; r++) pthread_create(&th[r], NULL, rank_main, &jobs[r]);
        for (int r = 0; r < W; r++) pthread_join(th[r], NULL);
        for (int r = 0; r < W; r++) { if (!jobs[r].ok) { fprintf(stderr, "rank %d of %d: proof differs or failed\n", r, W); return 1; } zk_ctx_destroy(fab.ctx[r]); }
        for (int r = 1; r < W; r++) if (jobs[r].gathers != jobs[0].gathers) { fprintf(stderr, "rank %d made %d all-gathers, rank 0 made %d\n", r, jobs[r].gathers, jobs[0].gathers); return 1; }
        printf("%d ranks (sharded tables, sharded keys, all-gather callback): every rank emitted the expected bytes, %d all-gathers per proof\n", W, jobs[0].gathers);
    }
    printf("capi_prove OK\n");
    return 0;
}
