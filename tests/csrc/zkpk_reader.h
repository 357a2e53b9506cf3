/* TEST-ONLY: reader of a ZKPK1 file (tools/dump_pk_blob.py — the HOST data a halo2 ProvingKey + ParamsKZG + witness + recorded rng stream + expected proof hold),
 * shared by the plain-C consumers of include/zkmi355.h: capi_prove.c (the call sequence of the Rust binding) and capi_faults.c (the exception barrier). */
#ifndef ZKPK_READER_H
#define ZKPK_READER_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkmi355.h"

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

typedef struct {
    zk_plonk_pk_host host;
    uint32_t k, n_advice, n_fixed, n_instance, L, P, draw_schedule;
    size_t n;
    const void* g; const void* g_lagrange;
    const void** advice; const void** inst; uint32_t* inst_len;
    stream st; const unsigned char* want; size_t want_len;
} zkpk;

static int zkpk_read(const char* path, zkpk* z) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
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
    zk_plonk_pk_host* host = &z->host;
    ZK_STRUCT_INIT(*host);
    host->k = k; host->cs_degree = cs_degree; host->blinding_factors = bf;
    host->n_fixed = n_fixed; host->n_advice = n_advice; host->n_instance = n_instance; host->n_lookups = L; host->n_perm_columns = P;
    host->perm_columns = lists; host->advice_queries = lists + 2 * P; host->n_advice_queries = n_aq;
    host->fixed_queries = lists + 2 * P + 2 * n_aq; host->n_fixed_queries = n_fq;
    host->lookup_table_key = lists + 2 * P + 2 * n_aq + 2 * n_fq;
    host->transcript_repr = transcript_repr; host->transcript = transcript; host->draw_schedule = draw_schedule;
    host->evaluator_zkq1_len = (size_t)take_u64(&r); host->evaluator_zkq1 = take(&r, host->evaluator_zkq1_len);
    const void** in_blob = (const void**)calloc(L + 1, sizeof(void*)); const void** tab_blob = (const void**)calloc(L + 1, sizeof(void*));
    size_t* in_len = (size_t*)calloc(L + 1, sizeof(size_t)); size_t* tab_len = (size_t*)calloc(L + 1, sizeof(size_t));
    for (uint32_t l = 0; l < L; l++) {
        in_len[l] = (size_t)take_u64(&r); in_blob[l] = take(&r, in_len[l]);
        tab_len[l] = (size_t)take_u64(&r); tab_blob[l] = take(&r, tab_len[l]);
    }
    host->lookup_input_zkq1 = in_blob; host->lookup_input_zkq1_len = in_len; host->lookup_table_zkq1 = tab_blob; host->lookup_table_zkq1_len = tab_len;
    z->g = take(&r, n * 64);
    z->g_lagrange = take(&r, n * 64);
    const void** fixed = (const void**)calloc(n_fixed + 1, sizeof(void*)); const void** sigma = (const void**)calloc(P + 1, sizeof(void*));
    for (uint32_t i = 0; i < n_fixed; i++) fixed[i] = take(&r, n * 32);
    for (uint32_t i = 0; i < P; i++) sigma[i] = take(&r, n * 32);
    host->fixed_values = fixed; host->sigma_values = sigma;
    z->advice = (const void**)calloc(n_advice + 1, sizeof(void*));
    for (uint32_t i = 0; i < n_advice; i++) z->advice[i] = take(&r, n * 32);
    z->inst = (const void**)calloc(n_instance + 1, sizeof(void*)); z->inst_len = (uint32_t*)calloc(n_instance + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n_instance; i++) { z->inst_len[i] = (uint32_t)take_u64(&r); z->inst[i] = take(&r, (size_t)z->inst_len[i] * 32); }
    z->st.draws = NULL; z->st.n = 0; z->st.at = 0; z->st.overrun = 0;
    z->st.n = take_u64(&r); z->st.draws = (const unsigned char*)take(&r, (size_t)z->st.n * 32);
    z->want_len = (size_t)take_u64(&r);
    z->want = (const unsigned char*)take(&r, z->want_len);
    z->k = k; z->n = n; z->n_advice = n_advice; z->n_fixed = n_fixed; z->n_instance = n_instance; z->L = L; z->P = P; z->draw_schedule = draw_schedule;
    return 0;
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
#endif
