/* TEST-ONLY: the exception barrier of the C ABI (include/zkmi355.h: "nothing throws or aborts"; csrc/abi_guard.h) driven from plain C through the fault hooks of
 * the EMULATOR build (csrc/capi.hip, -DZK_FAULT_INJECT: the n-th host allocation throws std::bad_alloc, the n-th std::thread start throws std::system_error).
 * For zk_quotient_program_load, zk_plonk_pk_build and zk_plonk_prove on the circuit of a ZKPK1 file: for a ladder of n the call must come back with a NEGATIVE
 * code and a text in zk_last_error — never a C++ exception, never an abort — and once n passes the call's allocation count it succeeds; after every failure the
 * SAME context must prove the expected bytes; at the end every device buffer the library took has been returned (emulator's hipMalloc / hipFree census).
 * usage: capi_faults FILE.zkpk   (links libzkmi355_emu.so; tests/test_abi_no_throw.py)   exit 0 = all of the above held */
#include "zkpk_reader.h"

void zk_test_fail_alloc(long nth);
void zk_test_fail_alloc_any_thread(long nth);
long zk_test_alloc_count(void);
void zk_test_fail_thread(int nth);
long zk_test_live_device_allocs(void);
int zk_test_alloc_hook_present(void);

#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zk_last_error(ctx)); return 1; } } while (0)
#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

static long next_rung(long n) { return n < 12 ? n + 1 : n + n / 3; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE.zkpk\n", argv[0]); return 1; }
    static zkpk z;
    if (zkpk_read(argv[1], &z)) return 1;
    const long live0 = zk_test_live_device_allocs();
    zk_ctx* ctx = NULL;
    if (zk_ctx_create(0, &ctx)) FAIL("zk_ctx_create failed");
    apply_tune(ctx);
    uint64_t h_g, h_gl, pk = 0, prog = 0;
    CK(zk_bases_register(ctx, z.g, z.n, &h_g));
    CK(zk_bases_register(ctx, z.g_lagrange, z.n, &h_gl));
    CK(zk_bases_enable_runs(ctx, h_gl));
    unsigned char* proof = (unsigned char*)malloc(z.want_len + 4096);
    size_t len = 0;
    int rc, failures;
    long n;
    int sections = getenv("ZK_FAULTS_SECTIONS") ? atoi(getenv("ZK_FAULTS_SECTIONS")) : 31;      /* bit i: inject in section i (debugging aid: which section leaks) */
    if (!zk_test_alloc_hook_present()) {                              /* an ASan / TSan build of the library: their runtimes own operator new */
        sections &= 16;
        printf("no allocation hook in this build of the library (sanitizer runtime): thread starts and the device-buffer census only\n");
    }

    /* ---- zk_quotient_program_load: the ZKQ1 compiler parses caller bytes into vectors, maps and shared_ptrs ---- */
    failures = 0;
    for (n = (sections & 1) ? 1 : 1000000000;; n = next_rung(n)) {
        zk_test_fail_alloc(n);
        rc = zk_quotient_program_load(ctx, z.host.evaluator_zkq1, z.host.evaluator_zkq1_len, &prog);
        zk_test_fail_alloc(0);
        if (rc == ZK_OK) { CK(zk_quotient_program_release(ctx, prog)); break; }
        if (rc != ZK_ERR_LIMIT || !strstr(zk_last_error(ctx), "out of host memory")) FAIL("zk_quotient_program_load with allocation %ld failing: rc %d, '%s'", n, rc, zk_last_error(ctx));
        failures++;
    }
    if (!failures && (sections & 1)) FAIL("zk_quotient_program_load: the fault hook never fired");
    printf("zk_quotient_program_load: %d injected allocation failures -> ZK_ERR_LIMIT each, then success (the call makes < %ld allocations) [%ld device buffers live]\n", failures, n, zk_test_live_device_allocs());

    /* ---- zk_plonk_pk_build ---- */
    failures = 0;
    for (n = (sections & 2) ? 1 : 1000000000;; n = next_rung(n)) {
        zk_test_fail_alloc(n);
        rc = zk_plonk_pk_build(ctx, &z.host, h_g, h_gl, &pk);
        zk_test_fail_alloc(0);
        if (rc == ZK_OK) break;
        if (rc != ZK_ERR_LIMIT) FAIL("zk_plonk_pk_build with allocation %ld failing: rc %d, '%s'", n, rc, zk_last_error(ctx));
        failures++;
    }
    if (!failures && (sections & 2)) FAIL("zk_plonk_pk_build: the fault hook never fired");
    printf("zk_plonk_pk_build: %d injected allocation failures -> ZK_ERR_LIMIT each, then the key (< %ld allocations) [%ld device buffers live]\n", failures, n, zk_test_live_device_allocs());

    /* ---- zk_plonk_prove: allocations of the calling thread ---- */
    failures = 0;
    int proofs_checked = 0;
    for (n = (sections & 4) ? 1 : 1000000000;; n = next_rung(n)) {
        z.st.at = 0;
        zk_test_fail_alloc(n);
        rc = zk_plonk_prove(ctx, pk, z.advice, 0, z.inst, z.inst_len, serve, &z.st, proof, z.want_len + 4096, &len);
        zk_test_fail_alloc(0);
        if (rc == ZK_OK) {
            if (len != z.want_len || memcmp(proof, z.want, len)) FAIL("zk_plonk_prove: proof differs after %d injected failures", failures);
            break;
        }
        if (rc != ZK_ERR_LIMIT) FAIL("zk_plonk_prove with allocation %ld failing: rc %d, '%s'", n, rc, zk_last_error(ctx));
        failures++;
        if (failures % 8 == 1) {                                      /* the same context proves the expected bytes right after a failed proof */
            z.st.at = 0;
            CK(zk_plonk_prove(ctx, pk, z.advice, 0, z.inst, z.inst_len, serve, &z.st, proof, z.want_len + 4096, &len));
            if (len != z.want_len || memcmp(proof, z.want, len)) FAIL("zk_plonk_prove: proof differs on the context of a failed proof (allocation %ld)", n);
            proofs_checked++;
        }
    }
    if (!failures && (sections & 4)) FAIL("zk_plonk_prove: the fault hook never fired");
    printf("zk_plonk_prove: %d injected allocation failures -> ZK_ERR_LIMIT each (%d followed by a good proof on the same context), then the expected bytes (< %ld allocations) [%ld device buffers live]\n", failures, proofs_checked, n, zk_test_live_device_allocs());

    /* ---- zk_plonk_prove: allocations of ANY thread (the rng thread's draw buffer, the side lane's jobs) ---- */
    int bad = 0, good = 0;
    for (n = (sections & 8) ? 1 : 1000000000; n < 400; n = next_rung(n)) {
        z.st.at = 0;
        zk_test_fail_alloc_any_thread(n);
        rc = zk_plonk_prove(ctx, pk, z.advice, 0, z.inst, z.inst_len, serve, &z.st, proof, z.want_len + 4096, &len);
        zk_test_fail_alloc_any_thread(0);
        if (rc == ZK_OK) { if (len != z.want_len || memcmp(proof, z.want, len)) FAIL("zk_plonk_prove: proof differs (any-thread allocation %ld)", n); good++; }
        else if (rc == ZK_ERR_LIMIT || rc == ZK_ERR_HIP) bad++;
        else FAIL("zk_plonk_prove with any-thread allocation %ld failing: rc %d, '%s'", n, rc, zk_last_error(ctx));
    }
    printf("zk_plonk_prove, failing allocation on whichever thread makes it: %d negative codes, %d unaffected proofs [%ld device buffers live]\n", bad, good, zk_test_live_device_allocs());

    /* ---- threads: the rng thread cannot start -> an error; the side lane's cannot -> the proof runs in one lane ---- */
    z.st.at = 0;
    if (sections & 16) {
    zk_test_fail_thread(1);
    rc = zk_plonk_prove(ctx, pk, z.advice, 0, z.inst, z.inst_len, serve, &z.st, proof, z.want_len + 4096, &len);
    zk_test_fail_thread(0);
    if (rc != ZK_ERR_HIP || !strstr(zk_last_error(ctx), "system_error")) FAIL("zk_plonk_prove without its rng thread: rc %d, '%s'", rc, zk_last_error(ctx));
    CK(zk_tune_set(ctx, "prover_side_lane", 2));
    z.st.at = 0;
    zk_test_fail_thread(2);
    rc = zk_plonk_prove(ctx, pk, z.advice, 0, z.inst, z.inst_len, serve, &z.st, proof, z.want_len + 4096, &len);
    zk_test_fail_thread(0);
    if (rc != ZK_OK || len != z.want_len || memcmp(proof, z.want, len)) FAIL("zk_plonk_prove without its side-lane thread: rc %d, '%s'", rc, zk_last_error(ctx));
    }
    z.st.at = 0;
    CK(zk_plonk_prove(ctx, pk, z.advice, 0, z.inst, z.inst_len, serve, &z.st, proof, z.want_len + 4096, &len));
    if (len != z.want_len || memcmp(proof, z.want, len)) FAIL("the last proof differs");
    printf("threads: no rng thread -> ZK_ERR_HIP (std::system_error), no side-lane thread -> one lane, same bytes [%ld device buffers live]\n", zk_test_live_device_allocs());

    /* ---- nothing leaked: keys, tables, pools and workspaces are all back ---- */
    CK(zk_plonk_pk_release(ctx, pk));
    CK(zk_bases_release(ctx, h_g)); CK(zk_bases_release(ctx, h_gl));
    zk_ctx_destroy(ctx);
    const long live1 = zk_test_live_device_allocs();
    if (live1 != live0) FAIL("device buffers leaked: %ld live before, %ld after", live0, live1);
    printf("device buffers: %ld live before the first call, %ld after zk_ctx_destroy\n", live0, live1);
    printf("capi_faults OK\n");
    free(proof);
    return 0;
}
