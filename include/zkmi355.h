/*
 * zkmi355.h — C ABI of libzkmi355.so: the MI355X (gfx950) backend for the halo2 KZG prover
 * hot path of CliqueOfficial/zk-dcap-verifier.
 *
 * The reference has no FFI of its own; it reaches this arithmetic only through
 *     create_proof   circuits/src/sgx_dcap_verifier.rs:814-822, crates/p256-ecdsa/src/base.rs:193-212
 *     keygen_vk/pk   circuits/src/sgx_dcap_verifier.rs:803,807,       crates/p256-ecdsa/src/base.rs:145
 *     gen_srs        circuits/src/sgx_dcap_verifier.rs:799,           crates/p256-ecdsa/src/base.rs:134
 * which bottom out in halo2_proofs 0.2.0 (zkwebauthn @ c254c75, Cargo.lock:1314-1327)
 * arithmetic::{best_multiexp, best_fft}, poly::EvaluationDomain::* and
 * plonk::evaluation::Evaluator::evaluate_h.  Each entry point below names the function of that
 * crate it replaces; INTEGRATION.md shows the Rust `extern "C"` shim a maintainer patches in with
 * Cargo [patch] (the mechanism the reference already uses, Cargo.toml:5-7).
 *
 * Conventions
 *   - Field elements are 32 bytes: 4 x u64 little-endian limbs in Montgomery form (R = 2^256),
 *     i.e. the in-memory representation of halo2curves::bn256::{Fr,Fq} — no conversion at the FFI.
 *   - G1Affine = {x, y} (64 B), identity = (0, 0).  G1 = {x, y, z} Jacobian (96 B).
 *   - All functions return 0 (ZK_OK) or a negative error code; nothing throws or aborts.
 *     zk_last_error() returns a human-readable description of the last failure on that context.
 *   - A context is bound to one GPU and is thread-safe (calls are serialised internally and
 *     block until the result is valid).  One process per GPU is the intended deployment; several
 *     contexts (one per device) may coexist in one process.
 *   - "host" pointers are ordinary process memory owned by the caller; the library never keeps a
 *     host pointer after the call returns.  "_dev" variants take device pointers (hipMalloc'ed
 *     or torch CUDA tensors) resident on the context's GPU.
 *   - There is NO CPU fallback: without a usable gfx950 device zk_ctx_create fails.
 */
#ifndef ZKMI355_H
#define ZKMI355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_ERR_ARG (-1)     /* bad argument (null pointer, size out of range, unknown handle) */
#define ZK_ERR_HIP (-2)     /* a HIP runtime call failed                                     */
#define ZK_ERR_NODEV (-3)   /* no usable GPU                                                  */
#define ZK_ERR_PROGRAM (-4) /* malformed / unsupported quotient program                       */
#define ZK_ERR_LIMIT (-5)   /* size limit of this build exceeded                              */
#define ZK_ERR_COMM (-6)    /* the caller's collective (zk_allgather_fn) reported a failure   */

typedef struct zk_ctx zk_ctx;

/* ---- ABI versioning ----------------------------------------------------------------------- *
 * Every struct that crosses this boundary (zk_quotient_args, zk_plonk_pk_desc, zk_plonk_pk_host) starts with `uint32_t struct_size`, which the caller sets to
 * ITS sizeof of the struct (ZK_STRUCT_INIT in C; size_of::<T>() in a Rust binding).  An entry point that receives another size than the library was built with
 * returns ZK_ERR_ARG ("struct_size N, expected M") before it reads any other field — a binding that has fallen behind the header fails loudly at its first
 * call instead of handing the library bytes past the end of its object.  Fields are only ever APPENDED, and each append bumps ZK_ABI_VERSION; a binding
 * asserts at start-up that zk_abi_version() is the ZK_ABI_VERSION it was written against and that zk_abi_struct_size(name) equals its own size of every
 * struct it declares (shim/halo2_proofs_mi355x/src/mi355x.rs does; tests/test_shim_abi.py diffs the declarations field by field). */
#define ZK_ABI_VERSION 4u
uint32_t zk_abi_version(void);
/* sizeof the named struct ("zk_quotient_args", "zk_plonk_pk_desc", "zk_plonk_pk_host") in this build of the library; 0 for an unknown name */
uint32_t zk_abi_struct_size(const char* struct_name);
#define ZK_STRUCT_INIT(s) do { memset(&(s), 0, sizeof(s)); (s).struct_size = (uint32_t)sizeof(s); } while (0)   /* needs <string.h> */

/* ---- context ------------------------------------------------------------------------------ */
int zk_ctx_create(int device_id, zk_ctx** out);
void zk_ctx_destroy(zk_ctx* ctx);
const char* zk_last_error(zk_ctx* ctx);
/* runtime tunables, e.g. "msm_c", "ntt_tile_log" (see DESIGN.md); unknown key -> ZK_ERR_ARG */
int zk_tune_set(zk_ctx* ctx, const char* key, int value);
int zk_tune_get(zk_ctx* ctx, const char* key, int* value);
/* per-kernel HIP-event timing, accumulated since zk_timing_enable(ctx, 1) (which also resets it).
 * zk_timing_get returns milliseconds for a kernel label ("msm_sort", "msm_accumulate", "msm_reduce",
 * "quotient"), the launch count for "<label>#n", the counters "msm_pairs" (point additions fed to the
 * accumulate kernel) and "msm_columns"; < 0 if unknown. */
int zk_timing_enable(zk_ctx* ctx, int on);
double zk_timing_get(zk_ctx* ctx, const char* label);

/* ---- device memory helpers (so callers need no HIP of their own) -------------------------- */
int zk_dev_alloc(zk_ctx* ctx, size_t bytes, void** dptr);
int zk_dev_free(zk_ctx* ctx, void* dptr);
int zk_dev_upload(zk_ctx* ctx, void* dptr, const void* host, size_t bytes);
int zk_dev_download(zk_ctx* ctx, void* host, const void* dptr, size_t bytes);
int zk_dev_copy(zk_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);   /* device -> device */
int zk_dev_zero(zk_ctx* ctx, void* dptr, size_t bytes);                           /* zero bytes (= Fr zero), ordered on the context's stream */
int zk_dev_sync(zk_ctx* ctx);
/* The witness crosses the boundary once per proof (create_proof receives host-owned circuits: sgx_dcap_verifier.rs:814-822, `&[circuit]`; halo2's prover
 * holds the synthesised advice columns as host Vec<Fr>).  zk_host_alloc hands out page-locked host memory the shim can synthesise / batch-invert the
 * columns into, so that the upload runs at link rate without a bounce buffer; zk_dev_upload_batch ships `count` columns of bytes_each in one call
 * (hosts / dptrs: HOST arrays of pointers; ordinary pageable memory works too, slower). */
int zk_host_alloc(zk_ctx* ctx, size_t bytes, void** hptr);
int zk_host_free(zk_ctx* ctx, void* hptr);
int zk_dev_upload_batch(zk_ctx* ctx, void* const* dptrs_dev, const void* const* hosts, size_t count, size_t bytes_each);

/* ---- MSM: replaces arithmetic::best_multiexp / ParamsKZG::{commit, commit_lagrange} -------- *
 * halo2_proofs src/arithmetic.rs best_multiexp(coeffs: &[Fr], bases: &[G1Affine]) -> G1 and
 * src/poly/kzg/commitment.rs commit{,_lagrange} (bases = params.g / params.g_lagrange, fixed per
 * SRS).  A base table is registered once (copied to HBM and expanded to its window multiples
 * 2^(c*j) * P_i so that every window shares one bucket set); each MSM then only ships scalars.
 * The window width c is chosen at registration from the TABLE's size (16 bits for 2^17 .. 2^21 points — the
 * prover's sizes, whose batches pay a bucket reduction per column — and 20 bits from 2^22 points: 13 windows per
 * scalar instead of 16 for single large MSMs; tune "msm_c" = 3 .. 22 before registering overrides it), so register
 * the table of the size you commit with — halo2's ParamsKZG::downsize(k) — rather than a prefix of a much larger SRS. */
int zk_bases_register(zk_ctx* ctx, const void* g1_affine_host, size_t n, uint64_t* handle);
int zk_bases_register_dev(zk_ctx* ctx, const void* g1_affine_dev, size_t n, uint64_t* handle);
int zk_bases_release(zk_ctx* ctx, uint64_t handle);
/* Optional second table for a registered base set: the window expansion of the prefix sums S_i = P_0 + ... + P_i (same size as the first).  With it,
 * zk_msm* commit a column whose neighbouring scalars are mostly EQUAL — the sorted permuted-input columns of halo2's lookup argument
 * (plonk/lookup/prover.rs permute_expression_pair), constant columns — through sum_i (a_i - a_(i+1)) S_i: zero differences cost nothing, so the work is that of
 * the column's run boundaries.  Chosen per column from a count of non-zero scalars vs non-zero differences; results are the same group elements.
 * One-time cost of a registration; shared by zk_bases_share. */
int zk_bases_enable_runs(zk_ctx* ctx, uint64_t handle);
/* Several contexts on one GPU (one per host thread that proves concurrently) use ONE expanded table: `ctx` receives a handle of its own onto the table
 * `owner_handle` of `owner` (same device); the HBM copy is freed when the last handle is released / the last holding context destroyed. */
int zk_bases_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_handle, uint64_t* handle);
/* out_jacobian: 96 B, always normalised: (x, y, mont(1)) or (0, 0, 0) for the identity — a valid
 * halo2curves G1 value.  n may be smaller than the registered table (prefix is used). */
int zk_msm(zk_ctx* ctx, uint64_t bases, const void* scalars_host, size_t n, void* out_jacobian);
int zk_msm_dev(zk_ctx* ctx, uint64_t bases, const void* scalars_dev, size_t n, void* out_jacobian);
/* `count` MSMs of n scalars each against the SAME table in one call (e.g. all advice columns of
 * a proof phase: the loop `for col in advice { params.commit_lagrange(col) }` of halo2_proofs
 * src/plonk/prover.rs).  scalars: HOST array of `count` column pointers (host / device memory);
 * out_jacobian: count x 96 B.  The latency-bound bucket-reduction phases are shared by the batch. */
int zk_msm_batch(zk_ctx* ctx, uint64_t bases, const void* const* scalars_host, size_t count, size_t n, void* out_jacobian);
int zk_msm_batch_dev(zk_ctx* ctx, uint64_t bases, const void* const* scalars_dev, size_t count, size_t n, void* out_jacobian);
/* unnormalised partial result as 128 B XYZZ (X, Y, ZZ, ZZZ) for multi-GPU sharding: partial
 * results of the ranks are exchanged (RCCL all-gather) and combined with zk_g1_sum_xyzz. */
int zk_msm_partial_dev(zk_ctx* ctx, uint64_t bases, const void* scalars_dev, size_t n, void* out_xyzz_host);
int zk_g1_sum_xyzz(const void* xyzz_host, size_t count, void* out_jacobian);
/* the same for a whole commitment phase: `count` columns against one (sharded) table, out_xyzz_host = count x 128 B; and the combination of the
 * `parts` ranks' results (xyzz_host laid out [part][column], as an all-gather delivers them) into count normalised points (count x 96 B). */
int zk_msm_batch_partial_dev(zk_ctx* ctx, uint64_t bases, const void* const* scalars_dev, size_t count, size_t n, void* out_xyzz_host);
int zk_g1_sum_xyzz_batch(const void* xyzz_host, size_t parts, size_t count, void* out_jacobian);

/* fixed-base batch: out[i] = [s_i] * G1::generator(), affine.  Replaces the n fixed-base
 * multiplications of ParamsKZG::setup (halo2_proofs src/poly/kzg/commitment.rs), reached from
 * gen_srs (sgx_dcap_verifier.rs:799); also used to build synthetic SRS-shaped tables. */
int zk_g1_fixed_base_mul_dev(zk_ctx* ctx, const void* scalars_dev, size_t n, void* out_affine_dev);

/* best_fft over G = G1 (the "EC-FFT" of ParamsKZG::setup's g_to_lagrange): out[j] = [scale] sum_i [omega^(i j)] in[i], affine in
 * and out (DEVICE, 2^log_n x 64 B); omega, scale: HOST 32 B Montgomery, scale may be NULL.  One-time setup work. */
int zk_g1_ntt_dev(zk_ctx* ctx, const void* g1_affine_in_dev, uint32_t log_n, const void* omega, const void* scale, void* g1_affine_out_dev);

/* G1Affine::{from_bytes, to_bytes} in bulk — the point encoding of ParamsKZG::{read, write} (halo2_proofs src/poly/kzg/commitment.rs), i.e. of the SRS file
 * params/kzg_bn254_{k}.srs that gen_srs caches (sgx_dcap_verifier.rs:799; bin/src/main.rs:227-231).  32 bytes per point: little-endian canonical x, y parity in
 * bit `sign_bit`: 255 = halo2curves 0.3.1 (stack A; identity = all zero), 254 = halo2curves-axiom 0.5.2 (stack B; bit 255 = identity).  Decompression costs one
 * square root per point ((x^3 + 3)^((p+1)/4)); *n_invalid counts encodings that are not curve points (then the call returns ZK_ERR_ARG). */
int zk_g1_decompress_dev(zk_ctx* ctx, const void* bytes_dev, size_t n, uint32_t sign_bit, void* out_affine_dev, uint32_t* n_invalid);
int zk_g1_compress_dev(zk_ctx* ctx, const void* affine_dev, size_t n, uint32_t sign_bit, void* bytes_dev);

/* ---- NTT: replaces arithmetic::best_fft (G = Fr) and the EvaluationDomain wrappers --------- *
 * halo2_proofs src/arithmetic.rs best_fft(a, omega, log_n): in place, natural order in and out,
 * out[j] = sum_i a[i] * omega^(i*j).  omega: 32 B Montgomery.                                   */
int zk_ntt(zk_ctx* ctx, void* a_host, uint32_t log_n, const void* omega);
int zk_ntt_dev(zk_ctx* ctx, void* a_dev, uint32_t log_n, const void* omega);
/* halo2_proofs src/poly/domain.rs — fused forms (j = cs.degree(), as EvaluationDomain::new(j,k)) */
int zk_lagrange_to_coeff_dev(zk_ctx* ctx, void* a_dev, uint32_t k);                 /* ifft * n^-1  */
int zk_coeff_to_lagrange_dev(zk_ctx* ctx, void* a_dev, uint32_t k);
/* coeff (2^k) -> extended coset evaluations (2^extended_k): zeta^(i mod 3) scaling, zero pad, NTT */
int zk_coeff_to_extended_dev(zk_ctx* ctx, const void* coeff_dev, uint32_t k, uint32_t extended_k, void* out_dev);
/* inverse of the above, in place on 2^extended_k values; entries [0, out_len) are the result    */
int zk_extended_to_coeff_dev(zk_ctx* ctx, void* a_ext_dev, uint32_t k, uint32_t extended_k);
/* a[i] *= t_evaluations[i mod 2^(extended_k-k)] (1/(X^n - 1) on the coset)                      */
int zk_divide_by_vanishing_poly_dev(zk_ctx* ctx, void* a_ext_dev, uint32_t k, uint32_t extended_k);
/* batched forms: `count` columns per call (HOST arrays of DEVICE pointers) — one launch per pass for the
 * whole phase, e.g. the advice-column loop of create_proof / evaluate_h                                */
int zk_ntt_batch_dev(zk_ctx* ctx, void* const* cols_dev, size_t count, uint32_t log_n, const void* omega);
int zk_lagrange_to_coeff_batch_dev(zk_ctx* ctx, void* const* cols_dev, size_t count, uint32_t k);
int zk_coeff_to_extended_batch_dev(zk_ctx* ctx, const void* const* coeffs_dev, void* const* outs_dev, size_t count, uint32_t k, uint32_t extended_k);
int zk_lagrange_to_coeff(zk_ctx* ctx, void* a_host, uint32_t k);
int zk_coeff_to_extended(zk_ctx* ctx, const void* coeff_host, uint32_t k, uint32_t extended_k, void* out_host);
int zk_extended_to_coeff(zk_ctx* ctx, void* a_ext_host, uint32_t k, uint32_t extended_k);

/* ---- element-wise Fr vector ops used around the NTTs (device) ------------------------------ */
int zk_fr_mul_dev(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n);
int zk_fr_add_dev(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n);
int zk_fr_sub_dev(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n);
int zk_fr_scale_dev(zk_ctx* ctx, const void* a, const void* scalar_host, void* out, size_t n);
/* Fq flavour of mul (used by the parity tests of the curve field) */
int zk_fq_mul_dev(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n);

/* ---- grand products (SURVEY 8f "next 1") ------------------------------------------------------ *
 * halo2_proofs src/plonk/permutation/prover.rs Argument::commit — the row loops of ONE column set:
 *   z[0] = z_init;  z[i+1] = z[i] * prod_j (v_j[i] + delta^(j0+j) beta omega^i + gamma) / prod_j (v_j[i] + beta sigma_j[i] + gamma)
 *   rows [n - blinding_factors, n) <- blinding (HOST, the caller's Fr::random draws);  *last_z_out = z[n - blinding_factors - 1]
 * values / sigmas: HOST arrays of `count` DEVICE columns (Lagrange basis, n = 2^k rows); delta_start = DELTA^(j0) with j0 the index of
 * the set's first column in cs.permutation.columns; beta, gamma, delta_start, z_init, last_z_out: HOST 32 B.  z_dev: DEVICE n x 32 B. */
int zk_permutation_product_dev(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t count, uint32_t k, const void* beta,
                               const void* gamma, const void* delta_start, const void* z_init, const void* blinding, uint32_t blinding_factors,
                               void* z_dev, void* last_z_out);
/* halo2_proofs src/plonk/lookup/prover.rs Permuted::commit_product:
 *   z[0] = 1;  z[i+1] = z[i] * (cin[i] + beta)(ctab[i] + gamma) / ((pin[i] + beta)(ptab[i] + gamma));  last rows <- blinding */
int zk_lookup_product_dev(zk_ctx* ctx, const void* compressed_input, const void* compressed_table, const void* permuted_input,
                          const void* permuted_table, uint32_t k, const void* beta, const void* gamma, const void* blinding,
                          uint32_t blinding_factors, void* z_dev);

/* permutation::Argument::commit for ALL column sets in one call: values / sigmas = the n_columns equality-enabled columns in cs.permutation.columns order
 * (HOST arrays of DEVICE columns), cut into sets of chunk_len = cs.degree() - 2; set s starts at the last unblinded value of set s - 1 and delta powers
 * continue across sets; blinding: HOST n_sets x blinding_factors x 32 B; z_devs: HOST array of n_sets DEVICE outputs. */
int zk_permutation_product_all_dev(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t n_columns, uint32_t chunk_len, uint32_t k,
                                   const void* beta, const void* gamma, const void* blinding, uint32_t blinding_factors, void* const* z_devs);
/* commit_product of `count` lookups in one call: cols4 = HOST array of 4*count DEVICE columns (compressed_input, compressed_table, permuted_input,
 * permuted_table per lookup); blinding: HOST count x blinding_factors x 32 B; z_devs: HOST array of count DEVICE outputs (n x 32 B each). */
int zk_lookup_product_batch_dev(zk_ctx* ctx, const void* const* cols4, size_t count, uint32_t k, const void* beta, const void* gamma, const void* blinding,
                                uint32_t blinding_factors, void* const* z_devs);

/* halo2_proofs src/plonk/lookup/prover.rs permute_expression_pair (SURVEY 8f "next 4"): input / table = the theta-compressed
 * expressions over the n = 2^k rows (DEVICE); the first n - (blinding_factors + 1) rows are permuted (A' sorted, S' aligned),
 * the remaining rows take blind_input / blind_table (HOST, (blinding_factors + 1) x 32 B each — the caller's Fr::random draws).
 * An input value absent from the table returns ZK_ERR_ARG (halo2: Error::ConstraintSystemFailure). */
int zk_lookup_permute_dev(zk_ctx* ctx, const void* input, const void* table, uint32_t k, uint32_t blinding_factors, const void* blind_input,
                          const void* blind_table, void* out_input, void* out_table);

/* all lookup arguments of a proof in one call (the `for lookup in lookups { lookup.commit_permuted(..) }` loop of create_proof): inputs /
 * tables / out_inputs / out_tables are HOST arrays of `count` DEVICE columns; blind_inputs / blind_tables: HOST, count x (blinding_factors + 1) x 32 B. */
int zk_lookup_permute_batch_dev(zk_ctx* ctx, const void* const* inputs, const void* const* tables, size_t count, uint32_t k, uint32_t blinding_factors,
                                const void* blind_inputs, const void* blind_tables, void* const* out_inputs, void* const* out_tables);

/* ---- evaluation phase (SURVEY 8f "next 2") ----------------------------------------------------- *
 * halo2_proofs src/arithmetic.rs eval_polynomial(poly, point): out[q] = polys[q](points[q]) for `count` queries of n coefficients
 * each (a polynomial queried at several rotations appears several times).  polys: HOST array of DEVICE pointers; points, out: HOST. */
int zk_eval_polynomial_batch_dev(zk_ctx* ctx, const void* const* polys_dev, size_t count, size_t n, const void* points, void* out);
/* halo2_proofs src/arithmetic.rs kate_division(a, b): q(X) = (a(X) - a(b)) / (X - b); a: n coefficients, q: n - 1 (DEVICE); b: HOST 32 B */
int zk_kate_division_dev(zk_ctx* ctx, const void* a_dev, size_t n, const void* b, void* q_dev);
/* halo2_proofs src/poly/kzg/multiopen/shplonk/prover.rs — the `poly * power_of_y … reduce(acc + &poly)` combinations of create_proof:
 * out[i] = sum_j scalars[j] * polys[j][i], i < n, in one pass.  polys: HOST array of `count` DEVICE polynomials (n coefficients each);
 * scalars: HOST count x 32 B; out_dev: DEVICE n x 32 B (may alias one of the inputs). */
int zk_fr_lincomb_dev(zk_ctx* ctx, const void* const* polys_dev, const void* scalars, size_t count, size_t n, void* out_dev);

/* ---- quotient: replaces plonk::evaluation::Evaluator::evaluate_h ---------------------------- *
 * halo2_proofs src/plonk/evaluation.rs.  The compiled GraphEvaluator of a proving key is uploaded
 * once as a "ZKQ1" blob (layout in DESIGN.md / INTEGRATION.md), then run per proof on
 * device-resident extended cosets.  Column pointer arrays are HOST arrays of DEVICE pointers.     */
typedef struct zk_quotient_args {
    uint32_t struct_size;            /* sizeof(zk_quotient_args) of the caller (ABI versioning, above) */
    const void* const* fixed;        /* n_fixed cosets                                        */
    const void* const* advice;       /* n_advice cosets                                       */
    const void* const* instance;     /* n_instance cosets                                     */
    const void* l0;
    const void* l_last;
    const void* l_active_row;
    const void* const* perm_cosets;  /* sigma cosets, one per permutation column              */
    const void* const* perm_products;/* z cosets, one per set                                 */
    uint32_t n_sets;
    const void* const* lookup_product;
    const void* const* lookup_input;
    const void* const* lookup_table;
    const void* challenges;          /* HOST: n_challenges x 32 B                             */
    const void* beta;                /* HOST 32 B each                                        */
    const void* gamma;
    const void* theta;
    const void* y;
    void* out;                       /* DEVICE: 2^extended_k x 32 B, overwritten              */
} zk_quotient_args;
int zk_quotient_program_load(zk_ctx* ctx, const void* blob, size_t len, uint64_t* prog);
int zk_quotient_program_release(zk_ctx* ctx, uint64_t prog);
/* a compiled program is immutable: `ctx` receives a handle of its own onto the program `owner_prog` of `owner` (same device), so that several contexts
 * (one per host thread that proves concurrently) run ONE compiled Evaluator; it is freed with its last handle */
int zk_quotient_program_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_prog, uint64_t* prog);
/* size of the compiled micro-program: instructions, live-value slots (the first is a register, the rest LDS), columns */
int zk_quotient_program_info(zk_ctx* ctx, uint64_t prog, uint32_t* n_instr, uint32_t* n_slots, uint32_t* n_columns);
/* Generated kernels (csrc/quotient_jit.hip).  With tune "quot_jit" = 1 at zk_quotient_program_load / zk_plonk_pk_build a program whose extended domain is larger than its
 * domain is ALSO turned into straight-line kernels written for it — one statement per micro-op on the same field functions, "quot_jit_group" products per kernel — and compiled
 * for gfx950 by hiprtc inside the load (13 s at level 1 and about 45 s at level 2 for the sgx-shaped program on a box that has not seen it; comgr caches the code objects, the next load takes 0.1 s).  Every
 * zk_quotient_run*_dev and zk_plonk_prove on that program, from any context that holds or borrows it, then runs those kernels instead of the micro-op interpreter: the same field
 * elements (the same proof bytes), about a fifth less kernel time on the quotient, +2.7-3.0 % proofs per hour at k = 19 (profiles/r05/run303).  A load that asks for them and cannot
 * have them (no libhiprtc.so, a compile error) FAILS with ZK_ERR_PROGRAM: the executor is never swapped silently.  Default 0: the interpreter.
 * "quot_jit" = 1 generates what a proof on ONE GPU launches: the high and low degree parts of a split program ("quot_degree_split", the default) and nothing for the whole
 * program — zk_quotient_run_dev / _coset_dev / _coset_rows_dev on it (a sharded proof's units, a proof that turns the split off) stay on the interpreter — or the whole program
 * when there is no split; that halves the compile (compile time is linear in the products generated).  "quot_jit" = 2 generates all three.
 * zk_quotient_program_kernels: how many generated kernels `prog` runs (its high and low degree parts included); 0 = the interpreter. */
int zk_quotient_program_kernels(zk_ctx* ctx, uint64_t prog, uint32_t* n_kernels);
/* opcode census of the compiled micro-program (the arithmetic a row costs — what the kernel's roofline is priced from):
 * counts[0..7] = add, sub, mul, sqr, dbl, neg, mov, mul-add instructions (a fused fold value*y + a*b — two products, one reduction — counts as a mul-add); counts[8] = column / constant operands read from memory per row */
int zk_quotient_program_opmix(zk_ctx* ctx, uint64_t prog, uint32_t counts[9]);
int zk_quotient_run_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args);
/* The same on ONE coset of the extended domain (the rows coset, coset + 2^(extended_k-k), ...): every column of `args` holds that coset's n = 2^k values
 * (zk_coeff_to_coset_batch_dev), args->out receives the coset's n numerator values.  The 2^(extended_k-k) cosets are independent, so the quotient of a
 * proof can be split over GPUs; zk_fr_interleave_dev puts the gathered cosets back into the order Evaluator::evaluate_h returns. */
int zk_quotient_run_coset_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t coset);
/* ... and on a slice of that coset's rows: rows [row_lo, row_lo + row_count) — row_count a power of two dividing row_lo — with the columns still the
 * coset's complete n values (rotations reach outside the slice); args->out receives row_count values.  This is the unit when a proof's quotient is split over
 * MORE ranks than there are cosets (8 GPUs at extended_k = k + 2): ranks that share a coset each evaluate a part of its rows. */
int zk_quotient_run_coset_rows_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t coset, uint64_t row_lo, uint64_t row_count);
/* Degree split of the numerator.  h's numerator is sum_i y^(N-1-i) id_i over the identities halo2 folds with y (gates, permutation, lookups); an identity of
 * degree d in the columns contributes a share of h(X) of degree below (d - 1) n, which d - 1 cosets of the size-n domain determine.  A program whose constraint system has
 * degree >= 4 is therefore compiled a second and third time: its HIGH part (identities of degree > 3) is evaluated on every row as before, its LOW part (degree <= 3: halo2-lib's
 * gates, the boundary / ordering identities of every lookup and permutation set — about half of the arithmetic of the sgx-shaped program) only on the rows of cosets 0 and 1, and
 * joins through zk_cosets_to_pieces_dev(.., pieces = 2, ..): h = (pieces of the high part) + (the two pieces of the low part).  Same field elements as the unsplit evaluation
 * whenever the witness satisfies the circuit (every identity then vanishes on the domain, so both shares are polynomials); for a witness that violates it the (invalid) proofs
 * differ — tune "quot_degree_split" = 0 before zk_quotient_program_load / zk_plonk_pk_build keeps halo2's bytes there too.
 * DEFAULT: on (1).  Where the setting is read: (1) when a program is loaded — that decides whether the two parts exist, and the PROGRAM records it (zk_quotient_program_split
 * is the way to ask; the tunable says nothing about a key built earlier); (2) at every zk_plonk_create_proof / zk_plonk_prove, which takes the split route only when the tunable is on
 * AND the key's program carries the parts.  So 0 at proof time switches the split off for a key that has it; 1 at proof time cannot switch it on for a key built with 0 — that proof
 * silently runs unsplit (same bytes for a satisfied circuit, about 2 % slower).  Set it once, before the key is built, and leave it.
 * zk_quotient_program_split: *low_cosets = 2 when `prog` carries the two parts, 0 when it does not (then only zk_quotient_run*_dev above apply).
 * zk_quotient_run_high_dev: as zk_quotient_run_dev, high part only.  zk_quotient_run_low_dev: columns of the whole extended domain as for zk_quotient_run_dev, low part on the
 * rows of cosets 0 .. low_cosets-1; args->out receives low_cosets x n values, coset-major.  zk_quotient_run_coset_part_dev: as zk_quotient_run_coset_dev, part 1 = high, 2 = low. */
int zk_quotient_program_split(zk_ctx* ctx, uint64_t prog, uint32_t* low_cosets, uint32_t* n_instr_high, uint32_t* n_instr_low);
int zk_quotient_program_part_opmix(zk_ctx* ctx, uint64_t prog, uint32_t part, uint32_t counts[9]);   /* zk_quotient_program_opmix of part 1 = high / 2 = low */
int zk_quotient_run_high_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args);
int zk_quotient_run_low_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t low_cosets);
int zk_quotient_run_coset_part_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t coset, uint32_t part);
int zk_coeff_to_coset_batch_dev(zk_ctx* ctx, const void* const* coeffs_dev, void* const* outs_dev, size_t count, uint32_t k, uint32_t extended_k, uint32_t coset);
int zk_fr_interleave_dev(zk_ctx* ctx, const void* const* cosets_dev, size_t count, size_t n, void* out_dev);   /* out[i * count + j] = cosets[j][i] */
/* The pieces of h(X) straight from the numerator's values on cosets 0 .. pieces-1 (vanishing::Argument::construct: divide_by_vanishing_poly + extended_to_coeff + the split
 * into cs_degree - 1 polynomials of n coefficients).  deg h < pieces * n, and on coset j X^n is a constant s_j, so `pieces` cosets determine h: per coset one inverse size-n
 * transform, then a pieces x pieces Vandermonde solve per coefficient index.  The 2^(extended_k - k) - pieces other cosets of the extended domain need never be evaluated —
 * a quarter of the extended transforms and of the quotient rows at cs_degree = 4 (halo2-lib circuits), nothing when cs_degree - 1 is a power of two (the sgx circuit: 5).
 * numer_dev[j]: n values (DEVICE, clobbered), out_dev[i]: n coefficients (DEVICE, distinct buffers).  pieces <= min(8, 2^(extended_k - k)).  Same field elements as the
 * extended route (h is unique). */
int zk_cosets_to_pieces_dev(zk_ctx* ctx, void* const* numer_dev, uint32_t pieces, uint32_t k, uint32_t extended_k, void* const* out_dev);

/* proving-key level form — the shape of halo2's own call (polynomials in, polynomial out):
 * zk_pk_load uploads what keygen_pk holds for the evaluator — fixed columns, permutation (sigma) columns, l0, l_last,
 * l_active_row — either as coefficient-form polynomials of n = 2^k (form 0; expanded with coeff_to_extended on the
 * device) or as the extended cosets pk already stores (form 1), all HOST pointers.
 * zk_evaluate_h takes the proof's COEFFICIENT-form polynomials (HOST, n x 32 B each: advice, instance, permutation
 * products z, and per lookup the product / permuted input / permuted table polynomials), expands them on the device,
 * runs the program and returns on the HOST either the extended numerator (finish = 0: 2^extended_k x 32 B, what
 * Evaluator::evaluate_h returns) or, with finish = 1, h(X) itself after divide_by_vanishing_poly and
 * extended_to_coeff ((cs_degree - 1) * n coefficients, what vanishing::Argument::construct splits and commits). */
int zk_pk_load(zk_ctx* ctx, uint64_t prog, const void* const* fixed, const void* const* sigma, const void* l0, const void* l_last,
               const void* l_active_row, int form, uint64_t* pk);
int zk_pk_release(zk_ctx* ctx, uint64_t pk);
int zk_evaluate_h(zk_ctx* ctx, uint64_t pk, const void* const* advice_polys, const void* const* instance_polys,
                  const void* const* perm_product_polys, const void* const* lookup_product_polys, const void* const* lookup_input_polys,
                  const void* const* lookup_table_polys, const void* challenges, const void* beta, const void* gamma, const void* theta,
                  const void* y, int finish, void* out);

/* ---- the whole per-proof path: replaces plonk::create_proof + ProverSHPLONK ------------------------------------- *
 * halo2_proofs src/plonk/prover.rs create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK, Challenge255, _, Blake2bWrite, _> as the reference calls it
 * (circuits/src/sgx_dcap_verifier.rs:814-822), single circuit instance, no user challenges: every O(n) step runs through the entry points above in the
 * order of INTEGRATION.md's phase table, columns stay in HBM from the advice commitment to the last SHPLONK commitment, Fiat-Shamir hashing (Blake2b,
 * Challenge255), point encoding (y parity in bit 255) and the rotation-set bookkeeping run on the host inside this call.  This is the native (C++) form of
 * the phase-batched prover; zk-dcap-verifier_amd/plonk/prover.py is its Python twin and both emit the same bytes for the same inputs and draws.
 *
 * The proving key is described by pointers to what keygen_pk leaves behind, all resident on the context's GPU (the caller keeps ownership):
 * fixed / sigma columns in Lagrange, coefficient and extended-coset form, l0 / l_last / l_active_row cosets, the compiled Evaluator (zk_quotient_program_load)
 * and, per lookup, the two expression programs that theta-compress its input and table expressions over the 2^k rows (ZKQ1 programs with extended_k = k);
 * lookups with equal lookup_table_key share their compressed table.  Queries are (column, rotation) pairs in cs.advice_queries / cs.fixed_queries order. */
/* One proof over several GPUs (BASELINE configs[4]): the collective between the ranks is the CALLER's (RCCL ncclAllGather over xGMI; anything else
 * in tests).  Gather `bytes` bytes at send_dev of every rank into recv_dev (rank r's block at recv_dev + r * bytes) — DEVICE pointers on the context's GPU —
 * and return 0 once recv_dev is complete and send_dev may be overwritten (the library has synchronised its own stream before the call and uses the data right
 * after it).  Every rank calls it the same number of times with the same sizes: once per commitment phase (128-byte points) and once per proof for the
 * quotient's numerators.
 * A rank whose proof fails on its own (out of memory, a HIP error, a witness outside a lookup table) enters the NEXT exchange once more with a block whose first 32
 * bytes are 0xFF and returns its error; the other ranks find the mark and return ZK_ERR_COMM from that same exchange, so all ranks leave the proof together.  What the
 * library cannot cover is the collective itself hanging or a rank dying: the callback MUST enforce a timeout and return non-zero when it expires. */
typedef int (*zk_allgather_fn)(void* user, const void* send_dev, void* recv_dev, size_t bytes);

typedef struct zk_plonk_pk_desc {
    uint32_t struct_size;                     /* sizeof(zk_plonk_pk_desc) of the caller (ABI versioning) */
    uint32_t k, extended_k, cs_degree, blinding_factors;
    uint32_t n_fixed, n_advice, n_instance, n_lookups, n_perm_columns;
    const uint32_t* perm_columns;             /* n_perm_columns x (column_type: 0 advice 1 fixed 2 instance, index) */
    const uint32_t* advice_queries; uint32_t n_advice_queries;   /* pairs (column, rotation as int32) */
    const uint32_t* fixed_queries;  uint32_t n_fixed_queries;
    uint64_t srs_g, srs_g_lagrange;           /* zk_bases_register handles of params.g / params.g_lagrange */
    uint64_t program;                         /* the proving key's Evaluator */
    const uint64_t* lookup_input_programs;    /* n_lookups */
    const uint64_t* lookup_table_programs;    /* n_lookups */
    const uint32_t* lookup_table_key;         /* n_lookups */
    const void* const* fixed_values; const void* const* fixed_polys; const void* const* fixed_cosets;   /* DEVICE, n_fixed each */
    const void* const* sigma_values; const void* const* sigma_polys; const void* const* sigma_cosets;   /* DEVICE, n_perm_columns each */
    const void* l0; const void* l_last; const void* l_active_row;                                       /* DEVICE extended cosets */
    const void* transcript_repr;              /* HOST 32 B: vk.transcript_repr, canonical little endian */
    uint32_t transcript;                      /* 0: Blake2bWrite + Challenge255, y-parity flag in bit 255 (stack A, sgx_dcap_verifier.rs:813);
                                               * 1: snark-verifier PoseidonTranscript<NativeLoader> (T = 3, RATE = 2, R_F = 8, R_P = 57), flag in bit 254 (stack B gen_proof, base.rs:200-212);
                                               * 2: snark-verifier EvmTranscript (Keccak-256, 32-byte big-endian words, uncompressed points; gen_evm_proof_shplonk, base.rs:193-199) */
    uint32_t draw_schedule;                   /* order in which the caller's rng is consumed (see zk_rng_fn): 1 = halo2_proofs v2023_01_20 (PSE; stack A). The only schedule: any other value is ZK_ERR_ARG */
    /* ---- one proof over shard_world GPUs, one process / context per GPU; shard_world <= 1 = single GPU and everything below is ignored ---------------------- *
     * MSM: srs_g / srs_g_lagrange are tables of n / shard_world points holding bases [shard_rank * n / shard_world, ...) of params.g / params.g_lagrange; every
     *   commitment phase is one partial MSM batch per rank on its slice of the scalars + ONE all-gather of 128-byte XYZZ points, summed on every rank.
     * Quotient: the extended domain is 2^(extended_k - k) interleaved cosets of the 2^k domain and evaluate_h never mixes them: unit u = (coset u / parts, the
     *   (u % parts)-th slice of n / parts rows), parts = shard_world / cosets when there are more ranks than cosets (else 1); rank r evaluates units
     *   [r * slots, (r + 1) * slots), slots = ceil(units / shard_world), on size-n coset NTTs of the proof's columns, and ONE all-gather carries the numerators.
     *   The proving key's cosets come per coset: for this rank's cosets in ascending order, n values per column — fixed_cosets / sigma_cosets / l0.. are unused.
     * Everything else (lookups, grand products, iNTTs, evaluations, SHPLONK arithmetic) is computed on every rank from the same witness and the same rng stream,
     * so every rank emits the same proof — byte for byte the single-GPU prover's. */
    uint32_t shard_world, shard_rank;
    zk_allgather_fn allgather; void* allgather_user;
    void* xchg_send; void* xchg_recv; size_t xchg_cap;   /* optional caller-owned DEVICE exchange buffers (send: xchg_cap bytes, recv: shard_world x xchg_cap), e.g. two
                                                          * torch tensors so that the callback can hand them to torch.distributed; NULL = the library allocates its own */
    const void* const* coset_fixed;           /* [this rank's cosets][n_fixed]        DEVICE, n x 32 B each */
    const void* const* coset_sigma;           /* [this rank's cosets][n_perm_columns]                         */
    const void* const* coset_l;               /* [this rank's cosets][3]: l0, l_last, l_active_row            */
    /* shard_world <= 1 with coset_l set and cs_degree - 1 < 2^(extended_k - k): the three arrays hold cosets 0 .. cs_degree-2 and the extended forms above are not read —
     * the numerator is evaluated on those cosets only and the pieces of h(X) come from zk_cosets_to_pieces_dev (what zk_plonk_pk_build sets up by default) */
} zk_plonk_pk_desc;
/* the caller's RNG (`&mut rng` of create_proof): fill out_fr with n uniform field elements as Montgomery limbs (n x 32 B).  Called from a helper thread of the
 * library, once per Fr::random block.  draw_schedule 1 follows halo2_proofs v2023_01_20 draw by draw ([3P-MEM]) — including the Blind(Fr::random) every
 * commitment draws, which KZG discards but which advances the stream:
 *   plonk/prover.rs            per advice column its n - usable_rows blinding rows; then one Blind per advice column
 *   lookup/prover.rs           per lookup: blinding_factors + 1 permuted-input rows, as many permuted-table rows (permute_expression_pair), two Blinds (commit_values)
 *   permutation/prover.rs      per column set: blinding_factors rows, one Blind
 *   lookup/prover.rs           per lookup product: blinding_factors rows, one Blind
 *   vanishing/prover.rs        commit: the n coefficients of the random polynomial, one Blind;  construct: one Blind per h(X) piece
 * so a proof consumes exactly the draws the CPU create_proof would and leaves the caller's rng in the same state (all draws are complete when the call returns ZK_OK).
 * Stack B's prover (halo2-axiom 0.4.2) is NOT claimed draw for draw: that fork commits with Blind::default() and is believed not to draw the Blinds ([3P-MEM], source
 * absent here), so under transcripts 1 / 2 the proofs verify and are deterministic in the rng stream, but byte parity with the axiom CPU prover is out of scope until its
 * schedule is pinned by shim/p256_k18_driver's dump.  Multi-phase advice and the Challenge API are not modelled either (zk_plonk_pk_build's caller must not pass such circuits). */
typedef void (*zk_rng_fn)(void* user, size_t n, void* out_fr);
/* advice: n_advice columns of 2^k x 32 B (HOST, or DEVICE when advice_on_device — then consumed: their last rows take the blinding values, and they hold coefficient
 * forms or blinded values afterwards, whichever route the proof took); instances: HOST,
 * instance_lens[c] canonical 32-byte values per instance column.  The proof (32 bytes per commitment and per evaluation; 64 per commitment under transcript 2) is written to proof_out;
 * *proof_len receives its length (ZK_ERR_LIMIT when proof_cap is too small).  Errors of the entry points it drives are returned as they are
 * (e.g. ZK_ERR_ARG from zk_lookup_permute_batch_dev for a lookup input outside its table: halo2's Error::ConstraintSystemFailure). */
/* One proof alone does not fill the GPU (a third of its commitment phases is the MSM's sort, reduction tail and host folds): a single-GPU proof on the extended domain
 * therefore runs lagrange_to_coeff + coeff_to_extended of every phase's columns on a HELPER context the library keeps for `ctx` (own stream and workspaces, a helper host
 * thread) while the phase's commitments run on `ctx` — tunable "prover_side_lane": 1 (default) when at most two proofs are in flight in the process, 2 always, 0 never.
 * No byte of the proof depends on it; with three and more proofs in flight (one context per proving thread) the other proofs fill the GPU and it stays off. */
int zk_plonk_create_proof(zk_ctx* ctx, const zk_plonk_pk_desc* pk, const void* const* advice, int advice_on_device, const void* const* instances,
                          const uint32_t* instance_lens, zk_rng_fn rng, void* rng_user, void* proof_out, size_t proof_cap, size_t* proof_len);
/* ---- the proving key as the library's own object: what a Rust / C caller uses instead of filling zk_plonk_pk_desc by hand ------------------------------ *
 * zk_plonk_pk_build is the device half of halo2's keygen_pk (src/plonk/keygen.rs; reference call sites sgx_dcap_verifier.rs:807, p256-ecdsa base.rs:145): from the
 * HOST data a halo2 ProvingKey holds — pk.fixed_values, pk.permutation.permutations (Lagrange columns of n = 2^k x 32 B), the constraint system's shape and the
 * Evaluator / lookup expressions serialised as ZKQ1 blobs — it uploads the columns and derives on the GPU what keygen_pk derives on the CPU: coefficient forms
 * (lagrange_to_coeff), extended cosets (coeff_to_extended) and l0 / l_last / l_active_row; extended_k follows EvaluationDomain::new(cs_degree, k).  Nothing of
 * `host` is referenced after the call.  The key lives in HBM once per process: zk_plonk_pk_share gives another context of the same device (one context per
 * concurrently proving host thread) a handle onto the same columns and compiled programs — pass that context's own SRS handles (zk_bases_share).
 * zk_plonk_prove = zk_plonk_create_proof on the descriptor the library built.  With values_on_device the columns are DEVICE pointers (borrowed, must outlive the key). */
typedef struct zk_plonk_pk_host {
    uint32_t struct_size;                     /* sizeof(zk_plonk_pk_host) of the caller (ABI versioning) */
    uint32_t k, cs_degree, blinding_factors;
    uint32_t n_fixed, n_advice, n_instance, n_lookups, n_perm_columns;
    const uint32_t* perm_columns;             /* as zk_plonk_pk_desc */
    const uint32_t* advice_queries; uint32_t n_advice_queries;
    const uint32_t* fixed_queries;  uint32_t n_fixed_queries;
    const void* evaluator_zkq1; size_t evaluator_zkq1_len;                      /* Evaluator::to_zkq1 (shim/halo2_proofs_mi355x/src/evaluation_zkq1.rs), extended_k as derived */
    const void* const* lookup_input_zkq1; const size_t* lookup_input_zkq1_len;   /* n_lookups expression programs (extended_k = k): Horner in theta over input_expressions */
    const void* const* lookup_table_zkq1; const size_t* lookup_table_zkq1_len;   /* ... over table_expressions */
    const uint32_t* lookup_table_key;         /* n_lookups: equal key = structurally equal table expressions (their compressed column is computed once) */
    const void* const* fixed_values;          /* n_fixed columns, n x 32 B Montgomery, HOST (DEVICE when values_on_device) */
    const void* const* sigma_values;          /* n_perm_columns columns: pk.permutation.permutations */
    uint32_t values_on_device;
    const void* transcript_repr;              /* HOST 32 B */
    uint32_t transcript, draw_schedule;       /* as zk_plonk_pk_desc */
    /* one proof over several GPUs (zk_plonk_pk_desc, last block): with shard_world > 1 the key keeps, instead of whole extended cosets, only the n-value cosets
     * this rank's quotient units need, and srs_g / srs_g_lagrange of zk_plonk_pk_build are this rank's table slices; zk_plonk_prove then calls `allgather` */
    uint32_t shard_world, shard_rank;
    zk_allgather_fn allgather; void* allgather_user;
} zk_plonk_pk_host;
int zk_plonk_pk_build(zk_ctx* ctx, const zk_plonk_pk_host* host, uint64_t srs_g, uint64_t srs_g_lagrange, uint64_t* pk);
int zk_plonk_pk_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_pk, uint64_t srs_g, uint64_t srs_g_lagrange, uint64_t* pk);
int zk_plonk_pk_release(zk_ctx* ctx, uint64_t pk);
/* the descriptor behind a key handle (valid until the handle is released): for callers that drive single phases themselves */
int zk_plonk_pk_descriptor(zk_ctx* ctx, uint64_t pk, const zk_plonk_pk_desc** desc);
int zk_plonk_prove(zk_ctx* ctx, uint64_t pk, const void* const* advice, int advice_on_device, const void* const* instances, const uint32_t* instance_lens,
                   zk_rng_fn rng, void* rng_user, void* proof_out, size_t proof_cap, size_t* proof_len);
/* wall milliseconds of the nine phases (instances, advice, lookups, grand products, random poly, h numerator, h commit, evaluations, SHPLONK) of the
 * calling thread's last zk_plonk_create_proof */
int zk_plonk_last_phase_ms(double out[9]);
/* return the per-proof device buffers zk_plonk_create_proof keeps for reuse on this context (zk_ctx_destroy does it too) */
int zk_plonk_trim(zk_ctx* ctx);

/* library / build identification */
const char* zk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI355_H */
