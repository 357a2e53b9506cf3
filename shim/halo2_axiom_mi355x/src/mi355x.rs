//! MI355X backend binding for halo2-axiom 0.4.2 (crates.io, Cargo.lock:1181-1200) — stack B of the reference: crates/p256-ecdsa (gen_pk base.rs:145,
//! gen_proof / gen_evm_proof_shplonk base.rs:193-212) and bin/ (main.rs:233-253) reach best_multiexp / best_fft through this crate.  `mod mi355x;` in lib.rs.
//!
//! Everything that does not name a halo2 type is shared with stack A: this file `include!`s ../../halo2_proofs_mi355x/src/mi355x.rs after bringing the
//! axiom-flavoured names into scope.  [3P-MEM] halo2-axiom depends on the curve crate as `halo2curves = { package = "halo2curves-axiom", version = "0.5" }`, so
//! inside this crate the path is `halo2curves::bn256` exactly as in stack A; what differs is ff 0.13 (`Fr::ONE` instead of `Fr::one()`) and best_fft's
//! signature (arithmetic.rs.patch).  The start-up asserts of `gpu()` are what guards the layout claim (size_of Fr / G1Affine / G1, Montgomery R, the generator):
//! halo2curves-axiom 0.5.2 (Cargo.lock:1346-1370) keeps bn256::Fr as `[u64; 4]` in Montgomery form with R = 2^256 — if a future version changes that, the
//! assert fires at first use instead of producing wrong proofs.  Uncompiled in the build image (no rustc there).
#![allow(dead_code)]

// the shared binding: extern "C" block, Gpu / table cache (pub(crate) fn table_for), try_msm, try_msm_batch, try_ntt, MIN_LEN — written against names both
// stacks have (`halo2curves::bn256::{Fr, G1Affine, G1}`, `Fr::from(1u64)` rather than ff 0.12's `Fr::one()` / ff 0.13's `Fr::ONE`).
include!("../../halo2_proofs_mi355x/src/mi355x.rs");

/// best_fft of halo2-axiom carries precomputed FFTData and an `inverse` flag; the butterflies are the same transform with omega or omega^-1 ([3P-MEM]
/// src/fft/mod.rs: `fft(a, omega, log_n, data, inverse)` -> parallel / recursive / baseline variants, all computing out[j] = sum_i a[i] w^(ij) with
/// w = omega, resp. omega_inv when `inverse` — the 1/n scaling stays in EvaluationDomain::ifft).  The caller passes the root it actually transforms with.
pub fn try_ntt_axiom<G: 'static, S>(a: &mut [G], omega: &S, omega_inv: &S, log_n: u32, inverse: bool) -> bool {
    try_ntt::<G, S>(a, if inverse { omega_inv } else { omega }, log_n)
}
