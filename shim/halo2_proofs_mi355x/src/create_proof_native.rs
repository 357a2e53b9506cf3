//! One-call form of the integration: `create_proof` hands the whole proof to `zk_plonk_prove` (include/zkmi355.h; csrc/prover.hip).
//! `mod create_proof_native;` next to `mod mi355x;` and `mod pk_desc;` — hooked into plonk/prover.rs by prover_native.patch, INSIDE create_proof right after
//! witness synthesis (`batch_invert_assigned`), because that is where the advice columns exist and nothing random has been drawn yet:
//!
//!     if let Some(done) = crate::create_proof_native::try_create_proof::<Scheme, E, R, T>(params, pk, instances, &advice_values, &mut rng, transcript) {
//!         return done;          // Ok(()) with the proof written into `transcript`, or the Err the CPU body would have returned
//!     }
//!
//! Guards (any failing -> None -> the original body continues, draw for draw, as if the hook were absent):
//!   * the scheme is KZG over bn256 (TypeId of the params type) and HALO2_MI355X != 0 and a gfx950 context exists;
//!   * ONE circuit instance (the reference passes `&[circuit]`, circuits/src/sgx_dcap_verifier.rs:814-822) in ONE advice phase, no Challenge API (pk_desc::key_for);
//!   * the transcript is `Blake2bWrite<_, G1Affine, Challenge255<_>>` (by type name: the library hashes with Blake2b / Challenge255 itself) and NOTHING has been
//!     absorbed since `init` except what create_proof absorbed (vk, instances) — the library re-absorbs those, then every commitment and evaluation;
//!   * n >= 2^12.
//! The proof comes back as bytes; `replay` feeds them through the caller's transcript (write_point / write_scalar, and a squeeze wherever create_proof squeezes),
//! so the writer holds the same bytes AND the same hash state as after the CPU body — for any `W: Write`, without touching Blake2bWrite's private fields.
//! `R: Send`: zk_plonk_prove calls the draw callback from a helper thread of the library (so that the draws of phase p + 1 overlap the kernels of phase p) while the
//! calling thread blocks — the exclusive borrow travels to that thread and back, which is exactly what `Send` licenses.  OsRng (what the reference passes,
//! sgx_dcap_verifier.rs:819) and every seedable rng are Send; create_proof's own bound becomes `R: RngCore + Send` in prover_native.patch (a ThreadRng caller
//! wraps it or keeps the CPU prover).
//! Witness synthesis (`WitnessCollection`, plonk/prover.rs) stays the CPU code it is.  Uncompiled in the build image (no rustc there).
use std::any::{type_name, TypeId};
use std::ffi::c_void;
use std::os::raw::c_int;

use ff::{Field, PrimeField};
use group::GroupEncoding;
use halo2curves::bn256::{Bn256, Fr, G1Affine};
use halo2curves::CurveAffine;
use rand_core::RngCore;

use crate::mi355x::{gpu, ZkCtx, MIN_LEN};
use crate::plonk::{Error, ProvingKey};
use crate::poly::commitment::{CommitmentScheme, Params};
use crate::poly::kzg::commitment::ParamsKZG;
use crate::poly::{LagrangeCoeff, Polynomial};
use crate::transcript::{EncodedChallenge, TranscriptWrite};

type ZkRngFn = extern "C" fn(user: *mut c_void, n: usize, out_fr: *mut c_void);

extern "C" {
    fn zk_plonk_prove(ctx: *mut ZkCtx, pk: u64, advice: *const *const c_void, advice_on_device: c_int, instances: *const *const c_void,
                      instance_lens: *const u32, rng: ZkRngFn, rng_user: *mut c_void, proof_out: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> c_int;
}

/// `Fr::random(&mut rng)` n times, written as the 4 x u64 Montgomery limbs Fr is in memory (layout asserted by mi355x::gpu()).  The library calls this from
/// ONE helper thread, block by block, in halo2's own order (zk_plonk_pk_desc.draw_schedule = 1: blinding rows, the Blind of every commitment, the random
/// polynomial, the h-piece Blinds) and has made every draw when zk_plonk_prove returns: `rng` is left exactly where the CPU body would leave it.
extern "C" fn draw<R: RngCore + Send>(user: *mut c_void, n: usize, out_fr: *mut c_void) {
    let rng = unsafe { &mut *(user as *mut R) };
    let out = unsafe { std::slice::from_raw_parts_mut(out_fr as *mut Fr, n) };
    for v in out.iter_mut() {
        *v = Fr::random(&mut *rng);
    }
}

/// Proof layout of create_proof + ProverSHPLONK for one circuit: commitments per phase, then the evaluations, then SHPLONK's two points.
struct Layout { advice: usize, lookups: usize, sets: usize, pieces: usize, evals: usize }

/// Feed `proof` through the caller's transcript exactly as the CPU body would have: write the phase's points, squeeze where create_proof squeezes.
/// Generic over the curve (only trait methods are used), so no reinterpretation of the transcript is needed: C is G1Affine by the caller's TypeId guard.
fn replay<C: CurveAffine, E: EncodedChallenge<C>, T: TranscriptWrite<C, E>>(t: &mut T, proof: &[u8], l: &Layout) -> Result<(), Error> {
    let bad = |what: &str| Error::Transcript(std::io::Error::new(std::io::ErrorKind::Other, format!("mi355x: bad {what} in the returned proof")));
    let mut at = 0usize;
    let mut point = |t: &mut T| -> Result<(), Error> {
        let mut repr = <C as GroupEncoding>::Repr::default();
        repr.as_mut().copy_from_slice(&proof[at..at + 32]);
        at += 32;
        let p: C = Option::from(C::from_bytes(&repr)).ok_or_else(|| bad("point"))?;
        t.write_point(p).map_err(Error::from)
    };
    for _ in 0..l.advice { point(t)?; }
    let _theta = t.squeeze_challenge();
    for _ in 0..2 * l.lookups { point(t)?; }
    let _beta = t.squeeze_challenge();
    let _gamma = t.squeeze_challenge();
    for _ in 0..l.sets + l.lookups + 1 { point(t)?; }            // permutation products, lookup products, the vanishing argument's random polynomial
    let _y = t.squeeze_challenge();
    for _ in 0..l.pieces { point(t)?; }
    let _x = t.squeeze_challenge();
    drop(point);
    for _ in 0..l.evals {
        let mut repr = <C::Scalar as PrimeField>::Repr::default();
        repr.as_mut().copy_from_slice(&proof[at..at + 32]);
        at += 32;
        let s: C::Scalar = Option::from(C::Scalar::from_repr(repr)).ok_or_else(|| bad("scalar"))?;
        t.write_scalar(s).map_err(Error::from)?;
    }
    let _y2 = t.squeeze_challenge();
    let _v = t.squeeze_challenge();
    for _ in 0..2 {                                               // SHPLONK: h(X), squeeze u, the linearisation quotient
        let mut repr = <C as GroupEncoding>::Repr::default();
        repr.as_mut().copy_from_slice(&proof[at..at + 32]);
        at += 32;
        let p: C = Option::from(C::from_bytes(&repr)).ok_or_else(|| bad("point"))?;
        t.write_point(p).map_err(Error::from)?;
        if at + 32 == proof.len() { let _u = t.squeeze_challenge(); }
    }
    debug_assert_eq!(at, proof.len());
    Ok(())
}

/// See the module comment.  `advice_values`: the columns of the single advice phase after batch_invert_assigned, blinding rows NOT yet filled.
pub fn try_create_proof<Scheme, E, R, T>(params: &Scheme::ParamsProver, pk: &ProvingKey<Scheme::Curve>, instances: &[&[&[Scheme::Scalar]]],
                                         advice_values: &[Polynomial<Scheme::Scalar, LagrangeCoeff>], rng: &mut R, transcript: &mut T) -> Option<Result<(), Error>>
where
    Scheme: CommitmentScheme + 'static,
    Scheme::ParamsProver: 'static,
    E: EncodedChallenge<Scheme::Curve>,
    R: RngCore + Send, // the library draws through `&mut R` on ITS helper thread while this thread blocks in zk_plonk_prove: moving a `&mut R` across threads needs R: Send
    T: TranscriptWrite<Scheme::Curve, E>,
{
    if TypeId::of::<Scheme::ParamsProver>() != TypeId::of::<ParamsKZG<Bn256>>() || instances.len() != 1 {
        return None;
    }
    let tn = type_name::<T>();
    if !(tn.contains("Blake2bWrite") && tn.contains("Challenge255")) {
        return None;
    }
    // Scheme::Curve == G1Affine and Scheme::Scalar == Fr from here on (the TypeId check): reinterpret the generic references
    let params: &ParamsKZG<Bn256> = unsafe { &*(params as *const _ as *const ParamsKZG<Bn256>) };
    let pk: &ProvingKey<G1Affine> = unsafe { &*(pk as *const _ as *const ProvingKey<G1Affine>) };
    let n = params.n() as usize;
    if n < MIN_LEN {
        return None;
    }
    let g = gpu()?;
    let key = crate::pk_desc::key_for(g, params, pk)?;
    let cs = &pk.vk.cs;
    if advice_values.len() != cs.num_advice_columns || instances[0].len() != cs.num_instance_columns {
        return None;
    }

    let adv: Vec<*const c_void> = advice_values.iter().map(|c| c.as_ptr() as *const c_void).collect();
    let canon: Vec<Vec<[u8; 32]>> = instances[0].iter().map(|c| c.iter().map(|v| { let mut b = [0u8; 32]; b.copy_from_slice(v.to_repr().as_ref()); b }).collect()).collect();
    let inst: Vec<*const c_void> = canon.iter().map(|c| c.as_ptr() as *const c_void).collect();
    let lens: Vec<u32> = canon.iter().map(|c| c.len() as u32).collect();
    let chunk = cs.degree() - 2;
    let p = cs.permutation.get_columns().len();
    let layout = Layout {
        advice: cs.num_advice_columns, lookups: cs.lookups.len(), sets: (p + chunk - 1) / chunk, pieces: cs.degree() - 1,
        evals: cs.advice_queries.len() + cs.fixed_queries.len() + 1 + p + (if p > 0 { 3 * ((p + chunk - 1) / chunk) - 1 } else { 0 }) + 5 * cs.lookups.len(),
    };
    let cap = 32 * (layout.advice + 2 * layout.lookups + layout.sets + layout.lookups + 1 + layout.pieces + layout.evals + 2);
    let mut proof = vec![0u8; cap];
    let mut len = 0usize;
    let rc = unsafe {
        zk_plonk_prove(g.ctx, key, adv.as_ptr(), 0, inst.as_ptr(), lens.as_ptr(), draw::<R>, rng as *mut R as *mut c_void,
                       proof.as_mut_ptr() as *mut c_void, proof.len(), &mut len)
    };
    if rc != 0 {
        g.complain("zk_plonk_prove");
        // ZK_ERR_ARG from the lookup phase = an input outside its table: the CPU body reports exactly that (Error::ConstraintSystemFailure) — but part of
        // `rng` has been consumed, so re-running the CPU body here would not reproduce a seeded proof.  Surface the error instead.
        return Some(Err(Error::ConstraintSystemFailure));
    }
    debug_assert_eq!(len, cap);
    proof.truncate(len);
    // (vk and the instance scalars were absorbed by create_proof before the hook; the library absorbed the same values on its side)
    Some(replay::<Scheme::Curve, E, T>(transcript, &proof, &layout))
}
