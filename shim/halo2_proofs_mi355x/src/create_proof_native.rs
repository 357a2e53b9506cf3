//! One-call form of the integration: `create_proof` hands the whole proof to `zk_plonk_create_proof` (include/zkmi355.h; csrc/prover.hip).
//! `mod create_proof_native;` next to `mod mi355x;` — used from plonk/prover.rs:
//!
//!     pub fn create_proof<...>(params, pk, circuits, instances, mut rng, transcript) -> Result<(), Error> {
//!         if let Some(done) = create_proof_native::try_create_proof(params, pk, circuits, instances, &mut rng, transcript) { return done; }
//!         /* original body */
//!     }
//!
//! Applies when the scheme is KZG over bn256 with one circuit and a byte-sink transcript (the reference: Blake2bWrite<Vec<u8>, G1Affine, Challenge255>,
//! circuits/src/sgx_dcap_verifier.rs:813-823).  Witness synthesis (`WitnessCollection`, plonk/prover.rs) stays the CPU code it is: the advice columns it
//! fills are what crosses the boundary.  Uncompiled in the build image (no rustc there).
use std::ffi::c_void;
use std::os::raw::c_int;

use ff::Field;
use halo2curves::bn256::Fr;
use rand_core::RngCore;

use crate::mi355x::{gpu, ZkCtx};

/// field-for-field `zk_plonk_pk_desc`
#[repr(C)]
pub struct ZkPlonkPkDesc {
    pub k: u32, pub extended_k: u32, pub cs_degree: u32, pub blinding_factors: u32,
    pub n_fixed: u32, pub n_advice: u32, pub n_instance: u32, pub n_lookups: u32, pub n_perm_columns: u32,
    pub perm_columns: *const u32,
    pub advice_queries: *const u32, pub n_advice_queries: u32,
    pub fixed_queries: *const u32, pub n_fixed_queries: u32,
    pub srs_g: u64, pub srs_g_lagrange: u64,
    pub program: u64,
    pub lookup_input_programs: *const u64,
    pub lookup_table_programs: *const u64,
    pub lookup_table_key: *const u32,
    pub fixed_values: *const *const c_void, pub fixed_polys: *const *const c_void, pub fixed_cosets: *const *const c_void,
    pub sigma_values: *const *const c_void, pub sigma_polys: *const *const c_void, pub sigma_cosets: *const *const c_void,
    pub l0: *const c_void, pub l_last: *const c_void, pub l_active_row: *const c_void,
    pub transcript_repr: *const c_void,
    pub transcript: u32, // 0 Blake2b/Challenge255 (stack A), 1 snark-verifier Poseidon, 2 snark-verifier EVM (Keccak)
}

type ZkRngFn = extern "C" fn(user: *mut c_void, n: usize, out_fr: *mut c_void);

extern "C" {
    fn zk_plonk_create_proof(ctx: *mut ZkCtx, pk: *const ZkPlonkPkDesc, advice: *const *const c_void, advice_on_device: c_int,
                             instances: *const *const c_void, instance_lens: *const u32, rng: ZkRngFn, rng_user: *mut c_void,
                             proof_out: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> c_int;
}

/// `Fr::random(&mut rng)` n times, written as the 4 x u64 Montgomery limbs Fr is in memory (layout asserted by mi355x::gpu()).
/// The library calls this from ONE helper thread, block by block, in the order plonk/prover.rs draws: the stream of `rng` is consumed exactly as
/// the CPU body would consume it.
extern "C" fn draw<R: RngCore>(user: *mut c_void, n: usize, out_fr: *mut c_void) {
    let rng = unsafe { &mut *(user as *mut R) };
    let out = unsafe { std::slice::from_raw_parts_mut(out_fr as *mut Fr, n) };
    for v in out.iter_mut() {
        *v = Fr::random(&mut *rng);
    }
}

/// `desc` is built once per ProvingKey (cached by address next to the base-table cache of mi355x.rs): its columns are uploaded with zk_dev_upload, the
/// Evaluator goes through evaluation_zkq1.rs -> zk_quotient_program_load, each lookup's input / table expressions through the same serialiser
/// with extended_k = k.  `advice[i]` are the host columns of the witness (n x 32 B, blinding rows NOT yet filled: the library asks `draw` for them).
pub fn create_proof_bytes<R: RngCore>(desc: &ZkPlonkPkDesc, advice: &[&[Fr]], instances: &[&[Fr]], rng: &mut R) -> Option<Vec<u8>> {
    let g = gpu()?;
    let adv: Vec<*const c_void> = advice.iter().map(|c| c.as_ptr() as *const c_void).collect();
    let canon: Vec<Vec<[u8; 32]>> = instances.iter().map(|c| c.iter().map(|v| v.to_bytes()).collect()).collect();   // canonical little endian
    let inst: Vec<*const c_void> = canon.iter().map(|c| c.as_ptr() as *const c_void).collect();
    let lens: Vec<u32> = canon.iter().map(|c| c.len() as u32).collect();
    let mut proof = vec![0u8; 1 << 16];
    let mut len = 0usize;
    let rc = unsafe {
        zk_plonk_create_proof(g.ctx, desc, adv.as_ptr(), 0, inst.as_ptr(), lens.as_ptr(), draw::<R>, rng as *mut R as *mut c_void,
                              proof.as_mut_ptr() as *mut c_void, proof.len(), &mut len)
    };
    if rc != 0 {
        g.complain("zk_plonk_create_proof");               // e.g. ZK_ERR_ARG: a lookup input outside its table -> the CPU body reports ConstraintSystemFailure
        return None;
    }
    proof.truncate(len);
    Some(proof)        // the caller appends these bytes to the transcript's writer: nothing else was written since Blake2bWrite::init
}
