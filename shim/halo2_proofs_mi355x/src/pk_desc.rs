//! `ProvingKey` -> the library's proving-key object (include/zkmi355.h: zk_plonk_pk_host, zk_plonk_pk_build).  `mod pk_desc;` next to `mod mi355x;`.
//!
//! What crosses the FFI is only what keygen_pk LEFT ON THE HOST: pointers to pk.fixed_values and pk.permutation.permutations (Lagrange columns, n x 32 B each),
//! the constraint system's shape and query lists, and the Evaluator / lookup expressions as ZKQ1 blobs (evaluation_zkq1.rs).  The library uploads the columns and
//! derives coefficient forms, extended cosets and l0 / l_last / l_active_row on the GPU (the device half of keygen_pk), so nothing of pk.fixed_cosets /
//! pk.permutation.cosets / pk.l0.. (the 64 MiB-per-column host copies at k = 19) is read at all.  One key object per ProvingKey, cached by address like the SRS
//! tables of mi355x.rs; tests/csrc/capi_prove.c is the same call sequence in C and runs in the build image (tools/dump_pk_blob.py writes its input).
//!
//! [3P-MEM] field / method names are PSE halo2 v2023_01_20 (zkwebauthn @ c254c75) as remembered: ProvingKey { vk, l0, l_last, l_active_row, fixed_values,
//! fixed_polys, fixed_cosets, permutation: permutation::ProvingKey { permutations, polys, cosets }, ev }, VerifyingKey { domain, fixed_commitments, permutation,
//! cs, cs_degree, transcript_repr }, ParamsKZG { k, n, g, g_lagrange, g2, s_g2 }.  Uncompiled in the build image (no rustc there).
use std::collections::HashMap;
use std::ffi::c_void;
use std::os::raw::c_int;
use std::sync::Mutex;

use ff::PrimeField;
use halo2curves::bn256::{Bn256, Fr, G1Affine};

use crate::mi355x::{gpu, Gpu, ZkCtx};
use crate::plonk::{Any, ProvingKey};
use crate::poly::kzg::commitment::ParamsKZG;

/// the caller's collective of a proof over several GPUs (`zk_allgather_fn`); a single-GPU binding passes None
pub type ZkAllgatherFn = extern "C" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes: usize) -> c_int;

/// field-for-field `zk_plonk_pk_host` (include/zkmi355.h) — tests/test_shim_abi.py diffs the two declarations name by name and type by type, and
/// `mi355x::gpu()` asserts `size_of::<ZkPlonkPkHost>() == zk_abi_struct_size("zk_plonk_pk_host")` before the first call.
#[repr(C)]
pub struct ZkPlonkPkHost {
    pub struct_size: u32,    // size_of::<ZkPlonkPkHost>(): the library refuses any other size than its own (ABI versioning)
    pub k: u32, pub cs_degree: u32, pub blinding_factors: u32,
    pub n_fixed: u32, pub n_advice: u32, pub n_instance: u32, pub n_lookups: u32, pub n_perm_columns: u32,
    pub perm_columns: *const u32,
    pub advice_queries: *const u32, pub n_advice_queries: u32,
    pub fixed_queries: *const u32, pub n_fixed_queries: u32,
    pub evaluator_zkq1: *const c_void, pub evaluator_zkq1_len: usize,
    pub lookup_input_zkq1: *const *const c_void, pub lookup_input_zkq1_len: *const usize,
    pub lookup_table_zkq1: *const *const c_void, pub lookup_table_zkq1_len: *const usize,
    pub lookup_table_key: *const u32,
    pub fixed_values: *const *const c_void,
    pub sigma_values: *const *const c_void,
    pub values_on_device: u32,
    pub transcript_repr: *const c_void,
    pub transcript: u32,     // 0 Blake2bWrite / Challenge255 (stack A)
    pub draw_schedule: u32,  // 1 = halo2's order of Fr::random draws (the only value a binding passes)
    // one proof over several GPUs (0 / 0 / None / null: the whole proof on this process's GPU)
    pub shard_world: u32, pub shard_rank: u32,
    pub allgather: Option<ZkAllgatherFn>,
    pub allgather_user: *mut c_void,
}

extern "C" {
    fn zk_plonk_pk_build(ctx: *mut ZkCtx, host: *const ZkPlonkPkHost, srs_g: u64, srs_g_lagrange: u64, pk: *mut u64) -> c_int;
    fn zk_plonk_pk_release(ctx: *mut ZkCtx, pk: u64) -> c_int;
}

/// (address of the ProvingKey, address of params.g) -> key handle.  A ProvingKey never moves while a prover uses it (create_proof takes `&pk`).
static KEYS: Mutex<Option<HashMap<(usize, usize), u64>>> = Mutex::new(None);

fn column_type_code(any: &Any) -> u32 {
    match any {
        Any::Advice(_) => 0, // `Any::Advice` without payload in forks that predate multi-phase advice
        Any::Fixed => 1,
        Any::Instance => 2,
    }
}

/// The key object for (params, pk) on the process's GPU context, built on first use.  None = this ProvingKey cannot take the one-call path (user challenges /
/// multi-phase advice, or a library error — reported through Gpu::complain): the caller runs the CPU body.
pub fn key_for(g: &'static Gpu, params: &ParamsKZG<Bn256>, pk: &ProvingKey<G1Affine>) -> Option<u64> {
    let id = (pk as *const _ as usize, params.g.as_ptr() as usize);
    let mut guard = KEYS.lock().unwrap();
    let map = guard.get_or_insert_with(HashMap::new);
    if let Some(h) = map.get(&id) {
        return Some(*h);
    }
    let cs = &pk.vk.cs;
    // the native prover implements halo2's single-phase flow: one advice phase, no Challenge API (the reference's circuits use neither)
    if cs.num_challenges != 0 || cs.advice_column_phase.iter().any(|p| p.0 != 0) {
        return None;
    }
    let k = params.k;
    let n = 1usize << k;
    let srs_g = g.table_for(&params.g[..n])?;               // pub(crate) in mi355x.rs; also enables the run-length twin (zk_bases_enable_runs)
    let srs_gl = g.table_for(&params.g_lagrange[..n])?;

    let u32s = |pairs: Vec<(u32, u32)>| -> Vec<u32> { pairs.into_iter().flat_map(|(a, b)| [a, b]).collect() };
    let perm_columns = u32s(cs.permutation.get_columns().iter().map(|c| (column_type_code(c.column_type()), c.index() as u32)).collect());
    // (column, rotation) in cs.advice_queries / cs.fixed_queries order: the order create_proof evaluates and the verifier reads them
    let advice_queries = u32s(cs.advice_queries.iter().map(|(c, r)| (c.index() as u32, r.0 as u32)).collect());
    let fixed_queries = u32s(cs.fixed_queries.iter().map(|(c, r)| (c.index() as u32, r.0 as u32)).collect());

    let extended_k = pk.vk.domain.extended_k();
    let evaluator = pk.ev.to_zkq1(cs, k, extended_k);
    let in_blobs: Vec<Vec<u8>> = cs.lookups.iter().map(|l| crate::plonk::evaluation::lookup_expression_zkq1::<G1Affine>(cs, k, &l.input_expressions)).collect();
    let tab_blobs: Vec<Vec<u8>> = cs.lookups.iter().map(|l| crate::plonk::evaluation::lookup_expression_zkq1::<G1Affine>(cs, k, &l.table_expressions)).collect();
    // lookups whose table expressions are structurally equal share one compressed table column per proof: the Debug rendering is the structural key
    let mut seen: HashMap<String, u32> = HashMap::new();
    let table_key: Vec<u32> = cs.lookups.iter().map(|l| { let next = seen.len() as u32; *seen.entry(format!("{:?}", l.table_expressions)).or_insert(next) }).collect();

    let ptrs = |v: &Vec<Vec<u8>>| -> Vec<*const c_void> { v.iter().map(|b| b.as_ptr() as *const c_void).collect() };
    let lens = |v: &Vec<Vec<u8>>| -> Vec<usize> { v.iter().map(|b| b.len()).collect() };
    let (in_p, in_l, tab_p, tab_l) = (ptrs(&in_blobs), lens(&in_blobs), ptrs(&tab_blobs), lens(&tab_blobs));
    let fixed: Vec<*const c_void> = pk.fixed_values.iter().map(|p| p.as_ptr() as *const c_void).collect();
    let sigma: Vec<*const c_void> = pk.permutation.permutations.iter().map(|p| p.as_ptr() as *const c_void).collect();
    debug_assert!(pk.fixed_values.iter().chain(pk.permutation.permutations.iter()).all(|p| p.len() == n));
    let repr = pk.vk.transcript_repr.to_repr(); // canonical little endian, what hash_into feeds common_scalar

    let host = ZkPlonkPkHost {
        struct_size: std::mem::size_of::<ZkPlonkPkHost>() as u32,
        k, cs_degree: cs.degree() as u32, blinding_factors: cs.blinding_factors() as u32,
        n_fixed: cs.num_fixed_columns as u32, n_advice: cs.num_advice_columns as u32, n_instance: cs.num_instance_columns as u32,
        n_lookups: cs.lookups.len() as u32, n_perm_columns: (perm_columns.len() / 2) as u32,
        perm_columns: perm_columns.as_ptr(),
        advice_queries: advice_queries.as_ptr(), n_advice_queries: (advice_queries.len() / 2) as u32,
        fixed_queries: fixed_queries.as_ptr(), n_fixed_queries: (fixed_queries.len() / 2) as u32,
        evaluator_zkq1: evaluator.as_ptr() as *const c_void, evaluator_zkq1_len: evaluator.len(),
        lookup_input_zkq1: in_p.as_ptr(), lookup_input_zkq1_len: in_l.as_ptr(),
        lookup_table_zkq1: tab_p.as_ptr(), lookup_table_zkq1_len: tab_l.as_ptr(),
        lookup_table_key: table_key.as_ptr(),
        fixed_values: fixed.as_ptr(), sigma_values: sigma.as_ptr(), values_on_device: 0,
        transcript_repr: repr.as_ref().as_ptr() as *const c_void,
        transcript: 0,
        draw_schedule: 1,
        shard_world: 0, shard_rank: 0, allgather: None, allgather_user: std::ptr::null_mut(),
    };
    let mut handle = 0u64;
    if unsafe { zk_plonk_pk_build(g.ctx, &host, srs_g, srs_gl, &mut handle) } != 0 {
        g.complain("zk_plonk_pk_build");
        return None;
    }
    map.insert(id, handle);
    Some(handle)
}

/// Drop the device copy of a key (e.g. from `impl Drop for ProvingKey`, or when a long-lived service rotates circuits).
pub fn forget(pk: &ProvingKey<G1Affine>) {
    let (Some(g), Ok(mut guard)) = (gpu(), KEYS.lock()) else { return };
    if let Some(map) = guard.as_mut() {
        let addr = pk as *const _ as usize;
        map.retain(|(p, _), h| {
            if *p == addr {
                unsafe { zk_plonk_pk_release(g.ctx, *h) };
                false
            } else {
                true
            }
        });
    }
}
