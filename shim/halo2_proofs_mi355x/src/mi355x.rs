//! MI355X backend binding for halo2_proofs 0.2.0 (zkwebauthn/halo2 @ c254c75) — `mod mi355x;` in lib.rs.
//! Binds include/zkmi355.h of zk-dcap-verifier_amd.  Every function here is best effort: on ANY failure it returns None / false and the caller
//! runs the original CPU body, so `create_proof` (circuits/src/sgx_dcap_verifier.rs:814-822) keeps its semantics with or without a GPU.
//! Uncompiled in the build image (no rustc there); field / point layouts are asserted at start-up instead of trusted.
use std::any::TypeId;
use std::collections::HashMap;
use std::ffi::{c_void, CStr};
use std::os::raw::{c_char, c_int};
use std::sync::{Mutex, OnceLock};

use halo2curves::bn256::{Fr, G1Affine, G1};
use halo2curves::CurveAffine;

#[repr(C)]
pub struct ZkCtx {
    _p: [u8; 0],
}

/// ZK_ABI_VERSION of the include/zkmi355.h this file was written against
pub const ZK_ABI_VERSION: u32 = 4;

/// field-for-field `zk_quotient_args`
#[repr(C)]
pub struct ZkQuotientArgs {
    pub struct_size: u32, // size_of::<ZkQuotientArgs>()
    pub fixed: *const *const c_void,
    pub advice: *const *const c_void,
    pub instance: *const *const c_void,
    pub l0: *const c_void,
    pub l_last: *const c_void,
    pub l_active_row: *const c_void,
    pub perm_cosets: *const *const c_void,
    pub perm_products: *const *const c_void,
    pub n_sets: u32,
    pub lookup_product: *const *const c_void,
    pub lookup_input: *const *const c_void,
    pub lookup_table: *const *const c_void,
    pub challenges: *const c_void,
    pub beta: *const c_void,
    pub gamma: *const c_void,
    pub theta: *const c_void,
    pub y: *const c_void,
    pub out: *mut c_void,
}

extern "C" {
    fn zk_abi_version() -> u32;
    fn zk_abi_struct_size(struct_name: *const c_char) -> u32;
    fn zk_ctx_create(device: c_int, out: *mut *mut ZkCtx) -> c_int;
    fn zk_last_error(ctx: *mut ZkCtx) -> *const c_char;
    fn zk_bases_register(ctx: *mut ZkCtx, g1_affine: *const c_void, n: usize, handle: *mut u64) -> c_int;
    fn zk_bases_release(ctx: *mut ZkCtx, handle: u64) -> c_int;
    fn zk_bases_enable_runs(ctx: *mut ZkCtx, handle: u64) -> c_int;
    fn zk_msm(ctx: *mut ZkCtx, bases: u64, scalars: *const c_void, n: usize, out_jac: *mut c_void) -> c_int;
    fn zk_msm_batch(ctx: *mut ZkCtx, bases: u64, scalars: *const *const c_void, count: usize, n: usize, out_jac: *mut c_void) -> c_int;
    fn zk_ntt(ctx: *mut ZkCtx, a: *mut c_void, log_n: u32, omega: *const c_void) -> c_int;
    pub fn zk_quotient_program_load(ctx: *mut ZkCtx, blob: *const c_void, len: usize, prog: *mut u64) -> c_int;
    pub fn zk_pk_load(ctx: *mut ZkCtx, prog: u64, fixed: *const *const c_void, sigma: *const *const c_void, l0: *const c_void,
                      l_last: *const c_void, l_active_row: *const c_void, form: c_int, pk: *mut u64) -> c_int;
    pub fn zk_evaluate_h(ctx: *mut ZkCtx, pk: u64, advice: *const *const c_void, instance: *const *const c_void,
                         perm_products: *const *const c_void, lookup_product: *const *const c_void, lookup_input: *const *const c_void,
                         lookup_table: *const *const c_void, challenges: *const c_void, beta: *const c_void, gamma: *const c_void,
                         theta: *const c_void, y: *const c_void, finish: c_int, out: *mut c_void) -> c_int;
    // page-locked staging memory + one-call upload of a proof's advice columns (the PCIe hop of the witness)
    pub fn zk_host_alloc(ctx: *mut ZkCtx, bytes: usize, hptr: *mut *mut c_void) -> c_int;
    pub fn zk_host_free(ctx: *mut ZkCtx, hptr: *mut c_void) -> c_int;
    pub fn zk_dev_upload_batch(ctx: *mut ZkCtx, dptrs: *const *mut c_void, hosts: *const *const c_void, count: usize, bytes_each: usize) -> c_int;
    // + the device-resident forms (zk_dev_*, zk_*_dev, zk_*_batch_dev) for a prover.rs that keeps columns in HBM between phases
}

/// One registered (window-expanded) table per base ARRAY, keyed by its base pointer.  `params.g` / `params.g_lagrange` never move while a
/// `ParamsKZG` lives; `commit()` of a shorter polynomial passes a PREFIX of the same array, which `zk_msm` serves from the same table
/// (include/zkmi355.h: "n may be smaller than the registered table"), so a shorter slice must never re-register (512 MiB per table at k = 19).
struct Table {
    handle: u64,
    len: usize,
}
pub struct Gpu {
    pub ctx: *mut ZkCtx,
    tables: Mutex<HashMap<usize, Table>>,
}
unsafe impl Send for Gpu {}
unsafe impl Sync for Gpu {}
static GPU: OnceLock<Option<Gpu>> = OnceLock::new();

pub const MIN_LEN: usize = 1 << 12; // below this the CPU body beats a launch

pub fn gpu() -> Option<&'static Gpu> {
    GPU.get_or_init(|| {
        if std::env::var("HALO2_MI355X").map(|v| v == "0").unwrap_or(false) {
            return None;
        }
        // layout guards: the FFI passes Rust memory as-is
        assert_eq!(std::mem::size_of::<Fr>(), 32);
        assert_eq!(std::mem::size_of::<G1Affine>(), 64);
        assert_eq!(std::mem::size_of::<G1>(), 96);
        let one: [u64; 4] = unsafe { std::mem::transmute(Fr::from(1u64)) }; // (spelled so that ff 0.12 and ff 0.13 both compile it: this file is shared with stack B)
        assert_eq!(one, [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]); // R mod r
        let gen: [u64; 8] = unsafe { std::mem::transmute(G1Affine::generator()) };
        assert_eq!(&gen[..4], &[0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]); // mont(1) in Fq
        // ABI guards: the library this process loaded is the one these declarations describe (a mismatch is a build error of the integration, not a
        // reason to fall back silently: refuse loudly)
        assert_eq!(unsafe { zk_abi_version() }, ZK_ABI_VERSION, "libzkmi355.so ABI version differs from the binding's");
        assert_eq!(unsafe { zk_abi_struct_size(b"zk_quotient_args\0".as_ptr() as *const c_char) } as usize, std::mem::size_of::<ZkQuotientArgs>());
        assert_eq!(unsafe { zk_abi_struct_size(b"zk_plonk_pk_host\0".as_ptr() as *const c_char) } as usize, std::mem::size_of::<crate::pk_desc::ZkPlonkPkHost>());
        let dev = std::env::var("HALO2_MI355X_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx = std::ptr::null_mut();
        if unsafe { zk_ctx_create(dev, &mut ctx) } != 0 {
            return None; // no usable gfx950: the CPU bodies run
        }
        Some(Gpu { ctx, tables: Mutex::new(HashMap::new()) })
    })
    .as_ref()
}

impl Gpu {
    pub(crate) fn complain(&self, what: &str) {
        let msg = unsafe { CStr::from_ptr(zk_last_error(self.ctx)) };
        eprintln!("mi355x: {what} failed ({msg:?}); falling back to the CPU body");
    }

    /// handle of the table that serves `bases` (registering, or re-registering when a LONGER slice of the same array shows up)
    pub(crate) fn table_for(&self, bases: &[G1Affine]) -> Option<u64> {
        let key = bases.as_ptr() as usize;
        let mut map = self.tables.lock().unwrap();
        if let Some(t) = map.get(&key) {
            if t.len >= bases.len() {
                return Some(t.handle);
            }
            unsafe { zk_bases_release(self.ctx, t.handle) };
        }
        let mut h = 0u64;
        if unsafe { zk_bases_register(self.ctx, bases.as_ptr() as _, bases.len(), &mut h) } != 0 {
            self.complain("zk_bases_register");
            map.remove(&key);
            return None;
        }
        // prefix-sum twin of the table: columns with long runs of equal scalars (sorted lookup inputs, grand products over unused rows) are then committed at
        // the cost of their run boundaries.  Doubles the table's HBM footprint; HALO2_MI355X_RUNS=0 skips it.  Failure is not an error (the direct path remains).
        if std::env::var("HALO2_MI355X_RUNS").map(|v| v != "0").unwrap_or(true) {
            unsafe { zk_bases_enable_runs(self.ctx, h) };
        }
        map.insert(key, Table { handle: h, len: bases.len() });
        Some(h)
    }
}

/// arithmetic::best_multiexp redirect
pub fn try_msm<C: CurveAffine + 'static>(coeffs: &[C::Scalar], bases: &[C]) -> Option<C::Curve> {
    if TypeId::of::<C>() != TypeId::of::<G1Affine>() || coeffs.len() < MIN_LEN {
        return None;
    }
    let g = gpu()?;
    let bases: &[G1Affine] = unsafe { std::slice::from_raw_parts(bases.as_ptr() as *const G1Affine, bases.len()) };
    let h = g.table_for(bases)?;
    let mut out = [0u64; 12];
    if unsafe { zk_msm(g.ctx, h, coeffs.as_ptr() as _, coeffs.len(), out.as_mut_ptr() as _) } != 0 {
        g.complain("zk_msm");
        return None;
    }
    // (x, y, mont(1)) or (0, 0, 0): a valid G1 under Jacobian and homogeneous conventions alike
    Some(unsafe { std::mem::transmute_copy::<[u64; 12], C::Curve>(&out) })
}

/// One commitment PHASE of create_proof (all advice columns; every lookup's permuted pair; all grand products; the h pieces): `polys` are the
/// columns of the phase, all against `bases` (params.g_lagrange or params.g).  Results come back in order, so the transcript sees what it saw.
pub fn try_msm_batch(polys: &[&[Fr]], bases: &[G1Affine]) -> Option<Vec<G1>> {
    let g = gpu()?;
    let n = polys.first()?.len();
    if n < MIN_LEN || polys.iter().any(|p| p.len() != n) {
        return None;
    }
    let h = g.table_for(bases)?;
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr() as *const c_void).collect();
    let mut out = vec![[0u64; 12]; polys.len()];
    if unsafe { zk_msm_batch(g.ctx, h, ptrs.as_ptr(), polys.len(), n, out.as_mut_ptr() as _) } != 0 {
        g.complain("zk_msm_batch");
        return None;
    }
    Some(out.iter().map(|o| unsafe { std::mem::transmute_copy::<[u64; 12], G1>(o) }).collect())
}

/// `params.commit_lagrange(poly, blind)` for every column of a phase — what prover_phases.patch calls (both stacks).  Generic over the scheme as
/// create_proof is: anything but KZG over bn256 declines.  (KZG ignores `blind`, so the batch needs none.)
pub fn commit_lagrange_batch<P: 'static, S: 'static, Cv: 'static>(params: &P, cols: &[&[S]]) -> Option<Vec<Cv>> {
    use crate::poly::kzg::commitment::ParamsKZG;
    use halo2curves::bn256::Bn256;
    if TypeId::of::<P>() != TypeId::of::<ParamsKZG<Bn256>>() || TypeId::of::<S>() != TypeId::of::<Fr>() || TypeId::of::<Cv>() != TypeId::of::<G1>() {
        return None;
    }
    let params: &ParamsKZG<Bn256> = unsafe { &*(params as *const P as *const ParamsKZG<Bn256>) };
    let n = cols.first()?.len();
    let cols: &[&[Fr]] = unsafe { std::slice::from_raw_parts(cols.as_ptr() as *const &[Fr], cols.len()) };
    let out = try_msm_batch(cols, &params.g_lagrange[..n])?;
    let mut out = std::mem::ManuallyDrop::new(out);
    Some(unsafe { Vec::from_raw_parts(out.as_mut_ptr() as *mut Cv, out.len(), out.capacity()) }) // Cv == G1 by the TypeId check
}

/// arithmetic::best_fft redirect (G = Fr only; the EC-FFT of ParamsKZG::setup stays on the CPU in this thin form)
pub fn try_ntt<G: 'static, S>(a: &mut [G], omega: &S, log_n: u32) -> bool {
    if TypeId::of::<G>() != TypeId::of::<Fr>() || a.len() < MIN_LEN || std::mem::size_of::<S>() != 32 {
        return false;
    }
    match gpu() {
        Some(g) => {
            let ok = unsafe { zk_ntt(g.ctx, a.as_mut_ptr() as _, log_n, omega as *const S as _) } == 0;
            if !ok {
                g.complain("zk_ntt");
            }
            ok
        }
        None => false,
    }
}
