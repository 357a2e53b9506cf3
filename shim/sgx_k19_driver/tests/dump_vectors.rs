//! k = 19 proof of the reference circuit with a SEEDED rng + vector dump (ZKV1 files consumed by tests/test_rust_vectors.py).
//! Same call sequence as the reference's test (circuits/src/sgx_dcap_verifier.rs:790-844) — MockProver, gen_srs, keygen_vk, keygen_pk, create_proof,
//! verify_proof — with k from the environment (default 19; needs ECDSA_CONFIG=src/configs/ecdsa_circuit.tmp.config, :163-168) and ChaCha20 instead of
//! OsRng so that proof bytes are reproducible.  Run with HALO2_MI355X=0 to dump REFERENCE vectors, with the backend on to compare.
//! Kind 5 records how many rng calls precede each transcript squeeze (a counting wrapper around the ChaCha20 + a recording wrapper around Blake2bWrite): the
//! one datum that settles zk_plonk_pk_desc.draw_schedule — tests/test_rust_vectors.py compares it with plonk/prover.py draw_plan.
//! Uncompiled in the build image (no rustc there).
use std::io::Write;

use halo2_base::halo2_proofs::{
    arithmetic::{best_fft, best_multiexp},
    dev::MockProver,
    halo2curves::bn256::{Bn256, Fr, G1Affine, G1},
    halo2curves::group::Curve,
    plonk::{create_proof, keygen_pk, keygen_vk, verify_proof},
    poly::{
        commitment::{Params, ParamsProver},
        kzg::{
            commitment::KZGCommitmentScheme,
            multiopen::{ProverSHPLONK, VerifierSHPLONK},
            strategy::SingleStrategy,
        },
    },
    transcript::{Blake2bRead, Blake2bWrite, Challenge255, TranscriptReadBuffer, TranscriptWriterBuffer},
};
use halo2_base::utils::fs::gen_srs;
use halo2_base64::sgx_dcap_verifier::SgxDcapVerifierCircuit; // `pub mod sgx_dcap_verifier` in circuits/src/lib.rs; SgxDcapVerifierCircuit::new at :252-257
use halo2_base::halo2_proofs::transcript::{Transcript, TranscriptWrite};
use rand::{RngCore, SeedableRng};
use std::cell::Cell;
use std::rc::Rc;
use rand_chacha::ChaCha20Rng;

fn raw<T>(v: &[T]) -> &[u8] {
    unsafe { std::slice::from_raw_parts(v.as_ptr() as *const u8, std::mem::size_of_val(v)) }
}

/// ZKV1: b"ZKV1" | kind u32 | kind-specific little-endian payload (tests/test_rust_vectors.py)
fn dump(dir: &str, name: &str, kind: u32, parts: &[&[u8]]) {
    let mut f = std::fs::File::create(format!("{dir}/{name}.zkv")).unwrap();
    f.write_all(b"ZKV1").unwrap();
    f.write_all(&kind.to_le_bytes()).unwrap();
    for p in parts {
        f.write_all(p).unwrap();
    }
}

/// The seeded rng behind a counter: every `Fr::random(&mut rng)` of create_proof — blinding rows, the Blind of each commitment, the random polynomial —
/// advances `calls` (in units of next_u64; `per_fr` calibrates one Fr::random).  zk_plonk_pk_desc.draw_schedule = 1 claims this exact sequence
/// (include/zkmi355.h zk_rng_fn); kind 5 below is what settles it.
struct CountingRng {
    inner: ChaCha20Rng,
    calls: Rc<Cell<u64>>,
}
impl RngCore for CountingRng {
    fn next_u32(&mut self) -> u32 {
        self.calls.set(self.calls.get() + 1);
        self.inner.next_u32()
    }
    fn next_u64(&mut self) -> u64 {
        self.calls.set(self.calls.get() + 1);
        self.inner.next_u64()
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        self.calls.set(self.calls.get() + ((dest.len() + 7) / 8) as u64);
        self.inner.fill_bytes(dest)
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand::Error> {
        self.fill_bytes(dest);
        Ok(())
    }
}

/// Blake2bWrite behind a recorder: at every squeeze_challenge it notes how far the rng has advanced, and which transcript byte offset the proof has reached.
struct RecordingTranscript {
    inner: Blake2bWrite<Vec<u8>, G1Affine, Challenge255<G1Affine>>,
    calls: Rc<Cell<u64>>,
    points: u64,
    scalars: u64,
    at_squeeze: Vec<[u64; 3]>, // (rng calls so far, points written so far, scalars written so far)
}
impl Transcript<G1Affine, Challenge255<G1Affine>> for RecordingTranscript {
    fn squeeze_challenge(&mut self) -> Challenge255<G1Affine> {
        self.at_squeeze.push([self.calls.get(), self.points, self.scalars]);
        self.inner.squeeze_challenge()
    }
    fn common_point(&mut self, point: G1Affine) -> std::io::Result<()> {
        self.inner.common_point(point)
    }
    fn common_scalar(&mut self, scalar: Fr) -> std::io::Result<()> {
        self.inner.common_scalar(scalar)
    }
}
impl TranscriptWrite<G1Affine, Challenge255<G1Affine>> for RecordingTranscript {
    fn write_point(&mut self, point: G1Affine) -> std::io::Result<()> {
        self.points += 1;
        self.inner.write_point(point)
    }
    fn write_scalar(&mut self, scalar: Fr) -> std::io::Result<()> {
        self.scalars += 1;
        self.inner.write_scalar(scalar)
    }
}

fn rand_fr(rng: &mut ChaCha20Rng, n: usize) -> Vec<Fr> {
    use halo2_base::halo2_proofs::halo2curves::group::ff::Field;
    (0..n).map(|_| Fr::random(&mut *rng)).collect()
}

#[test]
fn prove_k19_and_dump_vectors() {
    let k: u32 = std::env::var("SGX_K").ok().and_then(|s| s.parse().ok()).unwrap_or(19);
    let dir = std::env::var("ZK_VECTOR_DIR").unwrap_or_else(|_| "./zk_vectors".into());
    std::fs::create_dir_all(&dir).unwrap();
    let cert = std::fs::read_to_string(std::env::var("SGX_LEAF_CERT_B64").expect("SGX_LEAF_CERT_B64 = file with the 1696 base64 characters of :769")).unwrap();
    let characters: Vec<u8> = cert.trim().bytes().collect();
    assert_eq!(characters.len(), 1696);
    let circuit = SgxDcapVerifierCircuit::<Fr>::new(characters);

    MockProver::run(k, &circuit, vec![]).unwrap().assert_satisfied();
    let params = gen_srs(k);
    let vk = keygen_vk(&params, &circuit).unwrap();
    // vk.cs: the census bench.py's synthetic circuits estimate — A, F, I, L, equality columns, gates, degree, blinding factors
    let cs = vk.cs();
    let census = format!(
        "{{\"k\": {k}, \"num_advice_columns\": {}, \"num_fixed_columns\": {}, \"num_instance_columns\": {}, \"lookups\": {}, \"permutation_columns\": {}, \"degree\": {}, \"blinding_factors\": {}, \"advice_queries\": {}, \"fixed_queries\": {}}}\n",
        cs.num_advice_columns(), cs.num_fixed_columns(), cs.num_instance_columns(), cs.lookups().len(), cs.permutation().get_columns().len(),
        cs.degree(), cs.blinding_factors(), cs.advice_queries().len(), cs.fixed_queries().len());
    std::fs::write(format!("{dir}/vk_cs.json"), census).unwrap();
    let pk = keygen_pk(&params, vk, &circuit).unwrap();

    // ---- kind 1: MSM  (n u64 | n x 32 scalars | n x 64 bases | 96 result {x, y, z}) against params.g_lagrange, seeds as BASELINE.md §3 ----------------
    let mut rng = ChaCha20Rng::seed_from_u64(20241008);
    for log_n in [10u32, 16, k] {
        let n = 1usize << log_n;
        let scalars = rand_fr(&mut rng, n);
        let bases: Vec<G1Affine> = params.get_g()[..n].to_vec();
        let r: G1 = best_multiexp(&scalars, &bases);
        dump(&dir, &format!("msm_2p{log_n}"), 1, &[&(n as u64).to_le_bytes(), raw(&scalars), raw(&bases), raw(&[r])]);
    }
    // ---- kind 2: NTT  (log_n u32 | omega 32 | input | output) with the domain's own omega ------------------------------------------------------------
    let domain = pk.get_vk().get_domain();
    for log_n in [10u32, k] {
        let n = 1usize << log_n;
        let mut omega = domain.get_omega();
        for _ in log_n..k {
            omega = omega * omega; // omega_k^(2^(k - log_n)) has order 2^log_n
        }
        let input = rand_fr(&mut rng, n);
        let mut a = input.clone();
        best_fft(&mut a, omega, log_n);
        dump(&dir, &format!("ntt_2p{log_n}"), 2, &[&log_n.to_le_bytes(), raw(&[omega]), raw(&input), raw(&a)]);
    }
    // ---- kind 4: the proof itself under a seeded rng (proof_len u64 | proof | first draws of the rng stream for cross-checking the sampler) ----------
    let mut prng = ChaCha20Rng::seed_from_u64(7);
    let mut transcript = Blake2bWrite::<_, _, Challenge255<_>>::init(vec![]);
    create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, Blake2bWrite<Vec<u8>, G1Affine, Challenge255<G1Affine>>, _>(
        &params, &pk, &[circuit], &[&[]], &mut prng, &mut transcript).unwrap();
    let proof = transcript.finalize();
    let mut probe = ChaCha20Rng::seed_from_u64(7);
    let first_draws: Vec<u64> = (0..8).map(|_| probe.next_u64()).collect();
    dump(&dir, &format!("proof_sgx_k{k}_seed7"), 4, &[&(proof.len() as u64).to_le_bytes(), &proof, raw(&first_draws)]);

    // ---- kind 5: the draw schedule — how far the rng stands at every transcript squeeze of the SAME proof (must reproduce the kind-4 bytes) -----------------
    //   per_fr u32 (next_u64 calls of one Fr::random) | n_squeezes u32 | n_squeezes x (calls u64, points written u64, scalars written u64) | total calls u64
    let calls = Rc::new(Cell::new(0u64));
    let mut crng = CountingRng { inner: ChaCha20Rng::seed_from_u64(7), calls: calls.clone() };
    let per_fr = {
        use halo2_base::halo2_proofs::halo2curves::group::ff::Field;
        let c = Rc::new(Cell::new(0u64));
        let _ = Fr::random(CountingRng { inner: ChaCha20Rng::seed_from_u64(1), calls: c.clone() });
        c.get() as u32
    };
    let mut rec = RecordingTranscript { inner: Blake2bWrite::<_, _, Challenge255<_>>::init(vec![]), calls: calls.clone(), points: 0, scalars: 0, at_squeeze: vec![] };
    let circuit2 = SgxDcapVerifierCircuit::<Fr>::new(cert.trim().bytes().collect());
    create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, RecordingTranscript, _>(
        &params, &pk, &[circuit2], &[&[]], &mut crng, &mut rec).unwrap();
    assert_eq!(rec.inner.finalize(), proof, "the recording transcript changed the proof");
    let flat: Vec<u64> = rec.at_squeeze.iter().flat_map(|t| t.iter().copied()).collect();
    dump(&dir, &format!("draws_sgx_k{k}_seed7"), 5, &[&per_fr.to_le_bytes(), &(rec.at_squeeze.len() as u32).to_le_bytes(), raw(&flat), &calls.get().to_le_bytes()]);

    let strategy = SingleStrategy::new(&params);
    let mut rt = Blake2bRead::<_, _, Challenge255<_>>::init(&proof[..]);
    assert!(verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, Blake2bRead<&[u8], G1Affine, Challenge255<G1Affine>>, SingleStrategy<'_, Bn256>>(
        params.verifier_params(), pk.get_vk(), strategy, &[&[]], &mut rt).is_ok());
    // kind 3 (evaluate_h: ZKQ1 blob + coefficient polys + challenges + h) needs a hook inside create_proof; it is dumped by the patched crate when
    // ZK_VECTOR_DIR is set (evaluation_zkq1.rs: to_zkq1 + the arguments of evaluate_h), not from here.
}
