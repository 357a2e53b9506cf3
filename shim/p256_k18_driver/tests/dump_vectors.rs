//! Stack B (halo2-axiom 0.4.2 + snark-verifier): prove the reference's p256 vector (crates/p256-ecdsa/src/base.rs:293-312) with ECDSAProver at k = 18 — native
//! Poseidon proof (gen_proof, base.rs:200-212) and EVM proof (gen_evm_proof_shplonk, base.rs:193-199) — then VERIFY each through a recording transcript and
//! dump every transcript event in order (ZKV1 kind 6).  That sequence — which scalars and points are absorbed, where the challenges are squeezed, and the
//! challenge values — is the sponge framing (padding, empty-chunk permutation, point absorption as (x mod r, y mod r); Keccak buffer rules) that this repo's
//! PoseidonTranscript / EvmTranscript mirrors restate from memory (zk-dcap-verifier_amd/{poseidon,keccak,transcript}.py, csrc/prover.hip): with the file under
//! tests/golden/rust/, tests/test_rust_vectors.py replays the SAME proof bytes through the mirror and must squeeze the SAME challenges.
//! The proofs need OsRng (snark-verifier-sdk draws it itself), so the bytes differ run to run; the event sequence of each dumped proof is self-contained.
//! Run with HALO2_MI355X=0 to dump REFERENCE vectors.  Uncompiled in the build image (no rustc there).
use std::io::{Read, Write};

use common::halo2_proofs::{
    plonk::verify_proof,
    poly::commitment::ParamsProver,
    poly::kzg::{commitment::KZGCommitmentScheme, multiopen::VerifierSHPLONK, strategy::SingleStrategy},
    transcript::{EncodedChallenge, Transcript, TranscriptRead},
};
use common::halo2curves::bn256::{Bn256, Fr, G1Affine};
use common::halo2curves::ff::PrimeField;
use common::halo2curves::CurveAffine;
use common::snark_verifier::system::halo2::transcript::evm::EvmTranscript;
use common::snark_verifier_sdk::{halo2::PoseidonTranscript, NativeLoader};
use p256_ecdsa::{ECDSAInput, ECDSAProver};

/// ZKV1 kind 6 payload: which u32 (1 Poseidon, 2 EVM) | proof_len u64 | proof | n_events u32 | events
///   event = tag u8 followed by: 'c' common_scalar  + 32 B canonical LE      'p' common_point + 64 B (x, y canonical LE)
///                                'S' read_scalar    + 32 B (the value read)  'P' read_point   + 64 B (the point read)
///                                'Q' squeeze        + 32 B (the challenge as a scalar)
struct Recorder<T> {
    inner: T,
    events: Vec<u8>,
    n: u32,
}
fn scalar_bytes(s: &Fr) -> [u8; 32] {
    let mut b = [0u8; 32];
    b.copy_from_slice(s.to_repr().as_ref());
    b
}
fn point_bytes(p: &G1Affine) -> [u8; 64] {
    let c = p.coordinates().unwrap();
    let mut b = [0u8; 64];
    b[..32].copy_from_slice(c.x().to_repr().as_ref());
    b[32..].copy_from_slice(c.y().to_repr().as_ref());
    b
}
impl<T> Recorder<T> {
    fn push(&mut self, tag: u8, data: &[u8]) {
        self.events.push(tag);
        self.events.extend_from_slice(data);
        self.n += 1;
    }
}
impl<E: EncodedChallenge<G1Affine>, T: Transcript<G1Affine, E>> Transcript<G1Affine, E> for Recorder<T> {
    fn squeeze_challenge(&mut self) -> E {
        let c = self.inner.squeeze_challenge();
        let s = c.get_scalar();
        self.push(b'Q', &scalar_bytes(&s));
        c
    }
    fn common_point(&mut self, point: G1Affine) -> std::io::Result<()> {
        self.push(b'p', &point_bytes(&point));
        self.inner.common_point(point)
    }
    fn common_scalar(&mut self, scalar: Fr) -> std::io::Result<()> {
        self.push(b'c', &scalar_bytes(&scalar));
        self.inner.common_scalar(scalar)
    }
}
impl<E: EncodedChallenge<G1Affine>, T: TranscriptRead<G1Affine, E>> TranscriptRead<G1Affine, E> for Recorder<T> {
    fn read_point(&mut self) -> std::io::Result<G1Affine> {
        let p = self.inner.read_point()?;
        self.push(b'P', &point_bytes(&p));
        Ok(p)
    }
    fn read_scalar(&mut self) -> std::io::Result<Fr> {
        let s = self.inner.read_scalar()?;
        self.push(b'S', &scalar_bytes(&s));
        Ok(s)
    }
}

fn dump(dir: &str, name: &str, which: u32, proof: &[u8], events: &[u8], n: u32) {
    let mut f = std::fs::File::create(format!("{dir}/{name}.zkv")).unwrap();
    f.write_all(b"ZKV1").unwrap();
    f.write_all(&6u32.to_le_bytes()).unwrap();
    f.write_all(&which.to_le_bytes()).unwrap();
    f.write_all(&(proof.len() as u64).to_le_bytes()).unwrap();
    f.write_all(proof).unwrap();
    f.write_all(&n.to_le_bytes()).unwrap();
    f.write_all(events).unwrap();
}

#[test]
fn prove_p256_and_dump_transcript_events() {
    let dir = std::env::var("ZK_VECTOR_DIR").unwrap_or_else(|_| "./zk_vectors".into());
    std::fs::create_dir_all(&dir).unwrap();
    // the reference's own test vector, crates/p256-ecdsa/src/base.rs:293-296
    let msghash = "9c8adb93585642008f6defe84b014d3db86e65ec158f32c1fe8b78974123c264";
    let signature = "89e7242b7a0be99f7c668a8bdbc1fcaf6fa7562dd28538dbab4b059e9d6955c2c434593d3ccb0e7e5825effb14e251e6e5efb738d6042647ed2e2faac9191718";
    let pubkey = "04cd8fdae57e9fcc6638b7e0bdf1cfe6eb4783c29ed13916f10c121c70b7173dd61291422f9ef68a1b6a7e9cccbe7cc2c0738f81a996f7e62e9094c1f80bc0d788";
    let input = ECDSAInput::try_from_hex(msghash, signature, pubkey).unwrap();
    let prover = ECDSAProver::default(); // keygen at k = 18 on first use (base.rs:133-165), params/ next to the crate
    let instances = input.as_instances();
    // (`params` / `pk` are private fields of ECDSAProver: add `pub fn params(&self) -> &ParamsKZG<Bn256>` and `pub fn vk(&self) -> &VerifyingKey<G1Affine>` next to
    //  gen_evm_verifier, or read params/kzg_bn254_18.srs + params/vk.bin the way bin/src/main.rs:225-231 does — two read-only accessors, no behavioural change)
    let params = prover.params();
    let vk = prover.vk();

    // ---- native proof, Poseidon transcript (which = 1) --------------------------------------------------------------------------------------------------
    let proof = prover.create_proof(input, false).unwrap();
    assert_eq!(proof.len(), 1504); // 13 + 2 compressed points, 32 scalars (the layout of bin/assets/proof.bin)
    let mut rec = Recorder { inner: PoseidonTranscript::<NativeLoader, &[u8]>::new::<0>(proof.as_slice()), events: vec![], n: 0 };
    verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, _, _, _>(
        params.verifier_params(), vk, SingleStrategy::new(params), &[&[instances.as_slice()]], &mut rec).unwrap();
    dump(&dir, "p256_k18_poseidon_events", 1, &proof, &rec.events, rec.n);

    // ---- EVM proof, Keccak transcript (which = 2) ---------------------------------------------------------------------------------------------------------
    let proof = prover.create_proof(input, true).unwrap();
    let mut rec = Recorder { inner: EvmTranscript::<G1Affine, NativeLoader, _, _>::new(proof.as_slice()), events: vec![], n: 0 };
    verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, _, _, _>(
        params.verifier_params(), vk, SingleStrategy::new(params), &[&[instances.as_slice()]], &mut rec).unwrap();
    dump(&dir, "p256_k18_evm_events", 2, &proof, &rec.events, rec.n);
    let _ = std::io::stdout().flush();
    let _ = (&mut std::io::empty()).read(&mut []);
}
