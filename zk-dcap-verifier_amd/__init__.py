"""zk-dcap-verifier_amd — MI355X (gfx950) backend for the halo2 KZG prover hot path of
CliqueOfficial/zk-dcap-verifier: BN254 G1 MSM, Fr NTT and the quotient evaluation behind
``create_proof`` (reference call sites: circuits/src/sgx_dcap_verifier.rs:799-822,
crates/p256-ecdsa/src/base.rs:134,145,193-212).

The product is the C-ABI library ``libzkmi355.so`` (include/zkmi355.h).  This package is the
host-side mirror of the halo2_proofs interfaces that library sits under — same names, argument
meaning and error behaviour — so that tests read like the reference's own:

    arithmetic.best_multiexp / best_fft          (halo2_proofs::arithmetic)
    domain.EvaluationDomain                      (halo2_proofs::poly::domain)
    kzg.ParamsKZG.commit / commit_lagrange       (halo2_proofs::poly::kzg::commitment)
    evaluation.Evaluator.evaluate_h              (halo2_proofs::plonk::evaluation)
    permutation.permutation_commit / lookup_commit_product   (plonk::{permutation,lookup}::prover grand products)
    plonk.keygen / plonk.create_proof / plonk.ProverSHPLONK  (plonk::{keygen_vk, keygen_pk, create_proof}, multiopen::ProverSHPLONK)
    transcript.Blake2bWrite / Blake2bRead                    (halo2_proofs::transcript — host, unchanged)

There is no CPU fallback: importing works anywhere, but creating a Backend without the HIP
library or without a GPU raises.
"""
from ._lib import Backend, ZkError, LIB_PATH, default_backend  # noqa: F401
from . import arithmetic, domain, kzg, evaluation, permutation, fields, transcript, plonk  # noqa: F401

__all__ = ["Backend", "ZkError", "LIB_PATH", "default_backend", "arithmetic", "domain", "kzg", "evaluation", "permutation", "plonk", "transcript", "fields"]
