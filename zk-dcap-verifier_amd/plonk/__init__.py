"""halo2_proofs::plonk — the prover side the reference calls (keygen_vk / keygen_pk / create_proof at
circuits/src/sgx_dcap_verifier.rs:803,807,814-822), as a host-side mirror over the MI355X product API.

    ConstraintSystem, Assembly, Expression (Advice / Fixed / Instance / Constant …)      circuit description
    keygen(params, cs, fixed_columns, assembly) -> ProvingKey                            keygen_vk + keygen_pk
    create_proof(params, pk, advice_columns, instances, rng, transcript)                 plonk::create_proof + ProverSHPLONK (Python twin; multi-GPU driver)
    NativeProver(params, pk).create_proof(advice_columns, instances, rng) -> bytes       the same per-proof path in C++ (zk_plonk_create_proof)
    MockProver.run(k, cs, fixed, advice, instances, assembly).assert_satisfied()         dev::MockProver (host-only constraint check)

verify_proof is not part of the product (verifier side, out of scope); the tests carry their own
pure-Python verifier as the acceptance check, outside this package.
"""
from .circuit import ADVICE, FIXED, INSTANCE, Assembly, ConstraintSystem, LookupArgument  # noqa: F401
from .dev import MockProver, VerifyFailure  # noqa: F401
from .expression import Advice, Constant, Expression, Fixed, Instance  # noqa: F401
from .keygen import ProvingKey, VerifyingKey, compile_program, keygen  # noqa: F401
from .prover import create_proof  # noqa: F401
from .native import NativeProver, create_proof_native  # noqa: F401
from .shplonk import ProverSHPLONK  # noqa: F401
