#!/usr/bin/env python3
"""Regenerate the golden proofs under tests/golden/ with the INDEPENDENT CPU prover (oracle/prover.py: Python integers, quotient from its
definition, MSMs on the C oracle — no product code, no kernel emulator):
    toy_proof_k6_seed7.bin     the toy circuit of tests/test_create_proof.py, k = 6, np.random.default_rng(7)
    sgx_shaped_k8_seed3.bin    the sgx_dcap_verifier-shaped circuit (tools/sgx_shaped_circuit.py: 25 advice, 11 lookups, 16 equality columns), k = 8, rng 3
    reference_exact_k9_seed3.bin  census B of the same tool (the reference's base64 sub-circuit built exactly + chip estimate), k = 9, rng 3
The Fr::random draws follow halo2_proofs v2023_01_20 (incl. the Blind of every commitment).  All with the SRS trapdoor TAU of the tests.  The GPU prover (and the emulated kernels) must emit exactly these bytes, and the pure-Python
verifier must accept them.  They are NOT outputs of the reference (no Rust toolchain here; the reference holds no stack-A proof): they replace the
round-1 goldens, which the emulator build of the product's own kernels had produced."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402


def oracle_proof(k, tau, cs, fixed, copies, advice, instances, seed):
    """circuit description + witness (Montgomery arrays or int lists) -> (keys, proof bytes) from the CPU prover"""
    import oracle as orc
    import prover as op
    ints = lambda col: orc.fr_to_ints(col) if isinstance(col, np.ndarray) else [int(v) for v in col]
    params = op.Params(k, tau)
    keys = op.keygen(params, cs, [ints(c) for c in fixed], copies)
    return keys, op.create_proof(params, keys, [ints(c) for c in advice], instances, np.random.default_rng(seed))


def toy(k=6, seed=7):
    import test_create_proof as t
    cs, fixed, asm, advice, instances = t.toy_circuit(k)
    return t, cs, instances, oracle_proof(k, t.TAU, cs, fixed, asm.copies, advice, instances, seed)


def sgx_shaped(k=8, seed=3, census="chip_estimate"):
    import test_create_proof as t
    import sgx_shaped_circuit as sc
    import zk_dcap_verifier_amd as z

    class FieldCalc:                      # build() uses its backend argument as a field calculator for d = a + b*c only: do that on the oracle here
        def __init__(self):
            import oracle as orc
            self.orc = orc

        class _Buf:
            def __init__(self, a):
                self.a = np.array(a, dtype=np.uint64)

            def download(self, shape):
                return self.a.reshape(shape)

            def free(self):
                pass

        def to_device(self, a):
            return FieldCalc._Buf(a)

        def fr_mul_dev(self, a, b, out, n):
            out.a = self.orc.fr_mul(a.a[:n], b.a[:n])

        def fr_add_dev(self, a, b, out, n):
            out.a = self.orc.fr_add(a.a[:n], b.a[:n])

        def fr_scale_dev(self, a, scalar, out, n):
            out.a = self.orc.fr_mul(a.a[:n], np.repeat(np.asarray(scalar, dtype=np.uint64).reshape(1, 4), n, axis=0))
    cs, fixed, asm, advice = sc.build(z, FieldCalc(), k, census=census)
    return t, cs, [], oracle_proof(k, t.TAU, cs, fixed, asm.copies, advice, [], seed)


def main():
    import verifier
    cases = [("toy_proof_k6_seed7.bin", toy()), ("sgx_shaped_k8_seed3.bin", sgx_shaped()), ("reference_exact_k9_seed3.bin", sgx_shaped(9, 3, "reference_exact"))]
    for name, (t, cs, instances, (keys, proof)) in cases:
        keys.cs, keys.k = cs, keys.k
        assert verifier.verify_proof(keys, t.TAU, instances, proof) is True, name
        out = os.path.join(ROOT, "tests", "golden", name)
        open(out, "wb").write(proof)
        print("wrote", out, len(proof), "bytes (CPU prover, accepted by verify_proof)")


if __name__ == "__main__":
    main()
