"""create_proof -> verify_proof round trip: the reference's own acceptance test, restated for the GPU prover.

`test_sgx_dcap_verifier_pass` (circuits/src/sgx_dcap_verifier.rs:763-847) pins the prover in one way only: keygen ->
create_proof -> verify_proof must ACCEPT (and the p256-ecdsa test does the same, crates/p256-ecdsa/src/base.rs:214-247).
Here the same sequence runs with the product API as the prover (every O(n) step on the device: commitments, lookup
compression/permutation, grand products, NTTs, evaluate_h, evaluations, SHPLONK) and oracle/verifier.py as verify_proof.
Negative controls: a witness that violates a gate / a copy constraint / a lookup, a tampered proof, wrong instances.
"""
import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.plonk import ADVICE, INSTANCE, Advice, Fixed, Instance
from zk_dcap_verifier_amd.fields import R_MOD, fr_mont_array
from zk_dcap_verifier_amd.transcript import Blake2bWrite

TAU = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA % R_MOD   # SURVEY App. C.7 (any value works)
GOLDEN_PROOF = "toy_proof_k6_seed7.bin"      # tools/gen_golden_proof.py: bytes of the INDEPENDENT CPU prover (oracle/prover.py); see tests/golden/README.md
GOLDEN_SGX = "sgx_shaped_k8_seed3.bin"       # the same for the sgx_dcap_verifier-shaped circuit at k = 8, rng seed 3
GOLDEN_REF_EXACT = "reference_exact_k9_seed3.bin"   # ... and for census B (the reference's base64 sub-circuit built exactly) at k = 9


def _golden(name=GOLDEN_PROOF):
    import os
    from conftest import ROOT
    return open(os.path.join(ROOT, "tests", "golden", name), "rb").read()




def toy_circuit(k, with_lookup=True, tamper=None):
    """3 gates (one with a rotation, one querying the instance column), copy constraints over 3 advice + 1 instance column (2 permutation sets at degree 5),
    one lookup whose input is a product expression.  Returns (cs, fixed columns, assembly, advice columns, instances)."""
    n = 1 << k
    cs = plonk.ConstraintSystem(num_fixed_columns=4, num_advice_columns=3, num_instance_columns=1)
    a, b, c = Advice(0), Advice(1), Advice(2)
    q, q2, t = Fixed(0), Fixed(1), Fixed(2)
    cs.create_gate(q * (a * b - c))
    cs.create_gate(q2 * (Advice(0, 1) - a - 1))
    cs.create_gate(Fixed(3) * (c - Instance(0)))                     # an instance QUERY inside a gate (the verifier evaluates it itself)
    if with_lookup:
        cs.lookup([(q * a, t)])
    for col in ((ADVICE, 0), (ADVICE, 1), (ADVICE, 2), (INSTANCE, 0)):
        cs.enable_equality(*col)
    u = cs.usable_rows(k)
    A = [(i % 8) + 1 for i in range(n)]
    B = [((i // 2) % 5) + 2 for i in range(n)]                       # b[2i+1] == b[2i]
    C = [x * y % R_MOD for x, y in zip(A, B)]
    Q = [1 if i < u else 0 for i in range(n)]
    Q2 = [1 if (i % 8 != 7 and i + 1 < u) else 0 for i in range(n)]
    T = [i if i < 16 else 0 for i in range(n)]
    inst = [C[0], C[3]]
    asm = plonk.Assembly(cs, k)
    asm.copies = []                                                   # logged for the independent CPU prover (oracle/prover.py)
    for i in range(0, u - 1, 2):
        asm.copy((ADVICE, 1, i), (ADVICE, 1, i + 1))
    asm.copy((ADVICE, 2, 0), (INSTANCE, 0, 0))
    asm.copy((INSTANCE, 0, 1), (ADVICE, 2, 3))
    asm.copy((ADVICE, 0, 0), (ADVICE, 0, 8))                        # a[0] == a[8] (both 1): a cycle inside one column
    if tamper == "gate":
        C[5] = (C[5] + 1) % R_MOD
    elif tamper == "copy":
        B[7] = B[7] + 1
        C[7] = A[7] * B[7] % R_MOD                                   # gate still holds, b[6] == b[7] does not
    elif tamper == "lookup":
        A[9], A[10] = 200, 201                                      # outside the table; keep both gates satisfied on those rows
        C[9], C[10] = A[9] * B[9] % R_MOD, A[10] * B[10] % R_MOD
        Q2[8] = Q2[9] = Q2[10] = 0
    elif tamper == "instance":
        inst = [C[0], (C[3] + 1) % R_MOD]
    QI = [1] + [0] * (n - 1)                                         # row 0: c[0] == instance[0]
    return cs, [Q, Q2, T, QI], asm, [fr_mont_array(A), fr_mont_array(B), fr_mont_array(C)], [inst]


def test_mock_prover_mirrors_the_first_step_of_the_reference_test():
    """`MockProver::run(k, &circuit, vec![]).unwrap().assert_satisfied()` (sgx_dcap_verifier.rs:790-794): the witness of the toy circuit
    satisfies every gate, lookup and copy constraint; each tampered witness is reported for the right reason."""
    cs, fixed, asm, advice, instances = toy_circuit(5)
    plonk.MockProver.run(5, cs, fixed, advice, instances, asm).assert_satisfied()
    for what, needle in (("gate", "gate 0"), ("copy", "copy constraint"), ("lookup", "lookup 0"), ("instance", "copy constraint")):
        cs, fixed, asm, advice, instances = toy_circuit(5, tamper=what)
        with pytest.raises(plonk.VerifyFailure) as e:
            plonk.MockProver.run(5, cs, fixed, advice, instances, asm).assert_satisfied()
        assert needle in str(e.value), (what, str(e.value))


def prove(be, k, seed=1, **kw):
    """the reference test's sequence (sgx_dcap_verifier.rs:790-823): MockProver, gen_srs, keygen_vk/pk, create_proof"""
    cs, fixed, asm, advice, instances = toy_circuit(k, **kw)
    if k <= 10 and not kw.get("tamper"):
        plonk.MockProver.run(k, cs, fixed, advice, instances, asm).assert_satisfied()
    params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    tr = Blake2bWrite()
    info = plonk.create_proof(params, pk, advice, instances, np.random.default_rng(seed), tr)
    proof = tr.finalize()
    vk = pk.vk
    pk.release()
    params.release()
    return vk, instances, proof, info


def _round_trip(be, k, with_lookup=True):
    import verifier
    vk, instances, proof, info = prove(be, k, with_lookup=with_lookup)
    assert len(proof) == 32 * (info["commitments"] + info["evals"])
    assert verifier.verify_proof(vk, TAU, instances, proof) is True
    # the proof is bound to its public inputs and to every byte of itself
    assert verifier.verify_proof(vk, TAU, [[instances[0][0], (instances[0][1] + 1) % R_MOD]], proof) is False
    bad = bytearray(proof)
    bad[-40] ^= 1                                                     # inside the first SHPLONK commitment / last evaluations
    try:
        ok = verifier.verify_proof(vk, TAU, instances, bytes(bad))
    except ValueError:
        ok = False
    assert ok is False
    ev_off = 32 * (info["commitments"] - 2)                          # first evaluation word (the 2 SHPLONK points come last)
    bad = bytearray(proof)
    bad[ev_off] ^= 1
    assert verifier.verify_proof(vk, TAU, instances, bytes(bad)) is False
    return proof


def _rejects(be, k, what):
    import verifier
    try:
        vk, instances, proof, _ = prove(be, k, tamper=what)
    except z.ZkError:
        assert what == "lookup"                                       # permute_expression_pair refuses an input outside the table
        return
    assert what != "lookup"
    assert verifier.verify_proof(vk, TAU, instances, proof) is False


def test_create_proof_round_trip_emulated(emu, orc):
    _round_trip(emu, 6)


def test_create_proof_without_lookups_emulated(emu, orc):
    _round_trip(emu, 5, with_lookup=False)                            # degree 3: extended_k = k + 1, one permutation column per set


@pytest.mark.parametrize("what", ["gate", "copy", "lookup", "instance"])
def test_create_proof_negative_controls_emulated(emu, orc, what):
    _rejects(emu, 5, what)


def test_proof_bytes_are_a_function_of_the_seed(emu, orc):
    """with the RNG pinned the proof is reproducible byte for byte (seed 7 must give the committed golden bytes:
    test_golden_proof_is_accepted_and_reproduced_on_the_emulator); another seed gives another (valid) proof of the same length."""
    import verifier
    vk, instances, p8, _ = prove(emu, 6, seed=8)
    assert p8 != _golden() and len(p8) == len(_golden())
    assert verifier.verify_proof(vk, TAU, instances, p8) is True


@pytest.mark.gpu
@pytest.mark.parametrize("k", [6, 10, 13])
def test_create_proof_round_trip_gpu(gpu, orc, k):
    _round_trip(gpu, k)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["gate", "copy", "lookup", "instance"])
def test_create_proof_negative_controls_gpu(gpu, orc, what):
    _rejects(gpu, 9, what)


def _sgx_shaped(be, k, by_cosets=False, census="chip_estimate"):
    """The circuit shape bench.py proves at k = 19 (25 advice, 18 fixed, 11 lookups of 4-5 expressions, 16 equality columns,
    24 gates, degree 5; tools/sgx_shaped_circuit.py) at a size the Python verifier handles in a second."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    import verifier
    gpu = be
    cs, fixed, asm, advice = sc.build(z, gpu, k, census=census)
    if k <= 10:
        plonk.MockProver.run(k, cs, fixed, advice, [], asm).assert_satisfied()
    params = z.kzg.ParamsKZG.setup(k, TAU, backend=gpu)
    pk = plonk.keygen(params, cs, fixed, asm)
    tr = Blake2bWrite()
    info = plonk.create_proof(params, pk, advice, [], np.random.default_rng(3), tr)
    proof = tr.finalize()
    if by_cosets:                       # the multi-GPU quotient unit (all four cosets on this rank): same bytes as the whole-domain quotient
        params.quotient_by_cosets = True
        pk2 = plonk.keygen(params, cs, fixed, asm)
        tr2 = Blake2bWrite()
        plonk.create_proof(params, pk2, advice, [], np.random.default_rng(3), tr2)
        assert tr2.finalize() == proof
        pk2.release()
    n_sets = -(-len(cs.permutation_columns) // cs.permutation_chunk_len())
    assert info["commitments"] == 25 + 3 * 11 + n_sets + 1 + 4 + 2 and len(proof) == 32 * (info["commitments"] + info["evals"])
    if k == 9 and census == "reference_exact":
        assert proof == _golden(GOLDEN_REF_EXACT)
    if k == 8 and census == "chip_estimate":                          # same SRS / witness / RNG stream as the independent CPU prover's golden: the bytes must be identical
        assert proof == _golden(GOLDEN_SGX)
    assert verifier.verify_proof(pk.vk, TAU, [], proof) is True
    bad = bytearray(proof)
    bad[32 * (info["commitments"] - 2) + 5] ^= 4                     # first evaluation word
    assert verifier.verify_proof(pk.vk, TAU, [], bytes(bad)) is False
    pk.release()
    params.release()


def test_reference_exact_census_proof_verifies_emulated(emu, orc):
    """census B (tools/sgx_shaped_circuit.py build_reference_exact): the base64 part exactly as the reference configures / assigns it
    (sgx_dcap_verifier.rs:139-238, 260-329; table/mod.rs:24-149) + the chip estimate: MockProver satisfied, proof accepted, tampering rejected"""
    _sgx_shaped(emu, 9, census="reference_exact")


@pytest.mark.gpu
@pytest.mark.parametrize("k", [9, 11, 19])
def test_reference_exact_census_proof_verifies_gpu(gpu, orc, k):
    _sgx_shaped(gpu, k, census="reference_exact")


def test_sgx_shaped_circuit_proof_verifies_emulated(emu, orc):
    _sgx_shaped(emu, 6, by_cosets=True)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [12, 19])
def test_sgx_shaped_circuit_quotient_by_cosets_gpu(gpu, orc, k):
    _sgx_shaped(gpu, k, by_cosets=True)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [8, 12, 17, 19, 21])      # 19: BASELINE configs[1] at full size; 21: the size of configs[4]
def test_sgx_shaped_circuit_proof_verifies_gpu(gpu, orc, k):
    _sgx_shaped(gpu, k)


@pytest.mark.gpu
@pytest.mark.parametrize("c", [17, 20])
def test_proof_bytes_do_not_depend_on_the_msm_window_gpu(gpu, orc, c):
    """SRS tables of 2^22 points and more are registered with 20-bit windows (csrc/msm.hip pick_c): the prover's batched commitments, the run-length twin
    tables and the two-level sort then run on wide windows.  A commitment is a group element, so the proof bytes must not move: the k = 8 golden of the
    independent CPU prover reproduced with the window forced wide, and a k = 13 proof equal to the default window's."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    gpu.tune(msm_c=c)
    try:
        _sgx_shaped(gpu, 8)                                            # asserts the golden bytes at k = 8
        proofs = []
        for cc in (c, 0):
            gpu.tune(msm_c=cc)
            cs, fixed, asm, advice = sc.build(z, gpu, 13)
            params = z.kzg.ParamsKZG.setup(13, TAU, backend=gpu)
            pk = plonk.keygen(params, cs, fixed, asm)
            tr = Blake2bWrite()
            plonk.create_proof(params, pk, advice, [], np.random.default_rng(5), tr)
            proofs.append(tr.finalize())
            pk.release()
            params.release()
        assert proofs[0] == proofs[1]
    finally:
        gpu.tune(msm_c=0)


def test_cpu_prover_reproduces_the_committed_goldens(orc):
    """tests/golden/{toy_proof_k6_seed7, sgx_shaped_k8_seed3}.bin are what oracle/prover.py (independent CPU prover: Python integers, quotient from
    its definition) emits — regenerate them here and compare, so the goldens cannot drift from their generator; verify_proof accepts both."""
    import os, sys
    import verifier
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_golden_proof as gg
    for name, (t_, cs, instances, (keys, proof)) in ((GOLDEN_PROOF, gg.toy()), (GOLDEN_SGX, gg.sgx_shaped()),
                                                     (GOLDEN_REF_EXACT, gg.sgx_shaped(9, 3, "reference_exact"))):
        assert proof == _golden(name), name
        assert verifier.verify_proof(keys, TAU, instances, proof) is True


def test_sgx_shaped_golden_is_reproduced_by_the_emulated_kernels(emu, orc):
    """the sgx-shaped circuit at k = 8 through the product prover on the emulated kernels: byte-identical to the CPU prover's proof"""
    _sgx_shaped(emu, 8)


def test_golden_proof_is_accepted_and_reproduced_on_the_emulator(emu, orc):
    import verifier
    vk, instances, proof, _ = prove(emu, 6, seed=7)
    assert proof == _golden()
    assert verifier.verify_proof(vk, TAU, instances, _golden()) is True


@pytest.mark.gpu
def test_gpu_emits_the_golden_proof_bytes(gpu, orc):
    """Same SRS, same witness, same seeded RNG, same transcript => the GPU kernels must emit byte-for-byte the proof the independent CPU prover
    (oracle/prover.py) emitted (every MSM / NTT / quotient / sort result is a canonical value): the north star's bit-exactness claim, end to end.
    The sgx-shaped golden is compared in test_sgx_shaped_circuit_proof_verifies_gpu[8]."""
    assert prove(gpu, 6, seed=7)[2] == _golden()


def p256_shaped_circuit(k):
    """the census of the reference's stack-B circuit (tools/p256_shaped_circuit.py: bench.py proves it too, as BASELINE configs[0])"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import p256_shaped_circuit as p256
    return p256.build(k)


def _p256_shape(be, k):
    """configs[0] of BASELINE.json (crates/p256-ecdsa, CPU plumbing case) as a parity case: our prover, given a circuit with that census,
    emits a proof with EXACTLY the layout of the reference's golden proof.bin — 13 G1 points, 32 scalars, 2 G1 points = 1504 bytes
     — and verify_proof accepts it with the 15 instances (and rejects it with one of them changed)."""
    import os
    import verifier
    from conftest import ROOT
    cs, fixed, asm, advice, instances = p256_shaped_circuit(k)
    params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    tr = Blake2bWrite()
    info = plonk.create_proof(params, pk, advice, instances, np.random.default_rng(18), tr)
    proof = tr.finalize()
    ref = bytes.fromhex(open(os.path.join(ROOT, "tests", "golden", "proof.bin")).read().strip()[2:])
    assert len(proof) == len(ref) == 1504 and info["commitments"] == 13 + 2 and info["evals"] == 32
    assert verifier.verify_proof(pk.vk, TAU, instances, proof) is True
    wrong = [list(instances[0])]
    wrong[0][14] = (wrong[0][14] + 1) % R_MOD
    assert verifier.verify_proof(pk.vk, TAU, wrong, proof) is False
    # same word classes as proof.bin: words 0-12 and 45-46 are curve points, words 13-44 canonical scalars
    from zk_dcap_verifier_amd.transcript import point_from_bytes
    for w in list(range(13)) + [45, 46]:
        assert point_from_bytes(proof[32 * w:32 * w + 32]) is not None
    for w in range(13, 45):
        assert int.from_bytes(proof[32 * w:32 * w + 32], "little") < R_MOD
    pk.release()
    params.release()


def test_p256_ecdsa_shaped_proof_has_the_layout_of_proof_bin_emulated(emu, orc):
    _p256_shape(emu, 7)


@pytest.mark.gpu
def test_p256_ecdsa_shaped_proof_has_the_layout_of_proof_bin_gpu(gpu, orc):
    _p256_shape(gpu, 12)


def _by_cosets_proof(be, k=6):
    """The quotient taken coset by coset (the multi-GPU sharding unit, here all cosets on one rank) must give the same proof bytes."""
    import zk_dcap_verifier_amd as z
    params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
    params.quotient_by_cosets = True
    cs, fixed, asm, advice, instances = toy_circuit(k)
    pk = plonk.keygen(params, cs, fixed, asm)
    assert pk.coset_parts is not None and not pk.fixed_cosets
    tr = Blake2bWrite()
    plonk.create_proof(params, pk, advice, instances, np.random.default_rng(7), tr)
    pk.release()
    params.release()
    return tr.finalize()


def test_emulated_quotient_by_cosets_gives_golden_proof(emu):
    assert _by_cosets_proof(emu) == _golden()


@pytest.mark.gpu
def test_gpu_quotient_by_cosets_gives_golden_proof(gpu):
    assert _by_cosets_proof(gpu) == _golden()
