"""h(X) from cs_degree - 1 cosets (zk_cosets_to_pieces_dev; ProvingKey.pieces_from_cosets, the default on one GPU whenever cs_degree - 1 is not a power of two).
deg h < (cs_degree - 1) n, so the numerator's values on cs_degree - 1 cosets of the extended domain determine it: the remaining cosets — a quarter of the extended
transforms and quotient rows at cs_degree = 4, the shape of the reference's stack-B circuits (crates/p256-ecdsa) — are never evaluated.  For a witness that satisfies the
circuit the proof is the one the extended route (halo2's own: divide_by_vanishing_poly + extended_to_coeff on the whole domain) gives, byte for byte — checked here against the
extended-route key, the Python twin and the independent CPU prover on SATISFIED circuits of degree 4, 6, 7 and 8.  For an unsatisfied witness h is not a polynomial, the two
routes truncate different things and the (invalid) proofs differ from the first h commitment on: keygen(piece_cosets=False) / tune quot_piece_cosets=0 keeps halo2's bytes there
too, which is what tests/test_random_circuits.py (random, unsatisfied witnesses) runs on."""
import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.fields import R_MOD
from zk_dcap_verifier_amd.plonk import ADVICE, INSTANCE, Advice, Fixed
from zk_dcap_verifier_amd.transcript import Blake2bWrite

import random_circuits as rc
import test_create_proof as tcp


def graded_circuit(k, t):
    """a SATISFIED circuit of degree 3 + t: the product gate q * (a * b - c) times a^t, a running-sum gate with a rotation, copy constraints over all three advice columns"""
    n = 1 << k
    cs = plonk.ConstraintSystem(num_fixed_columns=2, num_advice_columns=3, num_instance_columns=1)
    a, b, c = Advice(0), Advice(1), Advice(2)
    g = Fixed(0) * (a * b - c)
    for _ in range(t):
        g = g * a
    cs.create_gate(g)
    cs.create_gate(Fixed(1) * (Advice(0, 1) - a - 1))
    for col in ((ADVICE, 0), (ADVICE, 1), (ADVICE, 2), (INSTANCE, 0)):
        cs.enable_equality(*col)
    assert cs.degree() == 3 + t
    u = cs.usable_rows(k)
    A = [(i % 8) + 1 for i in range(n)]
    B = [((i // 2) % 5) + 2 for i in range(n)]
    Cc = [x * y % R_MOD for x, y in zip(A, B)]
    Q = [1 if i < u else 0 for i in range(n)]
    Q2 = [1 if (i % 8 != 7 and i + 1 < u) else 0 for i in range(n)]
    asm = plonk.Assembly(cs, k)
    asm.copies = []
    for i in range(0, u - 1, 2):
        asm.copy((ADVICE, 1, i), (ADVICE, 1, i + 1))
    asm.copy((ADVICE, 2, 0), (INSTANCE, 0, 0))
    asm.copy((ADVICE, 0, 0), (ADVICE, 0, 8))
    from zk_dcap_verifier_amd.fields import fr_mont_array
    return cs, [Q, Q2], asm, [fr_mont_array(col) for col in (A, B, Cc)], [[Cc[0]]]


def _both_routes(be, k, circuit, seed, oracle=True):
    cs, fixed, asm, advice, instances = circuit
    if k <= 8:
        plonk.MockProver.run(k, cs, fixed, advice, instances, asm).assert_satisfied()
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    proofs = {}
    for route in (True, False):
        pk = plonk.keygen(params, cs, fixed, asm, piece_cosets=route)
        assert pk.pieces_from_cosets == route and (not pk.fixed_cosets) == route
        proofs["native", route] = plonk.NativeProver(params, pk).create_proof([a.copy() for a in advice], instances, np.random.default_rng(seed))
        tr = Blake2bWrite()
        plonk.create_proof(params, pk, [a.copy() for a in advice], instances, np.random.default_rng(seed), tr)
        proofs["twin", route] = tr.finalize()
        pk.release()
    params.release()
    assert len(set(proofs.values())) == 1, {key: v[:8].hex() for key, v in proofs.items()}
    if oracle:
        assert proofs["native", True] == rc.oracle_proof(k, tcp.TAU, cs, fixed, asm, advice, instances, seed)
    return proofs["native", True]


@pytest.mark.parametrize("t", [1, 3, 4, 5])
def test_pieces_from_cosets_on_satisfied_circuits_emulated(emu, orc, t):
    """degree 4 (3 of 4 cosets), 6, 7, 8 (5, 6, 7 of 8): the native prover and the twin on both kinds of key, and the independent CPU prover, emit one proof"""
    _both_routes(emu, 5, graded_circuit(5, t), seed=40 + t)


def test_p256_shaped_circuit_takes_three_of_four_cosets_emulated(emu, orc):
    """the census of the reference's stack-B circuit (crates/p256-ecdsa, bin/assets/proof.bin: degree 4, three h pieces)"""
    _both_routes(emu, 7, tcp.p256_shaped_circuit(7), seed=18, oracle=False)


def test_power_of_two_piece_counts_keep_the_extended_domain(emu, orc):
    cs, fixed, asm, advice, instances = tcp.toy_circuit(5)                     # degree 5: four pieces on four cosets
    params = z.kzg.ParamsKZG.setup(5, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    assert not pk.pieces_from_cosets and pk.coset_parts is None and pk.fixed_cosets
    pk.release()
    params.release()


@pytest.mark.gpu
@pytest.mark.parametrize("k,t", [(10, 1), (12, 4)])
def test_pieces_from_cosets_on_satisfied_circuits_gpu(gpu, orc, k, t):
    _both_routes(gpu, k, graded_circuit(k, t), seed=k + t, oracle=k <= 10)


@pytest.mark.gpu
def test_p256_shaped_circuit_takes_three_of_four_cosets_gpu(gpu, orc):
    _both_routes(gpu, 14, tcp.p256_shaped_circuit(14), seed=18, oracle=False)
