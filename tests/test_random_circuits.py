"""Differential test on RANDOM constraint systems (tests/random_circuits.py): the native prover (zk_plonk_create_proof) and its Python twin must emit, byte
for byte, the proof the independent CPU prover (oracle/prover.py: Python integers, quotient from its definition) computes for the same circuit, witness and
RNG stream.  The goldens pin three fixed shapes; this sweeps what they do not: degrees 3-9 (extended_k - k = 1, 2, 3), rotations up to +-3, instance columns,
lookups sharing a table, equality over advice / fixed / instance columns, unsatisfied gates (a numerator not divisible by X^n - 1)."""
import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.transcript import Blake2bWrite

import random_circuits as rc
import test_create_proof as tcp


def _check(be, k, seed, twin=True):
    cs, fixed, asm, advice, instances = rc.random_circuit(k, seed)
    want = rc.oracle_proof(k, tcp.TAU, cs, fixed, asm, advice, instances, seed)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    # these witnesses do NOT satisfy their gates: h(X) is not a polynomial, and only the extended route with every identity on every row is halo2's on such input
    # (tests/test_piece_cosets.py; the degree split of the quotient program, include/zkmi355.h)
    be.tune(quot_degree_split=0)
    try:
        return _check_unsplit(be, k, seed, twin, cs, fixed, asm, advice, instances, want, params)
    finally:
        be.tune(quot_degree_split=1)


def _check_unsplit(be, k, seed, twin, cs, fixed, asm, advice, instances, want, params):
    pk = plonk.keygen(params, cs, fixed, asm, piece_cosets=False)
    shape = dict(seed=seed, degree=cs.degree(), e=pk.domain.extended_k - k, advice=cs.num_advice_columns, fixed=cs.num_fixed_columns, instance=cs.num_instance_columns,
                 lookups=len(cs.lookups), perm=len(cs.permutation_columns), gates=len(cs.gates), bf=cs.blinding_factors())
    got = plonk.NativeProver(params, pk).create_proof([a.copy() for a in advice], instances, np.random.default_rng(seed))
    assert got == want, ("native", shape, next(i for i in range(0, len(want), 32) if got[i:i + 32] != want[i:i + 32]) // 32 if len(got) == len(want) else (len(got), len(want)))
    if twin:
        tr = Blake2bWrite()
        plonk.create_proof(params, pk, [a.copy() for a in advice], instances, np.random.default_rng(seed), tr)
        assert tr.finalize() == want, ("python twin", shape)
    pk.release()
    params.release()
    return shape


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_circuits_emulated(emu, orc, seed):
    _check(emu, 5 + seed % 2, seed)


def test_the_sweep_covers_what_it_claims():
    """the generator really reaches every extended_k offset, shared tables, instance columns and fixed-column equality within the seeds the tests use"""
    es, shared, inst, fixed_eq, rots = set(), False, False, False, set()
    for seed in range(1, 13):
        cs, *_ = rc.random_circuit(5 + seed % 2, seed)
        e = 0
        while (1 << e) < cs.degree() - 1:
            e += 1
        es.add(e)
        tabs = [tuple(lk.table_expressions) for lk in cs.lookups]
        shared |= len(set(tabs)) < len(tabs)
        inst |= cs.num_instance_columns > 0
        fixed_eq |= any(t == plonk.FIXED for t, _ in cs.permutation_columns)
        rots |= {r for _, r in cs.advice_queries()}
    assert es >= {1, 2, 3} and shared and inst and fixed_eq and min(rots) <= -2 and max(rots) >= 2, (es, shared, inst, fixed_eq, rots)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_random_circuits_gpu(gpu, orc, seed):
    _check(gpu, 6 + seed % 3, seed)
