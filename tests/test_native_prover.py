"""zk_plonk_create_proof — the native (C++) per-proof path over the C ABI — against the goldens of the independent CPU prover (oracle/prover.py) and against
its Python twin plonk.create_proof: same SRS, witness and seeded draws => the same bytes; the proofs are accepted by verify_proof; a witness outside a lookup
table is refused with halo2's ConstraintSystemFailure."""
import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.transcript import Blake2bWrite

import test_create_proof as tcp


def _toy(be, k, seed):
    import verifier
    cs, fixed, asm, advice, instances = tcp.toy_circuit(k)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    native = plonk.NativeProver(params, pk)
    proof = native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(seed))
    tr = Blake2bWrite()
    plonk.create_proof(params, pk, [a.copy() for a in advice], instances, np.random.default_rng(seed), tr)
    assert proof == tr.finalize()
    assert verifier.verify_proof(pk.vk, tcp.TAU, instances, proof) is True
    # device-resident witness: same bytes
    dev = [be.to_device(a) for a in advice]
    assert native.create_proof(dev, instances, np.random.default_rng(seed)) == proof
    pk.release()
    params.release()
    return proof


def _sgx(be, k, census, golden=None):
    import os, sys
    import verifier
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    cs, fixed, asm, advice = sc.build(z, be, k, census=census)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    proof = plonk.NativeProver(params, pk).create_proof(advice, [], np.random.default_rng(3))
    if golden:
        assert proof == tcp._golden(golden)
    assert verifier.verify_proof(pk.vk, tcp.TAU, [], proof) is True
    pk.release()
    params.release()


def test_native_prover_emits_the_golden_toy_proof_emulated(emu, orc):
    assert _toy(emu, 6, 7) == tcp._golden(tcp.GOLDEN_PROOF)


def test_native_prover_emits_the_sgx_shaped_goldens_emulated(emu, orc):
    _sgx(emu, 8, "chip_estimate", tcp.GOLDEN_SGX)
    _sgx(emu, 9, "reference_exact", tcp.GOLDEN_REF_EXACT)


def _split_on_off(be, k, census):
    """the degree split of the quotient (low-degree identities on two cosets, include/zkmi355.h) changes no byte of a valid proof: same circuit, witness and draws with the
    split compiled in (the default: asserted through zk_quotient_program_split) and without it"""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    proofs = []
    for split in (1, 0):
        be.tune(quot_degree_split=split)
        try:
            cs, fixed, asm, advice = sc.build(z, be, k, census=census)
            params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
            pk = plonk.keygen(params, cs, fixed, asm)
            sp = be.quotient_program_split(pk.evaluator.handle)
            assert (sp["low_cosets"] == 2 and sp["instructions_high"] > 0 and sp["instructions_low"] > 0) if split else sp["low_cosets"] == 0, sp
            proofs.append(plonk.NativeProver(params, pk).create_proof(advice, [], np.random.default_rng(3)))
            pk.release()
            params.release()
        finally:
            be.tune(quot_degree_split=1)
    assert proofs[0] == proofs[1]
    return proofs[0]


def test_degree_split_changes_no_proof_byte_emulated(emu, orc):
    assert _split_on_off(emu, 8, "chip_estimate") == tcp._golden(tcp.GOLDEN_SGX)


@pytest.mark.gpu
def test_degree_split_changes_no_proof_byte_gpu(gpu, orc):
    assert _split_on_off(gpu, 8, "chip_estimate") == tcp._golden(tcp.GOLDEN_SGX)
    assert _split_on_off(gpu, 9, "reference_exact") == tcp._golden(tcp.GOLDEN_REF_EXACT)
    _split_on_off(gpu, 13, "chip_estimate")


def _side_lane_on_off(be, k, census, device_columns):
    """the side lane of zk_plonk_create_proof (prover.hip: a phase's lagrange_to_coeff + coeff_to_extended on the context's helper context while the phase's commitments run)
    changes no byte: always (2), never (0), and the default (1: on for a proof that is alone); with the side lane the caller's device columns keep their values (without it
    they end as coefficient forms)"""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    cs, fixed, asm, advice = sc.build(z, be, k, census=census)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    prover = plonk.NativeProver(params, pk)
    proofs = []
    try:
        for mode in (2, 0, 1, 2):
            be.tune(prover_side_lane=mode)
            cols = [be.to_device(a) for a in advice] if device_columns else advice
            proofs.append(prover.create_proof(cols, [], np.random.default_rng(3)))
            if device_columns:
                if mode == 2:
                    assert all((c.download(a.shape)[:64] == a[:64]).all() for c, a in zip(cols[:3], advice[:3]))     # (the last rows are the blinding rows: the prover writes those)
                for c in cols:
                    c.free()
    finally:
        be.tune(prover_side_lane=1)
    pk.release()
    params.release()
    assert all(p == proofs[0] for p in proofs)
    return proofs[0]


def test_side_lane_changes_no_proof_byte_emulated(emu, orc):
    assert _side_lane_on_off(emu, 8, "chip_estimate", False) == tcp._golden(tcp.GOLDEN_SGX)
    _side_lane_on_off(emu, 7, "chip_estimate", True)


@pytest.mark.gpu
def test_side_lane_changes_no_proof_byte_gpu(gpu, orc):
    assert _side_lane_on_off(gpu, 8, "chip_estimate", True) == tcp._golden(tcp.GOLDEN_SGX)
    _side_lane_on_off(gpu, 14, "chip_estimate", True)


def test_draw_schedule_leaves_the_callers_rng_where_halo2_would(emu, orc):
    """A proof consumes exactly draw_plan's draws — blinding rows, random polynomial AND the Blind(Fr::random) of every commitment (advice,
    permuted pairs, grand products, random polynomial, h pieces) — and all of them are made before zk_plonk_create_proof returns, so a caller that proves twice
    with one seeded rng (the shape of a Rust caller's `&mut rng`) gets the same two proofs from the native prover and from the Python twin; the running
    totals per challenge are what shim/sgx_k19_driver's counting RNG dumps (tests/test_rust_vectors.py kind 5)."""
    from zk_dcap_verifier_amd.plonk.prover import draw_plan
    cs, fixed, asm, advice, instances = tcp.toy_circuit(6)
    params = z.kzg.ParamsKZG.setup(6, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    native = plonk.NativeProver(params, pk)
    rng_n, rng_t = np.random.default_rng(11), np.random.default_rng(11)
    first_n = native.create_proof([a.copy() for a in advice], instances, rng_n)
    second_n = native.create_proof([a.copy() for a in advice], instances, rng_n)
    out = []
    for _ in range(2):
        tr = Blake2bWrite()
        plonk.create_proof(params, pk, [a.copy() for a in advice], instances, rng_t, tr)
        out.append(tr.finalize())
    assert [first_n, second_n] == out and first_n != second_n
    assert rng_n.integers(0, 1 << 62) == rng_t.integers(0, 1 << 62)          # both streams stand at the same place afterwards
    bf, n = cs.blinding_factors(), 1 << 6
    chunk = cs.permutation_chunk_len()
    n_sets = -(-len(cs.permutation_columns) // chunk)
    plan = draw_plan(cs.num_advice_columns, len(cs.lookups), n_sets, cs.degree() - 1, n, bf)
    total = sum(c for _, _, c, _ in plan)
    A, L = cs.num_advice_columns, len(cs.lookups)
    assert total == A * (bf + 1) + A + L * (2 * (bf + 1) + 2) + n_sets * (bf + 1) + L * (bf + 1) + n + 1 + (cs.degree() - 1)
    pk.release()
    params.release()


def test_native_prover_refuses_a_lookup_input_outside_the_table(emu, orc):
    cs, fixed, asm, advice, instances = tcp.toy_circuit(5, tamper="lookup")
    params = z.kzg.ParamsKZG.setup(5, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    with pytest.raises(z.ZkError):
        plonk.NativeProver(params, pk).create_proof(advice, instances, np.random.default_rng(1))
    pk.release()
    params.release()


def test_native_prover_reports_a_short_output_buffer_and_its_phase_clock(emu, orc):
    """proof_cap too small -> ZK_ERR_LIMIT with the needed length reported, nothing written past the buffer; the phase clock covers the nine phases"""
    cs, fixed, asm, advice, instances = tcp.toy_circuit(5)
    params = z.kzg.ParamsKZG.setup(5, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    native = plonk.NativeProver(params, pk)
    proof = native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2))
    assert len(native.phase_ms) == 9 and all(v >= 0 for v in native.phase_ms.values()) and sum(native.phase_ms.values()) > 0
    native.proof_cap = len(proof) - 32
    with pytest.raises(z.ZkError):
        native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2))
    native.proof_cap = len(proof)                                    # exactly enough
    assert native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2)) == proof
    pk.release()
    params.release()


def test_native_prover_refuses_a_bad_descriptor_or_instance(emu, orc):
    """indices that would read outside the descriptor's arrays, an unknown transcript, a non-canonical instance value: ZK_ERR_ARG, no proof, no crash"""
    from zk_dcap_verifier_amd.fields import R_MOD
    cs, fixed, asm, advice, instances = tcp.toy_circuit(5)
    params = z.kzg.ParamsKZG.setup(5, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    native = plonk.NativeProver(params, pk)
    good = native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2))
    native.desc.transcript = 7
    with pytest.raises(z.ZkError):
        native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2))
    native.desc.transcript = 0
    n_adv = native.desc.n_advice
    native.desc.n_advice = 1                                          # the permutation and the queries name advice columns 1 and 2
    with pytest.raises(z.ZkError):
        native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2))
    native.desc.n_advice = n_adv
    with pytest.raises(z.ZkError):
        native.create_proof([a.copy() for a in advice], [[instances[0][0], R_MOD + 5]], np.random.default_rng(2))
    assert native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(2)) == good
    pk.release()
    params.release()


def test_native_provers_on_two_contexts_in_threads(built, orc):
    """one context + host thread per proof in flight (the bench's throughput mode): the second context borrows the first one's SRS tables; same bytes"""
    import threading
    from conftest import EMU_SO
    a, b = z.Backend(0, lib_path=EMU_SO), z.Backend(0, lib_path=EMU_SO)
    for be in (a, b):
        be.tune(msm_sort_threads=64, msm_sort_wgs=3, msm_block=32, msm_target_threads=64, msm_min_chunk=2, vec_block=32)
    cs, fixed, asm, advice, instances = tcp.toy_circuit(6)
    pa = z.kzg.ParamsKZG.setup(6, tcp.TAU, backend=a)
    pb = z.kzg.ParamsKZG.shared_with(pa, b)
    out = {}

    def run(name, params):
        pk = plonk.keygen(params, cs, fixed, asm)
        out[name] = plonk.NativeProver(params, pk).create_proof([c.copy() for c in advice], instances, np.random.default_rng(7))
        pk.release()
    ts = [threading.Thread(target=run, args=(n, p)) for n, p in (("a", pa), ("b", pb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out["a"] == out["b"] == tcp._golden(tcp.GOLDEN_PROOF)
    # ONE proving key for both contexts (ProvingKey.shared_with: columns and compiled programs shared, zk_quotient_program_share) proving at the same time
    pka = plonk.keygen(pa, cs, fixed, asm)
    pkb = plonk.ProvingKey.shared_with(pka, b)
    out.clear()

    def run_shared(name, params, pk):
        native = plonk.NativeProver(params, pk)
        out[name] = [native.create_proof([c.copy() for c in advice], instances, np.random.default_rng(7)) for _ in range(2)]
    ts = [threading.Thread(target=run_shared, args=(n, p, k_)) for n, p, k_ in (("a", pa, pka), ("b", pb, pkb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out["a"] == out["b"] == [tcp._golden(tcp.GOLDEN_PROOF)] * 2
    with pytest.raises(RuntimeError):                                 # the owner's columns are still borrowed: it refuses to free them under the borrower
        pka.release()
    pkb.release()
    assert plonk.NativeProver(pa, pka).create_proof([c.copy() for c in advice], instances, np.random.default_rng(7)) == tcp._golden(tcp.GOLDEN_PROOF)    # the owner is intact
    pka.release()
    pb.release()
    pa.release()
    b.close()
    a.close()


@pytest.mark.gpu
def test_native_prover_goldens_gpu(gpu, orc):
    assert _toy(gpu, 6, 7) == tcp._golden(tcp.GOLDEN_PROOF)
    _sgx(gpu, 8, "chip_estimate", tcp.GOLDEN_SGX)
    _sgx(gpu, 9, "reference_exact", tcp.GOLDEN_REF_EXACT)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [12, 19])
def test_native_prover_verifies_gpu(gpu, orc, k):
    _toy(gpu, 10, 5)
    _sgx(gpu, k, "chip_estimate")


def test_proof_bytes_replayed_through_a_fresh_transcript_reach_the_same_state(emu, orc):
    """shim/halo2_proofs_mi355x/src/create_proof_native.rs `replay`: the Rust hook gets the proof back as BYTES and feeds them through the caller's transcript —
    the phase's points, a squeeze wherever create_proof squeezes, the evaluations, SHPLONK's two points — so that the writer ends with the same bytes and the same
    hash state as after the CPU body.  Done here with the mirror's Blake2bWrite and the `Layout` arithmetic of that file; the reference state is the twin's
    transcript after plonk.create_proof."""
    from zk_dcap_verifier_amd.transcript import point_from_bytes
    cs, fixed, asm, advice, instances = tcp.toy_circuit(6)
    params = z.kzg.ParamsKZG.setup(6, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    proof = plonk.NativeProver(params, pk).create_proof([a.copy() for a in advice], instances, np.random.default_rng(5))
    twin = Blake2bWrite()
    plonk.create_proof(params, pk, [a.copy() for a in advice], instances, np.random.default_rng(5), twin)
    assert twin.finalize() == proof
    # what create_proof absorbed before the hook
    t = Blake2bWrite()
    pk.vk.hash_into(t)
    for col in instances:
        for v in col:
            t.common_scalar(v)
    chunk = cs.degree() - 2
    p_ = len(cs.permutation_columns)
    sets = -(-p_ // chunk) if p_ else 0
    L = len(cs.lookups)
    lay = dict(advice=cs.num_advice_columns, lookups=L, sets=sets, pieces=cs.degree() - 1,
               evals=len(cs.advice_queries()) + len(cs.fixed_queries()) + 1 + p_ + (3 * sets - 1 if p_ else 0) + 5 * L)
    assert len(proof) == 32 * (lay["advice"] + 2 * L + sets + L + 1 + lay["pieces"] + lay["evals"] + 2)      # `cap` of try_create_proof
    at = [0]

    def point():
        t.write_point(point_from_bytes(proof[at[0]:at[0] + 32]))
        at[0] += 32
    for _ in range(lay["advice"]):
        point()
    t.squeeze_challenge()
    for _ in range(2 * L):
        point()
    t.squeeze_challenge(), t.squeeze_challenge()
    for _ in range(sets + L + 1):
        point()
    t.squeeze_challenge()
    for _ in range(lay["pieces"]):
        point()
    t.squeeze_challenge()
    for _ in range(lay["evals"]):
        t.write_scalar(int.from_bytes(proof[at[0]:at[0] + 32], "little"))
        at[0] += 32
    t.squeeze_challenge(), t.squeeze_challenge()
    point()
    t.squeeze_challenge()
    point()
    assert at[0] == len(proof) and t.finalize() == proof
    assert t.state.digest() == twin.state.digest()
    pk.release()
    params.release()


def _full_chain_x4(be, k, table_bits):
    """BASELINE configs[4] / SURVEY 8d cfg 5: the full-DCAP-chain op-mix = cfg 2's census with the advice and lookup counts x 4 (synthetic: the reference's README lists that circuit
    as a roadmap item) — 100 advice columns, 44 lookups, 58 equality columns (20 permutation sets), 96 gates — through the native prover, accepted by verify_proof"""
    import os, sys
    import verifier
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    cs, fixed, asm, advice = sc.build(z, be, k, census="full_chain_x4", table_bits=table_bits)
    assert (cs.num_advice_columns, len(cs.lookups), len(cs.permutation_columns), cs.degree()) == (100, 44, 58, 5)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    proof = plonk.NativeProver(params, pk).create_proof(advice, [], np.random.default_rng(3))
    n_sets = -(-58 // 3)
    assert len(proof) >= 32 * (100 + 3 * 44 + n_sets + 1 + 4 + 2)
    assert verifier.verify_proof(pk.vk, tcp.TAU, [], proof) is True
    pk.release()
    params.release()


def test_full_chain_x4_census_emulated(emu, orc):
    _full_chain_x4(emu, 9, 6)


@pytest.mark.gpu
def test_full_chain_x4_census_gpu(gpu, orc):
    _full_chain_x4(gpu, 13, 12)


# (configs[4] at its own size, k = 21 — single context and sharded over two ranks — is tests/test_sharded_cfg5.py)
