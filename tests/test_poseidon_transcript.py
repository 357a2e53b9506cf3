"""Stack B's transcript: snark-verifier's PoseidonTranscript<NativeLoader> as the reference uses it for crates/p256-ecdsa
(base.rs:200-212 gen_proof; bin/src/main.rs:242 `PoseidonTranscript::<NativeLoader, &[u8]>::new::<0>(proof)`).

What IS pinned from outside this repo: the Poseidon permutation itself — the first round constant, MDS[0][0] and the hash of [1, 2] are the published
BN254 t = 3 (R_F = 8, R_P = 57) values every implementation of the reference parameter script reproduces (circomlib's poseidon_constants / its
`poseidon([1, 2])` test vector).  What is NOT: snark-verifier's sponge framing and point absorption ([3P-MEM]); two separate writings are cross-checked.
"""
import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk, poseidon
from zk_dcap_verifier_amd.fields import R_MOD
from zk_dcap_verifier_amd.transcript import PoseidonRead, PoseidonWrite, point_from_bytes, point_to_bytes

import test_create_proof as tcp


def test_poseidon_permutation_known_answers():
    rc, mds = poseidon.spec()
    assert len(rc) == 65 and len(rc[0]) == 3
    assert rc[0][0] == 0x0EE9A592BA9A9518D05986D656F40C2114C4993C11BB29938D21D47304CD8E6E       # C[0] of the BN254 t = 3 parameter set
    assert mds[0][0] == 0x109B7F411BA0E4C9B2B70CAF5C36A7B194BE7C11AD24378BFEDB68592BA8118B      # M[0][0]
    # poseidon([1, 2]) of circomlib / iden3 (state [0, 1, 2], output word 0)
    assert poseidon.permute([0, 1, 2])[0] == 7853200120776062878684798364095072458815029376092732009249414926327459813530


def test_two_writings_of_the_sponge_agree(pyref):
    import poseidon_ref as ref
    assert ref._C == poseidon.spec()
    rng = np.random.default_rng(5)
    for count in (0, 1, 2, 3, 4, 7):
        vals = [int.from_bytes(rng.bytes(31), "little") for _ in range(count)]
        a, b = poseidon.Sponge(), ref.Reader(b"")
        a.update(vals)
        for v in vals:
            b.common_scalar(v)
        assert a.squeeze() == b.squeeze() and a.squeeze() == b.squeeze(), count      # the second squeeze permutes an empty chunk


def test_point_flag_bit_254_round_trip(pyref):
    g2 = pyref.g1_mul(pyref.G1_GEN, 2)
    for pt in (pyref.G1_GEN, g2, pyref.g1_neg(g2), None):
        b = point_to_bytes(pt, 254)
        assert point_from_bytes(b, 254) == pt
        assert (b[31] & 0x80 == 0) or pt is None
    assert point_to_bytes(g2, 254)[31] & 0x40 == (g2[1] & 1) << 6


def _p256_poseidon(be, k):
    """the p256-ecdsa-shaped circuit (census of bin/assets/proof.bin) proved through PoseidonWrite: 1504 bytes with proof.bin's word classes AND its flag
    convention (y parity in bit 254, bit 255 clear), accepted by the oracle's verify_proof reading through the second Poseidon writing."""
    import os
    import poseidon_ref
    import verifier
    from conftest import ROOT
    cs, fixed, asm, advice, instances = tcp.p256_shaped_circuit(k)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    tr = PoseidonWrite()
    info = plonk.create_proof(params, pk, advice, instances, np.random.default_rng(18), tr)
    proof = tr.finalize()
    # the library's C++ create_proof with its own Poseidon (zk_plonk_pk_desc.transcript = 1): same bytes as the Python twin through transcript.PoseidonWrite
    assert plonk.NativeProver(params, pk, transcript="poseidon").create_proof([a.copy() for a in advice], instances, np.random.default_rng(18)) == proof
    ref = bytes.fromhex(open(os.path.join(ROOT, "tests", "golden", "proof.bin")).read().strip()[2:])
    assert len(proof) == len(ref) == 1504 and info["commitments"] == 15 and info["evals"] == 32
    assert verifier.verify_proof(pk.vk, tcp.TAU, instances, proof, reader=poseidon_ref.Reader) is True
    wrong = [list(instances[0])]
    wrong[0][3] = (wrong[0][3] + 1) % R_MOD
    assert verifier.verify_proof(pk.vk, tcp.TAU, wrong, proof, reader=poseidon_ref.Reader) is False
    bad = bytearray(proof)
    bad[32 * 20 + 1] ^= 2
    assert verifier.verify_proof(pk.vk, tcp.TAU, instances, bytes(bad), reader=poseidon_ref.Reader) is False
    # the product's own reader replays the same challenges
    rd = PoseidonRead(proof)
    rd.common_scalar(pk.vk.transcript_repr)
    for v in instances[0]:
        rd.common_scalar(v)
    for _ in range(3):
        rd.read_point()
    wr = PoseidonWrite()
    wr.common_scalar(pk.vk.transcript_repr)
    for v in instances[0]:
        wr.common_scalar(v)
    for w in range(3):
        wr.common_point(point_from_bytes(proof[32 * w:32 * w + 32], 254))
    assert rd.squeeze_challenge() == wr.squeeze_challenge()
    # layout diff against the reference's proof.bin: same word classes, same flag convention
    words = lambda blob: [blob[32 * i:32 * i + 32] for i in range(47)]
    for i, (mine, theirs) in enumerate(zip(words(proof), words(ref))):
        is_point = i < 13 or i >= 45
        assert mine[31] & 0x80 == 0 and theirs[31] & 0x80 == 0                       # bit 255 never set in either
        if is_point:
            assert point_from_bytes(mine, 254) is not None and point_from_bytes(theirs, 254) is not None
        else:
            assert int.from_bytes(mine, "little") < R_MOD and int.from_bytes(theirs, "little") < R_MOD
    assert any(words(proof)[i][31] & 0x40 for i in list(range(13)) + [45, 46])       # some y is odd: the flag is in use ...
    assert all(int.from_bytes(words(proof)[i], "little") < R_MOD for i in range(13, 45))   # ... and scalar words are plain canonical values
    pk.release()
    params.release()


def test_p256_shaped_poseidon_proof_emulated(emu, orc):
    _p256_poseidon(emu, 7)


@pytest.mark.gpu
def test_p256_shaped_poseidon_proof_gpu_at_the_real_k18(gpu, orc):
    """BASELINE configs[0] at its real size (k = 18, crates/p256-ecdsa/src/base.rs:134) on the GPU prover, Poseidon transcript"""
    _p256_poseidon(gpu, 18)
