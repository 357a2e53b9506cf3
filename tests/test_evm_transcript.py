"""Stack B's EVM transcript (snark_verifier_sdk::evm::gen_evm_proof_shplonk, crates/p256-ecdsa/src/base.rs:193-199): Keccak-256 is pinned by public known
answers; snark-verifier's EvmTranscript framing is [3P-MEM], cross-checked between two separate writings (product / oracle)."""
import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.fields import R_MOD
from zk_dcap_verifier_amd.keccak import keccak256
from zk_dcap_verifier_amd.transcript import EvmRead, EvmWrite

import test_create_proof as tcp


def test_keccak256_known_answers():
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # the Ethereum function selector of transfer(address,uint256)
    assert keccak256(b"transfer(address,uint256)")[:4].hex() == "a9059cbb"


def test_two_writings_of_keccak_and_of_the_transcript_agree(pyref):
    import evm_ref
    rng = np.random.default_rng(3)
    for n in (0, 1, 31, 32, 135, 136, 137, 271, 272, 1000):
        d = rng.bytes(n)
        assert evm_ref.keccak256(d) == keccak256(d), n
    w, r = EvmWrite(), evm_ref.Reader(b"")
    g2 = pyref.g1_mul(pyref.G1_GEN, 2)
    for tr in (w, r):
        tr.common_scalar(12345)
        tr.common_point(g2)
    assert w.squeeze_challenge() == r.squeeze()
    assert w.squeeze_challenge() == r.squeeze()                   # buffer = the previous hash (32 bytes): the 0x01 byte is appended
    w.common_scalar(7); r.common_scalar(7)
    assert w.squeeze_challenge() == r.squeeze()


def _p256_evm(be, k):
    """the p256-ecdsa-shaped circuit proved through EvmWrite: 15 points x 64 B + 32 scalars x 32 B = 1984 bytes, big endian, accepted by verify_proof
    reading through the oracle's own Keccak transcript; bound to its instances and to every byte"""
    import evm_ref
    import verifier
    cs, fixed, asm, advice, instances = tcp.p256_shaped_circuit(k)
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    tr = EvmWrite()
    info = plonk.create_proof(params, pk, advice, instances, np.random.default_rng(18), tr)
    proof = tr.finalize()
    # the library's C++ create_proof with its own Keccak transcript (zk_plonk_pk_desc.transcript = 2): same bytes as the Python twin through transcript.EvmWrite
    assert plonk.NativeProver(params, pk, transcript="evm").create_proof([a.copy() for a in advice], instances, np.random.default_rng(18)) == proof
    assert len(proof) == 15 * 64 + 32 * 32 and info["commitments"] == 15 and info["evals"] == 32
    assert verifier.verify_proof(pk.vk, tcp.TAU, instances, proof, reader=evm_ref.Reader) is True
    wrong = [list(instances[0])]
    wrong[0][0] = (wrong[0][0] + 1) % R_MOD
    assert verifier.verify_proof(pk.vk, tcp.TAU, wrong, proof, reader=evm_ref.Reader) is False
    bad = bytearray(proof)
    bad[13 * 64 + 5] ^= 1                                            # inside the first evaluation word
    assert verifier.verify_proof(pk.vk, tcp.TAU, instances, bytes(bad), reader=evm_ref.Reader) is False
    rd = EvmRead(proof)                                              # the product's reader parses the same stream
    pts = [rd.read_point() for _ in range(13)]
    assert all(p_ is not None for p_ in pts) and rd.read_scalar() < R_MOD
    pk.release()
    params.release()


def test_p256_shaped_evm_proof_emulated(emu, orc):
    _p256_evm(emu, 7)


@pytest.mark.gpu
def test_p256_shaped_evm_proof_gpu(gpu, orc):
    _p256_evm(gpu, 14)
