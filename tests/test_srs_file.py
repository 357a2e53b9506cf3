"""SRS file codec: G1 point (de)compression on the device and ParamsKZG.{write, read}.

Pins: (1) the definition — y^2 = x^3 + 3, parity flag — against Python integers; (2) the reference's only proof bytes,
bin/assets/proof.bin (bin/src/main.rs:275; stack B, flag in bit 254 — SURVEY App. B): every one of its 15 point words must
decompress to a curve point and compress back to the very same 32 bytes."""
import os

import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd.fields import P_MOD, fq_int
from zk_dcap_verifier_amd.transcript import point_to_bytes

TAU = 987654321987654321


def _proof_bin_words():
    from conftest import ROOT
    txt = open(os.path.join(ROOT, "tests", "golden", "proof.bin")).read().strip()
    raw = bytes.fromhex(txt[2:])
    return [raw[i:i + 32] for i in range(0, len(raw), 32)]


def _roundtrip_proof_bin(be):
    words = _proof_bin_words()
    pts = words[:13] + words[45:47]                                  # 13 commitments + the 2 SHPLONK points (App. B)
    arr = np.frombuffer(b"".join(pts), dtype=np.uint64).reshape(-1, 4).copy()
    b, d, b2 = be.to_device(arr), be.alloc(len(pts) * 64), be.alloc(len(pts) * 32)
    be.g1_decompress_dev(b, len(pts), 254, d)
    aff = d.download((len(pts), 8))
    for row, w in zip(aff, pts):
        x, y = fq_int(row[:4]), fq_int(row[4:])
        assert (y * y - x * x * x - 3) % P_MOD == 0
        assert x == int.from_bytes(w, "little") & ((1 << 254) - 1) and (y & 1) == (w[31] >> 6) & 1
    be.g1_compress_dev(d, len(pts), 254, b2)
    assert b2.download((len(pts), 4)).tobytes() == b"".join(pts)
    # an evaluation word that is NOT an x coordinate must be refused (App. B: 14 of the 32 scalar words are not on the curve)
    bad = [w for w in words[13:45] if pow((int.from_bytes(w, "little") ** 3 + 3) % P_MOD, (P_MOD - 1) // 2, P_MOD) != 1][0]
    bb = be.to_device(np.frombuffer(bad, dtype=np.uint64).reshape(1, 4).copy())
    with pytest.raises(z.ZkError):
        be.g1_decompress_dev(bb, 1, 254, d)


def _srs_roundtrip(be, k):
    n = 1 << k
    params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
    for sign_bit in (255, 254):
        blob = params.write(g2=bytes(range(64)), s_g2=bytes(range(64, 128)), sign_bit=sign_bit)
        assert len(blob) == 4 + 2 * n * 32 + 128 and blob[:4] == k.to_bytes(4, "little")
        # the encoding is the definition: x little-endian, parity of y in the flag bit (identity cannot occur in an SRS)
        for i in (0, 1, n - 1):
            x, y = fq_int(params.g_host[i][:4]), fq_int(params.g_host[i][4:])
            want = bytearray(x.to_bytes(32, "little"))
            want[31] |= (y & 1) << (7 if sign_bit == 255 else 6)
            assert blob[4 + 32 * i:4 + 32 * (i + 1)] == bytes(want)
            if sign_bit == 255:
                assert bytes(want) == point_to_bytes((x, y))         # the transcript's host encoder agrees with the kernel
        back = z.kzg.ParamsKZG.read(blob, backend=be, sign_bit=sign_bit)
        assert (back.g_host == params.g_host).all() and (back.g_lagrange_host == params.g_lagrange_host).all()
        assert back.g2 == bytes(range(64)) and back.s_g2 == bytes(range(64, 128))
        # the re-read tables commit to the same points
        poly = np.ascontiguousarray(params.g_host[:, :4] & np.uint64((1 << 60) - 1))
        assert (back.commit_lagrange(poly) == params.commit_lagrange(poly)).all()
        back.release()
    # a stream whose FIRST point is replaced by an evaluation word of proof.bin that is not an x coordinate must be refused
    bad_x = [w for w in _proof_bin_words()[13:45] if pow((int.from_bytes(w, "little") ** 3 + 3) % P_MOD, (P_MOD - 1) // 2, P_MOD) != 1][0]
    tampered = bytearray(params.write())
    tampered[4:36] = bad_x
    with pytest.raises(z.ZkError):
        z.kzg.ParamsKZG.read(bytes(tampered), backend=be)
    with pytest.raises(ValueError):
        z.kzg.ParamsKZG.read(params.write()[:-1], backend=be)
    params.release()


def test_emulated_proof_bin_points_roundtrip(emu):
    _roundtrip_proof_bin(emu)


def test_emulated_srs_file_roundtrip(emu):
    _srs_roundtrip(emu, 4)


@pytest.mark.gpu
def test_gpu_proof_bin_points_roundtrip(gpu):
    _roundtrip_proof_bin(gpu)


@pytest.mark.gpu
def test_gpu_srs_file_roundtrip(gpu):
    _srs_roundtrip(gpu, 12)


# ---- the SRS of the reference's own tests (halo2-lib gen_srs: ChaCha20, all-zero seed), as a known answer for the first machine with cargo -------------------------
def _srs_kat():
    import json
    from conftest import ROOT
    return json.load(open(os.path.join(ROOT, "tests", "golden", "srs_kat.json")))


def test_gen_srs_trapdoor_and_first_points_cpu(orc, pyref):
    """tau of `gen_srs(k)` re-derived here (ChaCha20 block pinned by its published vector, Fr::from_u512 of the first 64 keystream bytes: tools/gen_srs_kat.py) is the
    trapdoor SURVEY App. C.7 computed and every test SRS of this repo uses; the first 96 bytes of kzg_bn254_19.srs — k, G, [tau] G and the start of [tau^2] G,
    compressed with the flag in bit 255 — recomputed with big-integer curve arithmetic equal the committed ones (which the GPU's ParamsKZG.setup + write produced)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_srs_kat as gk
    import test_create_proof as tcp
    kat = _srs_kat()
    tau = gk.gen_srs_tau()
    assert tau == tcp.TAU == int(kat["tau_hex"], 16)
    G = (1, 2)
    want = (19).to_bytes(4, "little") + point_to_bytes(G) + point_to_bytes(pyref.g1_mul(G, tau)) + point_to_bytes(pyref.g1_mul(G, tau * tau))
    assert want[:96].hex() == kat["first_96_bytes_hex"] and kat["bytes_g1_part"] == 4 + 2 * 32 * (1 << 19)


@pytest.mark.gpu
def test_gen_srs_file_hash_gpu(gpu):
    """ParamsKZG.setup(19, tau) + write on the GPU (2^19 fixed-base multiplications, the EC-NTT g_to_lagrange, 2^20 point compressions) reproduces the committed SHA-256 of
    the file's G1 part"""
    import hashlib
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_srs_kat as gk
    kat = _srs_kat()
    params = z.kzg.ParamsKZG.setup(kat["k"], gk.gen_srs_tau(), backend=gpu)
    part = gk.g1_part_of(params)
    params.release()
    assert len(part) == kat["bytes_g1_part"] and hashlib.sha256(part).hexdigest() == kat["sha256_g1_part"] and part[:96].hex() == kat["first_96_bytes_hex"]
