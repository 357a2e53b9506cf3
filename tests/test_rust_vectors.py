"""Vectors dumped by the REFERENCE's Rust prover (shim/sgx_k19_driver, run with HALO2_MI355X=0) pin the oracle — and through it the HIP kernels —
against halo2 itself.  None can be produced in the build image (no Rust toolchain, un-vendored crates: SURVEY.md §0.4, §8c), so every test here SKIPS
until files appear under tests/golden/rust/; the writer below keeps the format honest in the meantime (a vector written by the oracle in the same
format must round-trip through the same loader).

ZKV1 format (little endian): b"ZKV1" | kind u32 | payload
  kind 1  MSM    n u64 | n x 32 B scalars (Fr, Montgomery limbs as Rust holds them) | n x 64 B bases (G1Affine) | 96 B result (G1 {x, y, z} Jacobian)
  kind 2  NTT    log_n u32 | omega 32 B | 2^log_n x 32 B input | 2^log_n x 32 B output of best_fft
  kind 4  proof  len u64 | proof bytes | 8 x u64 first draws of the seeded ChaCha20 stream
plus vk_cs.json — the real circuit's census (A, F, L, equality columns, degree), to replace the estimates of tools/sgx_shaped_circuit.py.
"""
import glob
import json
import os
import struct

import numpy as np
import pytest

from conftest import ROOT

DIR = os.path.join(ROOT, "tests", "golden", "rust")


def read_zkv(path):
    b = open(path, "rb").read()
    assert b[:4] == b"ZKV1", path
    kind = struct.unpack_from("<I", b, 4)[0]
    off = 8
    if kind == 1:
        n = struct.unpack_from("<Q", b, off)[0]
        off += 8
        sc = np.frombuffer(b, dtype="<u8", count=4 * n, offset=off).reshape(n, 4); off += 32 * n
        bases = np.frombuffer(b, dtype="<u8", count=8 * n, offset=off).reshape(n, 8); off += 64 * n
        res = np.frombuffer(b, dtype="<u8", count=12, offset=off)
        return {"kind": "msm", "scalars": sc, "bases": bases, "result": res}
    if kind == 2:
        log_n = struct.unpack_from("<I", b, off)[0]
        off += 4
        n = 1 << log_n
        omega = np.frombuffer(b, dtype="<u8", count=4, offset=off); off += 32
        a = np.frombuffer(b, dtype="<u8", count=4 * n, offset=off).reshape(n, 4); off += 32 * n
        out = np.frombuffer(b, dtype="<u8", count=4 * n, offset=off).reshape(n, 4)
        return {"kind": "ntt", "log_n": log_n, "omega": omega, "input": a, "output": out}
    if kind == 4:
        ln = struct.unpack_from("<Q", b, off)[0]
        off += 8
        return {"kind": "proof", "proof": b[off:off + ln], "first_draws": list(struct.unpack_from("<8Q", b, off + ln))}
    raise ValueError(f"{path}: unknown ZKV1 kind {kind}")


def write_zkv(path, kind, parts):
    with open(path, "wb") as f:
        f.write(b"ZKV1" + struct.pack("<I", kind))
        for p in parts:
            f.write(np.ascontiguousarray(p).tobytes() if isinstance(p, np.ndarray) else p)


def _vectors(kind):
    return sorted(p for p in glob.glob(os.path.join(DIR, "*.zkv")) if read_zkv(p)["kind"] == kind)


def _jac_to_affine(orc, jac12):
    """G1 {x, y, z}: halo2curves' bn256 G1 is Jacobian (x / z^2, y / z^3); orc.g1_to_affine uses the same convention"""
    return orc.g1_to_affine(np.ascontiguousarray(jac12).reshape(1, 12))[0]


def test_format_round_trip_with_oracle_written_vectors(orc, pyref, tmp_path):
    """the loader reads what the Rust driver writes: exercised here with vectors the ORACLE writes in the same layout"""
    import parity_cases as pc
    sc, bases = pc.msm_inputs(orc, pyref, 64, 3)
    res = orc.best_multiexp(sc, bases)
    write_zkv(tmp_path / "m.zkv", 1, [struct.pack("<Q", 64), sc, bases, res])
    v = read_zkv(tmp_path / "m.zkv")
    assert v["kind"] == "msm" and (v["scalars"] == sc).all() and (v["bases"] == bases).all() and (v["result"] == res).all()
    a = pc.rand_fr(orc, pyref, 32, 4)
    w = orc.fr_from_ints([pyref.omega(5)])[0]
    write_zkv(tmp_path / "n.zkv", 2, [struct.pack("<I", 5), w, a, orc.best_fft(a, w, 5)])
    v = read_zkv(tmp_path / "n.zkv")
    assert v["kind"] == "ntt" and v["log_n"] == 5 and (v["output"] == orc.best_fft(v["input"], v["omega"], 5)).all()
    write_zkv(tmp_path / "p.zkv", 4, [struct.pack("<Q", 5), b"hello", struct.pack("<8Q", *range(8))])
    assert read_zkv(tmp_path / "p.zkv") == {"kind": "proof", "proof": b"hello", "first_draws": list(range(8))}


@pytest.mark.skipif(not _vectors(1), reason="no Rust MSM vectors under tests/golden/rust (shim/README.md: needs a Rust toolchain)")
def test_oracle_msm_equals_rust_best_multiexp(orc):
    for p in _vectors(1):
        v = read_zkv(p)
        got = _jac_to_affine(orc, orc.best_multiexp(v["scalars"], v["bases"]))
        assert (got == _jac_to_affine(orc, v["result"])).all(), p


@pytest.mark.skipif(not _vectors(2), reason="no Rust NTT vectors under tests/golden/rust")
def test_oracle_fft_equals_rust_best_fft(orc):
    for p in _vectors(2):
        v = read_zkv(p)
        assert (orc.best_fft(v["input"], v["omega"], v["log_n"]) == v["output"]).all(), p


@pytest.mark.gpu
@pytest.mark.skipif(not (_vectors(1) or _vectors(2)), reason="no Rust vectors under tests/golden/rust")
def test_gpu_equals_rust_vectors(gpu, orc):
    import zk_dcap_verifier_amd as z
    for p in _vectors(1):
        v = read_zkv(p)
        got = z.arithmetic.best_multiexp(v["scalars"], v["bases"], backend=gpu)
        assert (got[:8] == _jac_to_affine(orc, v["result"])).all(), p
    for p in _vectors(2):
        v = read_zkv(p)
        a = v["input"].copy()
        z.arithmetic.best_fft(a, v["omega"], v["log_n"], backend=gpu)
        assert (a == v["output"]).all(), p


@pytest.mark.skipif(not os.path.exists(os.path.join(DIR, "vk_cs.json")), reason="no vk_cs.json under tests/golden/rust")
def test_census_of_the_real_circuit_is_recorded():
    cs = json.load(open(os.path.join(DIR, "vk_cs.json")))
    for key in ("k", "num_advice_columns", "num_fixed_columns", "lookups", "permutation_columns", "degree", "blinding_factors"):
        assert key in cs
    print("real sgx_dcap_verifier census:", cs)
