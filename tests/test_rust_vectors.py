"""Vectors dumped by the REFERENCE's Rust prover (shim/sgx_k19_driver, run with HALO2_MI355X=0) pin the oracle — and through it the HIP kernels —
against halo2 itself.  None can be produced in the build image (no Rust toolchain, un-vendored crates: SURVEY.md §0.4, §8c), so every test here SKIPS
until files appear under tests/golden/rust/; the writer below keeps the format honest in the meantime (a vector written by the oracle in the same
format must round-trip through the same loader).

ZKV1 format (little endian): b"ZKV1" | kind u32 | payload
  kind 1  MSM    n u64 | n x 32 B scalars (Fr, Montgomery limbs as Rust holds them) | n x 64 B bases (G1Affine) | 96 B result (G1 {x, y, z} Jacobian)
  kind 2  NTT    log_n u32 | omega 32 B | 2^log_n x 32 B input | 2^log_n x 32 B output of best_fft
  kind 4  proof  len u64 | proof bytes | 8 x u64 first draws of the seeded ChaCha20 stream
  kind 5  draws  per_fr u32 (rng calls of one Fr::random) | n_squeezes u32 | n_squeezes x (rng calls so far u64, points written u64, scalars written u64) | total calls u64
                 — the order in which create_proof consumes `&mut rng`, read off a counting RNG at every transcript squeeze: pins zk_plonk_pk_desc.draw_schedule
  kind 6  transcript events of VERIFYING a stack-B proof (shim/p256_k18_driver): which u32 (1 = PoseidonTranscript<NativeLoader>, 2 = EvmTranscript) | proof_len u64 | proof |
                 n_events u32 | events: tag u8 + payload — 'c' common_scalar 32 B, 'p' common_point 64 B (x, y), 'S' read_scalar 32 B, 'P' read_point 64 B, 'Q' squeeze 32 B
                 (all canonical little endian).  Replaying the SAME proof bytes through this repo's readers must reproduce every value: pins the sponge / buffer framing.
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
    if kind == 5:
        per_fr, nsq = struct.unpack_from("<II", b, off)
        off += 8
        rows = [list(struct.unpack_from("<3Q", b, off + 24 * i)) for i in range(nsq)]
        return {"kind": "draws", "per_fr": per_fr, "at_squeeze": rows, "total": struct.unpack_from("<Q", b, off + 24 * nsq)[0]}
    if kind == 6:
        which = struct.unpack_from("<I", b, off)[0]
        ln = struct.unpack_from("<Q", b, off + 4)[0]
        off += 12
        proof = b[off:off + ln]
        off += ln
        n_ev = struct.unpack_from("<I", b, off)[0]
        off += 4
        events = []
        for _ in range(n_ev):
            tag = chr(b[off])
            size = 64 if tag in "pP" else 32
            raw = b[off + 1:off + 1 + size]
            off += 1 + size
            events.append((tag, int.from_bytes(raw, "little") if size == 32 else (int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:], "little"))))
        assert off == len(b), path
        return {"kind": "events", "which": which, "proof": proof, "events": events}
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


def expected_squeeze_record(census: dict, blinds: bool = True):
    """What the Rust driver's kind-5 record must read for a circuit with this census (vk_cs.json keys) if halo2 draws in draw_plan's order (blinds = False: the same plan without the Blind(Fr::random) of the commitments — the alternative a dump could point at): per squeeze
    (Fr::random draws so far, points written, scalars written) — theta, beta, gamma, y, x, then SHPLONK's y, v, u — and the total number of draws."""
    from zk_dcap_verifier_amd.plonk.prover import draw_plan
    A, L, P, d, bf, k = (census[key] for key in ("num_advice_columns", "lookups", "permutation_columns", "degree", "blinding_factors", "k"))
    chunk = d - 2
    n_sets = -(-P // chunk) if P else 0
    plan = [it for it in draw_plan(A, L, n_sets, d - 1, 1 << k, bf) if blinds or it[0] != "blind"]
    upto = lambda names: sum(c for _, _, c, sq in plan if sq in names)
    n_evals = census["advice_queries"] + census["fixed_queries"] + 1 + P + (3 * n_sets - 1 if n_sets else 0) + 5 * L
    pts = [A, A + 2 * L, A + 2 * L, A + 2 * L + n_sets + L + 1, A + 2 * L + n_sets + L + 1 + (d - 1)]
    rows = [[upto({"theta"}), pts[0], 0], [upto({"theta", "beta"}), pts[1], 0], [upto({"theta", "beta"}), pts[2], 0],
            [upto({"theta", "beta", "y"}), pts[3], 0], [upto({"theta", "beta", "y", "x"}), pts[4], 0]]
    total = sum(c for _, _, c, _ in plan)
    rows += [[total, pts[4], n_evals], [total, pts[4], n_evals], [total, pts[4] + 1, n_evals]]
    return rows, total


def test_draw_record_round_trip_and_expected_counts(tmp_path):
    """the kind-5 loader on a record written here from draw_plan — and the plan with and without the Blind draws must be told apart by it (else the Rust dump could not settle anything)"""
    census = {"k": 8, "num_advice_columns": 25, "lookups": 11, "permutation_columns": 16, "degree": 5, "blinding_factors": 5, "advice_queries": 60, "fixed_queries": 30}
    rows1, total1 = expected_squeeze_record(census)
    rows0, total0 = expected_squeeze_record(census, blinds=False)
    assert total1 - total0 == 25 + 2 * 11 + 6 + 11 + 1 + 4 and rows1[0][0] == 25 * 6 + 25 and rows0[0][0] == 25 * 6        # one Blind per commitment of phases 2-7
    assert [r[0] for r in rows1] != [r[0] for r in rows0]
    flat = [v * 8 if i % 3 == 0 else v for r in rows1 for i, v in enumerate(r)]
    write_zkv(tmp_path / "d.zkv", 5, [struct.pack("<II", 8, len(rows1)), struct.pack("<%dQ" % len(flat), *flat), struct.pack("<Q", total1 * 8)])
    v = read_zkv(tmp_path / "d.zkv")
    assert v["kind"] == "draws" and v["per_fr"] == 8 and v["total"] == total1 * 8
    assert [[r[0] // v["per_fr"], r[1], r[2]] for r in v["at_squeeze"]] == rows1


@pytest.mark.skipif(not (_vectors(5) and os.path.exists(os.path.join(DIR, "vk_cs.json"))), reason="no Rust draw-count record under tests/golden/rust")
def test_draw_schedule_equals_rust_create_proof():
    """THE check of DESIGN.md 1's top open parity risk: halo2's create_proof, run under a counting rng, must have consumed exactly draw_plan's
    draws before each squeeze; the message says when the record matches the plan WITHOUT the Blind draws instead."""
    census = json.load(open(os.path.join(DIR, "vk_cs.json")))
    for p in _vectors(5):
        v = read_zkv(p)
        got = [[r[0] / v["per_fr"], r[1], r[2]] for r in v["at_squeeze"]]
        want1, total1 = expected_squeeze_record(census)
        want0, _ = expected_squeeze_record(census, blinds=False)
        assert got == want1 and v["total"] == total1 * v["per_fr"], (p, "matches the plan without Blind draws" if got == want0 else "matches neither plan", got, want1)


def replay_events(v):
    """feed the recorded proof through the PRODUCT's reader of that transcript flavour, event by event; returns the list of mismatches (empty = framing pinned)"""
    from zk_dcap_verifier_amd.transcript import EvmRead, PoseidonRead
    rd = {1: PoseidonRead, 2: EvmRead}[v["which"]](v["proof"])
    bad = []
    for i, (tag, val) in enumerate(v["events"]):
        if tag == "c":
            rd.common_scalar(val)
        elif tag == "p":
            rd.common_point(val)
        elif tag == "S":
            got = rd.read_scalar()
        elif tag == "P":
            got = rd.read_point()
        elif tag == "Q":
            got = rd.squeeze_challenge()
        if tag in "SPQ" and got != val:
            bad.append((i, tag, got, val))
    if rd.pos != len(v["proof"]):
        bad.append(("unread bytes", len(v["proof"]) - rd.pos))
    return bad


class _Recording:
    """wraps one of the ORACLE's readers (the second writing of each transcript) and records the event sequence in the kind-6 layout"""

    def __init__(self, inner):
        self.inner, self.events = inner, []

    pos = property(lambda self: self.inner.pos)
    proof = property(lambda self: self.inner.proof)

    def squeeze(self):
        c = self.inner.squeeze()
        self.events.append(("Q", c))
        return c

    def common_scalar(self, s):
        self.events.append(("c", int(s)))
        return self.inner.common_scalar(s)

    def common_point(self, pt):
        self.events.append(("p", pt))
        return self.inner.common_point(pt)

    def read_point(self):
        pt = self.inner.read_point()
        self.events.append(("P", pt))
        return pt

    def read_scalar(self):
        s = self.inner.read_scalar()
        self.events.append(("S", s))
        return s

    def payload(self, which, proof):
        out = [struct.pack("<IQ", which, len(proof)), proof, struct.pack("<I", len(self.events))]
        for tag, val in self.events:
            out.append(tag.encode() + (val.to_bytes(32, "little") if not isinstance(val, tuple) else val[0].to_bytes(32, "little") + val[1].to_bytes(32, "little")))
        return out


@pytest.mark.parametrize("which", [1, 2])
def test_transcript_event_record_round_trip(emu, orc, tmp_path, which):
    """kind 6 written HERE — events of the oracle verifier reading a p256-shaped proof of the product prover through the oracle's own Poseidon / Keccak reader —
    loads and replays cleanly through the product's reader (two separate writings of each transcript agree event by event), and a corrupted challenge is caught"""
    import evm_ref
    import poseidon_ref
    import verifier
    import test_create_proof as tcp
    import zk_dcap_verifier_amd as z
    from zk_dcap_verifier_amd import plonk
    cs, fixed, asm, advice, instances = tcp.p256_shaped_circuit(7)
    params = z.kzg.ParamsKZG.setup(7, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    proof = plonk.NativeProver(params, pk, transcript={1: "poseidon", 2: "evm"}[which]).create_proof([a.copy() for a in advice], instances, np.random.default_rng(18))
    held = {}

    def reader(proof_):
        held["rec"] = _Recording({1: poseidon_ref.Reader, 2: evm_ref.Reader}[which](proof_))
        return held["rec"]
    assert verifier.verify_proof(pk.vk, tcp.TAU, instances, proof, reader=reader) is True
    write_zkv(tmp_path / "e.zkv", 6, held["rec"].payload(which, proof))
    v = read_zkv(tmp_path / "e.zkv")
    assert v["kind"] == "events" and v["which"] == which and v["proof"] == proof and len(v["events"]) == len(held["rec"].events) > 40
    assert replay_events(v) == []
    i = next(i for i, e in enumerate(v["events"]) if e[0] == "Q")
    v["events"][i] = ("Q", (v["events"][i][1] + 1) % (1 << 200))
    assert replay_events(v) and replay_events(v)[0][0] == i
    pk.release()
    params.release()


@pytest.mark.skipif(not _vectors(6), reason="no Rust transcript-event records under tests/golden/rust (shim/p256_k18_driver)")
def test_transcript_events_equal_rust():
    """snark-verifier's PoseidonTranscript / EvmTranscript, as the reference's stack B runs them (crates/p256-ecdsa/src/base.rs:193-244): every value their
    verification of a real proof absorbed, read and squeezed must come out of this repo's readers on the same bytes"""
    for p in _vectors(6):
        bad = replay_events(read_zkv(p))
        assert not bad, (p, bad[:3])


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


@pytest.mark.skipif(not os.path.exists(os.path.join(DIR, "kzg_bn254_19.srs")), reason="no Rust-written SRS under tests/golden/rust (params/kzg_bn254_19.srs of a `cargo test` of the reference)")
def test_srs_file_equals_the_one_gen_srs_writes():
    """circuits/src/sgx_dcap_verifier.rs:799 `gen_srs(19)` caches params/kzg_bn254_19.srs: its G1 part must hash to tests/golden/srs_kat.json (this repo's ParamsKZG.setup + write
    on the GPU with the re-derived trapdoor: tools/gen_srs_kat.py) — pins gen_srs's seed, Fr::from_u512, setup's powers, g_to_lagrange and the point encoding at once."""
    import hashlib
    kat = json.load(open(os.path.join(os.path.dirname(DIR), "srs_kat.json")))
    data = open(os.path.join(DIR, "kzg_bn254_19.srs"), "rb").read()
    assert data[:96].hex() == kat["first_96_bytes_hex"], "k / G / [tau] G differ: another seed or encoding"
    assert len(data) == kat["bytes_g1_part"] + 128 and hashlib.sha256(data[:-128]).hexdigest() == kat["sha256_g1_part"]
