#!/usr/bin/env python3
"""A known answer for the day a Rust toolchain meets this repo: the SRS the reference's tests generate.

halo2-lib's `gen_srs(k)` (what circuits/src/sgx_dcap_verifier.rs:799 and crates/p256-ecdsa/src/base.rs:134 call) is
`ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed(Default::default()))` and caches the result as `params/kzg_bn254_{k}.srs`; `setup` draws ONE scalar,
`s = Fr::random(rng)` = `Fr::from_u512` of the first eight `next_u64` words = the first 64 bytes of the ChaCha20 keystream under the all-zero key and nonce,
read as a little-endian 512-bit integer and reduced mod r  ([3P-MEM]: halo2-base `utils::fs::gen_srs`, halo2_proofs `poly/kzg/commitment.rs` `ParamsKZG::setup`,
halo2curves `Fr::from_u512`, rand_chacha's block buffer — none of those sources is on this machine; the ChaCha20 block itself is pinned by its published vector).

This script derives that tau, builds the k = 19 parameters with the PRODUCT's ParamsKZG.setup (GPU: fixed-base powers + EC-NTT) and records, under tests/golden/srs_kat.json,
the SHA-256 of the file's G1 part (k | n compressed g | n compressed g_lagrange — everything but the two G2 points this library does not compute) and its first 96 bytes.
tests/test_srs_file.py recomputes tau and the first 96 bytes on the CPU (oracle big-integer curve arithmetic) and, on a GPU, the whole hash;
tests/test_rust_vectors.py compares both with `tests/golden/rust/kzg_bn254_19.srs` when a Rust run has put the reference's own file there.

usage (GPU box):  python tools/gen_srs_kat.py [k=19]"""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
CHACHA20_ZERO_BLOCK0 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                     "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")   # the published keystream of ChaCha20, zero key / nonce / counter


def chacha20_block(key: bytes, counter: int, nonce: bytes) -> bytes:
    """one 64-byte block of ChaCha20 (20 rounds, 64-bit counter | 64-bit nonce: Bernstein's layout, which rand_chacha's ChaCha20Rng uses with stream 0)"""
    def rotl(v, n):
        return ((v << n) & 0xFFFFFFFF) | (v >> (32 - n))
    st = list(struct.unpack("<4I", b"expand 32-byte k")) + list(struct.unpack("<8I", key)) + [counter & 0xFFFFFFFF, counter >> 32] + list(struct.unpack("<2I", nonce))
    x = st[:]

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(x, st)])


def gen_srs_tau() -> int:
    """the toxic waste of halo2-lib's gen_srs: Fr::from_u512 of the first 64 keystream bytes of ChaCha20Rng::from_seed([0; 32])"""
    ks = chacha20_block(bytes(32), 0, bytes(8))
    assert ks == CHACHA20_ZERO_BLOCK0, "ChaCha20 block function disagrees with the published zero-key vector"
    return int.from_bytes(ks, "little") % R_MOD


def g1_part_of(params) -> bytes:
    data = params.write()
    return data[:-128]                                                # k | g | g_lagrange (the two 64-byte G2 slots at the end are not this library's to compute)


def main():
    sys.path.insert(0, ROOT)
    import zk_dcap_verifier_amd as z
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    tau = gen_srs_tau()
    be = z.Backend(0)
    params = z.kzg.ParamsKZG.setup(k, tau, backend=be)
    part = g1_part_of(params)
    out = {"k": k, "tau_hex": "%064x" % tau, "bytes_g1_part": len(part), "sha256_g1_part": hashlib.sha256(part).hexdigest(), "first_96_bytes_hex": part[:96].hex(),
           "what": "kzg_bn254_%d.srs as halo2-lib's gen_srs writes it ([3P-MEM] derivation in tools/gen_srs_kat.py), without its last 128 bytes (g2, s_g2); "
                   "point flag in bit 255 (halo2curves 0.3.1)" % k}
    path = os.path.join(ROOT, "tests", "golden", "srs_kat.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))
    params.release()
    be.close()


if __name__ == "__main__":
    main()
