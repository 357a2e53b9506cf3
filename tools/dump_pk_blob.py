#!/usr/bin/env python3
"""Write a proving key + witness + draws as ONE flat file a plain-C program can consume: tests/csrc/capi_prove.c rebuilds the proving key with
zk_plonk_pk_build from the HOST data below, proves with zk_plonk_prove and compares with the expected proof bytes — the whole per-proof path through the C ABI
with no Python (and no zk_plonk_pk_desc marshalling by hand) in the loop, which is exactly what a Rust caller does (shim/halo2_proofs_mi355x/src/pk_desc.rs).

ZKPK1 layout (little endian, every section 8-byte aligned):
    "ZKPK" | u32 version = 1
    u32 x 12: k, cs_degree, blinding_factors, n_fixed, n_advice, n_instance, n_lookups, n_perm_columns, n_advice_queries, n_fixed_queries, transcript, draw_schedule
    u32 perm_columns[2 P] | advice_queries[2 AQ] | fixed_queries[2 FQ] | lookup_table_key[L]            (padded to 8 bytes)
    transcript_repr 32 B
    u64 len | ZKQ1 evaluator blob                                                                        (each blob padded to 8 bytes)
    L x ( u64 len | ZKQ1 input-expression blob | u64 len | ZKQ1 table-expression blob )
    n x 64 B params.g | n x 64 B params.g_lagrange                                                       (G1Affine, Montgomery)
    n_fixed x n x 32 B pk.fixed_values | P x n x 32 B pk.permutation.permutations                        (Lagrange columns, Montgomery)
    n_advice x n x 32 B witness (advice columns before blinding)
    n_instance x ( u64 len | len x 32 B canonical little-endian values )
    u64 n_draws | n_draws x 32 B   — the caller's Fr::random stream in the order create_proof asks for it (the C program's zk_rng_fn serves it sequentially)
    u64 proof_len | expected proof bytes (the golden of the independent CPU prover when the circuit has one)

usage: dump_pk_blob.py OUT.zkpk [toy|sgx|p256] [k] [seed]     (runs keygen on the emulator build when no GPU is present — setup only; the C program does the proving)
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p_)


def _pad8(b: bytes) -> bytes:
    return b + bytes(-len(b) % 8)


def build_blob(be, cs, fixed, asm, advice, instances, k, tau, seed, expected=None) -> bytes:
    import zk_dcap_verifier_amd as z
    from zk_dcap_verifier_amd import plonk
    from zk_dcap_verifier_amd.fields import rand_fr_array
    from zk_dcap_verifier_amd.plonk.prover import draw_plan
    from zk_dcap_verifier_amd.transcript import Blake2bWrite
    n = 1 << k
    params = z.kzg.ParamsKZG.setup(k, tau, backend=be)
    pk = plonk.keygen(params, cs, fixed, asm)
    aq, fq = cs.advice_queries(), cs.fixed_queries()
    keys, key_ids = {}, []
    for lk in cs.lookups:
        key_ids.append(keys.setdefault(tuple(lk.table_expressions), len(keys)))
    u32 = lambda vals: np.asarray(vals, dtype=np.int64).astype(np.uint32).tobytes()
    out = [b"ZKPK", struct.pack("<I", 1),
           struct.pack("<12I", k, cs.degree(), cs.blinding_factors(), cs.num_fixed_columns, cs.num_advice_columns, cs.num_instance_columns, len(cs.lookups),
                       len(cs.permutation_columns), len(aq), len(fq), 0, 1),
           _pad8(u32([v for t, i in cs.permutation_columns for v in (t, i)]) + u32([v for c, r in aq for v in (c, r)]) + u32([v for c, r in fq for v in (c, r)]) + u32(key_ids)),
           int(pk.vk.transcript_repr).to_bytes(32, "little")]
    blob = pk.program.to_blob()
    out += [struct.pack("<Q", len(blob)), _pad8(blob)]
    for a, b in pk.lookup_compressors:
        for ev_ in (a, b):
            bl = ev_.program.to_blob()
            out += [struct.pack("<Q", len(bl)), _pad8(bl)]
    out += [np.ascontiguousarray(params.g_host, dtype=np.uint64).tobytes(), np.ascontiguousarray(params.g_lagrange_host, dtype=np.uint64).tobytes()]
    out += [d.download((n, 4)).tobytes() for d in pk.fixed_values] + [d.download((n, 4)).tobytes() for d in pk.sigma_values]
    host_advice = [np.ascontiguousarray(a, dtype=np.uint64).reshape(n, 4).copy() for a in advice]
    out += [a.tobytes() for a in host_advice]
    for col in instances:
        out += [struct.pack("<Q", len(col)), b"".join(int(v).to_bytes(32, "little") for v in col)]
    chunk = cs.permutation_chunk_len()
    n_sets = -(-len(cs.permutation_columns) // chunk) if cs.permutation_columns else 0
    rng = np.random.default_rng(seed)
    draws = [rand_fr_array(rng, cnt) for _, _, cnt, _ in draw_plan(cs.num_advice_columns, len(cs.lookups), n_sets, cs.degree() - 1, n, cs.blinding_factors())]
    stream = np.concatenate(draws) if draws else np.zeros((0, 4), np.uint64)
    out += [struct.pack("<Q", stream.shape[0]), stream.tobytes()]
    if expected is None:                                             # no golden for this circuit: the Python twin's bytes (same kernels) are the reference
        tr = Blake2bWrite()
        plonk.create_proof(params, pk, [a.copy() for a in host_advice], instances, np.random.default_rng(seed), tr)
        expected = tr.finalize()
    out += [struct.pack("<Q", len(expected)), _pad8(expected)]
    pk.release()
    params.release()
    return b"".join(out)


def toy_blob(be, k=6, seed=7) -> bytes:
    import test_create_proof as t
    cs, fixed, asm, advice, instances = t.toy_circuit(k)
    golden = None
    if (k, seed) == (6, 7):
        golden = t._golden(t.GOLDEN_PROOF)
    return build_blob(be, cs, fixed, asm, advice, instances, k, t.TAU, seed, golden)


def sgx_blob(be, k=8, seed=3, census="chip_estimate") -> bytes:
    import sgx_shaped_circuit as sc
    import test_create_proof as t
    import zk_dcap_verifier_amd as z
    cs, fixed, asm, advice = sc.build(z, be, k, census=census)
    golden = None
    if (k, seed, census) == (8, 3, "chip_estimate"):
        golden = t._golden(t.GOLDEN_SGX)
    adv = [a.download((1 << k, 4)) if not isinstance(a, np.ndarray) else a for a in advice]
    return build_blob(be, cs, fixed, asm, adv, [], k, t.TAU, seed, golden)


def p256_blob(be, k=7, seed=18) -> bytes:
    """the census of the reference's stack-B circuit (degree 4: three h pieces, so zk_plonk_pk_build keeps three cosets of the key's columns and no extended form)"""
    import p256_shaped_circuit as p256
    import test_create_proof as t
    cs, fixed, asm, advice, instances = p256.build(k)
    return build_blob(be, cs, fixed, asm, advice, instances, k, t.TAU, seed, None)


def main():
    import zk_dcap_verifier_amd as z
    out = sys.argv[1]
    which = sys.argv[2] if len(sys.argv) > 2 else "toy"
    try:
        be = z.Backend(0)
    except z.ZkError:                                                # no GPU here: the emulator build of the same kernels does the (setup-time) keygen
        be = z.Backend(0, lib_path=os.path.join(ROOT, "tests", "csrc", "libzkmi355_emu.so"))
        be.tune(msm_sort_threads=64, msm_sort_wgs=3, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4, msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    k = int(sys.argv[3]) if len(sys.argv) > 3 else (6 if which == "toy" else 8)
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else (7 if which == "toy" else 3)
    blob = toy_blob(be, k, seed) if which == "toy" else p256_blob(be, k, seed) if which == "p256" else sgx_blob(be, k, seed)
    open(out, "wb").write(blob)
    print("wrote", out, len(blob), "bytes")


if __name__ == "__main__":
    main()
