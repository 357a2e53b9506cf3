#!/usr/bin/env python3
"""GPU-box probe: keygen + create_proof of the sgx-shaped synthetic circuit (tools/sgx_shaped_circuit.py) at a given k; prints per-phase
wall times.  (Timing only: the same circuit's proofs are VERIFIED at k = 8 ... 21 by tests/test_create_proof.py and, at k = 19, by bench.py's CPU leg.)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import zk_dcap_verifier_amd as z
import sgx_shaped_circuit as sc
from zk_dcap_verifier_amd.transcript import Blake2bWrite

TAU = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    be = z.Backend(0, os.environ.get("ZK_LIB"))        # ZK_LIB: A/B against another build of the library on the same box
    tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("ZK_TUNE", "").split(",") if kv}
    if tune:
        be.tune(**tune)
    out = {"k": k, "tune": tune, "host_witness": os.environ.get("ZK_HOST_WITNESS") == "1"}
    census = os.environ.get("ZK_CENSUS", "chip_estimate")
    instances = []
    t = time.time()
    if census == "p256":                               # BASELINE configs[0]: the reference's stack-B circuit shape (crates/p256-ecdsa, k = 18; degree 4, 3 advice, 1 lookup, 15 instance values)
        import p256_shaped_circuit as p256
        cs, fixed, asm, advice, instances = p256.build(k)
    else:
        cs, fixed, asm, advice = sc.build(z, be, k, census=census)
    out["census"] = census; out["build_witness_s"] = round(time.time() - t, 3)
    t = time.time(); params = z.kzg.ParamsKZG.setup(k, TAU, backend=be); out["srs_setup_s"] = round(time.time() - t, 3)
    if os.environ.get("ZK_BY_COSETS") == "1":          # the multi-GPU quotient unit on one GPU: 2^(ek-k) coset NTTs of size n instead of one of size 2^ek
        params.quotient_by_cosets = True
        out["quotient_by_cosets"] = True
    piece_cosets = os.environ.get("ZK_PIECE_COSETS", "1") != "0"      # 0: keep the extended forms although cs_degree - 1 cosets would do (A/B of zk_cosets_to_pieces_dev)
    t = time.time(); pk = z.plonk.keygen(params, cs, fixed, asm, piece_cosets=piece_cosets); out["keygen_s"] = round(time.time() - t, 3)
    out["pieces_from_cosets"] = pk.pieces_from_cosets
    out["program"] = be.quotient_program_info(pk.evaluator.handle)
    out["program"]["opmix"] = be.quotient_program_opmix(pk.evaluator.handle)
    n = 1 << k
    master = [be.to_device(a) for a in advice]
    work = [be.alloc(n * 32) for _ in advice]
    times, proof = [], None
    native = z.plonk.NativeProver(params, pk) if os.environ.get("ZK_PROVER") == "native" else None      # the C++ per-proof driver (zk_plonk_create_proof) instead of the Python twin
    out["prover"] = "native" if native else "python twin"
    for r in range(reps + 1):
        host_witness = os.environ.get("ZK_HOST_WITNESS") == "1"      # PCIe-inclusive: hand create_proof the host columns (as the Rust boundary would)
        if not host_witness:
            for w, m in zip(work, master):
                w.copy_from(m)
        be.sync()
        t = time.time()
        tr = Blake2bWrite()
        tm = {}
        if native:
            proof = native.create_proof(advice if host_witness else work, instances, np.random.default_rng(r))
            tm, info = dict(native.phase_ms), {"proof_bytes": len(proof)}
        else:
            info = z.plonk.create_proof(params, pk, advice if host_witness else work, instances, np.random.default_rng(r), tr, timings=tm)
            proof = tr.finalize()
        times.append(time.time() - t)
    be.timing(True)                                    # once more with HIP-event kernel timing (alone on the GPU: the event pairs bracket only this proof's kernels)
    for w, m in zip(work, master):
        w.copy_from(m)
    if native:
        native.create_proof(work, instances, np.random.default_rng(99))
    else:
        z.plonk.create_proof(params, pk, work, instances, np.random.default_rng(99), Blake2bWrite())
    out["kernel_ms"] = {lab: round(be.timing_get(lab)[0] or 0.0, 3) for lab in ("msm_sort", "msm_accumulate", "msm_reduce", "quotient")}
    out["msm_pairs"] = be.stat_get("msm_pairs")
    be.timing(False)
    out["create_proof_ms"] = [round(x * 1e3, 2) for x in times]
    out["phase_ms_last"] = {k_: round(v, 2) for k_, v in tm.items()}
    out["proof_bytes"] = len(proof)
    out["info"] = {k_: v for k_, v in info.items() if k_ != "h_eval"}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
