#!/usr/bin/env python3
"""GPU-box probe: the quotient executors on the sgx-shaped k = 19 key — interpreter vs the generated kernels (tune quot_jit) at several kernel sizes.  For every
setting: a fresh key (hiprtc compile time reported), then `reps` native proofs one at a time with kernel timing on: the "quotient" timer per proof (every executor
launch of the proof: the three numerator parts and the lookups' theta-compressions), the proof's wall time, and the proof bytes against the interpreter's.
usage: quot_jit_probe.py [k] [reps] -- each further argument `label:key=val,key=val`"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import zk_dcap_verifier_amd as z
import sgx_shaped_circuit as sc

TAU = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA
k, reps = int(sys.argv[1]), int(sys.argv[2])
ref = None
for spec in sys.argv[3:]:
    label, tune = spec.split(":")
    be = z.Backend(0)
    be.tune(prover_side_lane=0)
    if tune:
        be.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in tune.split(",")})
    cs, fixed, asm, advice = sc.build(z, be, k, census=os.environ.get("ZK_CENSUS", "chip_estimate"))
    params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
    t0 = time.time()
    pk = z.plonk.keygen(params, cs, fixed, asm)
    keygen_s = time.time() - t0
    jit_s = be.stat_get("quot_jit_compile_s")
    native = z.plonk.NativeProver(params, pk)
    n = 1 << k
    master = [be.to_device(a) for a in advice]
    work = [be.alloc(n * 32) for _ in advice]
    walls, qms, proof = [], [], None
    for r in range(reps + 1):
        for w, m in zip(work, master):
            w.copy_from(m)
        be.sync()
        be.timing(True)
        t0 = time.time()
        proof = native.create_proof(work, [], np.random.default_rng(3))
        wall = (time.time() - t0) * 1e3
        q = be.timing_get("quotient")
        be.timing(False)
        if r:
            walls.append(wall); qms.append(q[0])
    if ref is None:
        ref = proof
    print(json.dumps({"label": label, "keygen_s": round(keygen_s, 2), "jit_compile_s": round(jit_s, 2), "quotient_ms_per_proof": round(min(qms), 3), "quotient_launch_groups": q[1],
                      "proof_ms": round(min(walls), 2), "same_bytes": proof == ref}), flush=True)
    pk.release(); params.release()
    for b in master + work:
        b.free()
    be.close()
