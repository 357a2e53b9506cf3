#!/usr/bin/env python3
"""GPU-box probe: BASELINE configs[0] (the p256-shaped k = 18 circuit, Poseidon transcript) one proof at a time through zk_plonk_create_proof — per-phase wall times of the last
proof; under `rocprofv3 --kernel-trace` the kernel list of the same proof (tools/trace_gaps.py).  usage: p256_latency_probe.py [k=18] [proofs=6]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import zk_dcap_verifier_amd as z
import p256_shaped_circuit as p256
TAU = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
be = z.Backend(0)
tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("ZK_TUNE", "").split(",") if kv}
if tune:
    be.tune(**tune)
cs, fixed, asm, advice, inst = p256.build(k)
params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
pk = z.plonk.keygen(params, cs, fixed, asm)
prover = z.plonk.NativeProver(params, pk, transcript="poseidon")
dev = [be.to_device(a) for a in advice]
work = [be.alloc(a.nbytes) for a in advice]
ts = []
for r in range(reps):
    for w, m in zip(work, dev):
        w.copy_from(m)
    be.sync()
    t = time.time()
    proof = prover.create_proof(work, inst, np.random.default_rng(r))
    ts.append(round((time.time() - t) * 1e3, 2))
print(json.dumps({"k": k, "ms": ts, "phase_ms_last": {a: round(b, 2) for a, b in prover.phase_ms.items()}, "proof_bytes": len(proof), "tune": tune}))
