#!/usr/bin/env python3
"""GPU-box probe: latency of ONE proof alone on the GPU through the native prover (zk_plonk_create_proof) and through its Python twin, same circuit and witness."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import zk_dcap_verifier_amd as z
import sgx_shaped_circuit as sc
from zk_dcap_verifier_amd.transcript import Blake2bWrite
TAU = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
be = z.Backend(0)
cs, fixed, asm, advice = sc.build(z, be, k, census=os.environ.get("ZK_CENSUS", "chip_estimate"))
params = z.kzg.ParamsKZG.setup(k, TAU, backend=be)
pk = z.plonk.keygen(params, cs, fixed, asm)
native = z.plonk.NativeProver(params, pk)
master = [be.to_device(a) for a in advice]
work = [be.alloc((1 << k) * 32) for _ in advice]
out = {"k": k}
for name in ("native", "python", "native", "python"):
    ts = []
    for r in range(4):
        for w, m in zip(work, master):
            w.copy_from(m)
        be.sync()
        t = time.time()
        if name == "native":
            proof = native.create_proof(work, [], np.random.default_rng(r))
        else:
            tr = Blake2bWrite(); z.plonk.create_proof(params, pk, work, [], np.random.default_rng(r), tr); proof2 = tr.finalize()
        ts.append(round((time.time() - t) * 1e3, 2))
    out.setdefault(name, []).append(ts)
out["same_bytes"] = proof == proof2
out["native_phase_ms"] = native.phase_ms
tm = {}
for w, m in zip(work, master):
    w.copy_from(m)
tr = Blake2bWrite(); z.plonk.create_proof(params, pk, work, [], np.random.default_rng(1), tr, timings=tm)
out["python_phase_ms"] = {k_: round(v, 2) for k_, v in tm.items()}
print(json.dumps(out))
