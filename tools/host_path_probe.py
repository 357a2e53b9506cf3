#!/usr/bin/env python3
"""GPU-box probe: PCIe-inclusive rates of the host-buffer entry points (what the Rust shim calls when
columns live in ordinary process memory)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

def main():
    be = z.Backend(0)
    rng = np.random.default_rng(1)
    k = 19; n = 1 << k
    ks = be.to_device(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
    pts = be.alloc(n * 64)
    be.g1_fixed_base_mul(ks, n, pts)
    h = be.bases_register((pts, n))
    cols = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(8)]
    dcols = [be.to_device(c) for c in cols]
    def t(f, reps=5):
        f(); t0 = time.time()
        for _ in range(reps): f()
        return (time.time() - t0) / reps * 1e3
    res = {"msm_2^19_host_ms": t(lambda: be.msm(h, cols[0])), "msm_2^19_dev_ms": t(lambda: be.msm(h, dcols[0], n)),
           "msm_batch8_2^19_host_ms": t(lambda: be.msm_batch(h, cols)), "msm_batch8_2^19_dev_ms": t(lambda: be.msm_batch(h, dcols, n))}
    w = pow(7, (R - 1) >> k, R)
    wl = np.array([((w << 256) % R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    a = cols[1].copy()
    res["ntt_2^19_host_ms"] = t(lambda: be.ntt(a, k, wl))
    res["ntt_2^19_dev_ms"] = t(lambda: be.ntt_dev(dcols[1], k, wl))
    res["upload_16MiB_ms"] = t(lambda: dcols[2].upload(cols[2]))
    res["download_16MiB_ms"] = t(lambda: dcols[2].download((n, 4)))
    print(json.dumps(res))

if __name__ == "__main__":
    main()
