#!/usr/bin/env python3
"""GPU-box probe: counting-sort tunables of the batched MSM (workgroups per batch, threads)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z

def main():
    be = z.Backend(0)
    rng = np.random.default_rng(1)
    k = 19; n = 1 << k
    ks = be.to_device(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
    pts = be.alloc(n * 64)
    be.g1_fixed_base_mul(ks, n, pts)
    h = be.bases_register((pts, n))
    cols = [be.to_device(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)) for _ in range(25)]
    def run(nb, label, **tune):
        be.tune(**tune)
        be.msm_batch(h, cols[:nb], n)
        be.timing(True)
        t = time.time()
        for _ in range(3): be.msm_batch(h, cols[:nb], n)
        dt = (time.time() - t) / 3
        lab = {kk: be.timing_get(kk) for kk in ("msm_sort", "msm_accumulate", "msm_reduce")}
        be.timing(False)
        print(json.dumps({"nb": nb, **tune, "ms_per_msm": round(dt * 1e3 / nb, 3), "per_msm": {kk: round(v[0] / max(v[1], 1) / nb, 3) for kk, v in lab.items()}}), flush=True)
    for w in (256, 512, 1024, 2048, 4096):
        run(25, "batch_wgs", msm_sort_batch_wgs=w)
    be.tune(msm_sort_batch_wgs=2048)
    for w in (32, 64, 128, 256):
        run(1, "single_wgs", msm_sort_wgs=w)
    be.tune(msm_sort_wgs=256)
    for th in (256, 512, 1024):
        run(25, "threads", msm_sort_threads=th)

if __name__ == "__main__":
    main()
